// jpezy_capi.hip -- implementation of the C-ABI declared in include/jpezy_hip.h.
// Owns: per-GPU context (device id, stream, device constant tables, staging buffers, fallback counter).
// No CPU fallback: every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/jpezy_constants.h"
#include "../../include/jpezy_hip.h"
#include "jpezy_device.h"
#include "jpezy_entropy.h"
#include "jpezy_huffdec.h"
#include "jpezy_host_codec.h"
#include "jpezy_hostpipe.h"

using namespace jpezy_dev;

namespace {

thread_local std::string g_err;

int set_err(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}
int hip_err(hipError_t e, const char* what)
{
    return set_err(JPEZY_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                     \
    do {                                                  \
        hipError_t e__ = (expr);                          \
        if (e__ != hipSuccess) return hip_err(e__, #expr); \
    } while (0)

// No exception crosses the C ABI (include/jpezy_hip.h): every extern "C" body that allocates host memory is a
// function-try-block ending in JPEZY_CATCH.
#define JPEZY_CATCH                                                                                         \
    catch (const std::bad_alloc&) { return set_err(JPEZY_E_NOSPACE, "out of host memory"); }                \
    catch (const std::exception& e) { return set_err(JPEZY_E_HIP, std::string("unexpected exception: ") + e.what()); }

// coefficient buffers are moved with 16-byte accesses (one MCU = 768 or 512 bytes, so only the base matters)
inline bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

// Context-wide device tables (dequantiser constants, cached JFIF header) may be read by launches still in flight on ANY
// stream the caller drives this context with: before rewriting them, wait for the whole device; and never from inside
// a stream capture (a synchronisation there would invalidate the capture).
// own_stream_only: the context is a child of jpezy_decode_jpeg_batch -- it is only ever driven on its own stream, so waiting for
// that stream is enough (eight children that each stalled the whole device for every file with new tables serialised the batch).
int drain_before_table_rewrite(hipStream_t s, const char* what, bool own_stream_only = false)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
        return set_err(JPEZY_E_BADARG, std::string(what) + ": new tables/header cannot be uploaded while the stream is being captured; "
                                                            "make the first call with these arguments outside the capture");
    if (own_stream_only)
        HIP_TRY(hipStreamSynchronize(s));
    else
        HIP_TRY(hipDeviceSynchronize());
    return JPEZY_OK;
}

const int kQt[2][64] = { JPEZY_QT_LUMA_INIT, JPEZY_QT_CHROMA_INIT };
const unsigned char kZzInv[64] = JPEZY_ZZ_INV_INIT;   // natural index -> zig-zag position

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) return hip_err(e, "hipMalloc");
        cap = n;
        return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

namespace {

// Chunks of the streaming host-buffer entry points: MCU-row bands of a frame that is large against the chunk size, otherwise
// several whole frames.  Chunk k covers frames [f0, f0 + nf) and, in band mode (nf == 1), MCU rows [y0, y1) of frame f0.
struct HostChunk { int f0, nf, y0, y1; };

std::vector<HostChunk> plan_host_chunks(int W, int H, int n_frames, size_t bytes_per_px, size_t target)
{
    std::vector<HostChunk> out;
    const int mcu_rows = jpezy_mcu_rows(H);
    const size_t frame_bytes = (size_t)W * H * bytes_per_px;
    if (frame_bytes > 2 * target) {
        const size_t row_bytes = (size_t)16 * W * bytes_per_px;
        const int rows_per = (int)std::max<size_t>(1, target / row_bytes);
        for (int f = 0; f < n_frames; ++f)
            for (int y = 0; y < mcu_rows; y += rows_per) out.push_back({ f, 1, y, std::min(y + rows_per, mcu_rows) });
    } else {
        const int per = (int)std::max<size_t>(1, target / std::max<size_t>(frame_bytes, 1));
        for (int f = 0; f < n_frames; f += per) out.push_back({ f, std::min(per, n_frames - f), 0, mcu_rows });
    }
    return out;
}

}  // namespace

struct jpezy_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    DeviceTables* d_tab = nullptr;
    unsigned long long* d_counter = nullptr;
    double* d_dqscale = nullptr;   // [3][8][8]
    int* d_dqt = nullptr;          // [3][64]
    float* d_dqscale_f = nullptr;  // [8][8] luma constants in FP32 (decode tolerance mode)
    int dec_tolerance = 0;         // 0 = bit-exact decode (default), 1 = luma in FP32, output within one of the reference per channel
    uint16_t dq_cache[3][64];
    int coef_limit = 0;            // 2^15 / largest quantiser: the generic kernels and the tolerance mode of the fused kernel
    int coef_limit_exact = 0;      // 2^23 / largest quantiser: the fused kernel's exact mode (fast path + reference sum err by <= 1.2e-6 against a guard band of 3.8e-6)
    bool dq_valid = false;
    int force_exact = 0;           // 0 normal, 1 everything through the reference-order path, 2 (f32 variant) through level 2,
                                   // 3 (f32 variant) through the per-lane evaluator of the queue-overflow case
    int variant = 1;               // encode kernel: 0 = FP64 butterflies, 1 = FP32 first level (default)
#ifdef JPEZY_TRACE
    unsigned long long* d_trace = nullptr;
#endif
    DevBuf dump_t;                 // JPEZY_DUMP_T builds: level-1 t values of the last jpezy_fdct_quant_dev call
    DevBuf in[3], out, scratch;    // staging for the host-buffer entry points; scratch: samples of the generic decoder
    // GPU entropy coder (jpezy_entropy.hip): code tables + scratch
    jpezy_dev::entropy::CodeTables* d_codes = nullptr;
    DevBuf e_tmp, e_small, e_U, e_cnt, e_out, e_coef;
    DevBuf e_tt, e_fft;            // totals per tile (256 coded blocks: bits) and per piece (256 chunks of 64 bytes: 0xFF bytes)
    DevBuf e_S, e_base, e_ft;      // one-pass coder: tile streams, frame-relative tile bit offsets, first tile per 16 KB of output
    DevBuf e_status;               // per-frame error flags of the device-resident entropy path: zero between calls (cleared by their consumer)
    uint8_t* e_pinned = nullptr;   // pinned host staging of the stuffed streams
    size_t e_pinned_cap = 0;
    DevBuf h_scan, h_U, h_cnt, h_off, h_state, h_setup, h_small, h_dc;   // GPU Huffman decoder (jpezy_huffdec.hip)
    int h_last_passes = 0;         // synchronisation passes of the last jpezy_read_jpeg_gpu (0: the host decoder was used)
    size_t h_min_bytes = 64 << 10;    // scans shorter than this are decoded on the host: the GPU path has ~0.6 ms of fixed cost, the host decoder
                                      // takes ~11 us per KiB of a dense scan (tools/huffdec_threshold.py: they cross at 56 KiB; round 2: 3 ms, 256 KiB)
    static constexpr int B_DEPTH = 3;   // slices of jpezy_decode_jpeg_batch whose planes may be on their way to the host while the next one is decoded
    DevBuf b_scan, b_U, b_cnt, b_rb, b_state, b_prop, b_meta, b_coef, b_planes[B_DEPTH];   // jpezy_decode_jpeg_batch, batch form of the Huffman decoder
    int b_last_fast = 0;           // files of the last jpezy_decode_jpeg_batch call that took the batch form (diagnostic hook)
    uint8_t* b_pin = nullptr;      // pinned staging of the concatenated scans
    size_t b_pin_cap = 0;
    uint8_t* b_stage[B_DEPTH] = {};               // pinned staging of a slice's planes (one download per slice)
    size_t b_stage_cap[B_DEPTH] = {};
    DevBuf e_hdr;                  // JFIF header bytes of the device-resident variant (cached per W, H, comment)
    jpezy_host::HostPipe pipe;     // staging ring of the streaming host-buffer entry points (jpezy_hostpipe.h)
    size_t host_chunk_bytes = 4u << 20;   // bytes of input per chunk of that pipeline (jpezy_ctx_set_host_chunk_bytes)
    bool is_batch_child = false;       // a worker of jpezy_decode_jpeg_batch: only ever driven on its own stream
    std::vector<jpezy_ctx*> workers;   // jpezy_decode_jpeg_batch: one child context (stream, buffers, tables) per file in flight
    uint8_t e_hdr_host[1024];
    size_t e_hdr_len = 0;
};

extern "C" {

const char* jpezy_hip_last_error(void) { return g_err.c_str(); }

int jpezy_hip_is_experimental_build(void)
{
#ifdef JPEZY_EXPERIMENT
    return 1;
#else
    return 0;
#endif
}

int jpezy_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int jpezy_mcu_cols(int W) { return (W / 16) + ((W % 16) ? 1 : 0); }
int jpezy_mcu_rows(int H) { return (H / 16) + ((H % 16) ? 1 : 0); }
size_t jpezy_coeff_count(int W, int H, int gray)
{
    if (W <= 0 || H <= 0) return 0;
    return (size_t)jpezy_mcu_cols(W) * (size_t)jpezy_mcu_rows(H) * (gray ? 4 : 6) * 64;
}

jpezy_ctx* jpezy_ctx_create(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_err(JPEZY_E_NODEVICE, "no HIP device (the jpezy hot path has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= n) {
        set_err(JPEZY_E_NODEVICE, "device index out of range");
        return nullptr;
    }
    jpezy_ctx* c = new (std::nothrow) jpezy_ctx;
    if (!c) {
        set_err(JPEZY_E_NOSPACE, "out of host memory");
        return nullptr;
    }
    c->device = device;
    std::vector<DeviceTables> hbuf(1); // 33 KB: off the stack, and private to this call (contexts may be created concurrently)
    DeviceTables& h = hbuf[0];
    const double S = JPEZY_INV_SQRT2;
    int pos_of_row[8];                 // inverse of kPairRow: where the f32 kernel keeps coefficient row i of a block column
    for (int pp = 0; pp < 8; ++pp) pos_of_row[kPairRow[pp]] = pp;
    for (int t = 0; t < 2; ++t) {
        for (int j = 0; j < 8; ++j)
            for (int i = 0; i < 8; ++i) {
                const double cu = j ? 1.0 : S, cv = i ? 1.0 : S;
                h.qscale[t][j][i] = cu * cv / (4.0 * kQt[t][i * 8 + j]) * (double)(1 << QFRAC_BITS);
            }
        h.rq_dc[t] = 1.0 / kQt[t][0];
        for (int k = 0; k < 64; ++k) {
            h.qt[t][k] = kQt[t][k];
            h.qinv[t][k] = 1.0 / kQt[t][k];
        }
        for (int j = 0; j < 8; ++j)
            for (int i = 0; i < 8; ++i) {
                const double cu = j ? 1.0 : S, cv = i ? 1.0 : S;
                // the f32 kernel's 8-point transform leaves output 4 without its factor cos(pi/4); it is applied here
                const double k4 = 0x1.6a09e667f3bcdp-1;      // cos(pi/4), correctly rounded
                h.f32col[t][j].ks[pos_of_row[i]] = (float)(cu * cv / (4.0 * kQt[t][i * 8 + j]) * (i == 4 ? k4 : 1.0) * (j == 4 ? k4 : 1.0));
            }
    }
    for (int t = 0; t < 2; ++t)
        for (int sum = -8192; sum <= 8192; ++sum) {
            const double cu = S, cv = S;
            const int dct = (int)((double)sum * cu * cv / 4);          // int(sum * cu * cv / 4), no contraction (build flag)
            h.dcq[t][sum + 8192] = (signed char)(dct / kQt[t][0]);
        }
    {
        // f32 kernel, level 1: |t_fp32 - t| <= gamma_13 * S_i * S_j * 128 * ks + 2^-23 * |t|max, S_u = sum_x |cos_u(x)|
        // (13 roundings at most on any input->output path of the two butterfly passes; ks rounded to FP32 and the
        // fused product-and-bias fma(F, ks, delta1) rounded once: 2 * 2^-24 relative to |t| + delta1 <= 128 * S_i * S_j * ks
        // -- the few 1e-12 that delta1 adds to that rounding disappear in the 25 % margin).  tests/test_f32_error_bound.py re-derives it.
        static const double kCos[64] = JPEZY_COS_INIT;
        double S1[8];
        for (int u = 0; u < 8; ++u) {
            S1[u] = 0;
            for (int x = 0; x < 8; ++x) S1[u] += std::fabs(kCos[u * 8 + x]);
        }
        for (int t = 0; t < 2; ++t)
            for (int j = 0; j < 8; ++j) {
                double worst = 0;
                for (int i = 0; i < 8; ++i) {
                    if (i == 0 && j == 0) continue;            // DC: exact table lookup, no guard band
                    const double cu = j ? 1.0 : S, cv = i ? 1.0 : S;
                    const double ks = cu * cv / (4.0 * kQt[t][i * 8 + j]);
                    const double amp = 128.0 * S1[i] * S1[j] * ks;
                    const double bound = 13.0 * 0x1p-24 * amp + 0x1p-23 * amp;
                    if (bound > worst) worst = bound;
                }
                const float d1 = (float)(1.25 * worst);
                h.f32col[t][j].delta1[0] = h.f32col[t][j].delta1[1] = d1;
                h.f32col[t][j].th = d1 + d1;               // exact: a doubling
            }
    }
    for (int j = 0; j < 8; ++j)
        for (int hh = 0; hh < 2; ++hh) {
            uint32_t w = 0;
            for (int k = 0; k < 4; ++k) w |= (uint32_t)(2 * kZzInv[kPairRow[4 * hh + k] * 8 + j]) << (8 * k);
            for (int t = 0; t < 2; ++t) (hh ? h.f32col[t][j].zz_hi : h.f32col[t][j].zz_lo) = w;
        }
    bool ok = hipSetDevice(device) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_tab, sizeof(DeviceTables)) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_counter, sizeof(unsigned long long) * COUNTER_SHARDS) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_dqscale, sizeof(double) * 3 * 64) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_dqt, sizeof(int) * 3 * 64) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_dqscale_f, sizeof(float) * 64) == hipSuccess;
    ok = ok && hipMemcpy(c->d_tab, &h, sizeof h, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemset(c->d_counter, 0, sizeof(unsigned long long) * COUNTER_SHARDS) == hipSuccess;
    if (!ok) {
        set_err(JPEZY_E_HIP, std::string("context creation failed: ") + hipGetErrorString(hipGetLastError()));
        jpezy_ctx_destroy(c);
        return nullptr;
    }
    return c;
}

void jpezy_ctx_destroy(jpezy_ctx* c)
{
    if (!c) return;
    for (jpezy_ctx* w : c->workers) jpezy_ctx_destroy(w);
    c->workers.clear();
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->pipe.release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->d_tab) (void)hipFree(c->d_tab);
    if (c->d_counter) (void)hipFree(c->d_counter);
    if (c->d_dqscale) (void)hipFree(c->d_dqscale);
    if (c->d_dqt) (void)hipFree(c->d_dqt);
    if (c->d_dqscale_f) (void)hipFree(c->d_dqscale_f);
    for (auto& b : c->in) b.release();
    c->out.release();
    c->scratch.release();
    if (c->d_codes) (void)hipFree(c->d_codes);
    if (c->e_pinned) (void)hipHostFree(c->e_pinned);
    if (c->b_pin) (void)hipHostFree(c->b_pin);
    for (uint8_t* q : c->b_stage) if (q) (void)hipHostFree(q);
    for (DevBuf* b : { &c->b_scan, &c->b_U, &c->b_cnt, &c->b_rb, &c->b_state, &c->b_prop, &c->b_meta, &c->b_coef }) b->release();
    for (DevBuf& b : c->b_planes) b.release();
    for (DevBuf* b : { &c->e_tmp, &c->e_small, &c->e_U, &c->e_cnt, &c->e_out, &c->e_coef, &c->e_hdr, &c->e_status, &c->e_tt, &c->e_fft, &c->e_S, &c->e_base, &c->e_ft,
                       &c->dump_t, &c->h_scan, &c->h_U, &c->h_cnt, &c->h_off, &c->h_state, &c->h_setup, &c->h_small, &c->h_dc }) b->release();
    delete c;
}

int jpezy_ctx_sync(jpezy_ctx* c)
{
    if (!c) return set_err(JPEZY_E_BADARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return JPEZY_OK;
}

int jpezy_ctx_device(const jpezy_ctx* c) { return c ? c->device : -1; }
void* jpezy_ctx_stream(const jpezy_ctx* c) { return c ? (void*)c->stream : nullptr; }

void jpezy_ctx_set_force_exact(jpezy_ctx* c, int on)
{
    if (c) c->force_exact = on < 0 ? 0 : on > 3 ? 3 : on;
}

int jpezy_ctx_set_decode_tolerance(jpezy_ctx* c, int on)
{
    if (!c) return set_err(JPEZY_E_BADARG, "null context");
    if (on != 0 && on != 1) return set_err(JPEZY_E_BADARG, "decode tolerance: 0 (bit-exact) or 1 (within one per channel)");
    c->dec_tolerance = on;
    return JPEZY_OK;
}

int jpezy_ctx_set_variant(jpezy_ctx* c, int variant)
{
    if (!c) return set_err(JPEZY_E_BADARG, "null context");
    if (variant < 0 || variant > 1) return set_err(JPEZY_E_BADARG, "unknown kernel variant");
    c->variant = variant;
    return JPEZY_OK;
}

long jpezy_ctx_last_fallback_count(jpezy_ctx* c)
{
    if (!c) return -1;
    static thread_local unsigned long long shards[COUNTER_SHARDS];
    if (hipSetDevice(c->device) != hipSuccess) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(shards, c->d_counter, sizeof shards, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (hipMemset(c->d_counter, 0, sizeof shards) != hipSuccess) return -1;
    unsigned long long v = 0;
    for (unsigned long long s : shards) v += s;
    return (long)v;
}

static int check_dims(const jpezy_ctx* c, int W, int H, int n_frames)
{
    if (!c) return set_err(JPEZY_E_BADARG, "null context");
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return set_err(JPEZY_E_BADARG, "width/height must be in 1..65535 (16-bit SOF0 fields)");
    if (n_frames <= 0) return set_err(JPEZY_E_BADARG, "n_frames must be positive");
    return JPEZY_OK;
}

int jpezy_fdct_quant_dev(jpezy_ctx* c, const uint8_t* d_r, const uint8_t* d_g, const uint8_t* d_b,
                         size_t plane_stride, int W, int H, int gray, int n_frames, int16_t* d_coeffs, void* stream)
{
    if (int rc = check_dims(c, W, H, n_frames)) return rc;
    if (!d_r || !d_g || !d_b || !d_coeffs) return set_err(JPEZY_E_BADARG, "null device pointer");
    if (!aligned16(d_coeffs)) return set_err(JPEZY_E_BADARG, "d_coeffs must be 16-byte aligned");
    if (plane_stride < (size_t)W * H) return set_err(JPEZY_E_BADARG, "plane_stride smaller than W*H");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    EncParams p;
    p.r = d_r; p.g = d_g; p.b = d_b;
    p.plane_stride = plane_stride;
    p.coeffs = d_coeffs;
    p.coeffs_per_frame = jpezy_coeff_count(W, H, gray);
    p.tab = c->d_tab;
    p.dcq_luma = c->d_tab->dcq[0];       // address arithmetic only: d_tab is a device pointer
    p.dcq_chroma = c->d_tab->dcq[1];
    p.fallback_count = c->d_counter;
#ifdef JPEZY_TRACE
    if (!c->d_trace) HIP_TRY(hipMalloc((void**)&c->d_trace, sizeof(unsigned long long) * 13 * 65536));   // 4 words per wave + 9 phase stamps (JPEZY_TRACE=3)
    p.trace = c->d_trace;
#endif
#ifdef JPEZY_DUMP_T
    if (int rc = c->dump_t.reserve(p.coeffs_per_frame * (size_t)n_frames * sizeof(float))) return rc;
    HIP_TRY(hipMemsetAsync(c->dump_t.p, 0, p.coeffs_per_frame * (size_t)n_frames * sizeof(float), s));
    p.dump_t = (float*)c->dump_t.p;
#endif
    p.W = W; p.H = H;
    p.mcu_cols = jpezy_mcu_cols(W);
    p.mcu_rows = jpezy_mcu_rows(H);
    p.quads_per_row = (p.mcu_cols + 3) / 4;
    p.n_frames = n_frames;
    fast_div_setup((unsigned)p.quads_per_row, &p.qpr_magic, &p.qpr_shift);
    // the f32 kernel puts the frame index in grid.y (at most 65535): larger batches go out in chunks
    constexpr int kMaxFramesPerLaunch = 65535;
    for (int f0 = 0; f0 < n_frames; f0 += kMaxFramesPerLaunch) {
        EncParams q = p;
        q.n_frames = n_frames - f0 < kMaxFramesPerLaunch ? n_frames - f0 : kMaxFramesPerLaunch;
        q.r += (size_t)f0 * plane_stride; q.g += (size_t)f0 * plane_stride; q.b += (size_t)f0 * plane_stride;
        q.coeffs += (size_t)f0 * p.coeffs_per_frame;
        if (c->variant == 1)
            HIP_TRY(launch_fdct_quant_f32(q, gray != 0, c->force_exact, s));
        else
            HIP_TRY(launch_fdct_quant(q, gray != 0, c->force_exact != 0, s));
    }
    return JPEZY_OK;
}

#ifdef JPEZY_DUMP_T
int jpezy_debug_read_t(jpezy_ctx* c, float* host, size_t n)
{
    HIP_TRY(hipDeviceSynchronize());
    if (n * sizeof(float) > c->dump_t.cap) return set_err(JPEZY_E_BADARG, "debug_read_t: more than the last call dumped");
    HIP_TRY(hipMemcpy(host, c->dump_t.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return JPEZY_OK;
}
#endif

#ifdef JPEZY_TRACE
int jpezy_debug_read_trace(jpezy_ctx* c, unsigned long long* host, size_t n)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host, c->d_trace, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return JPEZY_OK;
}
#endif

void jpezy_ctx_set_host_chunk_bytes(jpezy_ctx* c, size_t n)
{
    if (c) c->host_chunk_bytes = n < 4096 ? 4096 : n;
}

int jpezy_fdct_quant(jpezy_ctx* c, const uint8_t* r, const uint8_t* g, const uint8_t* b, int W, int H, int gray,
                     int n_frames, int16_t* coeffs)
try {
    if (int rc = check_dims(c, W, H, n_frames)) return rc;
    if (!r || !g || !b || !coeffs) return set_err(JPEZY_E_BADARG, "null host pointer");
    HIP_TRY(hipSetDevice(c->device));
    const size_t plane = (size_t)W * H;
    const int mcu_cols = jpezy_mcu_cols(W), B = gray ? 4 : 6;
    const size_t ncoef = jpezy_coeff_count(W, H, gray);
    const std::vector<HostChunk> chunks = plan_host_chunks(W, H, n_frames, 3, c->host_chunk_bytes);
    // a chunk's planes sit one behind the other in its slot, P bytes apart (a multiple of 16: the aligned kernel stays usable)
    size_t P = 0, max_out = 0;
    for (const HostChunk& k : chunks) {
        const size_t rows = (size_t)std::min(H - k.y0 * 16, (k.y1 - k.y0) * 16);
        P = std::max(P, k.nf > 1 || k.y1 - k.y0 == jpezy_mcu_rows(H) ? plane * k.nf : rows * W);
        max_out = std::max(max_out, (size_t)k.nf * (k.y1 - k.y0) * mcu_cols * B * 128);
    }
    P = (P + 15) & ~(size_t)15;
    const uint8_t* src[3] = { r, g, b };
    int rc_kernel = JPEZY_OK;
    std::string err;
    auto plan = [&](int i) {
        const HostChunk& k = chunks[(size_t)i];
        jpezy_host::ChunkPlan p;
        const size_t rows = (size_t)std::min(H - k.y0 * 16, (k.y1 - k.y0) * 16);
        const size_t bytes = k.nf > 1 ? plane * k.nf : rows * W, off = (size_t)k.f0 * plane + (size_t)k.y0 * 16 * W;
        for (int q = 0; q < 3; ++q) p.in.push_back({ const_cast<uint8_t*>(src[q]) + off, bytes, (size_t)q * P });
        p.out.push_back({ coeffs + (size_t)k.f0 * ncoef + (size_t)k.y0 * mcu_cols * B * 64, (size_t)k.nf * (k.y1 - k.y0) * mcu_cols * B * 128, 0 });
        return p;
    };
    auto kernel = [&](int i, uint8_t* d_in, uint8_t* d_out, hipStream_t s) -> hipError_t {
        const HostChunk& k = chunks[(size_t)i];
        const int Hc = std::min(H - k.y0 * 16, (k.y1 - k.y0) * 16);
        const int rc = jpezy_fdct_quant_dev(c, d_in, d_in + P, d_in + 2 * P, k.nf > 1 ? plane : (size_t)Hc * W, W, Hc, gray, k.nf, (int16_t*)d_out, s);
        if (rc != JPEZY_OK) { rc_kernel = rc; return hipErrorLaunchFailure; }
        return hipSuccess;
    };
    const hipError_t e = c->pipe.run(c->device, c->stream, (int)chunks.size(), 3 * P, max_out, plan, kernel, &err);
    if (rc_kernel != JPEZY_OK) return rc_kernel;                    // message set by jpezy_fdct_quant_dev
    if (e != hipSuccess) return set_err(JPEZY_E_HIP, err.empty() ? std::string("host pipeline: ") + hipGetErrorString(e) : err);
    return JPEZY_OK;
}
JPEZY_CATCH

static int upload_dequant(jpezy_ctx* c, const uint16_t qt[4][64], const uint8_t comp_tq[3], hipStream_t s)
{
    uint16_t sel[3][64];
    for (int k = 0; k < 3; ++k) std::memcpy(sel[k], qt[comp_tq[k] & 3], sizeof sel[k]);
    if (c->dq_valid && !std::memcmp(sel, c->dq_cache, sizeof sel)) return JPEZY_OK;
    // A previous launch -- on this or another stream -- may still be reading the tables.
    if (int rc = drain_before_table_rewrite(s, "dequant_idct", c->is_batch_child && s == c->stream)) return rc;
    static thread_local double h_scale[3][8][8];
    static thread_local int h_qt[3][64];
    const double S = JPEZY_INV_SQRT2;
    for (int k = 0; k < 3; ++k)
        for (int u = 0; u < 8; ++u)
            for (int v = 0; v < 8; ++v) {
                const double cu = u ? 1.0 : S, cv = v ? 1.0 : S;
                // the reference's final / 4 (ref :667) is folded in here: an exact scaling of every intermediate value
                h_scale[k][u][v] = cu * cv * (double)sel[k][v * 8 + u] * 0.25;
                h_qt[k][v * 8 + u] = sel[k][v * 8 + u];
            }
    static thread_local float h_scale_f[8][8];
    for (int u = 0; u < 8; ++u)
        for (int v = 0; v < 8; ++v) h_scale_f[u][v] = (float)h_scale[0][u][v];
    HIP_TRY(hipMemcpy(c->d_dqscale, h_scale, sizeof h_scale, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_dqscale_f, h_scale_f, sizeof h_scale_f, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_dqt, h_qt, sizeof h_qt, hipMemcpyHostToDevice));
    std::memcpy(c->dq_cache, sel, sizeof sel);
    int qmax = 1;
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 64; ++i) qmax = sel[k][i] > qmax ? sel[k][i] : qmax;
    c->coef_limit = 32768 / qmax;
    // Fused kernel, exact mode: with |c * Q| <= 2^23 every dequantised input is below 2^21 and the two FP64 butterfly passes err
    // by at most (8 * 6 * 8 + 6 * 64) * 2^21 * 2^-53 = 1.8e-7.  The value they must truncate like -- the reference's own
    // reference-order sum of 64 such terms -- carries a rounding error of its own, at most 64 partial sums of up to 2^27 each
    // rounded to 2^-53 relative and divided by 4: 64 * 2^27 * 2^-53 / 4 = 2.4e-7 ... 1e-6 by the coarsest count.  Together
    // at most 1.2e-6 against the 2^-18 = 3.8e-6 guard band: a margin of about three (not twenty, as this comment used to say).
    // Every 8-bit quantiser table gives a limit >= 32768: no int16 coefficient can exceed it and the kernel without the range
    // test is launched (tests/test_gpu_tolerance.py::test_all_coefficients_at_the_int16_extremes_with_q255).
    c->coef_limit_exact = (1 << 23) / qmax;
    c->dq_valid = true;
    return JPEZY_OK;
}

int jpezy_dequant_idct_dev(jpezy_ctx* c, const int16_t* d_coeffs, const uint16_t qt[4][64], const uint8_t comp_tq[3],
                           size_t plane_stride, int W, int H, int gray, int n_frames, uint8_t* d_r, uint8_t* d_g,
                           uint8_t* d_b, void* stream)
{
    if (int rc = check_dims(c, W, H, n_frames)) return rc;
    if (!d_coeffs || !qt || !comp_tq || !d_r || !d_g || !d_b) return set_err(JPEZY_E_BADARG, "null pointer");
    if (!aligned16(d_coeffs)) return set_err(JPEZY_E_BADARG, "d_coeffs must be 16-byte aligned");
    if (plane_stride < (size_t)W * H) return set_err(JPEZY_E_BADARG, "plane_stride smaller than W*H");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    if (int rc = upload_dequant(c, qt, comp_tq, s)) return rc;
    DecParams p;
    p.coeffs = d_coeffs;
    p.coeffs_per_frame = jpezy_coeff_count(W, H, 0);
    p.r = d_r; p.g = d_g; p.b = d_b;
    p.plane_stride = plane_stride;
    p.dqscale = c->d_dqscale;
    p.dqscale_f = c->d_dqscale_f;
    p.dqt = c->d_dqt;
    p.coef_limit = c->dec_tolerance ? c->coef_limit : c->coef_limit_exact;
    p.fallback_count = c->d_counter;
    p.W = W; p.H = H;
    p.mcu_cols = jpezy_mcu_cols(W);
    p.mcu_rows = jpezy_mcu_rows(H);
    p.quads_per_row = (p.mcu_cols + 3) / 4;
    p.n_frames = n_frames;
    constexpr int kMaxFramesPerLaunch = 65535;           // grid.y
    for (int f0 = 0; f0 < n_frames; f0 += kMaxFramesPerLaunch) {
        DecParams q = p;
        q.n_frames = n_frames - f0 < kMaxFramesPerLaunch ? n_frames - f0 : kMaxFramesPerLaunch;
        q.coeffs += (size_t)f0 * p.coeffs_per_frame;
        q.r += (size_t)f0 * plane_stride; q.g += (size_t)f0 * plane_stride; q.b += (size_t)f0 * plane_stride;
        HIP_TRY(launch_dequant_idct(q, gray != 0, c->force_exact != 0, c->dec_tolerance != 0, s));
    }
    return JPEZY_OK;
}

int jpezy_dequant_idct(jpezy_ctx* c, const int16_t* coeffs, const uint16_t qt[4][64], const uint8_t comp_tq[3], int W,
                       int H, int gray, int n_frames, uint8_t* r, uint8_t* g, uint8_t* b)
try {
    if (int rc = check_dims(c, W, H, n_frames)) return rc;
    if (!coeffs || !qt || !comp_tq || !r || !g || !b) return set_err(JPEZY_E_BADARG, "null pointer");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = upload_dequant(c, qt, comp_tq, c->stream)) return rc;      // tables first: never rewritten while chunks are in flight
    const size_t plane = (size_t)W * H;
    const int mcu_cols = jpezy_mcu_cols(W);
    const size_t ncoef = jpezy_coeff_count(W, H, 0);
    const std::vector<HostChunk> chunks = plan_host_chunks(W, H, n_frames, 3, c->host_chunk_bytes);
    size_t P = 0, max_in = 0;
    for (const HostChunk& k : chunks) {
        const size_t rows = (size_t)std::min(H - k.y0 * 16, (k.y1 - k.y0) * 16);
        P = std::max(P, k.nf > 1 || k.y1 - k.y0 == jpezy_mcu_rows(H) ? plane * k.nf : rows * W);
        max_in = std::max(max_in, (size_t)k.nf * (k.y1 - k.y0) * mcu_cols * 6 * 128);
    }
    P = (P + 15) & ~(size_t)15;
    uint8_t* dst[3] = { r, g, b };
    int rc_kernel = JPEZY_OK;
    std::string err;
    auto plan = [&](int i) {
        const HostChunk& k = chunks[(size_t)i];
        jpezy_host::ChunkPlan p;
        const size_t rows = (size_t)std::min(H - k.y0 * 16, (k.y1 - k.y0) * 16);
        const size_t bytes = k.nf > 1 ? plane * k.nf : rows * W, off = (size_t)k.f0 * plane + (size_t)k.y0 * 16 * W;
        p.in.push_back({ const_cast<int16_t*>(coeffs) + (size_t)k.f0 * ncoef + (size_t)k.y0 * mcu_cols * 6 * 64,
                         (size_t)k.nf * (k.y1 - k.y0) * mcu_cols * 6 * 128, 0 });
        for (int q = 0; q < 3; ++q) p.out.push_back({ dst[q] + off, bytes, (size_t)q * P });
        return p;
    };
    auto kernel = [&](int i, uint8_t* d_in, uint8_t* d_out, hipStream_t s) -> hipError_t {
        const HostChunk& k = chunks[(size_t)i];
        const int Hc = std::min(H - k.y0 * 16, (k.y1 - k.y0) * 16);
        const int rc = jpezy_dequant_idct_dev(c, (const int16_t*)d_in, qt, comp_tq, k.nf > 1 ? plane : (size_t)Hc * W, W, Hc, gray, k.nf, d_out,
                                              d_out + P, d_out + 2 * P, s);
        if (rc != JPEZY_OK) { rc_kernel = rc; return hipErrorLaunchFailure; }
        return hipSuccess;
    };
    const hipError_t e = c->pipe.run(c->device, c->stream, (int)chunks.size(), max_in, 3 * P, plan, kernel, &err);
    if (rc_kernel != JPEZY_OK) return rc_kernel;
    if (e != hipSuccess) return set_err(JPEZY_E_HIP, err.empty() ? std::string("host pipeline: ") + hipGetErrorString(e) : err);
    return JPEZY_OK;
}
JPEZY_CATCH

// geometry + tables + the two launches of the any-layout decoder on device memory; asynchronous on stream s (the tables are
// uploaded synchronously when they changed since the last call)
static int generic_dev_core(jpezy_ctx* c, const int16_t* d_coeffs, const uint16_t qt[4][64], int ncomp, const uint8_t comp_h[3],
                            const uint8_t comp_v[3], const uint8_t comp_tq[3], int W, int H, int gray, int precision, uint8_t* d_r,
                            uint8_t* d_g, uint8_t* d_b, hipStream_t s, size_t* nblk_out, int n_frames = 1, size_t plane_stride = 0)
{
    if (ncomp != 1 && ncomp != 3) return set_err(JPEZY_E_UNSUPPORTED, "dimension not supported (the reference accepts 1 or 3)");
    GenericDecParams p;
    p.W = W; p.H = H; p.ncomp = ncomp; p.gray = gray != 0;
    p.level = precision == 8 ? 128 : 2048;                    // ref :654
    p.hmax = p.vmax = 0;
    p.blocks_per_mcu = 0;
    for (int k = 0; k < 3; ++k) { p.ch[k] = p.cv[k] = 1; p.blk_start[k] = 1 << 20; }
    for (int k = 0; k < ncomp; ++k) {
        p.ch[k] = comp_h[k]; p.cv[k] = comp_v[k];
        if (p.ch[k] < 1 || p.ch[k] > 4 || p.cv[k] < 1 || p.cv[k] > 4)
            return set_err(JPEZY_E_UNSUPPORTED, "sampling factors outside 1..4 (ITU-T T.81 B.2.2)");
        p.hmax = p.ch[k] > p.hmax ? p.ch[k] : p.hmax;
        p.vmax = p.cv[k] > p.vmax ? p.cv[k] : p.vmax;
        p.blk_start[k] = p.blocks_per_mcu;
        p.blocks_per_mcu += p.ch[k] * p.cv[k];
    }
    const int Hblock = (W >> 3) + ((W & 7) > 0), Vblock = (H >> 3) + ((H & 7) > 0);   // get_blocks, ref :166-169
    p.mcu_cols = Hblock / p.hmax + ((Hblock % p.hmax) ? 1 : 0);
    p.mcu_rows = Vblock / p.vmax + ((Vblock % p.vmax) ? 1 : 0);
    const size_t nblk = (size_t)p.mcu_cols * p.mcu_rows * p.blocks_per_mcu;
    if (nblk_out) *nblk_out = nblk;
    if (!d_coeffs) return JPEZY_OK;                            // geometry only
    if (n_frames < 1 || (n_frames > 1 && (plane_stride < (size_t)W * H || (plane_stride & 3))))
        return set_err(JPEZY_E_BADARG, "generic decoder, batch form: plane stride must hold a plane and be a multiple of 4");
    if (int rc = c->scratch.reserve(nblk * 64 * sizeof(int) * (size_t)n_frames)) return rc;
    p.n_frames = n_frames;
    p.plane_stride = plane_stride;
    // per-component dequantiser constants (fast path) and integer quantisers (reference-order path), cached in the context
    const uint8_t tq3[3] = { comp_tq[0], (uint8_t)(ncomp > 1 ? comp_tq[1] : 0), (uint8_t)(ncomp > 2 ? comp_tq[2] : 0) };
    if (int rc = upload_dequant(c, qt, tq3, s)) return rc;
    p.coeffs = d_coeffs;
    p.samples = (int*)c->scratch.p;
    p.qt = c->d_dqt;
    p.dqscale = c->d_dqscale;
    p.coef_limit = c->coef_limit;
    p.force_exact = c->force_exact != 0;
    p.fallback_count = c->d_counter;
    p.r = d_r; p.g = d_g; p.b = d_b;
    HIP_TRY(launch_dequant_idct_generic(p, s));
    return JPEZY_OK;
}

int jpezy_dequant_idct_generic_dev(jpezy_ctx* c, const int16_t* d_coeffs, const uint16_t qt[4][64], int ncomp, const uint8_t comp_h[3],
                                   const uint8_t comp_v[3], const uint8_t comp_tq[3], int precision, int W, int H, int gray,
                                   uint8_t* d_r, uint8_t* d_g, uint8_t* d_b, void* stream)
{
    if (int rc = check_dims(c, W, H, 1)) return rc;
    if (!d_coeffs || !qt || !comp_h || !comp_v || !comp_tq || !d_r || !d_g || !d_b) return set_err(JPEZY_E_BADARG, "null pointer");
    if (!aligned16(d_coeffs)) return set_err(JPEZY_E_BADARG, "d_coeffs must be 16-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    return generic_dev_core(c, d_coeffs, qt, ncomp, comp_h, comp_v, comp_tq, W, H, gray, precision, d_r, d_g, d_b, (hipStream_t)stream,
                            nullptr);
}

int jpezy_dequant_idct_generic_batch_dev(jpezy_ctx* c, const int16_t* d_coeffs, const uint16_t qt[4][64], int ncomp, const uint8_t comp_h[3],
                                         const uint8_t comp_v[3], const uint8_t comp_tq[3], int precision, int W, int H, int gray,
                                         int n_frames, size_t plane_stride, uint8_t* d_r, uint8_t* d_g, uint8_t* d_b, void* stream)
{
    if (int rc = check_dims(c, W, H, n_frames)) return rc;
    if (!d_coeffs || !qt || !comp_h || !comp_v || !comp_tq || !d_r || !d_g || !d_b) return set_err(JPEZY_E_BADARG, "null pointer");
    if (!aligned16(d_coeffs)) return set_err(JPEZY_E_BADARG, "d_coeffs must be 16-byte aligned");
    if (plane_stride < (size_t)W * H || (plane_stride & 3)) return set_err(JPEZY_E_BADARG, "plane_stride must hold a plane and be a multiple of 4");
    HIP_TRY(hipSetDevice(c->device));
    return generic_dev_core(c, d_coeffs, qt, ncomp, comp_h, comp_v, comp_tq, W, H, gray, precision, d_r, d_g, d_b, (hipStream_t)stream,
                            nullptr, n_frames, plane_stride);
}

static int dequant_idct_generic_impl(jpezy_ctx* c, const int16_t* coeffs, const uint16_t qt[4][64], int ncomp, const uint8_t comp_h[3],
                                     const uint8_t comp_v[3], const uint8_t comp_tq[3], int W, int H, int gray, int precision,
                                     uint8_t* r, uint8_t* g, uint8_t* b, bool coeffs_on_device = false)
{
    if (int rc = check_dims(c, W, H, 1)) return rc;
    if (!coeffs || !qt || !comp_h || !comp_v || !comp_tq || !r || !g || !b) return set_err(JPEZY_E_BADARG, "null pointer");
    HIP_TRY(hipSetDevice(c->device));
    size_t nblk = 0;
    if (int rc = generic_dev_core(c, nullptr, qt, ncomp, comp_h, comp_v, comp_tq, W, H, gray, precision, nullptr, nullptr, nullptr, c->stream,
                                  &nblk))
        return rc;
    const size_t plane = (size_t)W * H;
    for (int k = 0; k < 3; ++k)
        if (int rc = c->in[k].reserve(plane)) return rc;
    const int16_t* d_coeffs = coeffs;
    if (!coeffs_on_device) {
        if (int rc = c->out.reserve(nblk * 64 * sizeof(int16_t))) return rc;
        HIP_TRY(hipMemcpyAsync(c->out.p, coeffs, nblk * 64 * sizeof(int16_t), hipMemcpyHostToDevice, c->stream));
        d_coeffs = (const int16_t*)c->out.p;
    }
    if (int rc = generic_dev_core(c, d_coeffs, qt, ncomp, comp_h, comp_v, comp_tq, W, H, gray, precision, (uint8_t*)c->in[0].p,
                                  (uint8_t*)c->in[1].p, (uint8_t*)c->in[2].p, c->stream, nullptr))
        return rc;
    uint8_t* dst[3] = { r, g, b };
    for (int k = 0; k < 3; ++k) HIP_TRY(hipMemcpyAsync(dst[k], c->in[k].p, plane, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return JPEZY_OK;
}

int jpezy_dequant_idct_generic(jpezy_ctx* c, const int16_t* coeffs, const uint16_t qt[4][64], int ncomp, const uint8_t comp_h[3],
                               const uint8_t comp_v[3], const uint8_t comp_tq[3], int W, int H, int gray, uint8_t* r, uint8_t* g,
                               uint8_t* b)
{
    return dequant_idct_generic_impl(c, coeffs, qt, ncomp, comp_h, comp_v, comp_tq, W, H, gray, 8, r, g, b);
}

long jpezy_write_jpeg(const int16_t* coeffs, int W, int H, int gray, const char* comment, uint8_t* out, size_t cap)
try {
    std::string err;
    const long n = jpezy_host::write_jpeg(coeffs, W, H, gray != 0, comment, out, cap, &err);
    if (n < 0) g_err = err;
    return n;
}
JPEZY_CATCH

size_t jpezy_jpeg_bound(int W, int H) { return jpezy_host::jpeg_bound(W, H); }

int jpezy_write_jpeg_batch(const int16_t* coeffs, int W, int H, int gray, int n_frames, const char* comment, uint8_t* out,
                           size_t cap, long* sizes, int threads)
try {
    if (!coeffs || !out || !sizes || n_frames <= 0) return set_err(JPEZY_E_BADARG, "write_jpeg_batch: bad argument");
    const size_t cpf = jpezy_coeff_count(W, H, gray);
    if (!cpf) return set_err(JPEZY_E_BADARG, "write_jpeg_batch: bad dimensions");
    unsigned nt = threads > 0 ? (unsigned)threads : std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > (unsigned)n_frames) nt = (unsigned)n_frames;
    std::atomic<int> next{ 0 };
    std::atomic<int> failed{ 0 };
    auto work = [&]() {
        for (int f = next.fetch_add(1); f < n_frames; f = next.fetch_add(1)) {
            sizes[f] = jpezy_host::write_jpeg(coeffs + (size_t)f * cpf, W, H, gray != 0, comment, out + (size_t)f * cap, cap, nullptr);
            if (sizes[f] < 0) failed.store(1);
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    return failed.load() ? set_err(JPEZY_E_FORMAT, "write_jpeg_batch: at least one frame failed (see sizes[])") : JPEZY_OK;
}
JPEZY_CATCH

// ---- GPU entropy coding: same bytes as jpezy_write_jpeg, coefficients already on the device ----
namespace {

int ensure_code_tables(jpezy_ctx* c)
{
    if (c->d_codes) return JPEZY_OK;
    uint16_t code[4][256];
    uint8_t len[4][256];
    jpezy_host::enc_code_tables(code, len);
    std::vector<jpezy_dev::entropy::CodeTables> hv(1);       // 10 KB: off the stack
    jpezy_dev::entropy::CodeTables& h = hv[0];
    std::memset(&h, 0, sizeof h);
    for (int t = 0; t < 2; ++t) {      // DHT order: YDc, CDc, YAc, CAc
        for (int k = 0; k < 12; ++k) h.dc[t][k] = ((uint32_t)code[t][k] << 8) | len[t][k];
        for (int k = 0; k < 256; ++k) h.ac[t][k] = ((uint32_t)code[2 + t][k] << 8) | len[2 + t][k];
        for (int run = 0; run < 16; ++run)
            for (int v = -32; v < 32; ++v) {
                if (v == 0) continue;
                const int a = v < 0 ? -v : v;
                int sz = 0;
                while ((a >> sz) != 0) ++sz;
                const int k = (run << 4) | sz;
                const uint32_t bits = ((uint32_t)code[2 + t][k] << sz) | ((uint32_t)(v + (v >> 31)) & ((1u << sz) - 1u));
                h.fast[t][(run << 6) | (v + 32)] = (bits << 5) | (uint32_t)(len[2 + t][k] + sz);
            }
    }
    HIP_TRY(hipMalloc((void**)&c->d_codes, sizeof h));
    HIP_TRY(hipMemcpy(c->d_codes, &h, sizeof h, hipMemcpyHostToDevice));
    return JPEZY_OK;
}

// one chunk of frames, all resident in the scratch buffers
int entropy_chunk(jpezy_ctx* c, const int16_t* d_coeffs, int W, int H, int gray, int F, const char* comment, uint8_t* out,
                  size_t cap, long* sizes, bool* any_failed)
{
    namespace E = jpezy_dev::entropy;
    hipStream_t s = c->stream;
    const size_t nmcu = (size_t)jpezy_mcu_cols(W) * jpezy_mcu_rows(H);
    const size_t nblk = nmcu * 6;
    E::Job job;
    job.coeffs = d_coeffs;
    job.coeffs_per_frame = jpezy_coeff_count(W, H, gray);
    job.tables = c->d_codes;
    job.blocks_per_frame = (unsigned)nblk;
    job.bpm = gray ? 4 : 6;
    job.n_frames = F;

    // every block is coded once, into the stream of its tile (256 coded blocks of a frame); worst case 208 bytes per block
    const size_t tpf = E::tiles256(nblk), nt = tpf * (size_t)F, piece = E::assemble_piece_bytes(), chunk = E::chunk_bytes();
    const size_t u_stride = (nblk * 208 + 8 + piece - 1) / piece * piece, ft_stride = u_stride / piece;
    const size_t nchunks = u_stride / chunk * F;
    const bool self = E::assemble_scans_tiles_itself(tpf);
    if (int rc = c->e_tt.reserve(nt * sizeof(uint32_t))) return rc;                          // tile totals (bits)
    if (int rc = c->e_S.reserve(nt * E::tile_stream_bytes())) return rc;                     // tile streams
    if (int rc = c->e_U.reserve(u_stride * F)) return rc;                                    // unstuffed streams
    if (int rc = c->e_cnt.reserve(nchunks * sizeof(uint32_t))) return rc;                    // 0xFF bytes: per chunk inside its piece,
    if (int rc = c->e_fft.reserve(ft_stride * F * sizeof(uint32_t))) return rc;              //             per piece
    if (!self) {
        if (int rc = c->e_base.reserve((tpf + 1) * F * sizeof(unsigned long long))) return rc;   // frame-relative tile offsets
        if (int rc = c->e_ft.reserve(ft_stride * F * sizeof(uint32_t))) return rc;
    }
    // small arrays: [F] status u32 | [F] (unused) | [F] stream bytes | [F] 0xFF totals
    const size_t small_words = (size_t)F * 8;
    if (int rc = c->e_small.reserve(small_words * sizeof(unsigned long long))) return rc;
    unsigned* d_status = (unsigned*)c->e_small.p;
    unsigned long long* d_bytes = (unsigned long long*)c->e_small.p + 2 * F;
    unsigned long long* d_fftot = d_bytes + F;

    // 1. codes; 2. unstuffed streams, one per frame, with their 0xFF bytes counted; stream lengths
    HIP_TRY(hipMemsetAsync(d_status, 0, sizeof(unsigned) * F, s));
    HIP_TRY(E::launch_code_tiles(job, (uint32_t*)c->e_S.p, (uint32_t*)c->e_tt.p, d_status, s));
    if (!self)
        HIP_TRY(E::launch_tile_bases((const uint32_t*)c->e_tt.p, (unsigned)tpf, F, (unsigned long long*)c->e_base.p, d_bytes,
                                     (uint32_t*)c->e_ft.p, (unsigned)ft_stride, d_status, nullptr, s));
    HIP_TRY(E::launch_assemble((const uint32_t*)c->e_S.p, (const uint32_t*)c->e_tt.p, (const unsigned long long*)c->e_base.p, d_bytes,
                               (const uint32_t*)c->e_ft.p, (unsigned)ft_stride, (unsigned)tpf, F, (uint32_t*)c->e_U.p, u_stride / 4,
                               (uint32_t*)c->e_cnt.p, (uint32_t*)c->e_fft.p, d_status, nullptr, s));
    HIP_TRY(E::launch_ff_frame_totals((const uint32_t*)c->e_fft.p, d_bytes, u_stride / 4, F, d_fftot, s));
    std::vector<unsigned long long> nbytes(F), fftot(F);
    std::vector<unsigned> status(F);
    HIP_TRY(hipMemcpyAsync(nbytes.data(), d_bytes, sizeof(unsigned long long) * F, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(status.data(), d_status, sizeof(unsigned) * F, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(fftot.data(), d_fftot, sizeof(unsigned long long) * F, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));

    // 3. byte stuffing into a buffer sized from the actual lengths
    unsigned long long max_out = 0;
    for (int f = 0; f < F; ++f)
        if (nbytes[f] + fftot[f] > max_out) max_out = nbytes[f] + fftot[f];
    const size_t o_stride = ((size_t)max_out + 2 + 63) / 64 * 64;
    if (int rc = c->e_out.reserve(o_stride * F)) return rc;
    HIP_TRY(E::launch_stuff((const uint32_t*)c->e_U.p, u_stride / 4, d_bytes, F, (const uint32_t*)c->e_cnt.p, (const uint32_t*)c->e_fft.p,
                            (uint8_t*)c->e_out.p, o_stride, E::FilePlan{}, s));

    // 4. header + entropy-coded segment + EOI into the caller's buffers.  One device-to-host copy of all streams into a
    //    pinned staging buffer (per-frame copies into pageable memory cost more than the kernels for small frames).
    if (c->e_pinned_cap < o_stride * F) {
        if (c->e_pinned) (void)hipHostFree(c->e_pinned);
        c->e_pinned = nullptr;
        c->e_pinned_cap = 0;
        HIP_TRY(hipHostMalloc((void**)&c->e_pinned, o_stride * F, hipHostMallocDefault));
        c->e_pinned_cap = o_stride * F;
    }
    HIP_TRY(hipMemcpyAsync(c->e_pinned, c->e_out.p, o_stride * F, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // (four threads when there is much to hand out: a core copies ~25 GB/s -- 256 frames of 1080p noise, 168 MB: 11.9 -> 7 ms per call)
    std::atomic<int> failed{ 0 };
    auto hand_out = [&](int f0, int step) {
        for (int f = f0; f < F; f += step) {
            uint8_t* dst = out + (size_t)f * cap;
            if (status[f]) { sizes[f] = JPEZY_E_FORMAT; failed.store(1); continue; }
            const size_t hdr = jpezy_host::write_header(W, H, comment, dst, cap);
            const size_t body = (size_t)(nbytes[f] + fftot[f]);
            if (!hdr || hdr + body + 2 > cap) { sizes[f] = JPEZY_E_NOSPACE; failed.store(1); continue; }
            std::memcpy(dst + hdr, c->e_pinned + (size_t)f * o_stride, body);
            dst[hdr + body] = 0xFF;
            dst[hdr + body + 1] = 0xD9;
            sizes[f] = (long)(hdr + body + 2);
        }
    };
    const int n_copy = o_stride * (size_t)F > ((size_t)8 << 20) && F >= 4 ? 4 : 1;
    std::vector<std::thread> helpers;
    for (int t = 1; t < n_copy; ++t) helpers.emplace_back(hand_out, t, n_copy);
    hand_out(0, n_copy);
    for (auto& h : helpers) h.join();
    if (failed.load()) *any_failed = true;
    return JPEZY_OK;
}

}  // namespace

// Device-resident, asynchronous variant: everything is enqueued on `stream`, nothing is copied to the host.
int jpezy_write_jpeg_gpu_dev(jpezy_ctx* c, const int16_t* d_coeffs, int W, int H, int gray, int n_frames, const char* comment,
                             uint8_t* d_out, size_t out_stride, long long* d_sizes, void* stream)
{
    namespace E = jpezy_dev::entropy;
    if (int rc = check_dims(c, W, H, n_frames)) return rc;
    if (!d_coeffs || !d_out || !d_sizes) return set_err(JPEZY_E_BADARG, "write_jpeg_gpu_dev: null pointer");
    if (!aligned16(d_coeffs)) return set_err(JPEZY_E_BADARG, "write_jpeg_gpu_dev: d_coeffs must be 16-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = ensure_code_tables(c)) return rc;
    hipStream_t s = (hipStream_t)stream;
    // header bytes: cached on the device per (W, H, comment) -- uploaded outside any capture on first use
    uint8_t hdr[1024];
    const size_t hdr_len = jpezy_host::write_header(W, H, comment, hdr, sizeof hdr);
    if (!hdr_len) return set_err(JPEZY_E_BADARG, "write_jpeg_gpu_dev: comment too long");
    if (c->e_hdr_len != hdr_len || std::memcmp(c->e_hdr_host, hdr, hdr_len)) {
        if (int rc = c->e_hdr.reserve(sizeof hdr)) return rc;
        if (int rc = drain_before_table_rewrite(s, "write_jpeg_gpu_dev")) return rc;   // an earlier launch (any stream) may still read the old header
        HIP_TRY(hipMemcpy(c->e_hdr.p, hdr, hdr_len, hipMemcpyHostToDevice));
        std::memcpy(c->e_hdr_host, hdr, hdr_len);
        c->e_hdr_len = hdr_len;
    }
    const size_t nmcu = (size_t)jpezy_mcu_cols(W) * jpezy_mcu_rows(H);
    const size_t nblk = nmcu * 6;
    const size_t chunk = E::chunk_bytes();
    // worst case per block: 64 x (16-bit code + 10 value bits) = 208 bytes; whole 16 KB pieces (one workgroup of the
    // assembling / stuffing kernels each)
    const size_t piece = E::assemble_piece_bytes();
    const size_t u_stride = (nblk * 208 + 8 + piece - 1) / piece * piece;
    // frames per pass: worst-case streams below ~1 GiB, and at most 65535 (the frame index is a grid dimension)
    const int per = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((size_t)n_frames, 65535), ((size_t)1 << 30) / u_stride));
    const size_t cpf = jpezy_coeff_count(W, H, gray);
    const size_t tpf = E::tiles256(nblk);                           // tiles of one frame (a tile never straddles frames)
    const bool self = E::assemble_scans_tiles_itself(tpf);
    for (int f0 = 0; f0 < n_frames; f0 += per) {
        const int F = std::min(per, n_frames - f0);
        const size_t nchunks = u_stride / chunk * F, nt = tpf * F, nct = E::tiles256(nchunks);
        E::Job job;
        job.coeffs = d_coeffs + (size_t)f0 * cpf;
        job.coeffs_per_frame = cpf;
        job.tables = c->d_codes;
        job.blocks_per_frame = (unsigned)nblk;
        job.bpm = gray ? 4 : 6;
        job.n_frames = F;
        if (int rc = c->e_tt.reserve(nt * sizeof(uint32_t))) return rc;
        if (int rc = c->e_S.reserve(nt * E::tile_stream_bytes())) return rc;
        if (!self) {
            if (int rc = c->e_base.reserve((tpf + 1) * F * sizeof(unsigned long long))) return rc;
            if (int rc = c->e_ft.reserve(u_stride / piece * F * sizeof(uint32_t))) return rc;
        }
        if (int rc = c->e_small.reserve((size_t)F * 8 * sizeof(unsigned long long))) return rc;
        if (int rc = c->e_U.reserve(u_stride * F)) return rc;
        if (int rc = c->e_cnt.reserve(nchunks * sizeof(uint32_t))) return rc;
        if (int rc = c->e_fft.reserve(nct * sizeof(uint32_t))) return rc;
        if (c->e_status.cap < sizeof(unsigned) * (size_t)F) {      // grown (first call, never inside a capture): zero it once;
            if (int rc = c->e_status.reserve(sizeof(unsigned) * (size_t)F)) return rc;   // from then on tile_bases_kernel clears what it latches
            HIP_TRY(hipMemsetAsync(c->e_status.p, 0, c->e_status.cap, s));
        }
        unsigned* d_status = (unsigned*)c->e_status.p;
        unsigned* d_latched = (unsigned*)c->e_small.p;
        unsigned long long* d_bytes = (unsigned long long*)c->e_small.p + F;
        // every block coded once into its tile's stream; tile offsets; streams assembled and their 0xFF bytes counted; the
        // 0xFF offsets; files written (header, stuffed stream, EOI, size or verdict)
        HIP_TRY(E::launch_code_tiles(job, (uint32_t*)c->e_S.p, (uint32_t*)c->e_tt.p, d_status, s));
        // the coder may have raised per-frame error flags that only their consumer (tile_bases / assemble) clears: if the call ends
        // between the two, the flags are cleared here so that they do not leak into the context's next call
        hipError_t e_mid = hipSuccess;
        if (!self)
            e_mid = E::launch_tile_bases((const uint32_t*)c->e_tt.p, (unsigned)tpf, F, (unsigned long long*)c->e_base.p, d_bytes,
                                         (uint32_t*)c->e_ft.p, (unsigned)(u_stride / piece), d_status, d_latched, s);
        if (e_mid == hipSuccess)
            e_mid = E::launch_assemble((const uint32_t*)c->e_S.p, (const uint32_t*)c->e_tt.p, (const unsigned long long*)c->e_base.p, d_bytes,
                                       (const uint32_t*)c->e_ft.p, (unsigned)(u_stride / piece), (unsigned)tpf, F, (uint32_t*)c->e_U.p,
                                       u_stride / 4, (uint32_t*)c->e_cnt.p, (uint32_t*)c->e_fft.p, d_status, d_latched, s);
        if (e_mid != hipSuccess) {
            (void)hipMemsetAsync(d_status, 0, sizeof(unsigned) * (size_t)F, s);
            return hip_err(e_mid, "entropy stage (tile offsets / assembly)");
        }
        E::FilePlan plan;
        plan.hdr = (const uint8_t*)c->e_hdr.p;
        plan.hdr_len = hdr_len;
        plan.latched = d_latched;
        plan.sizes = d_sizes + f0;
        HIP_TRY(E::launch_stuff((const uint32_t*)c->e_U.p, u_stride / 4, d_bytes, F, (const uint32_t*)c->e_cnt.p, (const uint32_t*)c->e_fft.p,
                                d_out + (size_t)f0 * out_stride, out_stride, plan, s));
    }
    return JPEZY_OK;
}

int jpezy_write_jpeg_gpu_batch(jpezy_ctx* c, const int16_t* d_coeffs, int W, int H, int gray, int n_frames, const char* comment,
                               uint8_t* out, size_t cap, long* sizes)
try {
    if (int rc = check_dims(c, W, H, n_frames)) return rc;
    if (!d_coeffs || !out || !sizes) return set_err(JPEZY_E_BADARG, "write_jpeg_gpu: null pointer");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = ensure_code_tables(c)) return rc;
    const size_t nblk = (size_t)jpezy_mcu_cols(W) * jpezy_mcu_rows(H) * 6;
    // chunk the batch so that the worst-case unstuffed streams (208 bytes per block) stay below ~1 GiB
    const size_t worst = nblk * 208 + 4096;
    int per = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((size_t)n_frames, 65535), ((size_t)1 << 30) / worst));   // 65535: grid dimension
    bool any_failed = false;
    const size_t cpf = jpezy_coeff_count(W, H, gray);
    for (int f0 = 0; f0 < n_frames; f0 += per) {
        const int F = std::min(per, n_frames - f0);
        if (int rc = entropy_chunk(c, d_coeffs + (size_t)f0 * cpf, W, H, gray, F, comment, out + (size_t)f0 * cap, cap, sizes + f0, &any_failed))
            return rc;
    }
    return any_failed ? set_err(JPEZY_E_FORMAT, "write_jpeg_gpu: at least one frame failed (see sizes[])") : JPEZY_OK;
}
JPEZY_CATCH

long jpezy_write_jpeg_gpu(jpezy_ctx* c, const int16_t* d_coeffs, int W, int H, int gray, const char* comment, uint8_t* out, size_t cap)
{
    long size = 0;
    const int rc = jpezy_write_jpeg_gpu_batch(c, d_coeffs, W, H, gray, 1, comment, out, cap, &size);
    if (rc != JPEZY_OK && size >= 0) return rc;
    if (size == JPEZY_E_FORMAT) set_err(JPEZY_E_FORMAT, "write_jpeg_gpu: coefficient outside the Annex-K code tables");
    if (size == JPEZY_E_NOSPACE) set_err(JPEZY_E_NOSPACE, "write_jpeg_gpu: output buffer too small");
    return size;
}

// planar RGB on the host -> .jpg bytes on the host, both stages on the GPU (what encoder::encode does end to end)
long jpezy_encode_jpeg(jpezy_ctx* c, const uint8_t* r, const uint8_t* g, const uint8_t* b, int W, int H, int gray, const char* comment,
                       uint8_t* out, size_t cap)
try {
    if (int rc = check_dims(c, W, H, 1)) return rc;
    if (!r || !g || !b || !out) return set_err(JPEZY_E_BADARG, "encode_jpeg: null pointer");
    HIP_TRY(hipSetDevice(c->device));
    // the planes go up band by band (jpezy_hostpipe.h) while the bands before them are transformed into the frame's
    // coefficient buffer on the device; the Huffman stage then runs on the whole frame
    const size_t plane = (size_t)W * H;
    const int mcu_cols = jpezy_mcu_cols(W), B = gray ? 4 : 6;
    if (int rc = c->e_coef.reserve(jpezy_coeff_count(W, H, gray) * sizeof(int16_t))) return rc;
    const std::vector<HostChunk> chunks = plan_host_chunks(W, H, 1, 3, c->host_chunk_bytes);
    size_t P = 0;
    for (const HostChunk& k : chunks) P = std::max(P, (size_t)std::min(H - k.y0 * 16, (k.y1 - k.y0) * 16) * W);
    P = (P + 15) & ~(size_t)15;
    const uint8_t* src[3] = { r, g, b };
    int rc_kernel = JPEZY_OK;
    std::string err;
    auto plan = [&](int i) {
        const HostChunk& k = chunks[(size_t)i];
        jpezy_host::ChunkPlan p;
        const size_t rows = (size_t)std::min(H - k.y0 * 16, (k.y1 - k.y0) * 16);
        for (int q = 0; q < 3; ++q) p.in.push_back({ const_cast<uint8_t*>(src[q]) + (size_t)k.y0 * 16 * W, rows * W, (size_t)q * P });
        return p;
    };
    auto kernel = [&](int i, uint8_t* d_in, uint8_t*, hipStream_t s) -> hipError_t {
        const HostChunk& k = chunks[(size_t)i];
        const int Hc = std::min(H - k.y0 * 16, (k.y1 - k.y0) * 16);
        const int rc = jpezy_fdct_quant_dev(c, d_in, d_in + P, d_in + 2 * P, (size_t)Hc * W, W, Hc, gray, 1,
                                            (int16_t*)c->e_coef.p + (size_t)k.y0 * mcu_cols * B * 64, s);
        if (rc != JPEZY_OK) { rc_kernel = rc; return hipErrorLaunchFailure; }
        return hipSuccess;
    };
    const hipError_t e = c->pipe.run(c->device, c->stream, (int)chunks.size(), 3 * P, 0, plan, kernel, &err);
    if (rc_kernel != JPEZY_OK) return rc_kernel;
    if (e != hipSuccess) return set_err(JPEZY_E_HIP, err.empty() ? std::string("host pipeline: ") + hipGetErrorString(e) : err);
    (void)plane;
    return jpezy_write_jpeg_gpu(c, (const int16_t*)c->e_coef.p, W, H, gray, comment, out, cap);
}
JPEZY_CATCH

// ---- GPU Huffman decoding (decode side) ----
namespace {

// host decode + upload: the authoritative path for everything the GPU decoder does not take or is unsure about
int read_jpeg_host_to_device(jpezy_ctx* c, const uint8_t* data, size_t len, jpezy_frame_info* info, int16_t* d_coeffs, size_t total)
{
    if (total / 64 > 4 * len) return set_err(JPEZY_E_FORMAT, "scan too short for the declared dimensions");
    std::vector<int16_t> tmp(total);
    std::string err;
    const int rc = jpezy_host::read_jpeg(data, len, info, tmp.data(), tmp.size(), &err);
    if (rc < 0) { g_err = err; return rc; }
    HIP_TRY(hipMemcpy(d_coeffs, tmp.data(), total * sizeof(int16_t), hipMemcpyHostToDevice));
    return JPEZY_OK;
}

}  // namespace

// Launch schedule of the GPU Huffman decoder's synchronisation phase (jpezy_read_jpeg_gpu and the batch form).  A launch lets a corrected
// state travel `steps` subsequences inside a workgroup (a workgroup none of whose lanes has a new entry state leaves at once) and one step
// across a workgroup boundary; it reports the lanes that moved, the lanes it left pending and whether a workgroup's last lane moved -- nothing
// pending and no boundary moved means the states are the fixed point, i.e. the sequential decode.
//   first launch: every lane decodes once from its predecessor's proposed exit state (the confirmation) and the corrections travel up to seven
//     lanes on -- isolated wrong proposals, the usual case, settle here and the file is done after one launch and one look by the host
//     (round 2: confirmation, a 24-step launch and a launch that found nothing to do, with a host synchronisation after each).  More than half
//     of the proposals wrong at that first step: the stream does not synchronise (periodic data: flat areas) -- host decoder.
//   refinement launches of 64 steps for longer wrong runs (up to ~100 subsequences in the fuzzer's files), which decode such a stretch lane
//     after lane -- at a fraction of the host decoder's rate, so it only pays while the stretches are short.  They go on while they make
//     progress (round 2: a fixed six launches of 24 steps): two always run; from the third on the lanes that moved must be down to a
//     residue (<= 64) or have fallen to 3/4 of the launch before; never more than MAX_LAUNCHES.
// A file that drops out goes to the host decoder, whose result is the same.
// (tools/fuzz_huffdec.py with JPEZY_HUFFDEC_DEBUG=1 prints the lanes moved per launch; JPEZY_HUFFDEC_PATIENT=1 lifts the budget.)
struct RefineBudget {
    static constexpr int FIRST_STEPS = 8, STEPS = 64, MAX_LAUNCHES = 12;
    // may refinement launch `launch` (1-based, after the first launch) run, given the lanes that moved in the two launches before it?
    bool go_on(int launch, unsigned moved_before, unsigned moved_last) const
    {
        if (launch > MAX_LAUNCHES) return false;
        if (launch <= 2 || moved_last <= 64u) return true;
        return (unsigned long long)moved_last * 4u <= (unsigned long long)moved_before * 3u;
    }
};

namespace {

// ---- the batch form of the GPU Huffman decoder over a list of independent streams (jpezy_huffdec.h) ----
// A stream is an entropy-coded segment that starts in the known state (bit 0, block 0, DC, predictors 0): the scan of a file
// (jpezy_decode_jpeg_batch: one stream per file, every file with its own tables) or one restart interval of a scan
// (jpezy_read_jpeg_gpu: the intervals of a file share one set of tables).  All streams of a call have the same MCU structure.
struct DevStream {
    const uint8_t* scan;                // host memory: the segment, up to (not including) the marker that ends it
    size_t n;
    unsigned total_blocks;              // blocks the stream holds (whole MCUs)
    unsigned long long coeff_off;       // int16 offset of its first coefficient in the output
    unsigned setup;                     // index into the call's tables
};
struct StreamGeom {
    unsigned bpm, ncomp, cstart[3], ccount[3];      // blocks per MCU; component q owns blocks [cstart, cstart + ccount) of every MCU
};

// Decodes the streams into d_coef (device, coef_elems int16, zeroed here): one sequence of launches for all of them.  ok[k] = 1 for the
// streams that converged, decoded without an invalid code and ended inside their data; the others are the caller's to hand to the
// host decoder, whose verdict is the authoritative one.  setup_usable[j] = 0: tables the device form cannot express.
int huffdec_streams(jpezy_ctx* c, const std::vector<DevStream>& streams, const std::vector<jpezy_dev::huffdec::Setup>& setups,
                    const std::vector<char>& setup_usable, const StreamGeom& geom, int16_t* d_coef, size_t coef_elems, std::vector<char>& ok,
                    const std::function<void(const char*)>& lap, bool per_lane = false)
{
    namespace HD = jpezy_dev::huffdec;
    namespace E = jpezy_dev::entropy;
    hipStream_t s = c->stream;
    const unsigned nf = (unsigned)streams.size();
    const unsigned L = HD::subseq_bits();
    const size_t chunk = HD::chunk_bytes();
    ok.assign(nf, 0);

    // geometry
    std::vector<HD::BatchFile> F(nf);
    std::vector<unsigned> wg_file, wg_first;
    size_t total_chunks = 0, total_slots = 0, u_bytes = 0;
    std::vector<char> usable(nf, 1);
    for (unsigned k = 0; k < nf; ++k) {
        const DevStream& st = streams[k];
        HD::BatchFile& f = F[k];
        std::memset(&f, 0, sizeof f);
        f.chunk0 = (unsigned)total_chunks;
        f.n_chunks = (unsigned)((st.n + chunk - 1) / chunk);
        f.n_bytes = (unsigned)st.n;
        f.first_marker = ~0u;               // (st.n may be an upper bound: the device finds where the segment ends)
        f.sub0 = (unsigned)total_slots;
        f.n_sub_max = (unsigned)((st.n * 8 + L - 1) / L);
        f.u_off = u_bytes;
        const size_t ub = (((size_t)f.n_sub_max * L / 8 + 64) + 3) & ~(size_t)3;
        f.u_words = (unsigned)(ub / 4);
        f.coeff_off = st.coeff_off;
        f.total_blocks = st.total_blocks;
        f.nmcu = st.total_blocks / geom.bpm; f.bpm = geom.bpm; f.ncomp = geom.ncomp;
        for (unsigned q = 0; q < 3; ++q) { f.cstart[q] = geom.cstart[q]; f.ccount[q] = geom.ccount[q]; }
        f.setup = st.setup;
        usable[k] = setup_usable[st.setup];
        total_chunks += f.n_chunks;
        total_slots += ((size_t)f.n_sub_max + 255) / 256 * 256;           // a workgroup never straddles two streams
        u_bytes += ub;
        for (unsigned i0 = 0; i0 < f.n_sub_max; i0 += 256) { wg_file.push_back(k); wg_first.push_back(i0); }
    }
    if (total_chunks >= 0xFFFFFFFFull || total_slots >= 0xFFFFFFFFull) return set_err(JPEZY_E_BADARG, "GPU Huffman decoder: too much data for one call");
    const unsigned n_wg = (unsigned)wg_file.size();
    const unsigned ns = (unsigned)setups.size();

    // buffers
    const size_t scan_bytes = total_chunks * chunk;
    if (c->b_pin_cap < scan_bytes) {
        if (c->b_pin) (void)hipHostFree(c->b_pin);
        c->b_pin = nullptr; c->b_pin_cap = 0;
        HIP_TRY(hipHostMalloc((void**)&c->b_pin, scan_bytes + (scan_bytes >> 2) + 4096, hipHostMallocDefault));
        c->b_pin_cap = scan_bytes + (scan_bytes >> 2) + 4096;
    }
    const size_t meta_F = (sizeof(HD::BatchFile) * nf + 255) & ~(size_t)255, meta_S = (sizeof(HD::Setup) * ns + 255) & ~(size_t)255;
    const size_t meta_wg = ((size_t)n_wg * 4 + 255) & ~(size_t)255, meta_act = ((size_t)nf * 4 + 255) & ~(size_t)255;
    if (int rc = c->b_scan.reserve(scan_bytes + 64)) return rc;
    if (int rc = c->b_U.reserve(u_bytes + 64)) return rc;
    if (int rc = c->b_cnt.reserve(std::max(total_chunks, total_slots) * sizeof(uint32_t))) return rc;
    if (int rc = c->b_rb.reserve((std::max(total_chunks, total_slots) + 1) * sizeof(unsigned long long))) return rc;
    if (int rc = c->b_state.reserve(total_slots * 3 * sizeof(uint32_t))) return rc;
    if (int rc = c->b_prop.reserve((total_slots + 1) * sizeof(unsigned long long))) return rc;
    if (int rc = c->b_meta.reserve(meta_F + meta_S + 2 * meta_wg + meta_act)) return rc;
    if (int rc = c->e_tmp.reserve(E::scan_tmp_elems(std::max(total_chunks, total_slots)) * sizeof(unsigned long long))) return rc;
    uint8_t* meta = (uint8_t*)c->b_meta.p;
    HD::BatchFile* d_F = (HD::BatchFile*)meta;
    HD::Setup* d_S = (HD::Setup*)(meta + meta_F);
    unsigned* d_wg_file = (unsigned*)(meta + meta_F + meta_S);
    unsigned* d_wg_first = (unsigned*)(meta + meta_F + meta_S + meta_wg);
    unsigned* d_active = (unsigned*)(meta + meta_F + meta_S + 2 * meta_wg);

    // segments side by side (64-byte aligned, zero padded) in pinned memory: one upload
    for (unsigned k = 0; k < nf; ++k) {
        uint8_t* dst = c->b_pin + (size_t)F[k].chunk0 * chunk;
        std::memcpy(dst, streams[k].scan, streams[k].n);
        std::memset(dst + streams[k].n, 0, (size_t)F[k].n_chunks * chunk - streams[k].n);
    }
    HIP_TRY(hipMemcpyAsync(c->b_scan.p, c->b_pin, scan_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_F, F.data(), sizeof(HD::BatchFile) * nf, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_S, setups.data(), sizeof(HD::Setup) * ns, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_wg_file, wg_file.data(), (size_t)n_wg * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_wg_first, wg_first.data(), (size_t)n_wg * 4, hipMemcpyHostToDevice, s));
    lap("setup + scans up");

    // 1. stuffing out
    HIP_TRY(hipMemsetAsync(c->b_U.p, 0, u_bytes, s));
    HIP_TRY(HD::launch_unstuff_count_batch((const uint8_t*)c->b_scan.p, d_F, nf, (unsigned)total_chunks, (uint32_t*)c->b_cnt.p, s));
    HIP_TRY(E::launch_scan_u32((const uint32_t*)c->b_cnt.p, (unsigned long long*)c->b_rb.p, total_chunks, (unsigned long long*)c->e_tmp.p, s));
    HIP_TRY(HD::launch_unstuff_copy_batch((const uint8_t*)c->b_scan.p, d_F, nf, (unsigned)total_chunks, (const unsigned long long*)c->b_rb.p,
                                          (uint8_t*)c->b_U.p, s));
    lap("unstuff");
    if (per_lane) {
        // short streams with one set of tables (restart intervals of a few MCUs): a lane walks a whole stream -- no speculation, no
        // synchronisation launches, no scans; symbols, coefficients and DC predictors in one launch (jpezy_huffdec.hip)
        HIP_TRY(hipMemsetAsync(d_coef, 0, coef_elems * sizeof(int16_t), s));
        HIP_TRY(HD::launch_stream_per_lane(d_S, (const uint32_t*)c->b_U.p, d_F, nf, d_coef, s));
        HIP_TRY(hipMemcpyAsync(F.data(), d_F, sizeof(HD::BatchFile) * nf, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        lap("decode (a lane per stream)");
        for (unsigned k = 0; k < nf; ++k) {
            const unsigned n_eff = F[k].first_marker < F[k].n_bytes ? F[k].first_marker : F[k].n_bytes;
            const unsigned long long data_bits = ((unsigned long long)n_eff - F[k].removed) * 8;
            ok[k] = usable[k] && n_eff > 0 && !F[k].error && F[k].last_bit <= data_bits;
        }
        return JPEZY_OK;
    }
    // 2. speculation, confirmation, refinement -- one loop for all streams
    uint32_t* d_exit = (uint32_t*)c->b_state.p;
    uint32_t* d_last = d_exit + total_slots;
    unsigned* d_nblocks = (unsigned*)(d_last + total_slots);
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)d_exit, (int)0x80000000u, total_slots, s));
    HIP_TRY(hipMemsetAsync(d_last, 0xFF, total_slots * 4, s));
    HIP_TRY(hipMemsetAsync(d_nblocks, 0, total_slots * 4, s));
    HIP_TRY(HD::launch_speculate_batch(d_S, (const uint32_t*)c->b_U.p, d_F, d_wg_file, d_wg_first, n_wg, (unsigned)total_slots,
                                       (unsigned long long*)c->b_prop.p, d_exit, s));
    lap("speculate");
    std::vector<unsigned> active(nf), prev_moved(nf, 0u);
    std::vector<char> converged(nf, 0), dead(nf, 0);
    for (unsigned k = 0; k < nf; ++k) { active[k] = usable[k] ? 1u : 0u; dead[k] = !usable[k]; }
    RefineBudget budget;
    for (int pass = 0; pass <= RefineBudget::MAX_LAUNCHES; ++pass) {
        bool any = false;
        for (unsigned k = 0; k < nf; ++k) any = any || active[k];
        if (!any) break;
        HIP_TRY(hipMemcpyAsync(d_active, active.data(), (size_t)nf * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(HD::launch_sync_batch(d_S, (const uint32_t*)c->b_U.p, d_F, d_wg_file, d_wg_first, n_wg, d_active, d_exit, d_last, d_nblocks,
                                      pass == 0 ? RefineBudget::FIRST_STEPS : RefineBudget::STEPS, s));
        HIP_TRY(hipMemcpyAsync(F.data(), d_F, sizeof(HD::BatchFile) * nf, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (unsigned k = 0; k < nf; ++k) {
            if (!active[k]) continue;
            const unsigned moved = F[k].changed[0], pending = F[k].changed[1] + F[k].changed[2];
            if (F[k].n_sub == 0) { active[k] = 0; dead[k] = 1; continue; }
            if (pending == 0) { active[k] = 0; converged[k] = 1; continue; }
            // many proposals moved at the first look (periodic data never falls into step), or the refinement launches have
            // stopped paying for this stream (RefineBudget): the caller's other path
            if (pass == 0 ? F[k].changed[3] > F[k].n_sub / 2 + 16 : !budget.go_on(pass + 1, prev_moved[k], moved)) { active[k] = 0; dead[k] = 1; }
            prev_moved[k] = moved;
        }
        // reset the per-pass counters of the streams that go on (one launch: there may be tens of thousands of streams)
        bool any_left = false;
        for (unsigned k = 0; k < nf; ++k) any_left = any_left || active[k];
        if (any_left) {
            HIP_TRY(hipMemcpyAsync(d_active, active.data(), (size_t)nf * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(HD::launch_reset_changed_batch(d_F, d_active, nf, s));
        }
    }
    lap("confirm + refine");
    // 3. block index of every lane, coefficients, DC predictors -- for the streams that converged
    // (a vector of its own: the last refinement pass may still have an upload of `active` in flight from pageable memory)
    std::vector<unsigned> emit_active(nf);
    for (unsigned k = 0; k < nf; ++k) emit_active[k] = converged[k] ? 1u : 0u;
    HIP_TRY(hipMemcpyAsync(d_active, emit_active.data(), (size_t)nf * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->b_cnt.p, d_nblocks, total_slots * 4, hipMemcpyDeviceToDevice, s));
    unsigned long long* d_bb = (unsigned long long*)c->b_prop.p;          // (the proposals are dead: same buffer)
    HIP_TRY(E::launch_scan_u32((const uint32_t*)c->b_cnt.p, d_bb, total_slots, (unsigned long long*)c->e_tmp.p, s));
    HIP_TRY(hipMemsetAsync(d_coef, 0, coef_elems * sizeof(int16_t), s));
    HIP_TRY(HD::launch_emit_batch(d_S, (const uint32_t*)c->b_U.p, d_F, d_wg_file, d_wg_first, n_wg, d_active, d_exit, d_bb, d_coef, s));
    HIP_TRY(HD::launch_dc_prefix_batch(d_coef, d_F, d_active, nf, s));
    HIP_TRY(hipMemcpyAsync(F.data(), d_F, sizeof(HD::BatchFile) * nf, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    lap("emit + DC");
    for (unsigned k = 0; k < nf; ++k) {
        const unsigned n_eff = F[k].first_marker < F[k].n_bytes ? F[k].first_marker : F[k].n_bytes;
        const unsigned long long data_bits = ((unsigned long long)n_eff - F[k].removed) * 8;
        ok[k] = converged[k] && n_eff > 0 && !F[k].error && F[k].last_bit <= data_bits;
    }
    return JPEZY_OK;
}

// the device tables of one scan: Huffman tables + the table sequence of an MCU; false: something the device form cannot express
bool build_dev_setup(jpezy_dev::huffdec::Setup& S, const jpezy_host::ScanSetup& setup, const jpezy_frame_info& info, unsigned total_blocks)
{
    bool usable = true;
    std::memset(&S, 0, sizeof S);
    for (int td = 0; td < 3; ++td) {
        if (setup.present[td]) usable = build_dev_table(S.dc[td], setup.bits[td], setup.vals[td], setup.nvals[td], true) && usable;
        if (setup.present[4 + td]) usable = build_dev_table(S.ac[td], setup.bits[4 + td], setup.vals[4 + td], setup.nvals[4 + td], false) && usable;
    }
    S.total_blocks = total_blocks;
    // The decoder's state carries the block's position inside the MCU only to pick the tables.  It counts modulo the
    // PERIOD of the table sequence: with one table pair for every block (a one-component file, or all Td equal) a
    // decoder that has found the right bit position is in the right state whatever MCU phase it guessed.
    int seq[48], nb = 0;
    for (int q = 0; q < info.ncomp; ++q)
        for (int t = info.H[q] * info.V[q]; t > 0 && nb < 48; --t) seq[nb++] = setup.Td[q];
    int period = nb;
    for (int pd = 1; pd < nb; ++pd) {
        if (nb % pd) continue;
        bool same = true;
        for (int i = pd; i < nb && same; ++i) same = seq[i] == seq[i - pd];
        if (same) { period = pd; break; }
    }
    S.bpm = period;
    return jpezy_dev::huffdec::pack_td_sequence(seq, period, &S.tdmask) && usable;
}

StreamGeom stream_geom(const jpezy_frame_info& info)
{
    StreamGeom g;
    std::memset(&g, 0, sizeof g);
    g.bpm = (unsigned)info.blocks_per_mcu; g.ncomp = (unsigned)info.ncomp;
    for (unsigned q = 0, at = 0; q < (unsigned)info.ncomp && q < 3; ++q) {
        g.cstart[q] = at; g.ccount[q] = (unsigned)(info.H[q] * info.V[q]);
        at += g.ccount[q];
    }
    return g;
}

}  // namespace

int jpezy_read_jpeg_gpu(jpezy_ctx* c, const uint8_t* data, size_t len, jpezy_frame_info* info, int16_t* d_coeffs, size_t coeff_cap)
try {
    namespace HD = jpezy_dev::huffdec;
    namespace E = jpezy_dev::entropy;
    if (!c) return set_err(JPEZY_E_BADARG, "null context");
    jpezy_host::ScanSetup setup;
    std::string err;
    int rc = jpezy_host::parse_header(data, len, info, &setup, &err);
    if (rc < 0) { g_err = err; return rc; }
    if (!d_coeffs) return JPEZY_OK;
    c->h_last_passes = 0;
    HIP_TRY(hipSetDevice(c->device));
    const size_t nmcu = (size_t)info->mcu_cols * info->mcu_rows;
    const int bpm = info->blocks_per_mcu;
    const size_t total_blocks = nmcu * (size_t)bpm, total = total_blocks * 64;
    if (coeff_cap < total) return set_err(JPEZY_E_NOSPACE, "read_jpeg_gpu: coefficient buffer too small");
    if (!aligned16(d_coeffs)) return set_err(JPEZY_E_BADARG, "read_jpeg_gpu: d_coeffs must be 16-byte aligned");

    // what the GPU decoder takes: at most 48 blocks per MCU (3 components of 4 x 4), every selected table present
    bool gpu_ok = bpm <= 48 && total_blocks < 0xFFFFFFFFull && setup.scan_pos < len;
    for (int i = 0; i < info->ncomp && gpu_ok; ++i)
        gpu_ok = setup.Td[i] >= 0 && setup.Td[i] <= 2 && setup.present[setup.Td[i]] && setup.present[4 + setup.Td[i]];
    if (!gpu_ok) return read_jpeg_host_to_device(c, data, len, info, d_coeffs, total);
    if (info->restart_interval != 0) {
        // Restart intervals (DRI / RSTn, ref decoder/jpezy_decoder.hpp:152-163): every interval starts byte aligned, at an MCU boundary, with
        // the predictors at zero -- an entry point.  A scan that is REGULAR (exactly one RSTn behind every interval but the last, nothing else
        // before the marker that ends the scan) is decoded as that many independent streams by the batch form of the kernels; anything else --
        // a missing or extra marker, an interval that runs out of data, tables the device form cannot express -- is the host decoder's, whose
        // reading of such files (markers swallowed as data, predictors kept) is the reference's and nobody else's.
        const size_t Ri = (size_t)info->restart_interval, n_int = (nmcu + Ri - 1) / Ri;
        const uint8_t* scan = data + setup.scan_pos;
        const size_t n_all = len - setup.scan_pos;
        // intervals of a few KB: a lane per interval; longer ones: subsequences, speculation and synchronisation inside every interval
        // (a workgroup per 256 subsequences of an interval, so at most 65,536 of those)
        const bool per_lane = n_int >= 1 && n_all / n_int <= 4096;
        bool regular = n_int >= 1 && n_int <= (per_lane ? (size_t)1 << 20 : (size_t)65536) && n_all >= c->h_min_bytes;
        std::vector<DevStream> streams;
        size_t at = 0;
        while (regular) {
            const size_t seg = jpezy_host::entropy_segment_length(scan + at, n_all - at);       // bytes up to the next marker
            const size_t mk = at + seg;
            const unsigned i = (unsigned)streams.size();
            const size_t mcus = std::min(Ri, nmcu - (size_t)i * Ri);
            streams.push_back({ scan + at, seg, (unsigned)(mcus * bpm), (unsigned long long)i * Ri * bpm * 64, 0u });
            const bool rst = mk + 1 < n_all && scan[mk + 1] >= 0xD0 && scan[mk + 1] <= 0xD7;
            if (!rst) break;                                  // the marker that ends the scan (or the end of the data)
            at = mk + 2;
            if (streams.size() == n_int) regular = false;     // one marker too many
        }
        regular = regular && streams.size() == n_int;
        for (const DevStream& st : streams) regular = regular && st.n > 0;
        if (regular) {
            std::vector<HD::Setup> setups(1);
            std::vector<char> usable(1, build_dev_setup(setups[0], setup, *info, (unsigned)(Ri * bpm)) ? 1 : 0);
            std::vector<char> okv;
            const bool dbg = std::getenv("JPEZY_BATCH_DEBUG") != nullptr;
            auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
            double t_mark = now();
            auto lap = [&](const char* what) {
                if (!dbg) return;
                (void)hipStreamSynchronize(c->stream);
                const double t = now();
                std::fprintf(stderr, "  restart intervals (%zu streams): %-28s %.3f ms\n", streams.size(), what, (t - t_mark) * 1e3);
                t_mark = t;
            };
            if (usable[0] && huffdec_streams(c, streams, setups, usable, stream_geom(*info), d_coeffs, total, okv, lap, per_lane) == JPEZY_OK) {
                bool all = true;
                for (char v : okv) all = all && v;
                if (all) { c->h_last_passes = 1; return JPEZY_OK; }
            }
        }
        return read_jpeg_host_to_device(c, data, len, info, d_coeffs, total);
    }

    // The entropy-coded segment ends at the first marker (0xFF followed by anything but 0x00).  The device finds it while it counts the
    // stuffing (jpezy_huffdec.hip): the file goes up from the first scan byte to its end and the host never walks it -- a pass over a
    // 5 MB scan costs the host 0.3-0.5 ms, a third of the whole call.  n is the upper bound until then.
    const uint8_t* scan = data + setup.scan_pos;
    size_t n = len - setup.scan_pos;
    {
        // A file may carry a long tail behind its scan (a second image, appended data): what is uploaded, counted and allocated
        // for is capped at what the frame's blocks can take at most -- 64 coefficients of a 16-bit code plus 11 value bits
        // each, every byte stuffed: 432 bytes per block.  All blocks are decoded within that many bytes or the stream is bad.
        const size_t cap = (total / 64) * 432 + 4096;
        if (n > cap) n = cap;
    }
    if (n == 0 || n < c->h_min_bytes) return read_jpeg_host_to_device(c, data, len, info, d_coeffs, total);

    hipStream_t s = c->stream;
    const size_t chunk = HD::chunk_bytes(), nc = (n + chunk - 1) / chunk;
    const unsigned L = HD::subseq_bits();
    const unsigned n_sub_max = (unsigned)((n * 8 + L - 1) / L);      // before the stuffing is removed; buffers are sized for it
    unsigned n_sub = n_sub_max;
    const size_t u_bytes = ((size_t)n_sub * L / 8 + 64 + 3) & ~(size_t)3;
    if (int r2 = c->h_scan.reserve(n + 64)) return r2;
    if (int r2 = c->h_U.reserve(u_bytes)) return r2;
    if (int r2 = c->h_cnt.reserve(std::max(nc, (size_t)n_sub) * sizeof(uint32_t))) return r2;
    if (int r2 = c->h_off.reserve((std::max(nc, (size_t)n_sub) + 1) * sizeof(unsigned long long))) return r2;
    if (int r2 = c->h_state.reserve((size_t)n_sub * 3 * sizeof(uint32_t))) return r2;
    if (int r2 = c->h_setup.reserve(sizeof(HD::Setup))) return r2;
    if (int r2 = c->h_small.reserve(64)) return r2;
    size_t max_dc = 0;
    for (int i = 0; i < info->ncomp; ++i) max_dc = std::max(max_dc, nmcu * (size_t)(info->H[i] * info->V[i]));
    if (int r2 = c->h_dc.reserve(std::max((2 * max_dc + 2), (size_t)n_sub) * sizeof(unsigned long long))) return r2;
    if (int r2 = c->e_tmp.reserve(E::scan_tmp_elems(std::max(std::max(nc, (size_t)n_sub), max_dc)) * sizeof(unsigned long long))) return r2;

    // tables
    std::vector<HD::Setup> hs(1);
    HD::Setup& S = hs[0];
    std::memset(&S, 0, sizeof S);
    for (int td = 0; td < 3; ++td) {
        if (setup.present[td]) gpu_ok = build_dev_table(S.dc[td], setup.bits[td], setup.vals[td], setup.nvals[td], true) && gpu_ok;
        if (setup.present[4 + td]) gpu_ok = build_dev_table(S.ac[td], setup.bits[4 + td], setup.vals[4 + td], setup.nvals[4 + td], false) && gpu_ok;
    }
    if (!gpu_ok) return read_jpeg_host_to_device(c, data, len, info, d_coeffs, total);
    S.total_blocks = (unsigned)total_blocks;
    {
        // The decoder's state carries the block's position inside the MCU only to pick the tables.  It counts modulo the
        // PERIOD of the table sequence: with one table pair for every block (a one-component file, or all Td equal) a
        // decoder that has found the right bit position is in the right state whatever MCU phase it guessed.
        int seq[48], b = 0;
        for (int i = 0; i < info->ncomp; ++i)
            for (int k = info->H[i] * info->V[i]; k > 0; --k) seq[b++] = setup.Td[i];
        int period = bpm;
        for (int pd = 1; pd < bpm; ++pd) {
            if (bpm % pd) continue;
            bool ok = true;
            for (int i = pd; i < bpm && ok; ++i) ok = seq[i] == seq[i - pd];
            if (ok) { period = pd; break; }
        }
        S.bpm = period;
        if (!jpezy_dev::huffdec::pack_td_sequence(seq, period, &S.tdmask)) return read_jpeg_host_to_device(c, data, len, info, d_coeffs, total);
    }
    HIP_TRY(hipMemcpyAsync(c->h_setup.p, &S, sizeof S, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->h_scan.p, scan, n, hipMemcpyHostToDevice, s));

    // 1. find the end of the segment, remove the byte stuffing
    unsigned long long* d_marker = (unsigned long long*)c->h_small.p + 4;     // bytes 32..39 of h_small: first marker; 40..47: stuffing bytes removed
    HIP_TRY(hipMemsetAsync(c->h_U.p, 0, u_bytes, s));
    HIP_TRY(hipMemsetAsync(d_marker, 0xFF, sizeof(unsigned long long), s));
    HIP_TRY(hipMemsetAsync(d_marker + 1, 0, sizeof(unsigned long long), s));
    HIP_TRY(HD::launch_unstuff_count((const uint8_t*)c->h_scan.p, n, (uint32_t*)c->h_cnt.p, d_marker, s));
    HIP_TRY(E::launch_scan_u32((const uint32_t*)c->h_cnt.p, (unsigned long long*)c->h_off.p, nc, (unsigned long long*)c->e_tmp.p, s));
    HIP_TRY(HD::launch_unstuff_copy((const uint8_t*)c->h_scan.p, n, d_marker, (const unsigned long long*)c->h_off.p, (uint8_t*)c->h_U.p, d_marker + 1, s));
    unsigned long long seg[2] = { 0, 0 };
    HIP_TRY(hipMemcpyAsync(seg, d_marker, sizeof seg, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (seg[0] < n) n = (size_t)seg[0];
    const unsigned long long removed = seg[1];
    if (n == 0 || removed > n) return read_jpeg_host_to_device(c, data, len, info, d_coeffs, total);
    // only the subsequences that hold real data are decoded: behind them U is zero padding, which no decoder ever
    // falls into step on (a periodic stream), so it would be walked lane by lane
    n_sub = (unsigned)(((n - removed) * 8 + L - 1) / L);
    if (n_sub == 0) return read_jpeg_host_to_device(c, data, len, info, d_coeffs, total);

    // 2. synchronisation passes until no exit state changes
    uint32_t* d_exit = (uint32_t*)c->h_state.p;
    uint32_t* d_last = d_exit + n_sub;
    unsigned* d_nblocks = (unsigned*)(d_last + n_sub);
    unsigned* d_changed = (unsigned*)c->h_small.p + 4;     // [0] lanes that moved, [1] lanes left pending, [2] moved workgroup boundaries (bytes 16..31 of h_small)
    unsigned* d_error = (unsigned*)c->h_small.p + 1;
    unsigned long long* d_lastbit = (unsigned long long*)c->h_small.p + 1;
    // speculation: every lane decodes through its own and the next 12 subsequences from a guess; the farthest-travelled
    // proposal for every boundary becomes the initial exit state (h_dc doubles as the proposal scratch)
    HIP_TRY(HD::launch_speculate((const HD::Setup*)c->h_setup.p, (const uint32_t*)c->h_U.p, u_bytes / 4, n_sub,
                                 (unsigned long long*)c->h_dc.p, d_exit, d_last, d_nblocks, s));
    std::vector<uint32_t> dbg_spec;
    std::vector<unsigned> dbg_moved;
    const bool dbg = std::getenv("JPEZY_HUFFDEC_DEBUG") != nullptr;
    if (dbg) {
        dbg_spec.resize(n_sub);
        HIP_TRY(hipMemcpyAsync(dbg_spec.data(), d_exit, (size_t)n_sub * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    bool converged = false;
    int passes = 0;
    // confirmation and refinement (RefineBudget above)
    {
        unsigned moved = 0, mv[4] = { 0, 0, 0, 0 };
        bool pending = false;        // lanes left with a stale entry state, or a moved workgroup boundary: not the fixed point yet
        auto pass = [&](int max_inner) -> int {
            HIP_TRY(hipMemsetAsync(d_changed, 0, sizeof mv, s));
            HIP_TRY(HD::launch_sync((const HD::Setup*)c->h_setup.p, (const uint32_t*)c->h_U.p, u_bytes / 4, n_sub, d_exit, d_last, d_nblocks,
                                    d_changed, max_inner, s));
            HIP_TRY(hipMemcpyAsync(mv, d_changed, sizeof mv, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            moved = mv[0];
            pending = mv[1] != 0 || mv[2] != 0;
            ++passes;
            if (dbg) dbg_moved.push_back(moved);
            return JPEZY_OK;
        };
        if (int r2 = pass(RefineBudget::FIRST_STEPS)) return r2;
        converged = !pending;
        const unsigned wrong = mv[3];            // proposals the confirmation step did not bear out
        const bool patient = dbg && std::getenv("JPEZY_HUFFDEC_PATIENT") != nullptr;      // diagnostic: show where the launches would have led
        if (!converged && (wrong <= n_sub / 2 + 16 || patient)) {
            RefineBudget budget;
            unsigned prev = moved;
            for (int it = 1; !converged && (patient ? it <= 40 : budget.go_on(it, prev, moved)); ++it) {
                prev = moved;
                if (int r2 = pass(RefineBudget::STEPS)) return r2;
                converged = !pending;
            }
        }
    }
    if (dbg) {
        std::vector<uint32_t> fin(n_sub);
        HIP_TRY(hipMemcpy(fin.data(), d_exit, (size_t)n_sub * 4, hipMemcpyDeviceToHost));
        size_t same = 0, longest = 0, run = 0;
        for (unsigned i = 0; i < n_sub; ++i) {
            if (fin[i] == dbg_spec[i]) { ++same; run = 0; } else { ++run; if (run > longest) longest = run; }
        }
        std::string mv_s;
        for (unsigned m : dbg_moved) mv_s += " " + std::to_string(m);
        std::fprintf(stderr, "huffdec: %u subsequences, %zu speculative exit states already true, longest wrong run %zu, %d passes, converged %d; lanes moved per launch:%s\n",
                     n_sub, same, longest, passes, (int)converged, mv_s.c_str());
    }
    if (!converged) return read_jpeg_host_to_device(c, data, len, info, d_coeffs, total);

    // 3. global block index of every lane, coefficients, DC predictors
    HIP_TRY(hipMemcpyAsync(c->h_cnt.p, d_nblocks, (size_t)n_sub * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(E::launch_scan_u32((const uint32_t*)c->h_cnt.p, (unsigned long long*)c->h_off.p, n_sub, (unsigned long long*)c->e_tmp.p, s));
    HIP_TRY(hipMemsetAsync(d_coeffs, 0, total * sizeof(int16_t), s));
    HIP_TRY(hipMemsetAsync(d_error, 0, sizeof(unsigned), s));
    HIP_TRY(hipMemsetAsync(d_lastbit, 0xFF, sizeof(unsigned long long), s));
    HIP_TRY(HD::launch_emit((const HD::Setup*)c->h_setup.p, (const uint32_t*)c->h_U.p, u_bytes / 4, n_sub, d_exit,
                            (const unsigned long long*)c->h_off.p, d_coeffs, d_error, d_lastbit, s));
    {   // DC differences -> values, all components in three launches (round 2: gather, two-launch scan, scatter per component = twelve)
        const StreamGeom g = stream_geom(*info);
        HIP_TRY(HD::launch_dc_prefix(d_coeffs, g.bpm, g.ncomp, g.cstart, g.ccount, nmcu, (int*)c->h_dc.p, s));
    }
    unsigned error = 0;
    unsigned long long last_bit = 0;
    HIP_TRY(hipMemcpyAsync(&error, d_error, sizeof error, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&last_bit, d_lastbit, sizeof last_bit, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // an invalid code, or a last block that is not complete inside the real data: the host decoder decides
    if (error || last_bit > (unsigned long long)(n - removed) * 8) return read_jpeg_host_to_device(c, data, len, info, d_coeffs, total);
    c->h_last_passes = passes;
    return JPEZY_OK;
}
JPEZY_CATCH

// decoder::decode end to end (ref decoder/jpezy_decoder.hpp:76-134): .jpg bytes in, planar r,g,b out
int jpezy_decode_jpeg(jpezy_ctx* c, const uint8_t* data, size_t len, int gray, jpezy_frame_info* info, uint8_t* r, uint8_t* g, uint8_t* b,
                      size_t plane_cap)
try {
    if (!c || !info) return set_err(JPEZY_E_BADARG, "decode_jpeg: bad argument");
    int rc = jpezy_read_jpeg_gpu(c, data, len, info, nullptr, 0);           // header only
    if (rc < 0) return rc;
    if (!r || !g || !b) return JPEZY_OK;
    const int W = info->width, H = info->height;
    if (int rc2 = check_dims(c, W, H, 1)) return rc2;
    if (plane_cap < (size_t)W * H) return set_err(JPEZY_E_NOSPACE, "decode_jpeg: plane buffers too small");
    const size_t ncoef = (size_t)info->mcu_cols * info->mcu_rows * info->blocks_per_mcu * 64;
    // sized from untrusted SOF0 fields: a block costs at least 2 bits of scan (1-bit DC code + 1-bit EOB code)
    if (ncoef / 64 > 4 * len) return set_err(JPEZY_E_FORMAT, "decode_jpeg: scan too short for the declared dimensions");
    const uint8_t tq[3] = { (uint8_t)info->Tq[0], (uint8_t)info->Tq[1], (uint8_t)info->Tq[2] };
    const bool own_layout = info->ncomp == 3 && info->precision == 8 && info->H[0] == 2 && info->V[0] == 2 && info->H[1] == 1 &&
                            info->V[1] == 1 && info->H[2] == 1 && info->V[2] == 1;
    if (!own_layout) {   // any other baseline layout decode_mcu handles (:504-528): Huffman decoding on the device as for
                         // jpezy's own files (the host head for what that decoder declines), then the generic kernels
        HIP_TRY(hipSetDevice(c->device));
        if (int rc2 = c->out.reserve(ncoef * sizeof(int16_t))) return rc2;
        rc = jpezy_read_jpeg_gpu(c, data, len, info, (int16_t*)c->out.p, ncoef);
        if (rc < 0) return rc;
        const uint8_t hs[3] = { (uint8_t)info->H[0], (uint8_t)info->H[1], (uint8_t)info->H[2] };
        const uint8_t vs[3] = { (uint8_t)info->V[0], (uint8_t)info->V[1], (uint8_t)info->V[2] };
        return dequant_idct_generic_impl(c, (const int16_t*)c->out.p, info->qt, info->ncomp, hs, vs, tq, W, H, gray, info->precision, r,
                                         g, b, true);
    }
    // jpezy's own layout: Huffman decoding, dequantisation, IDCT and colour conversion all on the device
    HIP_TRY(hipSetDevice(c->device));
    if (int rc2 = c->out.reserve(ncoef * sizeof(int16_t))) return rc2;
    rc = jpezy_read_jpeg_gpu(c, data, len, info, (int16_t*)c->out.p, ncoef);
    if (rc < 0) return rc;
    const size_t plane = (size_t)W * H, stride = (plane + 15) & ~(size_t)15;
    uint8_t* dst[3] = { r, g, b };
    // three plain copies into the caller's planes: measured against bands through the pinned ring of the host-buffer entry points
    // (tools/measure_decode_single_raw.py, 4096 x 4096, planes the caller has touched before: 2.0 ms against 2.6 ms) -- the runtime's
    // pageable path moves 50 MB in 0.9 ms when the pages exist; what a caller pays for fresh pages is page faults, in either form
    for (int k = 0; k < 3; ++k)
        if (int rc2 = c->in[k].reserve(stride)) return rc2;
    if (int rc2 = jpezy_dequant_idct_dev(c, (const int16_t*)c->out.p, info->qt, tq, stride, W, H, gray, 1, (uint8_t*)c->in[0].p,
                                         (uint8_t*)c->in[1].p, (uint8_t*)c->in[2].p, c->stream))
        return rc2;
    for (int k = 0; k < 3; ++k) HIP_TRY(hipMemcpyAsync(dst[k], c->in[k].p, plane, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return JPEZY_OK;
}
JPEZY_CATCH

// ---- batch form (round 3): files of jpezy's own layout and one size go through the Huffman decoder TOGETHER ----
namespace {

struct FastFile {
    int index;                          // position in the caller's arrays
    jpezy_host::ScanSetup setup;
    const uint8_t* scan;
    size_t n;
};

// One slice of a group (same W x H, same layout, same quantiser tables): Huffman decoding of all files in one sequence of launches
// (huffdec_streams: a stream per file), ONE inverse-transform launch over the slice -- the fused kernel for jpezy's own 2x2,1x1,1x1
// layout, the generic kernels' batch form for every other layout decode_mcu handles --, the planes copied out per file.
// ok[k] = 1 for files decoded here; the others (not converged, irregular stream) are left to the per-file path, whose verdict --
// host decoder included -- is the authoritative one.
int decode_slice_fast(jpezy_ctx* c, const std::vector<FastFile>& files, const jpezy_frame_info& info, int gray, int plane_buf,
                      std::vector<char>& ok)
{
    namespace HD = jpezy_dev::huffdec;
    hipStream_t s = c->stream;
    const unsigned nf = (unsigned)files.size();
    const int W = info.width, H = info.height;
    const unsigned bpm = (unsigned)info.blocks_per_mcu;
    const size_t nmcu = (size_t)info.mcu_cols * info.mcu_rows, cpf = nmcu * bpm * 64, plane = (size_t)W * H, pstride = (plane + 15) & ~(size_t)15;
    const bool own_layout = info.ncomp == 3 && info.precision == 8 && info.H[0] == 2 && info.V[0] == 2 && info.H[1] == 1 && info.V[1] == 1 &&
                            info.H[2] == 1 && info.V[2] == 1;
    ok.assign(nf, 0);
    const bool dbg = std::getenv("JPEZY_BATCH_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(s);
        const double t = now();
        std::fprintf(stderr, "  batch slice (%u files): %-28s %.3f ms\n", nf, what, (t - t_mark) * 1e3);
        t_mark = t;
    };
    // the device tables are built once per DISTINCT set of Huffman tables (24 KB and ~10 us each: files of one encoder share theirs)
    auto same_tables = [](const jpezy_host::ScanSetup& a, const jpezy_host::ScanSetup& b0) {
        if (std::memcmp(a.Td, b0.Td, sizeof a.Td) || std::memcmp(a.present, b0.present, sizeof a.present) || std::memcmp(a.bits, b0.bits, sizeof a.bits)) return false;
        for (int t = 0; t < 8; ++t)
            if (a.present[t] && (a.nvals[t] != b0.nvals[t] || std::memcmp(a.vals[t], b0.vals[t], (size_t)a.nvals[t]))) return false;
        return true;
    };
    std::vector<DevStream> streams(nf);
    std::vector<HD::Setup> setups;
    std::vector<char> setup_usable;
    std::vector<unsigned> first_with;                       // file that brought setups[j]
    bool small = nf > 1;
    for (unsigned k = 0; k < nf; ++k) {
        unsigned j = (unsigned)first_with.size();
        for (unsigned q = (unsigned)first_with.size(); q-- > 0;)               // (the latest first: neighbours tend to match)
            if (same_tables(files[k].setup, files[first_with[q]].setup)) { j = q; break; }
        if (j == first_with.size()) {
            setups.emplace_back();
            setup_usable.push_back(build_dev_setup(setups.back(), files[k].setup, info, (unsigned)(nmcu * bpm)) ? 1 : 0);   // (any DHT the file carries)
            if (first_with.size() < 64) first_with.push_back(k);              // (many different tables: one each from there on, no more searching)
            j = (unsigned)setups.size() - 1;
        }
        streams[k] = { files[k].scan, files[k].n, (unsigned)(nmcu * bpm), (unsigned long long)k * cpf, j };
        small = small && files[k].n <= 4096;
    }
    // Thumbnails: scans of a few KB that all carry the same tables are walked by a lane each -- one launch instead of the speculation /
    // synchronisation chain over a workgroup per file
    const bool per_lane = small && setups.size() == 1;
    if (int rc = c->b_coef.reserve((size_t)nf * cpf * sizeof(int16_t))) return rc;
    if (int rc = huffdec_streams(c, streams, setups, setup_usable, stream_geom(info), (int16_t*)c->b_coef.p, (size_t)nf * cpf, ok, lap, per_lane)) return rc;
    // dequantisation + inverse transform + colour conversion of the whole slice in one launch (a file that failed decodes to garbage nobody reads)
    if (int rc = c->b_planes[plane_buf].reserve(3 * pstride * nf)) return rc;
    uint8_t* pl = (uint8_t*)c->b_planes[plane_buf].p;
    const uint8_t tq[3] = { (uint8_t)info.Tq[0], (uint8_t)info.Tq[1], (uint8_t)info.Tq[2] };
    if (own_layout) {
        if (int rc = jpezy_dequant_idct_dev(c, (const int16_t*)c->b_coef.p, info.qt, tq, pstride, W, H, gray, (int)nf, pl, pl + pstride * nf,
                                            pl + 2 * pstride * nf, s))
            return rc;
    } else {              // any other layout: the generic kernels over the slice (block loop over all frames, one plane launch with the frame as z)
        const uint8_t hs[3] = { (uint8_t)info.H[0], (uint8_t)info.H[1], (uint8_t)info.H[2] };
        const uint8_t vs[3] = { (uint8_t)info.V[0], (uint8_t)info.V[1], (uint8_t)info.V[2] };
        if (int rc = generic_dev_core(c, (const int16_t*)c->b_coef.p, info.qt, info.ncomp, hs, vs, tq, W, H, gray, info.precision, pl,
                                      pl + pstride * nf, pl + 2 * pstride * nf, s, nullptr, (int)nf, pstride))
            return rc;
    }
    HIP_TRY(hipStreamSynchronize(s));
    lap("IDCT");
    (void)plane;
    return JPEZY_OK;      // the planes of the files with ok[k] wait in b_planes[plane_buf]: [r | g | b][nf][pstride]
}

}  // namespace

// Many files: the per-file pipeline is latency-bound (small launches, five host synchronisations), so files are decoded
// concurrently -- up to 8 in flight, each on a child context of its own (stream, scratch, quantiser tables), one host
// thread per child.  Files are independent (ref decoder objects are per file): status[i] is file i's own result.
int jpezy_decode_jpeg_batch(jpezy_ctx* c, int n, const uint8_t* const* data, const size_t* len, int gray, jpezy_frame_info* info,
                            uint8_t* const* r, uint8_t* const* g, uint8_t* const* b, const size_t* plane_cap, int* status)
try {
    if (!c || n < 0 || (n > 0 && (!data || !len || !info || !r || !g || !b || !plane_cap || !status)))
        return set_err(JPEZY_E_BADARG, "decode_jpeg_batch: bad argument");
    if (n == 0) return JPEZY_OK;
    // Fast path (round 3): files are grouped by size, layout and quantiser tables and go through the batch form of the GPU Huffman
    // decoder and ONE inverse-transform launch per slice (the fused kernel for jpezy's own layout, the generic kernels for the others);
    // whatever that path declines or cannot settle (restart intervals, irregular streams, streams that do not converge) takes the
    // per-file path below, file by file as before.
    std::vector<char> done((size_t)n, 0);
    c->b_last_fast = 0;
    HIP_TRY(hipSetDevice(c->device));
    {
        struct Cand { FastFile ff; jpezy_frame_info info; bool good = false; };
        std::vector<Cand> all((size_t)n);
        // headers, spread over host threads
        auto prep = [&](int i) {
            Cand& cd = all[(size_t)i];
            std::string err;
            if (!data[i] || !r[i] || !g[i] || !b[i]) return;
            if (jpezy_host::parse_header(data[i], len[i], &cd.info, &cd.ff.setup, &err) < 0) return;
            const jpezy_frame_info& fi = cd.info;
            // what the batch form takes: every baseline layout the reference's decode_mcu handles (1 or 3 components, sampling factors
            // 1..4, at most 48 blocks per MCU), no restart intervals
            bool fits = (fi.ncomp == 1 || fi.ncomp == 3) && fi.restart_interval == 0 && fi.width > 0 && fi.height > 0 &&
                        fi.blocks_per_mcu >= 1 && fi.blocks_per_mcu <= 48 && fi.width <= 65535 && fi.height <= 65535;
            for (int q = 0; q < fi.ncomp && fits; ++q) fits = fi.H[q] >= 1 && fi.H[q] <= 4 && fi.V[q] >= 1 && fi.V[q] <= 4;
            if (!fits || cd.ff.setup.scan_pos >= len[i] || plane_cap[i] < (size_t)fi.width * fi.height) return;
            bool tabs = true;
            for (int q = 0; q < fi.ncomp && tabs; ++q)
                tabs = cd.ff.setup.Td[q] >= 0 && cd.ff.setup.Td[q] <= 2 && cd.ff.setup.present[cd.ff.setup.Td[q]] && cd.ff.setup.present[4 + cd.ff.setup.Td[q]];
            if (!tabs) return;
            // (the file goes up from its first scan byte to its end: the device finds the marker that ends the entropy-coded segment)
            const uint8_t* scan = data[i] + cd.ff.setup.scan_pos;
            size_t ns = len[i] - cd.ff.setup.scan_pos;
            const size_t nblk = (size_t)fi.mcu_cols * fi.mcu_rows * (size_t)fi.blocks_per_mcu;
            if (ns == 0 || nblk > 4 * len[i] || nblk >= 0xFFFFFFFFull) return;
            ns = std::min(ns, nblk * 432 + 4096);          // a long tail behind the scan is not uploaded (see jpezy_read_jpeg_gpu)
            if (ns >= 0xFFFFFFFFull) return;
            cd.ff.index = i; cd.ff.scan = scan; cd.ff.n = ns;
            cd.good = true;
        };
        {
            unsigned hwp = std::thread::hardware_concurrency();
            const int nt = (int)std::max(1u, std::min<unsigned>(std::min<unsigned>(hwp ? hwp : 4u, 8u), (unsigned)(n + 15) / 16));
            std::atomic<int> next{ 0 };
            auto work = [&] { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) prep(i); };
            std::vector<std::thread> pool;
            for (int t = 1; t < nt; ++t) pool.emplace_back(work);
            work();
            for (auto& t : pool) t.join();
        }
        std::vector<const Cand*> cand;
        for (const Cand& cd : all)
            if (cd.good) cand.push_back(&cd);
        // the planes of slice k go down to the caller's buffers on a thread and a stream of their own while slice k + 1 is decoded
        // (two plane buffers): a 1080p file is 6.2 MB of planes, PCIe is what bounds a batch
        hipStream_t s_down = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking));
        // (three plane buffers: with two, slice k waits for the planes of slice k - 2 to be delivered, and a slice of 1080p noise -- 1.7 ms of
        // decoding, 2.8 ms of download and hand-out -- then takes (1.7 + 2.8) / 2 = 2.25 ms; with three the link's 1.8 ms is the bound)
        std::thread drainer[jpezy_ctx::B_DEPTH];
        std::atomic<int> drain_err{ 0 };
        int slice_no = 0;
        auto join_all = [&] { for (auto& t : drainer) if (t.joinable()) t.join(); };
        std::vector<char> taken(cand.size(), 0);
        for (size_t a = 0; a < cand.size(); ++a) {
            if (taken[a]) continue;
            // the group of cand[a]: same size, same quantiser tables for the three components
            std::vector<FastFile> grp;
            auto same_group = [&](const jpezy_frame_info& x, const jpezy_frame_info& y) {
                if (x.width != y.width || x.height != y.height || x.ncomp != y.ncomp || x.precision != y.precision) return false;
                for (int q = 0; q < x.ncomp; ++q) {
                    if (x.H[q] != y.H[q] || x.V[q] != y.V[q]) return false;
                    if (std::memcmp(x.qt[x.Tq[q] & 3], y.qt[y.Tq[q] & 3], sizeof x.qt[0])) return false;
                }
                return true;
            };
            for (size_t k = a; k < cand.size(); ++k)
                if (!taken[k] && same_group(cand[a]->info, cand[k]->info)) { taken[k] = 1; grp.push_back(cand[k]->ff); }
            if (grp.size() < 2) continue;                                   // a single file gains nothing here
            const jpezy_frame_info& gi = cand[a]->info;
            // slices: a slice's chain of launches is latency (~1 ms whatever it holds), its planes go down while the next slice is decoded.  16
            // files of 1080p (100 MB of planes) balance the two; smaller files get proportionally more per slice -- ~100 MB of planes,
            // at most 512 files (JPEZY_BATCH_SLICE: development knob, a fixed count) -- and never more than ~1.5 GB of planes +
            // coefficients + the generic kernels' int samples at a time
            const size_t per_file = (size_t)gi.width * gi.height * 3 + (size_t)gi.mcu_cols * gi.mcu_rows * (size_t)gi.blocks_per_mcu * (128 + 256);
            static const size_t slice_knob = [] { const char* e = std::getenv("JPEZY_BATCH_SLICE"); const int v = e ? std::atoi(e) : 0; return (size_t)(v <= 0 ? 0 : v < 2 ? 2 : v > 512 ? 512 : v); }();
            const size_t by_planes = std::min<size_t>(512, std::max<size_t>(16, ((size_t)100 << 20) / std::max<size_t>((size_t)gi.width * gi.height * 3, 1)));
            const size_t slice_files = slice_knob ? slice_knob : by_planes;
            const size_t per_slice = std::max<size_t>(2, std::min<size_t>(slice_files, ((size_t)3 << 29) / std::max<size_t>(per_file, 1)));
            const size_t plane = (size_t)gi.width * gi.height, pstride = (plane + 15) & ~(size_t)15;
            for (size_t s0 = 0; s0 < grp.size(); s0 += per_slice) {
                std::vector<FastFile> slice(grp.begin() + s0, grp.begin() + std::min(grp.size(), s0 + per_slice));
                std::vector<char> okv;
                const int pb = slice_no % jpezy_ctx::B_DEPTH;
                if (drainer[pb].joinable()) drainer[pb].join();             // the slice that used this plane buffer has been delivered
                if (decode_slice_fast(c, slice, gi, gray, pb, okv) != JPEZY_OK) continue;      // (the per-file path reports what is wrong)
                ++slice_no;
                std::vector<int> idx;                                        // (k, caller index) of the files decoded here
                for (size_t k = 0; k < slice.size(); ++k)
                    if (okv[k]) {
                        const int i = slice[k].index;
                        info[i] = all[(size_t)i].info;
                        status[i] = JPEZY_OK;
                        done[(size_t)i] = 1;
                        ++c->b_last_fast;
                        idx.push_back((int)k); idx.push_back(i);
                    }
                const uint8_t* pl = (const uint8_t*)c->b_planes[pb].p;
                const size_t nfs = slice.size();
                const int device = c->device;
                // The slice's planes come down in ONE copy into pinned memory and are handed out with memcpy (four threads when there is
                // much to copy: a core moves ~25 GB/s, the link 56).  Three copies into the caller's pageable planes per file cost ~40 us
                // of driver time per file whatever their size (1,024 files of 256 x 256: 47 ms, PCIe would need 5) and reach 43 GB/s on
                // large ones (256 x 1080p, smooth content: 43.5 -> 35.6 ms; JPEZY_BATCH_DIRECT=1: the direct copies, for comparison).
                uint8_t* stage = nullptr;
                static const bool direct = std::getenv("JPEZY_BATCH_DIRECT") != nullptr;
                if (!direct && 3 * pstride * nfs <= ((size_t)256 << 20)) {       // (slices of very large pictures: no quarter-GB of pinned memory each)
                    const size_t need = 3 * pstride * nfs;
                    if (c->b_stage_cap[pb] < need) {
                        if (c->b_stage[pb]) (void)hipHostFree(c->b_stage[pb]);
                        c->b_stage[pb] = nullptr; c->b_stage_cap[pb] = 0;
                        if (hipHostMalloc((void**)&c->b_stage[pb], need + (need >> 2), hipHostMallocDefault) == hipSuccess) c->b_stage_cap[pb] = need + (need >> 2);
                    }
                    stage = c->b_stage_cap[pb] >= need ? c->b_stage[pb] : nullptr;         // (no pinned memory: the per-plane copies)
                }
                drainer[pb] = std::thread([=, &drain_err] {
                    if (hipSetDevice(device) != hipSuccess) { drain_err.store(1); return; }
                    if (stage) {
                        if (hipMemcpyAsync(stage, pl, 3 * pstride * nfs, hipMemcpyDeviceToHost, s_down) != hipSuccess ||
                            hipStreamSynchronize(s_down) != hipSuccess) { drain_err.store(1); return; }
                        auto hand_out = [&](size_t q0, size_t step) {
                            for (size_t q = q0; q + 1 < idx.size(); q += step) {
                                const size_t k = (size_t)idx[q];
                                const int i = idx[q + 1];
                                std::memcpy(r[i], stage + pstride * k, plane);
                                std::memcpy(g[i], stage + pstride * (nfs + k), plane);
                                std::memcpy(b[i], stage + pstride * (2 * nfs + k), plane);
                            }
                        };
                        const int nt = 3 * plane * (idx.size() / 2) > ((size_t)8 << 20) ? 4 : 1;
                        std::vector<std::thread> helpers;
                        for (int t = 1; t < nt; ++t) helpers.emplace_back(hand_out, (size_t)2 * t, (size_t)2 * nt);
                        hand_out(0, (size_t)2 * nt);
                        for (auto& h : helpers) h.join();
                        return;
                    }
                    for (size_t q = 0; q + 1 < idx.size(); q += 2) {
                        const size_t k = (size_t)idx[q];
                        const int i = idx[q + 1];
                        if (hipMemcpyAsync(r[i], pl + pstride * k, plane, hipMemcpyDeviceToHost, s_down) != hipSuccess ||
                            hipMemcpyAsync(g[i], pl + pstride * (nfs + k), plane, hipMemcpyDeviceToHost, s_down) != hipSuccess ||
                            hipMemcpyAsync(b[i], pl + pstride * (2 * nfs + k), plane, hipMemcpyDeviceToHost, s_down) != hipSuccess)
                            drain_err.store(1);
                    }
                    if (hipStreamSynchronize(s_down) != hipSuccess) drain_err.store(1);
                });
            }
        }
        join_all();
        (void)hipStreamDestroy(s_down);
        if (drain_err.load()) return set_err(JPEZY_E_HIP, "decode_jpeg_batch: copying the planes to the host failed");
    }
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 4;
    const int nw = (int)std::min<unsigned>(std::min<unsigned>((unsigned)n, hw), 8u);
    while ((int)c->workers.size() < nw) {
        jpezy_ctx* w = jpezy_ctx_create(c->device);
        if (!w) return JPEZY_E_HIP;                                  // message set by jpezy_ctx_create
        w->is_batch_child = true;
        c->workers.push_back(w);
    }
    // the per-file workers decode the files the grouped form declines: same knobs as the parent, or one batch would mix modes
    for (int k = 0; k < nw; ++k) {
        c->workers[k]->h_min_bytes = c->h_min_bytes;
        c->workers[k]->dec_tolerance = c->dec_tolerance;
        c->workers[k]->force_exact = c->force_exact;
    }
    std::vector<std::string> msg((size_t)n);
    auto work = [&](int k) {
        jpezy_ctx* w = c->workers[(size_t)k];
        for (int i = k; i < n; i += nw) {
            if (done[(size_t)i]) continue;
            status[i] = jpezy_decode_jpeg(w, data[i], len[i], gray, &info[i], r[i], g[i], b[i], plane_cap[i]);
            if (status[i] < 0) msg[(size_t)i] = g_err;               // this thread's message
        }
    };
    std::vector<std::thread> pool;
    for (int k = 1; k < nw; ++k) pool.emplace_back(work, k);
    work(0);
    for (auto& t : pool) t.join();
    for (int i = 0; i < n; ++i)
        if (status[i] < 0) return set_err(status[i], "decode_jpeg_batch: file " + std::to_string(i) + ": " + msg[(size_t)i]);
    return JPEZY_OK;
}
JPEZY_CATCH

int jpezy_ctx_last_huffdec_passes(jpezy_ctx* c) { return c ? c->h_last_passes : 0; }
int jpezy_ctx_last_batch_fast_count(jpezy_ctx* c) { return c ? c->b_last_fast : 0; }
void jpezy_ctx_set_huffdec_min_bytes(jpezy_ctx* c, size_t n) { if (c) c->h_min_bytes = n; }

int jpezy_read_jpeg(const uint8_t* data, size_t len, jpezy_frame_info* info, int16_t* coeffs, size_t coeff_cap)
try {
    std::string err;
    const int rc = jpezy_host::read_jpeg(data, len, info, coeffs, coeff_cap, &err);
    if (rc < 0) g_err = err;
    return rc;
}
JPEZY_CATCH

}  // extern "C"
