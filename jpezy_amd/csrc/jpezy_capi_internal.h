// jpezy_capi_internal.h -- what the translation units of the C-ABI share (internal): error plumbing, device buffers, the context.
// jpezy_capi.hip (context, the two transform stages), jpezy_capi_entropy.hip (Huffman coding, host and GPU),
// jpezy_capi_huffdec.hip (GPU Huffman decoding of one file, decoder::decode end to end), jpezy_capi_decode_batch.hip (the batch form).
#pragma once
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
#include <functional>
#include <vector>

#include "../../include/jpezy_constants.h"
#include "../../include/jpezy_hip.h"
#include "jpezy_device.h"
#include "jpezy_entropy.h"
#include "jpezy_huffdec.h"
#include "jpezy_host_codec.h"
#include "jpezy_hostpipe.h"

using namespace jpezy_dev;

namespace jpezy_capi {

inline thread_local std::string g_err;

inline int set_err(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}
inline int hip_err(hipError_t e, const char* what)
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
inline int drain_before_table_rewrite(hipStream_t s, const char* what, bool own_stream_only = false)
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

inline const int kQt[2][64] = { JPEZY_QT_LUMA_INIT, JPEZY_QT_CHROMA_INIT };
inline const unsigned char kZzInv[64] = JPEZY_ZZ_INV_INIT;   // natural index -> zig-zag position

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

// Chunks of the streaming host-buffer entry points: MCU-row bands of a frame that is large against the chunk size, otherwise
// several whole frames.  Chunk k covers frames [f0, f0 + nf) and, in band mode (nf == 1), MCU rows [y0, y1) of frame f0.
struct HostChunk { int f0, nf, y0, y1; };

inline std::vector<HostChunk> plan_host_chunks(int W, int H, int n_frames, size_t bytes_per_px, size_t target)
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

}  // namespace jpezy_capi
using namespace jpezy_capi;

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
#ifndef JPEZY_DEFAULT_VARIANT
#define JPEZY_DEFAULT_VARIANT 1
#endif
    int variant = JPEZY_DEFAULT_VARIANT;   // encode kernel: 0 = FP64 butterflies, 1 = FP32 first level (default), 2 = variant 1's arithmetic in persistent workgroups
    int n_cus = 0;                 // compute units of the device (grid of the persistent kernel)
    float dc_rq[2] = { 0, 0 }, dc_bias[2] = { 0, 0 };   // f32::dc_formula's constants; 0: the formula does not reproduce the DC table (jpezy_ctx_create)
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
    DevBuf h_scan, h_U, h_cnt, h_off, h_state, h_setup, h_small, h_dc, h_dcbuf;   // GPU Huffman decoder (jpezy_huffdec.hip)
    std::vector<uint8_t> h_setup_host;  // the device tables h_setup holds (jpezy_read_jpeg_gpu uploads them only when they change)
    const void* h_setup_dev = nullptr;  // ... and the allocation they were uploaded to
    uint8_t* h_fb_pin = nullptr;        // pinned buffer for the host decoder's coefficients (read_jpeg_host_to_device)
    size_t h_fb_cap = 0;
    int h_last_passes = 0;         // synchronisation passes of the last jpezy_read_jpeg_gpu (0: the host decoder was used)
    size_t h_min_bytes = 32 << 10;    // scans shorter than this are decoded on the host: the GPU path has ~0.32 ms of fixed cost, the host decoder
                                      // takes ~10.5 us per KiB of scan (tools/measure/huffdec_threshold.py, profiles/r04_huffdec_threshold.txt: they cross at
                                      // ~30 KiB; round 3: 0.6 ms, 64 KiB; round 2: 3 ms, 256 KiB)
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

// ---- helpers shared by the translation units (C linkage, hidden: not part of the ABI) ----
// streams of the batch form of the GPU Huffman decoder (jpezy_capi_huffdec.hip)
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

#define JPEZY_INTERNAL __attribute__((visibility("hidden")))
extern "C" {
JPEZY_INTERNAL int jpezy_internal_check_dims(const jpezy_ctx* c, int W, int H, int n_frames);
// geometry + tables + the two launches of the any-layout decoder on device memory (jpezy_capi.hip)
JPEZY_INTERNAL int jpezy_internal_generic_dev_core(jpezy_ctx* c, const int16_t* d_coeffs, const uint16_t qt[4][64], int ncomp, const uint8_t comp_h[3],
                                                   const uint8_t comp_v[3], const uint8_t comp_tq[3], int W, int H, int gray, int precision, uint8_t* d_r,
                                                   uint8_t* d_g, uint8_t* d_b, hipStream_t s, size_t* nblk_out, int n_frames = 1, size_t plane_stride = 0);
JPEZY_INTERNAL int jpezy_internal_dequant_idct_generic_impl(jpezy_ctx* c, const int16_t* coeffs, const uint16_t qt[4][64], int ncomp, const uint8_t comp_h[3],
                                                            const uint8_t comp_v[3], const uint8_t comp_tq[3], int W, int H, int gray, int precision,
                                                            uint8_t* r, uint8_t* g, uint8_t* b, bool coeffs_on_device = false);
// the GPU Huffman decoder over a list of independent streams, the device tables of one scan, the block structure of an MCU (jpezy_capi_huffdec.hip)
JPEZY_INTERNAL int jpezy_internal_huffdec_streams(jpezy_ctx* c, const std::vector<DevStream>& streams, const std::vector<jpezy_dev::huffdec::Setup>& setups,
                                                  const std::vector<char>& setup_usable, const StreamGeom& geom, int16_t* d_coef, size_t coef_elems,
                                                  std::vector<char>& ok, const std::function<void(const char*)>& lap, bool per_lane = false);
JPEZY_INTERNAL bool jpezy_internal_build_dev_setup(jpezy_dev::huffdec::Setup& S, const jpezy_host::ScanSetup& setup, const jpezy_frame_info& info, unsigned total_blocks);
JPEZY_INTERNAL StreamGeom jpezy_internal_stream_geom(const jpezy_frame_info& info);
}
