// jpezy_capi_multi.hip -- one batch of frames over several GPUs of one node, driven from ONE host process: the C-ABI entry
// jpezy_encode_batch_multi (include/jpezy_hip.h).  What is sharded is the loop a caller of the reference runs over independent
// encoder objects, one frame each (encoder/jpezy_encoder.hpp:38-77; inside a frame the MCU loop :55-67 is what the kernels replace);
// frames share nothing -- pre_DC and the bit cursor are per file (:180-181) -- so there is no data-path collective, only the
// gather of the results.
//
// One host thread per device.  A device encodes its contiguous shard (jpezy_shard_range, the rule of jpezy_amd/sharding.py) in
// chunks of chunk_frames on two alternating streams, each with a context of its own (the entropy stage keeps its scratch in the
// context): while chunk c runs its kernels, chunk c - 1's results travel -- to the root device over xGMI (hipMemcpyPeerAsync; the
// root's own chunks are written in place) when the consumer lives on that GPU, or straight to host memory over the device's own PCIe
// link.  .jpg files travel at their real length: their sizes come to the host first (a few bytes per frame), then one copy per file.
#include "jpezy_capi_internal.h"

#include <mutex>

namespace {

struct MultiJob {
    const int* devices;
    int n_dev;
    const uint8_t* r;
    const uint8_t* g;
    const uint8_t* b;
    int W, H, gray, n_frames, chunk;
    const char* comment;
    jpezy_multi_out out;
    size_t plane, cpf, dev_stride;      // bytes of a plane, int16 elements of a frame's coefficients, bytes reserved per .jpg on a device
};

struct Worker {
    int index = 0, dev = 0;
    long f0 = 0, nf = 0;
    int rc = JPEZY_OK;
    std::string err;
};

struct Slot {
    jpezy_ctx* ctx = nullptr;
    hipStream_t s = nullptr;
    DevBuf planes, coef, jpg, sizes;
    long long* h_sizes = nullptr;       // pinned
    long c_f0 = -1;                     // first frame and frames of the chunk whose .jpg files have not been sent on yet
    int c_nf = 0;
    bool jpg_in_place = false;
};

#define W_TRY(expr)                                                                  \
    do {                                                                             \
        hipError_t e__ = (expr);                                                     \
        if (e__ != hipSuccess) {                                                     \
            w.rc = JPEZY_E_HIP;                                                      \
            w.err = std::string(#expr) + ": " + hipGetErrorString(e__);              \
            return false;                                                            \
        }                                                                            \
    } while (0)
#define W_RC(expr)                                                                   \
    do {                                                                             \
        const int rc__ = (expr);                                                     \
        if (rc__ != JPEZY_OK) {                                                      \
            w.rc = rc__;                                                             \
            w.err = jpezy_hip_last_error();                                          \
            return false;                                                            \
        }                                                                            \
    } while (0)

// the .jpg files of the chunk a slot last coded: sizes to the caller's array, every file at its real length to its place
bool send_jpg(Worker& w, const MultiJob& J, Slot& sl, int root_dev)
{
    if (sl.c_f0 < 0) return true;
    W_TRY(hipStreamSynchronize(sl.s));                 // kernels of that chunk done, its sizes are in h_sizes
    for (int i = 0; i < sl.c_nf; ++i) {
        const long f = sl.c_f0 + i;
        const long long n = sl.h_sizes[i];
        J.out.jpg_sizes[f] = n;
        if (n <= 0 || sl.jpg_in_place) continue;        // a refused frame keeps its negative status; files written in place need no copy
        uint8_t* dst = J.out.jpg + (size_t)f * J.out.jpg_stride;
        const uint8_t* src = (const uint8_t*)sl.jpg.p + (size_t)i * J.dev_stride;
        if ((size_t)n > J.out.jpg_stride) { J.out.jpg_sizes[f] = JPEZY_E_NOSPACE; continue; }
        if (J.out.on_root_device)
            W_TRY(hipMemcpyPeerAsync(dst, root_dev, src, w.dev, (size_t)n, sl.s));
        else
            W_TRY(hipMemcpyAsync(dst, src, (size_t)n, hipMemcpyDeviceToHost, sl.s));
    }
    sl.c_f0 = -1;
    return true;
}

bool run_shard(Worker& w, const MultiJob& J, Slot (&slot)[2])
{
    const int root_dev = J.devices[0];
    const bool root = w.index == 0;
    W_TRY(hipSetDevice(w.dev));
    if (J.out.on_root_device && w.dev != root_dev) {
        const hipError_t e = hipDeviceEnablePeerAccess(root_dev, 0);        // xGMI peers: direct; without it the copies are staged by the runtime
        if (e != hipSuccess) (void)hipGetLastError();                       // already enabled / not supported: hipMemcpyPeerAsync works either way
    }
    const int chunk = J.chunk;
    for (Slot& sl : slot) {
        sl.ctx = jpezy_ctx_create(w.dev);
        if (!sl.ctx) { w.rc = JPEZY_E_HIP; w.err = jpezy_hip_last_error(); return false; }
        W_TRY(hipStreamCreateWithFlags(&sl.s, hipStreamNonBlocking));
        W_RC(sl.planes.reserve(3 * J.plane * (size_t)chunk));
        if (J.out.jpg) {
            W_RC(sl.sizes.reserve(sizeof(long long) * (size_t)chunk));
            W_TRY(hipHostMalloc((void**)&sl.h_sizes, sizeof(long long) * (size_t)chunk));
        }
    }
    long c = 0;
    for (long f = w.f0; f < w.f0 + w.nf; f += chunk, ++c) {
        Slot& sl = slot[c & 1];
        const int nf = (int)std::min<long>(chunk, w.f0 + w.nf - f);
        if (!send_jpg(w, J, sl, root_dev)) return false;      // (nothing left normally: sent when the following chunk was launched)
        W_TRY(hipStreamSynchronize(sl.s));                    // the slot's buffers are free again
        uint8_t* dp = (uint8_t*)sl.planes.p;
        const uint8_t* src[3] = { J.r, J.g, J.b };
        for (int q = 0; q < 3; ++q)
            W_TRY(hipMemcpyAsync(dp + (size_t)q * J.plane * chunk, src[q] + (size_t)f * J.plane, J.plane * (size_t)nf, hipMemcpyHostToDevice, sl.s));
        // coefficients: in place when this is the root and the consumer lives on it, otherwise into the slot and on from there
        const bool coef_in_place = J.out.coeffs && J.out.on_root_device && root;
        int16_t* d_coef;
        if (coef_in_place) d_coef = J.out.coeffs + (size_t)f * J.cpf;
        else {
            W_RC(sl.coef.reserve(J.cpf * sizeof(int16_t) * (size_t)chunk));
            d_coef = (int16_t*)sl.coef.p;
        }
        W_RC(jpezy_fdct_quant_dev(sl.ctx, dp, dp + J.plane * chunk, dp + 2 * J.plane * chunk, J.plane, J.W, J.H, J.gray, nf, d_coef, sl.s));
        if (J.out.coeffs && !coef_in_place) {
            int16_t* dst = J.out.coeffs + (size_t)f * J.cpf;
            const size_t bytes = J.cpf * sizeof(int16_t) * (size_t)nf;
            if (J.out.on_root_device) W_TRY(hipMemcpyPeerAsync(dst, root_dev, d_coef, w.dev, bytes, sl.s));
            else W_TRY(hipMemcpyAsync(dst, d_coef, bytes, hipMemcpyDeviceToHost, sl.s));
        }
        if (J.out.jpg) {
            sl.jpg_in_place = J.out.on_root_device && root;
            uint8_t* d_jpg;
            size_t stride;
            if (sl.jpg_in_place) { d_jpg = J.out.jpg + (size_t)f * J.out.jpg_stride; stride = J.out.jpg_stride; }
            else {
                W_RC(sl.jpg.reserve(J.dev_stride * (size_t)chunk));
                d_jpg = (uint8_t*)sl.jpg.p;
                stride = J.dev_stride;
            }
            W_RC(jpezy_write_jpeg_gpu_dev(sl.ctx, d_coef, J.W, J.H, J.gray, nf, J.comment, d_jpg, stride, (long long*)sl.sizes.p, sl.s));
            W_TRY(hipMemcpyAsync(sl.h_sizes, sl.sizes.p, sizeof(long long) * (size_t)nf, hipMemcpyDeviceToHost, sl.s));
            sl.c_f0 = f;
            sl.c_nf = nf;
        }
        // the chunk before this one has had this chunk's launch time to finish: send its files on while this chunk computes
        if (c > 0 && !send_jpg(w, J, slot[(c - 1) & 1], root_dev)) return false;
    }
    for (Slot& sl : slot)
        if (!send_jpg(w, J, sl, root_dev)) return false;
    for (Slot& sl : slot) W_TRY(hipStreamSynchronize(sl.s));
    return true;
}

void worker_main(Worker& w, const MultiJob& J)
{
    Slot slot[2];
    try {
        (void)run_shard(w, J, slot);
    } catch (const std::bad_alloc&) {
        w.rc = JPEZY_E_NOSPACE; w.err = "out of host memory";
    } catch (const std::exception& e) {
        w.rc = JPEZY_E_HIP; w.err = std::string("unexpected exception: ") + e.what();
    }
    (void)hipSetDevice(w.dev);
    for (Slot& sl : slot) {
        if (sl.s) { (void)hipStreamSynchronize(sl.s); (void)hipStreamDestroy(sl.s); }
        if (sl.h_sizes) (void)hipHostFree(sl.h_sizes);
        for (DevBuf* b : { &sl.planes, &sl.coef, &sl.jpg, &sl.sizes }) b->release();
        if (sl.ctx) jpezy_ctx_destroy(sl.ctx);
    }
}

}  // namespace

extern "C" {

void jpezy_shard_range(long n_units, int n_shards, int k, long* first, long* count)
{
    long lo = 0, n = 0;
    if (n_units > 0 && n_shards > 0 && k >= 0 && k < n_shards) {
        const long base = n_units / n_shards, extra = n_units % n_shards;
        lo = k * base + std::min<long>(k, extra);
        n = base + (k < extra ? 1 : 0);
    }
    if (first) *first = lo;
    if (count) *count = n;
}

int jpezy_encode_batch_multi(const int* devices, int n_dev, const uint8_t* r, const uint8_t* g, const uint8_t* b, int W, int H, int gray,
                             int n_frames, int chunk_frames, const char* comment, const jpezy_multi_out* out)
try {
    if (!devices || n_dev <= 0 || n_dev > 64) return set_err(JPEZY_E_BADARG, "encode_batch_multi: 1..64 devices");
    if (!r || !g || !b || !out) return set_err(JPEZY_E_BADARG, "encode_batch_multi: null pointer");
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return set_err(JPEZY_E_BADARG, "width/height must be in 1..65535 (16-bit SOF0 fields)");
    if (n_frames <= 0) return set_err(JPEZY_E_BADARG, "n_frames must be positive");
    if (!out->coeffs && !out->jpg) return set_err(JPEZY_E_BADARG, "encode_batch_multi: neither coefficients nor .jpg files asked for");
    if (out->jpg && (!out->jpg_sizes || out->jpg_stride == 0)) return set_err(JPEZY_E_BADARG, "encode_batch_multi: jpg needs jpg_sizes and jpg_stride");
    if (out->on_root_device && out->coeffs && !aligned16(out->coeffs)) return set_err(JPEZY_E_BADARG, "encode_batch_multi: coeffs on the root device must be 16-byte aligned");
    const int have = jpezy_hip_device_count();
    if (have <= 0) return set_err(JPEZY_E_NODEVICE, "no HIP device (the jpezy hot path has no CPU fallback)");
    for (int i = 0; i < n_dev; ++i)
        if (devices[i] < 0 || devices[i] >= have) return set_err(JPEZY_E_NODEVICE, "encode_batch_multi: device index out of range");
    MultiJob J;
    J.devices = devices; J.n_dev = n_dev;
    J.r = r; J.g = g; J.b = b;
    J.W = W; J.H = H; J.gray = gray != 0; J.n_frames = n_frames;
    J.chunk = chunk_frames > 0 ? chunk_frames : 16;
    J.comment = comment;
    J.out = *out;
    J.plane = (size_t)W * H;
    J.cpf = jpezy_coeff_count(W, H, gray);
    J.dev_stride = (jpezy_jpeg_bound(W, H) + 15) & ~(size_t)15;
    if (out->jpg)
        for (int f = 0; f < n_frames; ++f) out->jpg_sizes[f] = 0;       // (a shard that fails leaves its frames at 0, never at garbage)
    std::vector<Worker> workers((size_t)n_dev);
    for (int i = 0; i < n_dev; ++i) {
        workers[(size_t)i].index = i;
        workers[(size_t)i].dev = devices[i];
        jpezy_shard_range(n_frames, n_dev, i, &workers[(size_t)i].f0, &workers[(size_t)i].nf);
    }
    std::vector<std::thread> pool;
    for (int i = 1; i < n_dev; ++i)
        if (workers[(size_t)i].nf > 0) pool.emplace_back(worker_main, std::ref(workers[(size_t)i]), std::cref(J));
    if (workers[0].nf > 0) worker_main(workers[0], J);      // the calling thread drives the root device
    for (std::thread& t : pool) t.join();
    for (const Worker& w : workers)
        if (w.rc != JPEZY_OK) return set_err(w.rc, "encode_batch_multi, device " + std::to_string(w.dev) + " (shard " + std::to_string(w.index) + "): " + w.err);
    if (out->jpg)
        for (int f = 0; f < n_frames; ++f)
            if (out->jpg_sizes[f] < 0) return set_err(JPEZY_E_FORMAT, "encode_batch_multi: at least one frame failed (see jpg_sizes[])");
    return JPEZY_OK;
}
JPEZY_CATCH

}  // extern "C"
