// jpezy_capi_multi.hip -- batches of frames over several GPUs of one node, driven from ONE host process: the C-ABI entries
// jpezy_multi_create / jpezy_multi_encode / jpezy_multi_destroy and the one-shot jpezy_encode_batch_multi (include/jpezy_hip.h).
// What is sharded is the loop a caller of the reference runs over independent encoder objects, one frame each
// (encoder/jpezy_encoder.hpp:38-77; inside a frame the MCU loop :55-67 is what the kernels replace); frames share nothing -- pre_DC
// and the bit cursor are per file (:180-181) -- so there is no data-path collective, only the gather of the results.
//
// A handle owns one LANE per entry of `devices`: a context, an upload stream, a download stream per drainer and a ring of RING slots,
// each a pinned host buffer + a device buffer per direction.  A call shards the frames over the lanes (jpezy_shard_range) and every
// lane streams its shard through its ring in chunks of chunk_frames:
//
//   feeder threads   memcpy caller planes -> pinned slot, hipMemcpyAsync H2D on the upload stream, event
//                    (planes the caller has pinned itself -- hipHostMalloc / hipHostRegister -- go to the DMA engine as they are)
//   lane thread      waits (stream-side) for the upload, enqueues FDCT + Huffman stage + the chunk's file sizes on the context's stream
//   drainer threads  wait for the chunk's kernels, then move its results at their REAL length: .jpg files and coefficients through the
//                    pinned slot into the caller's host memory, or -- consumer on the root GPU -- by hipMemcpyPeerAsync into devices[0]'s
//                    memory over xGMI (the root lane's own results are written in place)
//
// so that upload of chunk c + 1, kernels of chunk c and download of chunk c - 1 overlap, PCIe runs in both directions, the caller's
// pageable memory never reaches the DMA engines (round 5 uploaded straight from it: 7.2 GB/s against the 46-50 GB/s of pinned
// memory, jpezy_hostpipe.h) and nothing is allocated inside a call once the handle is warm.
//
// The gather is hipMemcpyPeerAsync, not RCCL: one process owns every device, so a peer copy IS the point-to-point transfer over
// xGMI -- no communicator, no second library in the product's link line.  (RCCL is what the multi-process harness uses:
// jpezy_amd/sharding.py, bench.py --gpus N.)
#include "jpezy_capi_internal.h"

#include <condition_variable>
#include <memory>
#include <mutex>

namespace {

constexpr int RING = 6, MAX_FEED = 6, DEFAULT_FEED = 4, MAX_DRAIN = 4;

struct Job {                            // one jpezy_multi_encode call
    const uint8_t* src[3] = {};
    bool src_pinned = false;            // the caller's planes are pinned host memory: no staging copy
    int n_frames = 0;
    const char* comment = nullptr;
    jpezy_multi_out out = {};
};

struct Slot {
    uint8_t* pin_in = nullptr;          // [3][chunk][plane]
    DevBuf dev_in, dev_coef, dev_jpg, dev_sizes;
    long long* pin_sizes = nullptr;
    uint8_t* pin_coef = nullptr;
    uint8_t* pin_jpg = nullptr;         // files of a chunk, packed at their real lengths (grows on demand)
    size_t pin_jpg_cap = 0;
    hipEvent_t ev_up = nullptr, ev_k0 = nullptr, ev_k = nullptr;
};

struct Lane {
    int index = 0, dev = 0;
    jpezy_ctx* ctx = nullptr;
    hipStream_t s_up = nullptr, s_down[MAX_DRAIN] = {};
    Slot slot[RING];
    int ring = RING;                    // slots in use (the one-shot form of a small batch builds no more than its chunks need)
    bool peer_enabled = false;
    // per call
    long f0 = 0, nf = 0;
    int rc = JPEZY_OK;
    std::string err;
    jpezy_multi_lane_stats stats = {};
};

}  // namespace

struct jpezy_multi {
    std::vector<int> devices;
    int W = 0, H = 0, gray = 0, chunk = 1;
    int n_feed = 4;                             // feeder threads per lane: a core copies ~11 GB/s into pinned memory, the link takes ~50
    size_t plane = 0, cpf = 0, bound = 0;       // bytes of a plane, int16 elements of a frame's coefficients, jpezy_jpeg_bound rounded to 16
    std::vector<std::unique_ptr<Lane>> lanes;
};

namespace {

// the current device of the calling thread is the caller's business: every entry point puts it back
struct DeviceRestore {
    int dev = -1;
    DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) { dev = -1; (void)hipGetLastError(); } }
    ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
};

// joined on every way out of a scope; an exception on the way raises the lanes' failure flag first (see jpezy_hostpipe.h)
struct Joiner {
    std::vector<std::thread>& ts;
    std::function<void()> on_unwind;
    ~Joiner()
    {
        bool running = false;
        for (auto& t : ts) running = running || t.joinable();
        if (running && std::uncaught_exceptions() > 0 && on_unwind) on_unwind();
        for (auto& t : ts) if (t.joinable()) t.join();
    }
};

void release_lane(Lane& L)
{
    if (hipSetDevice(L.dev) != hipSuccess) (void)hipGetLastError();
    if (L.ctx) (void)jpezy_ctx_sync(L.ctx);
    if (L.s_up) { (void)hipStreamSynchronize(L.s_up); (void)hipStreamDestroy(L.s_up); }
    for (hipStream_t& s : L.s_down) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); s = nullptr; }
    L.s_up = nullptr;
    for (Slot& sl : L.slot) {
        if (sl.pin_in) (void)hipHostFree(sl.pin_in);
        if (sl.pin_sizes) (void)hipHostFree(sl.pin_sizes);
        if (sl.pin_coef) (void)hipHostFree(sl.pin_coef);
        if (sl.pin_jpg) (void)hipHostFree(sl.pin_jpg);
        sl.pin_in = sl.pin_coef = sl.pin_jpg = nullptr;
        sl.pin_sizes = nullptr;
        sl.pin_jpg_cap = 0;
        for (DevBuf* b : { &sl.dev_in, &sl.dev_coef, &sl.dev_jpg, &sl.dev_sizes }) b->release();
        for (hipEvent_t* e : { &sl.ev_up, &sl.ev_k0, &sl.ev_k }) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
    }
    if (L.ctx) jpezy_ctx_destroy(L.ctx);
    L.ctx = nullptr;
}

#define L_TRY(expr)                                                                  \
    do {                                                                             \
        hipError_t e__ = (expr);                                                     \
        if (e__ != hipSuccess) return hip_err(e__, #expr);                           \
    } while (0)

// what every call needs whatever it asks for: context, streams, events, the input ring, the coefficient buffers
int create_lane(const jpezy_multi& M, Lane& L)
{
    L_TRY(hipSetDevice(L.dev));
    L.ctx = jpezy_ctx_create(L.dev);
    if (!L.ctx) return JPEZY_E_HIP;                     // (message set by jpezy_ctx_create)
    L_TRY(hipStreamCreateWithFlags(&L.s_up, hipStreamNonBlocking));
    for (hipStream_t& s : L.s_down) L_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t in_bytes = 3 * M.plane * (size_t)M.chunk;
    for (int k = 0; k < L.ring; ++k) {
        Slot& sl = L.slot[k];
        L_TRY(hipEventCreateWithFlags(&sl.ev_up, hipEventDisableTiming));
        L_TRY(hipEventCreate(&sl.ev_k0));               // (timed: the lane's kernel time is the sum of its chunks' spans)
        L_TRY(hipEventCreate(&sl.ev_k));
        if (int rc = sl.dev_in.reserve(in_bytes)) return rc;
        if (int rc = sl.dev_coef.reserve(M.cpf * sizeof(int16_t) * (size_t)M.chunk)) return rc;
        if (int rc = sl.dev_sizes.reserve(sizeof(long long) * (size_t)M.chunk)) return rc;
        L_TRY(hipHostMalloc((void**)&sl.pin_sizes, sizeof(long long) * (size_t)M.chunk, hipHostMallocDefault));
    }
    return JPEZY_OK;
}

// per-call shared state of one lane's threads
struct LaneRun {
    std::mutex mu, up_mu;
    std::condition_variable cv;
    std::vector<int> state;             // per chunk: 0 nothing, 1 upload issued, 2 kernels enqueued, 3 delivered (guarded by mu)
    std::atomic<int> failed{ 0 };       // a jpezy_status
    std::string err;

    void set(int c, int v) { { std::lock_guard<std::mutex> lk(mu); state[(size_t)c] = v; } cv.notify_all(); }
    bool wait(int c, int v)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return state[(size_t)c] >= v || failed.load(); });
        return !failed.load();
    }
    void fail(int code, const std::string& what)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!failed.load()) { err = what; failed.store(code); }
        }
        cv.notify_all();
    }
    bool hip(hipError_t e, const char* what)
    {
        if (e == hipSuccess) return true;
        fail(JPEZY_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
        return false;
    }
};
#define R_TRY(expr) do { if (!R.hip((expr), #expr)) return; } while (0)

void run_lane(const jpezy_multi& M, Lane& L, const Job& J)
{
    const auto t_begin = std::chrono::steady_clock::now();
    const int chunk = M.chunk;
    const int n_chunks = (int)((L.nf + chunk - 1) / chunk);
    const int root_dev = M.devices[0], ring = L.ring, n_feed = M.n_feed;
    const bool to_root = J.out.on_root_device != 0;
    const bool in_place = to_root && L.index == 0;                      // the root lane writes its results where they belong
    const bool want_jpg = J.out.jpg != nullptr, want_coef = J.out.coeffs != nullptr;
    // bytes reserved per file in a slot's device buffer: never more than the caller gives a file (a longer one is refused anyway)
    const size_t dstride = std::min(M.bound, (J.out.jpg_stride + 15) & ~(size_t)15);
    // drainers: two bring a tenth of the upload's bytes back (.jpg files); coefficients to host memory are as many bytes as the planes
    // and want as many copying threads as the way up
#ifdef JPEZY_MULTI_FIXED_DRAIN       // A/B builds (tools/ab/ab_build.py)
    const int n_drain = JPEZY_MULTI_FIXED_DRAIN;
#else
    const int n_drain = want_coef && !to_root ? std::min(MAX_DRAIN, std::max(2, n_feed)) : 2;
#endif
    L.stats = {};
    L.stats.device = L.dev;
    L.stats.frames = L.nf;
    L.stats.staged = J.src_pinned ? 0 : 1;
    LaneRun R;
    R.state.assign((size_t)n_chunks, 0);
    if (!R.hip(hipSetDevice(L.dev), "hipSetDevice")) { L.rc = R.failed.load(); L.err = R.err; return; }

    // buffers that depend on what this call asks for (kept by the handle afterwards)
    {
        set_err(JPEZY_OK, "");
        int rc = JPEZY_OK;
        hipError_t e = hipSuccess;
        if (to_root && !in_place && L.dev != root_dev && !L.peer_enabled) {
            if (hipDeviceEnablePeerAccess(root_dev, 0) != hipSuccess) (void)hipGetLastError();   // already enabled / not supported: hipMemcpyPeerAsync works either way
            L.peer_enabled = true;
        }
        for (int k = 0; k < L.ring; ++k) {
            Slot& sl = L.slot[k];
            if (rc || e != hipSuccess) break;
            if (!J.src_pinned && !sl.pin_in) e = hipHostMalloc((void**)&sl.pin_in, 3 * M.plane * (size_t)chunk, hipHostMallocDefault);
            if (e == hipSuccess && want_jpg && !in_place) rc = sl.dev_jpg.reserve(dstride * (size_t)chunk);
            if (!rc && e == hipSuccess && want_coef && !to_root && !sl.pin_coef)
                e = hipHostMalloc((void**)&sl.pin_coef, M.cpf * sizeof(int16_t) * (size_t)chunk, hipHostMallocDefault);
        }
        if (e != hipSuccess) { L.rc = JPEZY_E_HIP; L.err = std::string("hipHostMalloc (staging ring): ") + hipGetErrorString(e); return; }
        if (rc) { L.rc = rc; L.err = jpezy_hip_last_error(); return; }
    }

    std::atomic<unsigned long long> bytes_up{ 0 }, bytes_down{ 0 };
    std::mutex stat_mu;
    double kernel_ms = 0;

    auto chunk_frames = [&](int c) { return (int)std::min<long>(chunk, L.nf - (long)c * chunk); };

    auto feeder = [&](int id) {
        R_TRY(hipSetDevice(L.dev));
        for (int c = id; c < n_chunks && !R.failed.load(); c += n_feed) {
            Slot& sl = L.slot[c % ring];
            if (c >= ring) {                                            // the slot's input buffers are free once the kernels of the chunk that used them last have run
                if (!R.wait(c - ring, 2)) return;
                R_TRY(hipEventSynchronize(sl.ev_k));
            }
            const int n = chunk_frames(c);
            const long f = L.f0 + (long)c * chunk;
            const size_t seg = M.plane * (size_t)n, qs = M.plane * (size_t)chunk;
            uint8_t* d_in = (uint8_t*)sl.dev_in.p;
            if (!J.src_pinned)
                for (int q = 0; q < 3; ++q) std::memcpy(sl.pin_in + (size_t)q * qs, J.src[q] + (size_t)f * M.plane, seg);
            {
                std::lock_guard<std::mutex> lk(R.up_mu);                // one stream, several feeders: keep copies + event together
                for (int q = 0; q < 3; ++q) {
                    const uint8_t* from = J.src_pinned ? J.src[q] + (size_t)f * M.plane : sl.pin_in + (size_t)q * qs;
                    R_TRY(hipMemcpyAsync(d_in + (size_t)q * qs, from, seg, hipMemcpyHostToDevice, L.s_up));
                }
                R_TRY(hipEventRecord(sl.ev_up, L.s_up));
            }
            bytes_up += 3 * seg;
            R.set(c, 1);
        }
    };

    auto drainer = [&](int id) {
        R_TRY(hipSetDevice(L.dev));
        hipStream_t sd = L.s_down[id];
        for (int c = id; c < n_chunks && !R.failed.load(); c += n_drain) {
            Slot& sl = L.slot[c % ring];
            if (!R.wait(c, 2)) return;
            R_TRY(hipEventSynchronize(sl.ev_k));                        // the chunk's kernels are done, its sizes are in pin_sizes
            {
                float ms = 0;
                if (hipEventElapsedTime(&ms, sl.ev_k0, sl.ev_k) == hipSuccess) { std::lock_guard<std::mutex> lk(stat_mu); kernel_ms += ms; }
                else (void)hipGetLastError();
            }
            const int n = chunk_frames(c);
            const long f = L.f0 + (long)c * chunk;
            size_t moved = 0;
            if (want_coef && !in_place) {
                const size_t bytes = M.cpf * sizeof(int16_t) * (size_t)n;
                int16_t* dst = J.out.coeffs + (size_t)f * M.cpf;
                if (to_root) R_TRY(hipMemcpyPeerAsync(dst, root_dev, sl.dev_coef.p, L.dev, bytes, sd));
                else R_TRY(hipMemcpyAsync(sl.pin_coef, sl.dev_coef.p, bytes, hipMemcpyDeviceToHost, sd));
                moved += bytes;
            }
            std::vector<size_t> off;
            if (want_jpg) {
                size_t total = 0;
                off.assign((size_t)n, 0);
                for (int i = 0; i < n; ++i) {
                    long long len = sl.pin_sizes[i];
                    if (len > 0 && (size_t)len > J.out.jpg_stride) len = JPEZY_E_NOSPACE;
                    J.out.jpg_sizes[f + i] = len;                       // a refused frame keeps its negative status
                    off[(size_t)i] = total;
                    if (len > 0) total += ((size_t)len + 63) & ~(size_t)63;
                }
                if (!in_place && !to_root && total > sl.pin_jpg_cap) {  // (first chunks of a handle, or content that codes longer than any before)
                    if (sl.pin_jpg) (void)hipHostFree(sl.pin_jpg);
                    sl.pin_jpg = nullptr;
                    sl.pin_jpg_cap = 0;
                    const size_t cap = std::max(total + total / 4, M.plane * (size_t)chunk / 2);
                    R_TRY(hipHostMalloc((void**)&sl.pin_jpg, cap, hipHostMallocDefault));
                    sl.pin_jpg_cap = cap;
                }
                if (!in_place)
                    for (int i = 0; i < n; ++i) {
                        const long long len = J.out.jpg_sizes[f + i];
                        if (len <= 0) continue;
                        const uint8_t* src = (const uint8_t*)sl.dev_jpg.p + (size_t)i * dstride;
                        if (to_root) R_TRY(hipMemcpyPeerAsync(J.out.jpg + (size_t)(f + i) * J.out.jpg_stride, root_dev, src, L.dev, (size_t)len, sd));
                        else R_TRY(hipMemcpyAsync(sl.pin_jpg + off[(size_t)i], src, (size_t)len, hipMemcpyDeviceToHost, sd));
                        moved += (size_t)len;
                    }
            }
            if (!in_place) R_TRY(hipStreamSynchronize(sd));
            if (!to_root) {
                if (want_coef) std::memcpy(J.out.coeffs + (size_t)f * M.cpf, sl.pin_coef, M.cpf * sizeof(int16_t) * (size_t)n);
                if (want_jpg)
                    for (int i = 0; i < n; ++i) {
                        const long long len = J.out.jpg_sizes[f + i];
                        if (len > 0) std::memcpy(J.out.jpg + (size_t)(f + i) * J.out.jpg_stride, sl.pin_jpg + off[(size_t)i], (size_t)len);
                    }
            }
            bytes_down += moved;
            R.set(c, 3);
        }
    };

    std::vector<std::thread> threads;
    {
        Joiner joiner{ threads, [&] { R.fail(JPEZY_E_HIP, "unexpected exception in the lane thread"); } };
        threads.reserve((size_t)(n_feed + n_drain));
        try {
            for (int k = 0; k < std::min(n_feed, n_chunks); ++k) threads.emplace_back(feeder, k);
            for (int k = 0; k < std::min(n_drain, n_chunks); ++k) threads.emplace_back(drainer, k);
        } catch (const std::exception&) {
            R.fail(JPEZY_E_HIP, "starting a copy thread failed");
        }
        hipStream_t sc = (hipStream_t)jpezy_ctx_stream(L.ctx);
        for (int c = 0; c < n_chunks && !R.failed.load(); ++c) {
            Slot& sl = L.slot[c % ring];
            if (!R.wait(c, 1)) break;
            if (c >= ring && !R.wait(c - ring, 3)) break;               // the slot's output buffers have been delivered
            const int n = chunk_frames(c);
            const long f = L.f0 + (long)c * chunk;
            const size_t qs = M.plane * (size_t)chunk;
            uint8_t* d_in = (uint8_t*)sl.dev_in.p;
            int16_t* d_coef = want_coef && in_place ? J.out.coeffs + (size_t)f * M.cpf : (int16_t*)sl.dev_coef.p;
            hipError_t e = hipStreamWaitEvent(sc, sl.ev_up, 0);
            if (e == hipSuccess) e = hipEventRecord(sl.ev_k0, sc);
            if (e != hipSuccess) { R.hip(e, "hipStreamWaitEvent / hipEventRecord"); break; }
            int rc = jpezy_fdct_quant_dev(L.ctx, d_in, d_in + qs, d_in + 2 * qs, M.plane, M.W, M.H, M.gray, n, d_coef, sc);
            if (rc == JPEZY_OK && want_jpg) {
                uint8_t* d_jpg = in_place ? J.out.jpg + (size_t)f * J.out.jpg_stride : (uint8_t*)sl.dev_jpg.p;
                rc = jpezy_write_jpeg_gpu_dev(L.ctx, d_coef, M.W, M.H, M.gray, n, J.comment, d_jpg, in_place ? J.out.jpg_stride : dstride,
                                              (long long*)sl.dev_sizes.p, sc);
                if (rc == JPEZY_OK) {
                    e = hipMemcpyAsync(sl.pin_sizes, sl.dev_sizes.p, sizeof(long long) * (size_t)n, hipMemcpyDeviceToHost, sc);
                    if (e != hipSuccess) { R.hip(e, "hipMemcpyAsync (file sizes)"); break; }
                }
            }
            if (rc != JPEZY_OK) { R.fail(rc, jpezy_hip_last_error()); break; }
            e = hipEventRecord(sl.ev_k, sc);
            if (e != hipSuccess) { R.hip(e, "hipEventRecord"); break; }
            R.set(c, 2);
        }
    }   // threads joined
    (void)hipStreamSynchronize((hipStream_t)jpezy_ctx_stream(L.ctx));
    if (R.failed.load()) {
        (void)hipStreamSynchronize(L.s_up);
        for (hipStream_t s : L.s_down) (void)hipStreamSynchronize(s);
        L.rc = R.failed.load();
        L.err = R.err;
    }
    L.stats.kernel_ms = kernel_ms;
    L.stats.bytes_up = bytes_up.load();
    L.stats.bytes_down = bytes_down.load();
    L.stats.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
}

void lane_main(const jpezy_multi& M, Lane& L, const Job& J)
{
    try {
        run_lane(M, L, J);
    } catch (const std::bad_alloc&) {
        L.rc = JPEZY_E_NOSPACE; L.err = "out of host memory";
    } catch (const std::exception& e) {
        L.rc = JPEZY_E_HIP; L.err = std::string("unexpected exception: ") + e.what();
    }
}

bool host_pinned(const void* p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }    // ordinary pageable memory: "invalid value"
    return a.type == hipMemoryTypeHost;
}

int check_devices(const int* devices, int n_dev, const char* who)
{
    if (!devices || n_dev <= 0 || n_dev > 64) return set_err(JPEZY_E_BADARG, std::string(who) + ": 1..64 devices");
    return JPEZY_OK;
}
int check_device_range(const int* devices, int n_dev, const char* who)
{
    const int have = jpezy_hip_device_count();
    if (have <= 0) return set_err(JPEZY_E_NODEVICE, "no HIP device (the jpezy hot path has no CPU fallback)");
    for (int i = 0; i < n_dev; ++i)
        if (devices[i] < 0 || devices[i] >= have) return set_err(JPEZY_E_NODEVICE, std::string(who) + ": device index out of range");
    return JPEZY_OK;
}
int check_out(const jpezy_multi_out* out, const char* who)
{
    if (!out) return set_err(JPEZY_E_BADARG, std::string(who) + ": null pointer");
    if (!out->coeffs && !out->jpg) return set_err(JPEZY_E_BADARG, std::string(who) + ": neither coefficients nor .jpg files asked for");
    if (out->jpg && (!out->jpg_sizes || out->jpg_stride == 0)) return set_err(JPEZY_E_BADARG, std::string(who) + ": jpg needs jpg_sizes and jpg_stride");
    if (out->on_root_device && out->coeffs && !aligned16(out->coeffs))
        return set_err(JPEZY_E_BADARG, std::string(who) + ": coeffs on the root device must be 16-byte aligned");
    return JPEZY_OK;
}

// frames per chunk when the caller does not say: about 28 MB of planes (1080p: 4 frames, 4096^2: 1).  tools/measure/native_multi_sweep.py
// on one device, 256 frames 1080p, GB/s of upload by (frames per chunk, feeder threads): 1 frame 39-40 whatever the feeders; 2 frames 42 / 47 /
// 47 with 2 / 3 / 4 feeders; 4 frames 42.7 / 49.3 / 49.2; 8 frames 49.4 / 48.6 / 47.5 -- against 50.3 for a pinned hipMemcpy
// (profiles/r06_native_multi_sweep.txt).  Three feeders keep a link busy; a ring of RING slots of that size is ~150 MB of pinned memory per lane.
int default_chunk(size_t plane)
{
    const size_t per = ((size_t)28 << 20) / std::max<size_t>(3 * plane, 1);
    return (int)std::min<size_t>(std::max<size_t>(per, 1), 64);
}

jpezy_multi* multi_create(const int* devices, int n_dev, int W, int H, int gray, int chunk_frames, long frames_hint)
{
    std::unique_ptr<jpezy_multi> M(new jpezy_multi);
    M->devices.assign(devices, devices + n_dev);
    M->W = W; M->H = H; M->gray = gray != 0;
    M->plane = (size_t)W * H;
    M->cpf = jpezy_coeff_count(W, H, gray);
    M->bound = (jpezy_jpeg_bound(W, H) + 15) & ~(size_t)15;
    M->chunk = chunk_frames > 0 ? chunk_frames : default_chunk(M->plane);
    {   // feeders: what the host's cores allow when every lane runs its own (a lane also has its thread and two drainers)
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        M->n_feed = (int)std::min<unsigned>(DEFAULT_FEED, std::max<unsigned>(2, hw / (2u * (unsigned)n_dev)));
    }
    if (frames_hint > 0) {                              // one-shot form: no ring slot larger than the largest shard
        const long largest = (frames_hint + n_dev - 1) / n_dev;
        M->chunk = (int)std::max<long>(1, std::min<long>(M->chunk, largest));
    }
    for (int i = 0; i < n_dev; ++i) {
        M->lanes.emplace_back(new Lane);
        M->lanes.back()->index = i;
        M->lanes.back()->dev = devices[i];
        if (frames_hint > 0) {                          // ... and no more slots than the shard has chunks
            long f0 = 0, nf = 0;
            jpezy_shard_range(frames_hint, n_dev, i, &f0, &nf);
            M->lanes.back()->ring = (int)std::max<long>(1, std::min<long>(RING, (nf + M->chunk - 1) / M->chunk));
        }
    }
    for (auto& L : M->lanes) {
        const int rc = create_lane(*M, *L);
        if (rc != JPEZY_OK) {
            const std::string why = jpezy_hip_last_error();
            for (auto& K : M->lanes) release_lane(*K);
            set_err(rc, "multi_create, device " + std::to_string(L->dev) + " (lane " + std::to_string(L->index) + "): " + why);
            return nullptr;
        }
    }
    return M.release();
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

jpezy_multi* jpezy_multi_create(const int* devices, int n_dev, int W, int H, int gray, int chunk_frames)
try {
    if (check_devices(devices, n_dev, "multi_create")) return nullptr;
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) { set_err(JPEZY_E_BADARG, "width/height must be in 1..65535 (16-bit SOF0 fields)"); return nullptr; }
    if (check_device_range(devices, n_dev, "multi_create")) return nullptr;
    DeviceRestore restore;
    return multi_create(devices, n_dev, W, H, gray, chunk_frames, 0);
} catch (const std::exception& e) {
    set_err(JPEZY_E_NOSPACE, std::string("multi_create: ") + e.what());
    return nullptr;
}

void jpezy_multi_destroy(jpezy_multi* m)
{
    if (!m) return;
    DeviceRestore restore;
    for (auto& L : m->lanes) release_lane(*L);
    delete m;
}

int jpezy_multi_chunk_frames(const jpezy_multi* m) { return m ? m->chunk : 0; }
int jpezy_multi_feeder_threads(const jpezy_multi* m) { return m ? m->n_feed : 0; }
int jpezy_multi_set_feeder_threads(jpezy_multi* m, int n)
{
    if (!m) return set_err(JPEZY_E_BADARG, "multi_set_feeder_threads: null handle");
    if (n < 1 || n > MAX_FEED) return set_err(JPEZY_E_BADARG, "multi_set_feeder_threads: 1.." + std::to_string(MAX_FEED));
    m->n_feed = n;
    return JPEZY_OK;
}

int jpezy_multi_encode(jpezy_multi* m, const uint8_t* r, const uint8_t* g, const uint8_t* b, int n_frames, const char* comment,
                       const jpezy_multi_out* out)
try {
    if (!m) return set_err(JPEZY_E_BADARG, "multi_encode: null handle");
    if (!r || !g || !b) return set_err(JPEZY_E_BADARG, "multi_encode: null pointer");
    if (int rc = check_out(out, "multi_encode")) return rc;
    if (n_frames <= 0) return set_err(JPEZY_E_BADARG, "n_frames must be positive");
    DeviceRestore restore;
    Job J;
    J.src[0] = r; J.src[1] = g; J.src[2] = b;
    J.src_pinned = host_pinned(r) && host_pinned(g) && host_pinned(b);
    J.n_frames = n_frames;
    J.comment = comment;
    J.out = *out;
    if (out->jpg)
        for (int f = 0; f < n_frames; ++f) out->jpg_sizes[f] = 0;       // (a shard that fails leaves its frames at 0, never at garbage)
    const int n_dev = (int)m->lanes.size();
    for (int i = 0; i < n_dev; ++i) {
        Lane& L = *m->lanes[(size_t)i];
        jpezy_shard_range(n_frames, n_dev, i, &L.f0, &L.nf);
        L.rc = JPEZY_OK;
        L.err.clear();
        L.stats = {};
        L.stats.device = L.dev;
    }
    {
        std::vector<std::thread> pool;
        Joiner joiner{ pool, nullptr };             // (lanes fail on their own; an exception here only has to wait for them)
        try {
            for (int i = 1; i < n_dev; ++i)
                if (m->lanes[(size_t)i]->nf > 0) pool.emplace_back(lane_main, std::cref(*m), std::ref(*m->lanes[(size_t)i]), std::cref(J));
        } catch (const std::exception&) {           // the lanes that did start run to their end; the call fails
            for (auto& t : pool) if (t.joinable()) t.join();
            return set_err(JPEZY_E_HIP, "multi_encode: starting a lane thread failed");
        }
        if (m->lanes[0]->nf > 0) lane_main(*m, *m->lanes[0], J);    // the calling thread drives the root device
    }
    for (const auto& L : m->lanes)
        if (L->rc != JPEZY_OK) return set_err(L->rc, "multi_encode, device " + std::to_string(L->dev) + " (shard " + std::to_string(L->index) + "): " + L->err);
    if (out->jpg)
        for (int f = 0; f < n_frames; ++f)
            if (out->jpg_sizes[f] < 0) return set_err(JPEZY_E_FORMAT, "multi_encode: at least one frame failed (see jpg_sizes[])");
    return JPEZY_OK;
}
JPEZY_CATCH

int jpezy_multi_last_stats(const jpezy_multi* m, jpezy_multi_lane_stats* stats, int cap)
{
    if (!m) return 0;
    const int n = (int)m->lanes.size();
    for (int i = 0; i < n && i < cap && stats; ++i) stats[i] = m->lanes[(size_t)i]->stats;
    return n;
}

int jpezy_encode_batch_multi(const int* devices, int n_dev, const uint8_t* r, const uint8_t* g, const uint8_t* b, int W, int H, int gray,
                             int n_frames, int chunk_frames, const char* comment, const jpezy_multi_out* out)
try {
    if (int rc = check_devices(devices, n_dev, "encode_batch_multi")) return rc;
    if (!r || !g || !b || !out) return set_err(JPEZY_E_BADARG, "encode_batch_multi: null pointer");
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return set_err(JPEZY_E_BADARG, "width/height must be in 1..65535 (16-bit SOF0 fields)");
    if (n_frames <= 0) return set_err(JPEZY_E_BADARG, "n_frames must be positive");
    if (int rc = check_out(out, "encode_batch_multi")) return rc;
    if (int rc = check_device_range(devices, n_dev, "encode_batch_multi")) return rc;
    DeviceRestore restore;
    jpezy_multi* m = multi_create(devices, n_dev, W, H, gray, chunk_frames, n_frames);
    if (!m) return set_err(JPEZY_E_HIP, std::string("encode_batch_multi: ") + jpezy_hip_last_error());
    const int rc = jpezy_multi_encode(m, r, g, b, n_frames, comment, out);
    const std::string why = rc != JPEZY_OK ? jpezy_hip_last_error() : "";
    jpezy_multi_destroy(m);
    return rc != JPEZY_OK ? set_err(rc, why) : JPEZY_OK;
}
JPEZY_CATCH

}  // extern "C"
