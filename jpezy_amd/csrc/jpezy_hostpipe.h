// jpezy_hostpipe.h -- streaming of the host-buffer entry points (internal; used by jpezy_capi.hip).
//
// jpezy_fdct_quant / jpezy_dequant_idct / jpezy_encode_jpeg take HOST buffers: what encoder::encode and decoder::decode hand
// over (ref encoder/jpezy_encoder.hpp:24-28, 58-67: the planes are std::vectors the object copied at construction, so a caller's
// buffers are fresh for every image).  Round 2 staged the whole batch -- one pageable copy in, the kernel, one copy out, one
// synchronisation: 5.7 ms per 4096x4096 frame around a 0.03 ms kernel, device memory proportional to the batch.
//
// Here a call is cut into chunks of a few MB (MCU-row bands of a large frame, or several small frames) that flow through a ring
// of RING slots, each a pinned host buffer + a device buffer per direction:
//
//   feeder threads   memcpy caller -> pinned slot, hipMemcpyAsync H2D on the upload stream, event
//   calling thread   waits (stream-side) for the upload, launches the kernel on the context's stream, event,
//                    hipMemcpyAsync D2H on the download stream, event
//   drainer threads  wait for the download, memcpy pinned slot -> caller
//
// so that the upload of chunk c + 1, the kernel of chunk c and the download of chunk c - 1 overlap, PCIe runs in both directions
// at once, and the device footprint is the ring, not the batch.  The caller's memory is never handed to the DMA engines: pinning
// fresh pages costs as much as copying them (tools/ubench/host_xfer.cpp: 1.7 ms per 2 x 48 MB registered, 0.6 ms per 48 MB copied
// by four threads), and a library-owned ring behaves the same whether or not the caller reuses its buffers.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace jpezy_host {

struct Segment {
    void* host;          // caller memory (source for uploads, destination for downloads)
    size_t bytes;
    size_t slot_off;     // offset inside the slot's pinned / device buffer
};

struct ChunkPlan {
    std::vector<Segment> in, out;
};

class HostPipe {
public:
    static constexpr int RING = 4;

    ~HostPipe() { release(); }

    void release()
    {
        for (int k = 0; k < RING; ++k) {
            if (pin_in_[k]) (void)hipHostFree(pin_in_[k]);
            if (pin_out_[k]) (void)hipHostFree(pin_out_[k]);
            if (dev_in_[k]) (void)hipFree(dev_in_[k]);
            if (dev_out_[k]) (void)hipFree(dev_out_[k]);
            pin_in_[k] = pin_out_[k] = nullptr;
            dev_in_[k] = dev_out_[k] = nullptr;
            if (ev_up_[k]) (void)hipEventDestroy(ev_up_[k]);
            if (ev_k_[k]) (void)hipEventDestroy(ev_k_[k]);
            if (ev_down_[k]) (void)hipEventDestroy(ev_down_[k]);
            ev_up_[k] = ev_k_[k] = ev_down_[k] = nullptr;
        }
        if (s_up_) (void)hipStreamDestroy(s_up_);
        if (s_down_) (void)hipStreamDestroy(s_down_);
        s_up_ = s_down_ = nullptr;
        cap_in_ = cap_out_ = 0;
    }

    // kernel(c, d_in, d_out, stream): enqueue chunk c's launches on `stream` (inputs at d_in + Segment::slot_off, outputs likewise);
    // returns a hipError_t.  plan(c): the chunk's segments.  Returns hipSuccess or the first error (message in *err).
    hipError_t run(int device, hipStream_t compute, int n_chunks, size_t max_in, size_t max_out,
                   const std::function<ChunkPlan(int)>& plan,
                   const std::function<hipError_t(int, uint8_t*, uint8_t*, hipStream_t)>& kernel, std::string* err)
    {
        if (n_chunks <= 0) return hipSuccess;
        hipError_t e = reserve(max_in, max_out);
        if (e != hipSuccess) { if (err) *err = "host pipeline: allocating the staging ring failed"; return e; }
        std::vector<ChunkPlan> plans((size_t)n_chunks);
        for (int c = 0; c < n_chunks; ++c) plans[(size_t)c] = plan(c);

        if (n_chunks == 1) {
            // a call that fits one chunk (small frames): nothing to overlap -- the plain staged copy on the calling thread,
            // without six thread starts and their event round trips
            for (const Segment& sg : plans[0].in) std::memcpy(pin_in_[0] + sg.slot_off, sg.host, sg.bytes);
            for (const Segment& sg : plans[0].in) {
                e = hipMemcpyAsync(dev_in_[0] + sg.slot_off, pin_in_[0] + sg.slot_off, sg.bytes, hipMemcpyHostToDevice, compute);
                if (e != hipSuccess) { if (err) *err = "host pipeline: hipMemcpyAsync (upload) failed"; return e; }
            }
            e = kernel(0, dev_in_[0], dev_out_[0], compute);
            for (const Segment& sg : plans[0].out) {
                if (e != hipSuccess) break;
                e = hipMemcpyAsync(pin_out_[0] + sg.slot_off, dev_out_[0] + sg.slot_off, sg.bytes, hipMemcpyDeviceToHost, compute);
            }
            const hipError_t es = hipStreamSynchronize(compute);
            if (e == hipSuccess) e = es;
            if (e != hipSuccess) { if (err) *err = std::string("host pipeline: ") + hipGetErrorString(e); return e; }
            for (const Segment& sg : plans[0].out) std::memcpy(sg.host, pin_out_[0] + sg.slot_off, sg.bytes);
            return hipSuccess;
        }

        // per-chunk progress: 0 nothing, 1 upload issued, 2 download issued, 3 delivered to the caller
        std::vector<std::atomic<int>> state((size_t)n_chunks);
        for (auto& s : state) s.store(0);
        std::atomic<int> failed{ 0 };
        std::mutex mu;
        std::condition_variable cv;
        auto set_state = [&](int c, int v) { { std::lock_guard<std::mutex> lk(mu); state[(size_t)c].store(v); } cv.notify_all(); };
        auto wait_state = [&](int c, int v) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return state[(size_t)c].load() >= v || failed.load(); });
            return !failed.load();
        };
        auto fail = [&](hipError_t code, const char* what) {
            int expected = 0;
            if (failed.compare_exchange_strong(expected, (int)code) && err) *err = std::string("host pipeline: ") + what + ": " + hipGetErrorString(code);
            cv.notify_all();
        };

        // three memcpy threads per direction (one core copies ~26 GB/s, PCIe moves ~56 GB/s each way: tools/ubench/host_xfer.cpp)
        const int n_feed = std::min(n_chunks, 3), n_drain = std::min(n_chunks, 3);
        auto feeder = [&](int id) {
            if (hipSetDevice(device) != hipSuccess) { fail(hipErrorInvalidDevice, "hipSetDevice"); return; }
            for (int c = id; c < n_chunks && !failed.load(); c += n_feed) {
                const int slot = c % RING;
                // the slot's input buffers are free once the kernel of the chunk that used them last has run
                if (c >= RING) {
                    if (!wait_state(c - RING, 2)) return;                 // its kernel has been enqueued (the event is recorded)
                    hipError_t e2 = hipEventSynchronize(ev_k_[slot]);
                    if (e2 != hipSuccess) { fail(e2, "hipEventSynchronize"); return; }
                }
                for (const Segment& sg : plans[(size_t)c].in) std::memcpy(pin_in_[slot] + sg.slot_off, sg.host, sg.bytes);
                {
                    std::lock_guard<std::mutex> lk(up_mu_);               // one stream, several feeders: keep copy + event together
                    for (const Segment& sg : plans[(size_t)c].in) {
                        hipError_t e2 = hipMemcpyAsync(dev_in_[slot] + sg.slot_off, pin_in_[slot] + sg.slot_off, sg.bytes, hipMemcpyHostToDevice, s_up_);
                        if (e2 != hipSuccess) { fail(e2, "hipMemcpyAsync (upload)"); return; }
                    }
                    hipError_t e2 = hipEventRecord(ev_up_[slot], s_up_);
                    if (e2 != hipSuccess) { fail(e2, "hipEventRecord"); return; }
                }
                set_state(c, 1);
            }
        };
        auto drainer = [&](int id) {
            if (hipSetDevice(device) != hipSuccess) { fail(hipErrorInvalidDevice, "hipSetDevice"); return; }
            for (int c = id; c < n_chunks && !failed.load(); c += n_drain) {
                const int slot = c % RING;
                if (!wait_state(c, 2)) return;
                hipError_t e2 = hipEventSynchronize(ev_down_[slot]);
                if (e2 != hipSuccess) { fail(e2, "hipEventSynchronize"); return; }
                for (const Segment& sg : plans[(size_t)c].out) std::memcpy(sg.host, pin_out_[slot] + sg.slot_off, sg.bytes);
                set_state(c, 3);
            }
        };
        // Joined on every way out of this scope: an exception below (a thread that cannot be started, an allocation, a
        // throwing kernel callback) first raises `failed` -- the workers' waits all watch it -- and then joins them, instead
        // of destroying joinable threads (std::terminate) before the C-ABI's catch can turn it into an error code.
        std::vector<std::thread> threads;
        struct Joiner {
            std::vector<std::thread>& ts;
            std::atomic<int>& failed;
            std::condition_variable& cv;
            ~Joiner()
            {
                bool running = false;
                for (auto& t : ts) running = running || t.joinable();
                if (running && std::uncaught_exceptions() > 0) {
                    int expected = 0;
                    failed.compare_exchange_strong(expected, (int)hipErrorUnknown);
                    cv.notify_all();
                }
                for (auto& t : ts) if (t.joinable()) t.join();
            }
        } joiner{ threads, failed, cv };
        threads.reserve((size_t)(n_feed + n_drain));
        try {
            for (int k = 0; k < n_feed; ++k) threads.emplace_back(feeder, k);
            for (int k = 0; k < n_drain; ++k) threads.emplace_back(drainer, k);
        } catch (const std::exception&) {
            fail(hipErrorOutOfMemory, "starting a copy thread");
        }

        for (int c = 0; c < n_chunks && !failed.load(); ++c) {
            const int slot = c % RING;
            if (!wait_state(c, 1)) break;
            if (c >= RING && !wait_state(c - RING, 3)) break;             // the slot's output buffers have been delivered
            hipError_t e2 = hipStreamWaitEvent(compute, ev_up_[slot], 0);
            if (e2 == hipSuccess) e2 = kernel(c, dev_in_[slot], dev_out_[slot], compute);
            if (e2 == hipSuccess) e2 = hipEventRecord(ev_k_[slot], compute);
            if (e2 == hipSuccess) e2 = hipStreamWaitEvent(s_down_, ev_k_[slot], 0);
            for (const Segment& sg : plans[(size_t)c].out) {
                if (e2 != hipSuccess) break;
                e2 = hipMemcpyAsync(pin_out_[slot] + sg.slot_off, dev_out_[slot] + sg.slot_off, sg.bytes, hipMemcpyDeviceToHost, s_down_);
            }
            if (e2 == hipSuccess) e2 = hipEventRecord(ev_down_[slot], s_down_);
            if (e2 != hipSuccess) { fail(e2, "launch / download"); break; }
            set_state(c, 2);
        }
        for (auto& t : threads) t.join();
        if (failed.load()) {
            (void)hipStreamSynchronize(compute);
            (void)hipStreamSynchronize(s_up_);
            (void)hipStreamSynchronize(s_down_);
            return (hipError_t)failed.load();
        }
        return hipStreamSynchronize(compute);        // (every download has been waited for by its drainer)
    }

private:
    hipError_t reserve(size_t in, size_t out)
    {
        hipError_t e = hipSuccess;
        if (!s_up_) e = hipStreamCreateWithFlags(&s_up_, hipStreamNonBlocking);
        if (e == hipSuccess && !s_down_) e = hipStreamCreateWithFlags(&s_down_, hipStreamNonBlocking);
        for (int k = 0; k < RING && e == hipSuccess; ++k) {
            if (!ev_up_[k]) e = hipEventCreateWithFlags(&ev_up_[k], hipEventDisableTiming);
            if (e == hipSuccess && !ev_k_[k]) e = hipEventCreateWithFlags(&ev_k_[k], hipEventDisableTiming);
            if (e == hipSuccess && !ev_down_[k]) e = hipEventCreateWithFlags(&ev_down_[k], hipEventDisableTiming);
        }
        if (e != hipSuccess) return e;
        if (in > cap_in_) {
            for (int k = 0; k < RING; ++k) {
                if (pin_in_[k]) (void)hipHostFree(pin_in_[k]);
                if (dev_in_[k]) (void)hipFree(dev_in_[k]);
                pin_in_[k] = nullptr; dev_in_[k] = nullptr;
            }
            cap_in_ = 0;
            for (int k = 0; k < RING; ++k) {
                if ((e = hipHostMalloc((void**)&pin_in_[k], in, hipHostMallocDefault)) != hipSuccess) return e;
                if ((e = hipMalloc((void**)&dev_in_[k], in)) != hipSuccess) return e;
            }
            cap_in_ = in;
        }
        if (out > cap_out_) {
            for (int k = 0; k < RING; ++k) {
                if (pin_out_[k]) (void)hipHostFree(pin_out_[k]);
                if (dev_out_[k]) (void)hipFree(dev_out_[k]);
                pin_out_[k] = nullptr; dev_out_[k] = nullptr;
            }
            cap_out_ = 0;
            for (int k = 0; k < RING; ++k) {
                if ((e = hipHostMalloc((void**)&pin_out_[k], out, hipHostMallocDefault)) != hipSuccess) return e;
                if ((e = hipMalloc((void**)&dev_out_[k], out)) != hipSuccess) return e;
            }
            cap_out_ = out;
        }
        return hipSuccess;
    }

    uint8_t* pin_in_[RING] = {};
    uint8_t* pin_out_[RING] = {};
    uint8_t* dev_in_[RING] = {};
    uint8_t* dev_out_[RING] = {};
    hipEvent_t ev_up_[RING] = {}, ev_k_[RING] = {}, ev_down_[RING] = {};
    hipStream_t s_up_ = nullptr, s_down_ = nullptr;
    size_t cap_in_ = 0, cap_out_ = 0;
    std::mutex up_mu_;
};

}  // namespace jpezy_host
