// jpezy_capi_decode_batch.hip -- the C-ABI, part 4: jpezy_decode_jpeg_batch (files of one layout and size go through the GPU Huffman
// decoder together), the host decoder entry point and the decoder's knobs.
#include "jpezy_capi_internal.h"

extern "C" {
// ---- batch form (round 3): files of jpezy's own layout and one size go through the Huffman decoder TOGETHER ----
namespace {

struct FastFile {
    int index;                          // position in the caller's arrays
    jpezy_host::ScanSetup setup;
    const uint8_t* scan;
    size_t n;
};

// One slice of a group (same W x H, same layout, same quantiser tables): Huffman decoding of all files in one sequence of launches
// (jpezy_internal_huffdec_streams: a stream per file), ONE inverse-transform launch over the slice -- the fused kernel for jpezy's own 2x2,1x1,1x1
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
            setup_usable.push_back(jpezy_internal_build_dev_setup(setups.back(), files[k].setup, info, (unsigned)(nmcu * bpm)) ? 1 : 0);   // (any DHT the file carries)
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
    if (int rc = jpezy_internal_huffdec_streams(c, streams, setups, setup_usable, jpezy_internal_stream_geom(info), (int16_t*)c->b_coef.p, (size_t)nf * cpf, ok, lap, per_lane)) return rc;
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
        if (int rc = jpezy_internal_generic_dev_core(c, (const int16_t*)c->b_coef.p, info.qt, info.ncomp, hs, vs, tq, W, H, gray, info.precision, pl,
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
