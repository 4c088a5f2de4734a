// jpezy_capi_huffdec.hip -- the C-ABI, part 3: the Huffman head of decoder::decode on the GPU (ref decoder/jpezy_decoder.hpp:583-642;
// SURVEY.md 8(f)-1, decode side): one file, the stream form shared with the batch, decoder::decode end to end.
#include "jpezy_capi_internal.h"

extern "C" {
// ---- GPU Huffman decoding (decode side) ----
namespace {

// host decode + upload: the authoritative path for everything the GPU decoder does not take or is unsure about
int read_jpeg_host_to_device(jpezy_ctx* c, const uint8_t* data, size_t len, jpezy_frame_info* info, int16_t* d_coeffs, size_t total)
{
    if (total / 64 > 4 * len) return set_err(JPEZY_E_FORMAT, "scan too short for the declared dimensions");
    std::string err;
    // The host decoder writes every coefficient, so its output needs no zeroing: up to 256 MB it goes to a pinned buffer the context keeps
    // (round 4: a fresh 50 MB vector per 4096^2 frame cost 8 ms of page faults and a pageable upload -- 14.6 ms around a 5 ms decode)
    const size_t bytes = total * sizeof(int16_t);
    // A worker of jpezy_decode_jpeg_batch keeps at most 32 MB pinned (eight or more workers that each fell back once on a large file
    // would otherwise hold gigabytes of page-locked memory for the life of their parent); a pinned allocation that fails is not an
    // error: the pageable path below does the same work.
    const size_t pin_limit = c->is_batch_child ? ((size_t)32 << 20) : ((size_t)256 << 20);
    if (bytes <= pin_limit) {
        if (c->h_fb_cap < bytes) {
            if (c->h_fb_pin) (void)hipHostFree(c->h_fb_pin);
            c->h_fb_pin = nullptr; c->h_fb_cap = 0;
            if (hipHostMalloc((void**)&c->h_fb_pin, bytes, hipHostMallocDefault) == hipSuccess) c->h_fb_cap = bytes;
            else { c->h_fb_pin = nullptr; (void)hipGetLastError(); }
        }
        if (c->h_fb_pin) {
            const int rc = jpezy_host::read_jpeg(data, len, info, (int16_t*)c->h_fb_pin, total, &err);
            if (rc < 0) { g_err = err; return rc; }
            HIP_TRY(hipMemcpy(d_coeffs, c->h_fb_pin, bytes, hipMemcpyHostToDevice));
            return JPEZY_OK;
        }
    }
    std::vector<int16_t> tmp(total);
    const int rc = jpezy_host::read_jpeg(data, len, info, tmp.data(), tmp.size(), &err);
    if (rc < 0) { g_err = err; return rc; }
    HIP_TRY(hipMemcpy(d_coeffs, tmp.data(), bytes, hipMemcpyHostToDevice));
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
//     of the lanes moved at that first step and more than a quarter still pending after it (scan_hopeless, jpezy_huffdec.h): the stream does
//     not synchronise (periodic data: flat areas) -- host decoder.
//   refinement launches of 64 steps for longer wrong runs (up to ~100 subsequences in the fuzzer's files), which decode such a stretch lane
//     after lane -- at a fraction of the host decoder's rate, so it only pays while the stretches are short.  They go on while they make
//     progress (round 2: a fixed six launches of 24 steps): two always run; from the third on the lanes that moved must be down to a
//     residue (<= 64) or have fallen to 3/4 of the launch before; never more than MAX_LAUNCHES.
// A file that drops out goes to the host decoder, whose result is the same.
// (tools/fuzz/fuzz_huffdec.py with JPEZY_HUFFDEC_DEBUG=1 prints the lanes moved per launch; JPEZY_HUFFDEC_PATIENT=1 lifts the budget.)
// The numbers are in units of 1,024 bits of stream: with subsequences of L bits a step is 1024 / L times cheaper and a correction has that
// many more lanes to cross.
struct RefineBudget {
    static constexpr int MAX_LAUNCHES = 12;
    int first_steps, steps;
    unsigned residue;
    int max_launches = MAX_LAUNCHES;
    explicit RefineBudget(unsigned subseq_bits)
    {
        const int k = subseq_bits >= 1024u ? 1 : (int)(1024u / subseq_bits);
        first_steps = 8 * k;
        steps = 64 * k > 256 ? 256 : 64 * k;
        residue = 64u * (unsigned)k;
    }
    // A periodic stream (flat areas) can hold the decoder in a wrong parse for ever: the speculative lanes, all at the same phase of the period,
    // agree with one another, nothing looks wrong at the first step, and the true state creeps down the scan one lane per step -- ~33 bits per
    // microsecond where the host decoder walks ~780.  The speculation counts the lanes that leave their subsequence in the state they entered it
    // (ScanState::periodic; chance says one in ten thousand); a scan in which most lanes do gets the launch that is enqueued blindly and no
    // further one.  (A flat 4080 x 4096 colour frame was refined for 50 ms, 64 lanes per launch, before the twelfth launch gave up:
    // profiles/r04_huffdec_periodic.txt.)
    RefineBudget(unsigned subseq_bits, unsigned n_sub, unsigned periodic) : RefineBudget(subseq_bits)
    {
        if ((unsigned long long)periodic * 5u > (unsigned long long)n_sub * 3u) max_launches = 1;
    }
    // may refinement launch `launch` (1-based, after the first launch) run, given the lanes that moved in the two launches before it?
    bool go_on(int launch, unsigned moved_before, unsigned moved_last) const
    {
        if (launch > max_launches) return false;
        if (launch <= 2 || moved_last <= residue) return true;
        return (unsigned long long)moved_last * 4u <= (unsigned long long)moved_before * 3u;
    }
};

// (shared with jpezy_capi_decode_batch.hip: declared in jpezy_capi_internal.h)


// Decodes the streams into d_coef (device, coef_elems int16, zeroed here): one sequence of launches for all of them.  ok[k] = 1 for the
// streams that converged, decoded without an invalid code and ended inside their data; the others are the caller's to hand to the
// host decoder, whose verdict is the authoritative one.  setup_usable[j] = 0: tables the device form cannot express.
int jpezy_internal_huffdec_streams(jpezy_ctx* c, const std::vector<DevStream>& streams, const std::vector<jpezy_dev::huffdec::Setup>& setups,
                    const std::vector<char>& setup_usable, const StreamGeom& geom, int16_t* d_coef, size_t coef_elems, std::vector<char>& ok,
                    const std::function<void(const char*)>& lap, bool per_lane)
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
    const RefineBudget budget(L);
    for (int pass = 0; pass <= RefineBudget::MAX_LAUNCHES; ++pass) {
        bool any = false;
        for (unsigned k = 0; k < nf; ++k) any = any || active[k];
        if (!any) break;
        HIP_TRY(hipMemcpyAsync(d_active, active.data(), (size_t)nf * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(HD::launch_sync_batch(d_S, (const uint32_t*)c->b_U.p, d_F, d_wg_file, d_wg_first, n_wg, d_active, d_exit, d_last, d_nblocks,
                                      pass == 0 ? budget.first_steps : budget.steps, s));
        HIP_TRY(hipMemcpyAsync(F.data(), d_F, sizeof(HD::BatchFile) * nf, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (unsigned k = 0; k < nf; ++k) {
            if (!active[k]) continue;
            const unsigned moved = F[k].changed[0], pending = F[k].changed[1] + F[k].changed[2];
            if (F[k].n_sub == 0) { active[k] = 0; dead[k] = 1; continue; }
            if (pending == 0) { active[k] = 0; converged[k] = 1; continue; }
            // many proposals moved at the first look (periodic data never falls into step), or the refinement launches have
            // stopped paying for this stream (RefineBudget): the caller's other path
            // (a periodic stream that has not settled at once: its true state creeps down the scan one lane per step -- RefineBudget above)
            const bool periodic = (unsigned long long)F[k].periodic * 5u > (unsigned long long)F[k].n_sub * 3u;
            if (pass == 0 ? periodic || HD::scan_hopeless(F[k].changed, F[k].n_sub) : !budget.go_on(pass + 1, prev_moved[k], moved)) { active[k] = 0; dead[k] = 1; }
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
bool jpezy_internal_build_dev_setup(jpezy_dev::huffdec::Setup& S, const jpezy_host::ScanSetup& setup, const jpezy_frame_info& info, unsigned total_blocks)
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

StreamGeom jpezy_internal_stream_geom(const jpezy_frame_info& info)
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



int jpezy_read_jpeg_gpu(jpezy_ctx* c, const uint8_t* data, size_t len, jpezy_frame_info* info, int16_t* d_coeffs, size_t coeff_cap)
try {
    namespace HD = jpezy_dev::huffdec;
    namespace E = jpezy_dev::entropy;
    if (!c) return set_err(JPEZY_E_BADARG, "null context");
    static const bool tprobe = std::getenv("JPEZY_HUFFDEC_TIMING") != nullptr;
    auto tnow = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tp0 = tnow();
    jpezy_host::ScanSetup setup;
    std::string err;
    int rc = jpezy_host::parse_header(data, len, info, &setup, &err);
    const double tp1 = tnow();
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
            std::vector<char> usable(1, jpezy_internal_build_dev_setup(setups[0], setup, *info, (unsigned)(Ri * bpm)) ? 1 : 0);
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
            if (usable[0] && jpezy_internal_huffdec_streams(c, streams, setups, usable, jpezy_internal_stream_geom(*info), d_coeffs, total, okv, lap, per_lane) == JPEZY_OK) {
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
    const unsigned n_sub_max = (unsigned)((n * 8 + L - 1) / L);      // before the stuffing is removed; buffers and launches are sized for it
    unsigned n_sub = n_sub_max;
    const size_t u_bytes = ((size_t)n_sub_max * L / 8 + 64 + 3) & ~(size_t)3;
    if (int r2 = c->h_scan.reserve(n + 64)) return r2;
    if (int r2 = c->h_U.reserve(u_bytes)) return r2;
    if (int r2 = c->h_cnt.reserve(std::max(nc, (size_t)n_sub) * sizeof(uint32_t))) return r2;
    if (int r2 = c->h_off.reserve((std::max(nc, (size_t)n_sub) + 1) * sizeof(unsigned long long))) return r2;
    if (int r2 = c->h_state.reserve((size_t)n_sub_max * (3 + 2 * (HD::emit_parts() - 1)) * sizeof(uint32_t))) return r2;
    if (int r2 = c->h_setup.reserve(sizeof(HD::Setup))) return r2;
    if (int r2 = c->h_small.reserve(sizeof(HD::ScanState))) return r2;
    if (int r2 = c->h_dcbuf.reserve(total_blocks * sizeof(int16_t) + 64)) return r2;
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
    // Round 4: the whole chain below is enqueued without the host looking in between (round 3: three synchronisations -- after the unstuffing,
    // after every synchronisation launch).  What the host used to fetch -- where the segment ends, how many subsequences hold data -- stays in a
    // ScanState on the device; launches are sized for the upper bound n_sub_max and their workgroups leave when they lie beyond the data.  A second
    // synchronisation launch and the tail (block counts, coefficient pass, DC sums) are enqueued blindly: they leave at once when there is
    // nothing for them (scan_settled in jpezy_huffdec.h).  The tables go up only when they differ from the ones the context already holds
    // (profiles/r04_huffdec_*).
    if (c->h_setup_dev != c->h_setup.p || c->h_setup_host.size() != sizeof S || std::memcmp(c->h_setup_host.data(), &S, sizeof S) != 0) {
        c->h_setup_dev = nullptr;            // (stays null if the upload fails: the next call uploads again)
        c->h_setup_host.assign(reinterpret_cast<const uint8_t*>(&S), reinterpret_cast<const uint8_t*>(&S) + sizeof S);
        // (from the context's own copy: it outlives the asynchronous upload)
        HIP_TRY(hipMemcpyAsync(c->h_setup.p, c->h_setup_host.data(), sizeof S, hipMemcpyHostToDevice, s));
        c->h_setup_dev = c->h_setup.p;
    }
    // what does not depend on the scan goes first: the zeroing of the unstuffed stream's buffer and of the coefficients (50 MB for a 4096^2
    // frame: 10 us) and the state record.  (They do not overlap the upload -- one stream; a side stream would hide their ~20 us behind it at the price
    // of four event calls.  The upload from pageable memory blocks the host for ~100 us before anything behind it can be enqueued.)
    HD::ScanState* d_st = (HD::ScanState*)c->h_small.p;
    HIP_TRY(hipMemsetAsync(c->h_U.p, 0, u_bytes, s));
    HIP_TRY(hipMemsetAsync(d_coeffs, 0, total * sizeof(int16_t), s));
    HIP_TRY(HD::launch_scan_state_init(d_st, s));
    const double tp2 = tnow();
    HIP_TRY(hipMemcpyAsync(c->h_scan.p, scan, n, hipMemcpyHostToDevice, s));
    const double tp3 = tnow();

    // 1. find the end of the segment, remove the byte stuffing
    HIP_TRY(HD::launch_unstuff_count((const uint8_t*)c->h_scan.p, n, (uint32_t*)c->h_cnt.p, d_st, s));
    HIP_TRY(E::launch_scan_u32((const uint32_t*)c->h_cnt.p, (unsigned long long*)c->h_off.p, nc, (unsigned long long*)c->e_tmp.p, s));
    HIP_TRY(HD::launch_unstuff_copy((const uint8_t*)c->h_scan.p, n, (const unsigned long long*)c->h_off.p, (uint8_t*)c->h_U.p, d_st, s));

    // 2. speculation: every lane walks from a guess three subsequences in front of its own through its own and leaves entry state, exit state,
    // blocks and marks of its own (jpezy_huffdec.hip).  Then synchronisation passes until no lane's entry state differs from its predecessor's exit.
    uint32_t* d_exit = (uint32_t*)c->h_state.p;
    uint32_t* d_last = d_exit + n_sub_max;
    unsigned* d_nblocks = (unsigned*)(d_last + n_sub_max);
    uint32_t* d_marks = (uint32_t*)(d_nblocks + n_sub_max);
    unsigned* d_mark_blocks = (unsigned*)(d_marks + (size_t)n_sub_max * (HD::emit_parts() - 1));
    HIP_TRY(HD::launch_speculate((const HD::Setup*)c->h_setup.p, (const uint32_t*)c->h_U.p, u_bytes / 4, n_sub_max, d_st, d_exit, d_last, d_nblocks,
                                 d_marks, d_mark_blocks, s));
    HD::ScanState st;
    std::vector<uint32_t> dbg_spec;
    std::vector<unsigned> dbg_moved;
    const bool dbg = std::getenv("JPEZY_HUFFDEC_DEBUG") != nullptr;
    if (dbg) {
        dbg_spec.resize(n_sub_max);
        HIP_TRY(hipMemcpyAsync(dbg_spec.data(), d_exit, (size_t)n_sub_max * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    const RefineBudget budget(L);
    auto sync_launch = [&](unsigned* changed, const unsigned* prev, int max_inner) -> hipError_t {
        return HD::launch_sync((const HD::Setup*)c->h_setup.p, (const uint32_t*)c->h_U.p, u_bytes / 4, n_sub_max, d_st, d_exit, d_last, d_nblocks,
                               d_marks, d_mark_blocks, changed, prev, max_inner, s);
    };
    HIP_TRY(sync_launch(d_st->changed, nullptr, budget.first_steps));           // confirmation + the first propagation steps
    HIP_TRY(sync_launch(d_st->changed2, d_st->changed, budget.steps));          // refinement launch 1, if there is anything left for it
    // 3. global block index of every lane, coefficients, DC predictors.  Enqueued BLINDLY behind the two synchronisation launches: the coefficient
    // and DC launches ask scan_settled() themselves and leave at once when the scan is not settled yet (3 % of the fuzz corpus: the host then
    // goes on with refinement launches and enqueues this tail again, unguarded) -- the usual file is decoded without the host looking once.
    auto enqueue_tail = [&](bool guarded) -> int {
        HIP_TRY(E::launch_scan_u32((const uint32_t*)d_nblocks, (unsigned long long*)c->h_off.p, n_sub_max, (unsigned long long*)c->e_tmp.p, s));
        // (the coefficients were zeroed at the start of the call; a guarded coefficient launch that left at once has written nothing)
        HIP_TRY(HD::launch_emit((const HD::Setup*)c->h_setup.p, (const uint32_t*)c->h_U.p, u_bytes / 4, n_sub_max, d_st, d_exit, d_marks, d_mark_blocks,
                                (const unsigned long long*)c->h_off.p, d_coeffs, (int16_t*)c->h_dcbuf.p, guarded, s));
        // DC differences -> values, all components in two launches (round 2: gather, two-launch scan, scatter per component = twelve)
        const StreamGeom g = jpezy_internal_stream_geom(*info);
        HIP_TRY(HD::launch_dc_prefix(d_coeffs, (int16_t*)c->h_dcbuf.p, g.bpm, g.ncomp, g.cstart, g.ccount, nmcu, (int*)c->h_dc.p, guarded ? d_st : nullptr, s));
        return JPEZY_OK;
    };
    if (!dbg)
        if (int r2 = enqueue_tail(true)) return r2;
    const double tp4 = tnow();
    HIP_TRY(hipMemcpyAsync(&st, d_st, sizeof st, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (tprobe) std::fprintf(stderr, "read_jpeg_gpu host us: parse %.1f, reserve+tables %.1f, scan upload call %.1f, enqueue %.1f, wait %.1f\n", tp1 - tp0, tp2 - tp1, tp3 - tp2, tp4 - tp3, tnow() - tp4);
    if (st.first_marker < n) n = (size_t)st.first_marker;
    const unsigned long long removed = st.removed;
    n_sub = st.n_sub;
    // only the subsequences that hold real data were decoded: behind them U is zero padding, which no decoder ever
    // falls into step on (a periodic stream), so it would be walked lane by lane
    if (n == 0 || removed > n || n_sub == 0 || n_sub > n_sub_max) return read_jpeg_host_to_device(c, data, len, info, d_coeffs, total);

    bool converged = false, tail_done = false;
    int passes = 1;
    {
        unsigned moved = st.changed[0];
        bool pending = st.changed[1] != 0 || st.changed[2] != 0;        // lanes left with a stale entry state, or a moved workgroup boundary: not the fixed point yet
        if (dbg) dbg_moved.push_back(moved);
        converged = !pending;
        const bool patient = dbg && std::getenv("JPEZY_HUFFDEC_PATIENT") != nullptr;      // diagnostic: show where the launches would have led
        auto pass = [&](int max_inner) -> int {          // further launches, one look each (rare: 1 % of the fuzz corpus)
            unsigned mv[4] = { 0, 0, 0, 0 };
            HIP_TRY(hipMemsetAsync(d_st->changed, 0, sizeof mv, s));
            HIP_TRY(sync_launch(d_st->changed, nullptr, max_inner));
            HIP_TRY(hipMemcpyAsync(mv, d_st->changed, sizeof mv, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            moved = mv[0];
            pending = mv[1] != 0 || mv[2] != 0;
            ++passes;
            if (dbg) dbg_moved.push_back(moved);
            return JPEZY_OK;
        };
        if (!converged && !HD::scan_hopeless(st.changed, n_sub)) {
            // the blind launch was refinement launch 1
            unsigned prev = moved;
            moved = st.changed2[0];
            pending = st.changed2[1] != 0 || st.changed2[2] != 0;
            ++passes;
            if (dbg) dbg_moved.push_back(moved);
            converged = !pending;
            const RefineBudget sized(L, n_sub, st.periodic);
            for (int it = 2; !converged && (patient ? it <= 40 : sized.go_on(it, prev, moved)); ++it) {
                prev = moved;
                if (int r2 = pass(budget.steps)) return r2;
                converged = !pending;
            }
        } else if (!converged && patient) {
            unsigned prev = moved;
            for (int it = 1; !converged && it <= 40; ++it) {
                prev = moved;
                if (int r2 = pass(budget.steps)) return r2;
                converged = !pending;
            }
            (void)prev;
        }
    }
    // the guarded tail ran iff the device saw what the host sees now (one function: scan_settled); it implies convergence
    tail_done = !dbg && converged && HD::scan_settled(st.changed, st.changed2, n_sub);
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
    if (!tail_done) {
        if (int r2 = enqueue_tail(false)) return r2;
        HIP_TRY(hipMemcpyAsync(&st, d_st, sizeof st, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    // an invalid code, or a last block that is not complete inside the real data: the host decoder decides
    if (st.error || st.last_bit > (unsigned long long)(n - removed) * 8) return read_jpeg_host_to_device(c, data, len, info, d_coeffs, total);
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
    if (int rc2 = jpezy_internal_check_dims(c, W, H, 1)) return rc2;
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
        return jpezy_internal_dequant_idct_generic_impl(c, (const int16_t*)c->out.p, info->qt, info->ncomp, hs, vs, tq, W, H, gray, info->precision, r,
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
    // (tools/measure/measure_decode_single_raw.py, 4096 x 4096, planes the caller has touched before: 2.0 ms against 2.6 ms) -- the runtime's
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

}  // extern "C"
