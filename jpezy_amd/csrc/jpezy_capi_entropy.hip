// jpezy_capi_entropy.hip -- the C-ABI, part 2: the Huffman tail of encoder::encode (ref encoder/jpezy_encoder.hpp:174-225): host writer,
// GPU entropy coder, encoder::encode end to end.
#include "jpezy_capi_internal.h"

extern "C" {
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
    if (int rc = jpezy_internal_check_dims(c, W, H, n_frames)) return rc;
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
    if (int rc = jpezy_internal_check_dims(c, W, H, n_frames)) return rc;
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
    if (int rc = jpezy_internal_check_dims(c, W, H, 1)) return rc;
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

}  // extern "C"
