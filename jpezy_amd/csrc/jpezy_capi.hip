// jpezy_capi.hip -- the C-ABI of include/jpezy_hip.h, part 1: the per-GPU context (device id, stream, device constant tables, staging
// buffers, fallback counter) and the two transform stages.  No CPU fallback: every compute entry point needs a HIP device.
#include "jpezy_capi_internal.h"

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
    if (hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->n_cus <= 0) c->n_cus = 256;
#ifdef JPEZY_WITH_LAB
    // laboratory builds only (tools/ab/ab_run.sh): the default encode kernel of every context from the environment.  The shipped
    // library reads no environment variable -- a stray one must not switch every context to another kernel
    if (const char* v = getenv("JPEZY_ENC_VARIANT")) {
        const int k = atoi(v);
        if (k >= 0 && k <= 3) c->variant = k;
    }
#endif
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
    // the table-free form of the same value (f32::dc_formula, persistent encode kernels): usable only if it reproduces the table for
    // every sum with these constants -- evaluated here with the kernel's own FP32 operations
    for (int t = 0; t < 2; ++t) {
        const float rq = 1.0f / (float)kQt[t][0], bias = 0.5f / (float)kQt[t][0];
        bool ok = true;
        for (int sum = -8192; sum <= 8192 && ok; ++sum) {
            const float d = std::trunc(std::fmaf(std::fabs((float)sum), 0.125f, -0.125f));
            const float q = std::trunc(std::fmaf(d, rq, bias));
            ok = (int)std::copysign(q, (float)sum) == (int)h.dcq[t][sum + 8192];
        }
        c->dc_rq[t] = ok ? rq : 0.f;
        c->dc_bias[t] = ok ? bias : 0.f;
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
    if (c->h_fb_pin) (void)hipHostFree(c->h_fb_pin);
    for (uint8_t* q : c->b_stage) if (q) (void)hipHostFree(q);
    for (DevBuf* b : { &c->b_scan, &c->b_U, &c->b_cnt, &c->b_rb, &c->b_state, &c->b_prop, &c->b_meta, &c->b_coef }) b->release();
    for (DevBuf& b : c->b_planes) b.release();
    for (DevBuf* b : { &c->e_tmp, &c->e_small, &c->e_U, &c->e_cnt, &c->e_out, &c->e_coef, &c->e_hdr, &c->e_status, &c->e_tt, &c->e_fft, &c->e_S, &c->e_base, &c->e_ft,
                       &c->dump_t, &c->h_scan, &c->h_U, &c->h_cnt, &c->h_off, &c->h_state, &c->h_setup, &c->h_small, &c->h_dc, &c->h_dcbuf }) b->release();
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
    if (variant < 0 || variant > 3) return set_err(JPEZY_E_BADARG, "unknown kernel variant");
#ifndef JPEZY_WITH_LAB
    if (variant >= 2)
        return set_err(JPEZY_E_UNSUPPORTED, "encode variants 2 and 3 (persistent kernels) exist only in laboratory builds (-DJPEZY_WITH_LAB, "
                                            "python -m jpezy_amd._build --lab): measured not faster than variant 1, they are not shipped");
#endif
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
#ifdef JPEZY_WITH_LAB
    // encode variant 2 (laboratory): a workgroup whose bounded spin ran out drains the launch and raises bit 40 of the last shard --
    // the coefficients of that launch are incomplete.  Surfaced as an error, not as a count.
    if (shards[COUNTER_SHARDS - 1] >> 40) {
        set_err(JPEZY_E_HIP, "encode variant 2: a persistent workgroup gave up waiting for its ring (spin cap); the last launch's coefficients are incomplete");
        return -1;
    }
#endif
    return (long)v;
}

int jpezy_internal_check_dims(const jpezy_ctx* c, int W, int H, int n_frames)
{
    if (!c) return set_err(JPEZY_E_BADARG, "null context");
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return set_err(JPEZY_E_BADARG, "width/height must be in 1..65535 (16-bit SOF0 fields)");
    if (n_frames <= 0) return set_err(JPEZY_E_BADARG, "n_frames must be positive");
    return JPEZY_OK;
}

int jpezy_fdct_quant_dev(jpezy_ctx* c, const uint8_t* d_r, const uint8_t* d_g, const uint8_t* d_b,
                         size_t plane_stride, int W, int H, int gray, int n_frames, int16_t* d_coeffs, void* stream)
{
    if (int rc = jpezy_internal_check_dims(c, W, H, n_frames)) return rc;
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
    for (int t = 0; t < 2; ++t) { p.dc_rq[t] = c->dc_rq[t]; p.dc_bias[t] = c->dc_bias[t]; }
    // the f32 kernel puts the frame index in grid.y (at most 65535): larger batches go out in chunks
    constexpr int kMaxFramesPerLaunch = 65535;
    for (int f0 = 0; f0 < n_frames; f0 += kMaxFramesPerLaunch) {
        EncParams q = p;
        q.n_frames = n_frames - f0 < kMaxFramesPerLaunch ? n_frames - f0 : kMaxFramesPerLaunch;
        q.r += (size_t)f0 * plane_stride; q.g += (size_t)f0 * plane_stride; q.b += (size_t)f0 * plane_stride;
        q.coeffs += (size_t)f0 * p.coeffs_per_frame;
#ifdef JPEZY_WITH_LAB
        if (c->variant == 3)
            HIP_TRY(launch_fdct_quant_f32_ps2(q, gray != 0, c->force_exact, c->n_cus, s));
        else if (c->variant == 2)
            HIP_TRY(launch_fdct_quant_f32_ps(q, gray != 0, c->force_exact, c->n_cus, s));
        else
#endif
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
    if (int rc = jpezy_internal_check_dims(c, W, H, n_frames)) return rc;
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
    if (int rc = jpezy_internal_check_dims(c, W, H, n_frames)) return rc;
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
    if (int rc = jpezy_internal_check_dims(c, W, H, n_frames)) return rc;
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
int jpezy_internal_generic_dev_core(jpezy_ctx* c, const int16_t* d_coeffs, const uint16_t qt[4][64], int ncomp, const uint8_t comp_h[3],
                            const uint8_t comp_v[3], const uint8_t comp_tq[3], int W, int H, int gray, int precision, uint8_t* d_r,
                            uint8_t* d_g, uint8_t* d_b, hipStream_t s, size_t* nblk_out, int n_frames, size_t plane_stride)
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
    if (int rc = jpezy_internal_check_dims(c, W, H, 1)) return rc;
    if (!d_coeffs || !qt || !comp_h || !comp_v || !comp_tq || !d_r || !d_g || !d_b) return set_err(JPEZY_E_BADARG, "null pointer");
    if (!aligned16(d_coeffs)) return set_err(JPEZY_E_BADARG, "d_coeffs must be 16-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    return jpezy_internal_generic_dev_core(c, d_coeffs, qt, ncomp, comp_h, comp_v, comp_tq, W, H, gray, precision, d_r, d_g, d_b, (hipStream_t)stream,
                            nullptr);
}

int jpezy_dequant_idct_generic_batch_dev(jpezy_ctx* c, const int16_t* d_coeffs, const uint16_t qt[4][64], int ncomp, const uint8_t comp_h[3],
                                         const uint8_t comp_v[3], const uint8_t comp_tq[3], int precision, int W, int H, int gray,
                                         int n_frames, size_t plane_stride, uint8_t* d_r, uint8_t* d_g, uint8_t* d_b, void* stream)
{
    if (int rc = jpezy_internal_check_dims(c, W, H, n_frames)) return rc;
    if (!d_coeffs || !qt || !comp_h || !comp_v || !comp_tq || !d_r || !d_g || !d_b) return set_err(JPEZY_E_BADARG, "null pointer");
    if (!aligned16(d_coeffs)) return set_err(JPEZY_E_BADARG, "d_coeffs must be 16-byte aligned");
    if (plane_stride < (size_t)W * H || (plane_stride & 3)) return set_err(JPEZY_E_BADARG, "plane_stride must hold a plane and be a multiple of 4");
    HIP_TRY(hipSetDevice(c->device));
    return jpezy_internal_generic_dev_core(c, d_coeffs, qt, ncomp, comp_h, comp_v, comp_tq, W, H, gray, precision, d_r, d_g, d_b, (hipStream_t)stream,
                            nullptr, n_frames, plane_stride);
}

int jpezy_internal_dequant_idct_generic_impl(jpezy_ctx* c, const int16_t* coeffs, const uint16_t qt[4][64], int ncomp, const uint8_t comp_h[3],
                                     const uint8_t comp_v[3], const uint8_t comp_tq[3], int W, int H, int gray, int precision,
                                     uint8_t* r, uint8_t* g, uint8_t* b, bool coeffs_on_device)
{
    if (int rc = jpezy_internal_check_dims(c, W, H, 1)) return rc;
    if (!coeffs || !qt || !comp_h || !comp_v || !comp_tq || !r || !g || !b) return set_err(JPEZY_E_BADARG, "null pointer");
    HIP_TRY(hipSetDevice(c->device));
    size_t nblk = 0;
    if (int rc = jpezy_internal_generic_dev_core(c, nullptr, qt, ncomp, comp_h, comp_v, comp_tq, W, H, gray, precision, nullptr, nullptr, nullptr, c->stream,
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
    if (int rc = jpezy_internal_generic_dev_core(c, d_coeffs, qt, ncomp, comp_h, comp_v, comp_tq, W, H, gray, precision, (uint8_t*)c->in[0].p,
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
    return jpezy_internal_dequant_idct_generic_impl(c, coeffs, qt, ncomp, comp_h, comp_v, comp_tq, W, H, gray, 8, r, g, b);
}

long jpezy_write_jpeg(const int16_t* coeffs, int W, int H, int gray, const char* comment, uint8_t* out, size_t cap)
try {
    std::string err;
    const long n = jpezy_host::write_jpeg(coeffs, W, H, gray != 0, comment, out, cap, &err);
    if (n < 0) g_err = err;
    return n;
}
JPEZY_CATCH

}  // extern "C"
