// jpezy_kernels_f32.hip -- encode kernel, variant 1: three precision levels, same bits as the reference.
//
// Level 1 (every coefficient): colour conversion as an FP32 estimate with a guard band (below), separable 8-point
//   butterflies in FP32.  A quantised coefficient t = F*cu*cv/(4Q) is accepted when it is further than delta1 from every
//   non-zero integer; delta1 = 1.25 x the worst-case FP32 error of t over the coefficients of the lane's block column
//   (DeviceTables::delta1, at most 1.06e-4 luma / 5.8e-5 chroma with the Annex-K tables; DESIGN.md "exactness").
// Level 2 (guard-band hits, ~0.1 per quad): the 8 lanes holding the block's rows recompute that one coefficient in FP64
//   from the integer samples; accepted when further than 1e-6 from every boundary m*Q, m != 0.
// Level 3 (true boundary cases, ~0.2 per quad): the 64 terms are added in the reference's exact order.
// Colour conversion: Y = trunc(fma chain in FP32) is exact unless the estimate is within 2^-12 of an integer, which
//   happens exactly when 299R+587G+114B is a multiple of 1000 (1 pixel in 1000): there the reference's own FP64
//   rounding decides and the FP64 formula is evaluated.  Same for Cb/Cr (multiples of 10000, 2^-14).
// The DC coefficient is a sum of integers (exact in FP32) and is read from a table built in the reference's FP64 order.
// Compiled with -ffp-contract=off; every FMA below is explicit.
#include "jpezy_device.h"
#include "../../include/jpezy_constants.h"

namespace jpezy_dev {
namespace f32 {

__constant__ double c_cos[64] = JPEZY_COS_INIT;
__constant__ unsigned char c_zzinv[64] = JPEZY_ZZ_INV_INIT;

#define JPEZY_S JPEZY_INV_SQRT2
#define K1 0x1.f6297cff75cb0p-1f
#define K2 0x1.d906bcf328d46p-1f
#define K3 0x1.a9b66290ea1a3p-1f
#define K4 0x1.6a09e667f3bcdp-1f
#define K5 0x1.1c73b39ae68c8p-1f
#define K6 0x1.87de2a6aea963p-2f
#define K7 0x1.8f8b83c69a60bp-3f
#define FMAF(a, b, c) __builtin_fmaf((a), (b), (c))
#ifndef JPEZY_F32_WAVES
#define JPEZY_F32_WAVES 5
#endif

// Level-1 guard bands on t = v/Q: DeviceTables::delta1[table][j].  Norm-wise bound of the FP32 error of F[i][j]:
// gamma_13 * sum|cos_i| * sum|cos_j| * 128 (at most 13 roundings on any input->output path, u = 2^-24), times the
// coefficient's scale factor cu*cv/(4Q), plus the roundings of ks and of the product.  tests/test_f32_error_bound.py
// recomputes the table and measures errors 10x smaller on adversarial blocks.
constexpr double DELTA2 = 1e-6;             // level-2 guard band on v (FP64 tree-sum error < 1e-9)

// LDS geometry in dwords (floats).  Column reads are ds_read_b32 over 32-lane groups (32 banks): conflict free
// when the per-MCU stride == 8 (mod 32); row writes are ds_write_b128 over 8-lane groups: pitch 20 keeps them apart.
constexpr int Y_PITCH = 20;
constexpr int Y_MCU = 16 * Y_PITCH + 8;     // 328
constexpr int C_PITCH = 8;
constexpr int C_COMP = 68;                  // Cb rows, then Cr rows 68 dwords later (== 4 mod 32)
constexpr int C_MCU = 136;                  // == 8 mod 32
constexpr int STG_BLK = 144;                // bytes per staged block (128 + 16 pad)
constexpr int CT_BYTES = 4 * C_MCU * 4;     // 2176: chroma tile, then the staging area behind it
constexpr int STG_BYTES = 4 * 6 * STG_BLK;  // 3456
constexpr int TILE_BYTES = (4 * Y_MCU * 4 > CT_BYTES + STG_BYTES) ? 4 * Y_MCU * 4 : CT_BYTES + STG_BYTES;   // 5632
#ifndef JPEZY_QUEUE_DWORDS
#define JPEZY_QUEUE_DWORDS 192   // 5632 + 768 = 6400 B per wave: 12800 B per workgroup = 10 LDS granules of 1280 B, 12 workgroups (6 waves/SIMD) per CU
#endif
constexpr int QUEUE_CAP = 2 * (JPEZY_QUEUE_DWORDS - 1);   // entries; beyond it every coefficient of the quad is resolved (pathological input)
constexpr int WAVE_LDS_DWORDS = TILE_BYTES / 4 + JPEZY_QUEUE_DWORDS;   // + count word + 16-bit entries

__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned magic, unsigned shift)
{
    const unsigned q = __umulhi(n, magic);
    return magic ? (((n - q) >> 1) + q) >> shift : n;
}

// wave-uniform "some lane": one v_cmp into an SGPR pair + s_cmp (HIP's __any goes through a 0/1 VGPR)
__device__ __forceinline__ bool wave_any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0ull; }

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// X[u] = sum_x x[x] * cos((2x+1)u*pi/16), except X[4], which is left WITHOUT its factor cos(pi/4): both passes'
// factors are folded into the quantiser scale ks (F32Column::ks carries cos(pi/4) per index 4).  X[0] is the plain
// sum (exact for integers below 2^24).
__device__ __forceinline__ void fdct8f(const float* x, float* X)
{
    const float s0 = x[0] + x[7], s1 = x[1] + x[6], s2 = x[2] + x[5], s3 = x[3] + x[4];
    const float d0 = x[0] - x[7], d1 = x[1] - x[6], d2 = x[2] - x[5], d3 = x[3] - x[4];
    const float e0 = s0 + s3, e1 = s1 + s2, e2 = s0 - s3, e3 = s1 - s2;
    X[0] = e0 + e1;
    X[4] = e0 - e1;
    X[2] = FMAF(e3, K6, e2 * K2);
    X[6] = FMAF(-e3, K2, e2 * K6);
    X[1] = FMAF(d3, K7, FMAF(d2, K5, FMAF(d1, K3, d0 * K1)));
    X[3] = FMAF(-d3, K5, FMAF(-d2, K1, FMAF(-d1, K7, d0 * K3)));
    X[5] = FMAF(d3, K3, FMAF(d2, K7, FMAF(-d1, K1, d0 * K5)));
    X[7] = FMAF(-d3, K1, FMAF(d2, K3, FMAF(-d1, K5, d0 * K7)));
}

// ---- colour conversion in the reference's exact FP64 order (ref encoder/jpezy_encoder.hpp:244-256) ----
__device__ __forceinline__ double ref_y(double r, double g, double b)
{
    return __builtin_trunc((0.2990 * r) + (0.5870 * g) + (0.1140 * b) - 128.0);
}
__device__ __forceinline__ double ref_cb(double r, double g, double b)
{
    return __builtin_trunc(-(0.1687 * r) - (0.3313 * g) + (0.5000 * b));
}
__device__ __forceinline__ double ref_cr(double r, double g, double b)
{
    return __builtin_trunc((0.5000 * r) - (0.4187 * g) - (0.0813 * b));
}

template <int B>
__device__ __forceinline__ float ubyte(uint32_t w)
{
    return (float)((w >> (8 * B)) & 0xFFu);      // selected as v_cvt_f32_ubyteB
}

// Colour conversion, level 1.  Y = trunc(y*) with y* = (299R + 587G + 114B - 128000) / 1000.  The FP32 estimate
//   t = fma(.114f, B, fma(.587f, G, fma(.299f, R, -128)))
// is within 2.4e-5 of y* (three roundings of at most 2^-18 each, three constants rounded to FP32: 1.2e-5), and y* is
// either an integer or at least 1e-3 away from one.  So trunc(t) = trunc(y*) unless t is within LUMA_EPS = 2^-12 of an
// integer -- exactly the pixels with y* integral (1 in 1000), where the reference's own FP64 rounding sequence decides
// and the FP64 formula is evaluated instead.  The test costs one subtraction and one add per pixel: d = t - trunc(t)
// is exact, e = |d| - 1/2, near an integer <=> |e| > 1/2 - eps; the eight |e| of a half row are reduced with v_max3.
// Chroma: c* = N / 10000 (N integer), FP32 error 1.7e-5, non-integral c* at least 1e-4 from an integer, CHROMA_EPS 2^-14.
constexpr float LUMA_EPS = 0x1p-12f;
constexpr float CHROMA_EPS = 0x1p-14f;

template <int B>
__device__ __forceinline__ float luma_px(uint32_t wr, uint32_t wg, uint32_t wb, float& e)
{
    const float t = FMAF(0.114f, ubyte<B>(wb), FMAF(0.587f, ubyte<B>(wg), FMAF(0.299f, ubyte<B>(wr), -128.f)));
    const float Yt = __builtin_truncf(t);
    e = __builtin_fabsf(t - Yt) - 0.5f;
    return Yt;
}
// two pixels at once: the three FMAs of the estimate as v_pk_fma_f32 (2.2 ns for two results against 2 x 1.4)
typedef float float2_t __attribute__((ext_vector_type(2)));
template <int B0, int B1>
__device__ __forceinline__ void luma_px2(uint32_t wr, uint32_t wg, uint32_t wb, float& y0, float& y1, float& e0, float& e1)
{
#ifdef JPEZY_NO_PK
    y0 = luma_px<B0>(wr, wg, wb, e0);
    y1 = luma_px<B1>(wr, wg, wb, e1);
#else
    const float2_t r = { ubyte<B0>(wr), ubyte<B1>(wr) }, g = { ubyte<B0>(wg), ubyte<B1>(wg) }, b = { ubyte<B0>(wb), ubyte<B1>(wb) };
    const float2_t c1 = { 0.299f, 0.299f }, c2 = { 0.587f, 0.587f }, c3 = { 0.114f, 0.114f }, c0 = { -128.f, -128.f };
    const float2_t t = __builtin_elementwise_fma(c3, b, __builtin_elementwise_fma(c2, g, __builtin_elementwise_fma(c1, r, c0)));
    y0 = __builtin_truncf(t.x);
    y1 = __builtin_truncf(t.y);
    e0 = __builtin_fabsf(t.x - y0) - 0.5f;
    e1 = __builtin_fabsf(t.y - y1) - 0.5f;
#endif
}
template <int B>
__device__ __forceinline__ float luma_px_ref(uint32_t wr, uint32_t wg, uint32_t wb)
{
    return (float)ref_y((double)ubyte<B>(wr), (double)ubyte<B>(wg), (double)ubyte<B>(wb));
}
__device__ __forceinline__ float absmax8(const float* e)
{
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(e[0]), __builtin_fabsf(e[1])),
                                           __builtin_fmaxf(__builtin_fabsf(e[2]), __builtin_fabsf(e[3]))),
                           __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(e[4]), __builtin_fabsf(e[5])),
                                           __builtin_fmaxf(__builtin_fabsf(e[6]), __builtin_fabsf(e[7]))));
}

// 8 luma samples of words wr[0..1] etc. (pixels 0..7 of a half row)
__device__ __forceinline__ void luma8(const uint32_t* wr, const uint32_t* wg, const uint32_t* wb, float* y)
{
    float e[8];
    luma_px2<0, 1>(wr[0], wg[0], wb[0], y[0], y[1], e[0], e[1]);
    luma_px2<2, 3>(wr[0], wg[0], wb[0], y[2], y[3], e[2], e[3]);
    __builtin_amdgcn_sched_barrier(0);   // 4 pixels at a time: more in flight only costs registers
    luma_px2<0, 1>(wr[1], wg[1], wb[1], y[4], y[5], e[4], e[5]);
    luma_px2<2, 3>(wr[1], wg[1], wb[1], y[6], y[7], e[6], e[7]);
    constexpr float TH = 0.5f - LUMA_EPS;
#ifdef JPEZY_ABL_NOCFLAG    // timing probe (wrong results, tools/ab_build.py): what the colour guard tests and their rare path cost
    if (false) {
#else
    if (wave_any(absmax8(e) > TH)) {   // one pixel in 1000: the reference's FP64 rounding decides
#endif
        bool f;
        f = __builtin_fabsf(e[0]) > TH; if (wave_any(f)) { if (f) y[0] = luma_px_ref<0>(wr[0], wg[0], wb[0]); }
        f = __builtin_fabsf(e[1]) > TH; if (wave_any(f)) { if (f) y[1] = luma_px_ref<1>(wr[0], wg[0], wb[0]); }
        f = __builtin_fabsf(e[2]) > TH; if (wave_any(f)) { if (f) y[2] = luma_px_ref<2>(wr[0], wg[0], wb[0]); }
        f = __builtin_fabsf(e[3]) > TH; if (wave_any(f)) { if (f) y[3] = luma_px_ref<3>(wr[0], wg[0], wb[0]); }
        f = __builtin_fabsf(e[4]) > TH; if (wave_any(f)) { if (f) y[4] = luma_px_ref<0>(wr[1], wg[1], wb[1]); }
        f = __builtin_fabsf(e[5]) > TH; if (wave_any(f)) { if (f) y[5] = luma_px_ref<1>(wr[1], wg[1], wb[1]); }
        f = __builtin_fabsf(e[6]) > TH; if (wave_any(f)) { if (f) y[6] = luma_px_ref<2>(wr[1], wg[1], wb[1]); }
        f = __builtin_fabsf(e[7]) > TH; if (wave_any(f)) { if (f) y[7] = luma_px_ref<3>(wr[1], wg[1], wb[1]); }
    }
}

// chroma sample (Cb on even-row lanes, Cr on odd-row lanes) of pixel byte B; k1..k3: this lane's three coefficients
template <int B>
__device__ __forceinline__ float chroma_px(uint32_t wr, uint32_t wg, uint32_t wb, float k1, float k2, float k3, float& e)
{
    const float t = FMAF(k3, ubyte<B>(wb), FMAF(k2, ubyte<B>(wg), k1 * ubyte<B>(wr)));
    const float Ct = __builtin_truncf(t);
    e = __builtin_fabsf(t - Ct) - 0.5f;
    return Ct;
}
template <int B0, int B1>
__device__ __forceinline__ void chroma_px2(uint32_t wr0, uint32_t wg0, uint32_t wb0, uint32_t wr1, uint32_t wg1, uint32_t wb1, float k1,
                                           float k2, float k3, float& c0, float& c1, float& e0, float& e1)
{
#ifdef JPEZY_NO_PK
    c0 = chroma_px<B0>(wr0, wg0, wb0, k1, k2, k3, e0);
    c1 = chroma_px<B1>(wr1, wg1, wb1, k1, k2, k3, e1);
#else
    const float2_t r = { ubyte<B0>(wr0), ubyte<B1>(wr1) }, g = { ubyte<B0>(wg0), ubyte<B1>(wg1) }, b = { ubyte<B0>(wb0), ubyte<B1>(wb1) };
    const float2_t q1 = { k1, k1 }, q2 = { k2, k2 }, q3 = { k3, k3 };
    const float2_t t = __builtin_elementwise_fma(q3, b, __builtin_elementwise_fma(q2, g, q1 * r));
    c0 = __builtin_truncf(t.x);
    c1 = __builtin_truncf(t.y);
    e0 = __builtin_fabsf(t.x - c0) - 0.5f;
    e1 = __builtin_fabsf(t.y - c1) - 0.5f;
#endif
}
template <int B>
__device__ __forceinline__ float chroma_px_ref(uint32_t wr, uint32_t wg, uint32_t wb, bool odd)
{
    const double r = (double)ubyte<B>(wr), g = (double)ubyte<B>(wg), b = (double)ubyte<B>(wb);
    return (float)(odd ? ref_cr(r, g, b) : ref_cb(r, g, b));
}

__device__ __forceinline__ double readlane_f64(double v, int src)   // src wave-uniform
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b & 0xFFFFFFFFull), src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Levels 2 and 3 for ONE coefficient (i, j) of one block, by the 8 lanes that hold the block's 8 rows of samples
// (w[0..7]: this lane's 8 samples, integers held as floats; part: this lane holds row y of the block; first/stride:
// lane of row 0 and lane distance between rows -- all but w, part, y wave-uniform).  ref jpezy_encoder.hpp:146-172.
template <int FORCE>
__device__ __forceinline__ int resolve_coef(const float* w, bool part, int y, int first, int stride, int i, int j,
                                            int Q, double qinv)
{
    double t[8];
    {
        const double cy = c_cos[i * 8 + y];
        const double* cj = c_cos + j * 8;
        // the reference's term (pic * cos[j][x]) * cos[i][y], plain multiplications
#pragma unroll
        for (int x = 0; x < 8; ++x) t[x] = (double)w[x] * cj[x] * cy;
    }
    const double cu = j ? 1.0 : JPEZY_S, cv = i ? 1.0 : JPEZY_S;
    // (i, j) in {0,4}x{0,4}: every cosine is +-cos(pi/4) or 1, the exact value of v is a multiple of 1/8 and t a multiple
    // of 1/(8Q) >= 1e-3: such a coefficient is queued only when it sits exactly on a boundary -- level 2 cannot decide
    const bool rational = ((i | j) & 3) == 0;
    if (FORCE == 2 || (FORCE == 0 && !rational)) {
        // level 2: accurate sum in any order -- row sums, then a butterfly over the 8 participating lanes
        double s = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        s = part ? s : 0.0;
        s += __shfl_xor(s, stride, 64);
        s += __shfl_xor(s, stride * 2, 64);
        s += __shfl_xor(s, stride * 4, 64);
        const double v2 = readlane_f64(s, first) * cu * cv / 4;
        const double mq = __builtin_rint(v2 * qinv);
        const bool ambiguous = mq != 0.0 && __builtin_fabs(v2 - mq * (double)Q) < DELTA2;   // wave-uniform
        if (!ambiguous) return (int)(v2 * qinv);                                              // trunc toward zero
    }
    // level 3: the reference's order, y outer, x inner; the running sum hops from row lane to row lane
    double S = 0;
#pragma unroll
    for (int yy = 0; yy < 8; ++yy) {
        double sl = S;
#pragma unroll
        for (int x = 0; x < 8; ++x) sl += t[x];
        S = readlane_f64(sl, first + yy * stride);
    }
    const int dct = (int)(S * cu * cv / 4);
    return dct / Q;
}

// Quantised DC of a block from the exact table (DeviceTables::dcq): sum8 = this lane's column sum of eight samples
// after the row pass -- on the j == 0 lane that is the block's integer sample sum (exact in FP32).  Issued right after
// the column reads, long before the value is needed, so the L2 latency never sits on a wave's critical path; written
// with fdct8f's own association so that the adds are shared with it.
__device__ __forceinline__ int dc_lookup(const float* x, const signed char* dcq)
{
    const float s0 = x[0] + x[7], s1 = x[1] + x[6], s2 = x[2] + x[5], s3 = x[3] + x[4];
    const float sum8 = (s0 + s3) + (s1 + s2);
    // index = sum + 8192, formed in FP32 (exact) and clamped there (non-DC lanes carry arbitrary values); an
    // unsigned index keeps the lookup a scalar-base + 32-bit-offset load
    const unsigned si = (unsigned)(__builtin_fminf(__builtin_fmaxf(sum8, -8192.f), 8192.f) + 8192.f);
    return dcq[si];
}

// quantise 8 coefficients of one block column.  Returns the smallest distance |t - rint(t)| over the column (the DC
// lane skips i == 0): below DELTA1 some coefficient MAY need level 2 -- the caller then looks coefficient by coefficient
// (a distance below DELTA1 around rint(t) == 0 is not a truncation boundary and is sorted out there, off the hot path).
// dc: the block's quantised DC (dc_lookup below) -- only the j == 0 lane uses it.
__device__ __forceinline__ float quant8f(const float* F, const float* ks, bool dc_lane, int dc, int* q)
{
    float d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float t = F[i] * ks[i];
        q[i] = (int)t;                                                    // v_cvt_i32_f32 truncates toward zero
        d[i] = __builtin_fabsf(t - __builtin_rintf(t));
    }
    if (dc_lane) { q[0] = dc; d[0] = 1.f; }
    // v_min3_f32: 3.5 instructions for 8 values
    return __builtin_fminf(__builtin_fminf(__builtin_fminf(d[0], d[1]), __builtin_fminf(d[2], d[3])),
                           __builtin_fminf(__builtin_fminf(d[4], d[5]), __builtin_fminf(d[6], d[7])));
}

// base: LDS address of this lane's FIRST block; blk_off: byte offset of the block to write (an immediate after inlining)
// delta1: this lane's level-1 guard band (DeviceTables::delta1, a function of the table and of the column j)
__device__ __forceinline__ void quant_block_column(const float* F, const float* ks, float delta1, int j, int dc,
                                                   bool live, char* base, uint32_t zz_lo, uint32_t zz_hi, int blk_off, int blk,
                                                   unsigned* queue, bool force
#ifdef JPEZY_DUMP_T
                                                   , float* dump_quad
#endif
                                                   , int qcap = QUEUE_CAP)
{
#ifdef JPEZY_DUMP_T   // diagnostic build: the level-1 values exactly as the guard test sees them
    if (live && dump_quad)
#pragma unroll
        for (int i = 0; i < 8; ++i) dump_quad[blk * 64 + i * 8 + j] = F[i] * ks[i];
#endif
    int q[8];
    const float dmin = quant8f(F, ks, j == 0, dc, q);
#ifdef JPEZY_ABL_NOGUARD    // timing probe (wrong results): what the coefficient guard tests and levels 2/3 cost
    const bool cand = false;
#else
    const bool cand = force || dmin < delta1;
#endif
    // rare on noisy content; flat content (exact zeros) enters and finds nothing to queue.  (Lanes that are not live
    // repeat the quad's last MCU, so leaving them in the vote changes nothing and keeps it a bare v_cmp + s_cmp.)
    if (wave_any(cand)) {
        if (cand && live) {       // Fully unrolled: a runtime index into F/ks would send both arrays to scratch
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float t = F[i] * ks[i];
                bool f = __builtin_fabsf(t - __builtin_rintf(t)) < delta1 && __builtin_fabsf(t) > 0.5f && !(i == 0 && j == 0);
                if (force) f = true;
                if (f) {
                    const unsigned slot = atomicAdd(&queue[0], 1u);
                    if (slot < (unsigned)qcap)
                        reinterpret_cast<unsigned short*>(queue + 1)[slot] = (unsigned short)((blk << 6) | (i * 8 + j));
                }
            }
        }
    }
    // zz_lo/zz_hi: byte i = LDS byte offset of natural coefficient (i, j) inside a block (2 * zig-zag index < 128);
    // packed so that the eight addresses cost two registers (one SDWA add per store instead)
#pragma unroll
    for (int ii = 1; ii <= 8; ++ii) {      // i = 0 last: it waits for the DC lookup
        const int i = ii & 7;
        const uint32_t off = ((i < 4 ? zz_lo : zz_hi) >> (8 * (i & 3))) & 0xFFu;
        *reinterpret_cast<int16_t*>(base + off + blk_off) = (int16_t)q[i];
    }
}

template <bool GRAY, bool ALIGNED, int FORCE>
__global__ __launch_bounds__(64 * WPB, JPEZY_F32_WAVES) void fdct_quant_f32_kernel(EncParams p)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[WPB][WAVE_LDS_DWORDS];
    constexpr int BPM = GRAY ? 4 : 6;

    // WPB waves per workgroup; the wave index is made an SGPR so that everything derived from it (quad position, plane
    // and coefficient base addresses, the LDS slice) is computed once on the scalar unit
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // One quad per wave, one launch of exactly as many waves as quads.  Measured alternatives: a grid-stride loop over a
    // grid sized to the resident workgroups -- same 4096^2 time (the kernel's tail comes from XCD-to-XCD variation, which
    // a static partition cannot balance either) and 8 more VGPRs; resident waves drawing quads from one device-scope
    // atomic counter -- 240 us instead of 30: 21 k draws on one address serialise at ~10 ns each; 4 resident waves per
    // SIMD, 4 quads each, the next quad's pixels prefetched during the current one (109 VGPRs) -- 34.7 us: waves started
    // together stay in the same phase of the quad (all in the LDS transposes, then all in the butterflies), whereas waves
    // of one-quad launches arrive staggered and overlap each other's latency-bound phases.
    const unsigned qidx = blockIdx.x * (unsigned)WPB + (unsigned)wave;          // quad index inside the frame
    if (qidx >= (unsigned)(p.mcu_rows * p.quads_per_row)) return;     // wave-uniform
    const int frame = (int)blockIdx.y;
    const int mcu_y = (int)fast_div(qidx, p.qpr_magic, p.qpr_shift);
    const int quad_x = (int)qidx - mcu_y * p.quads_per_row;

#ifdef JPEZY_TRACE
    const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    uint32_t* lds = lds_all[wave];
    float* ldsf = reinterpret_cast<float*>(lds);
    unsigned* queue = lds + TILE_BYTES / 4;                            // [0] = count, then 16-bit entries
    if (lane == 0) queue[0] = 0;
    const int row = lane >> 2, m = lane & 3;
    const int mcu_x_raw = quad_x * 4 + m;
    const bool live = mcu_x_raw < p.mcu_cols;
    const int mcu_x = live ? mcu_x_raw : p.mcu_cols - 1;
    const int W = p.W, H = p.H;
    const uint8_t* pr = p.r + (size_t)frame * p.plane_stride;
    const uint8_t* pg = p.g + (size_t)frame * p.plane_stride;
    const uint8_t* pb = p.b + (size_t)frame * p.plane_stride;
    const DeviceTables* tab = p.tab;
#ifdef JPEZY_DUMP_T
    float* dump_quad = p.dump_t ? p.dump_t + (size_t)frame * p.coeffs_per_frame + ((size_t)mcu_y * p.mcu_cols + (size_t)quad_x * 4) * (BPM * 64) : nullptr;
#define DUMP_ARG , dump_quad
#else
#define DUMP_ARG
#endif

    // ---- 1. this lane's 16-pixel row segment of the three planes ----
    uint32_t R[4], G[4], B[4];
    {
        const int y = min(mcu_y * 16 + row, H - 1);                   // edge replication, ref :101
        const unsigned rowoff = (unsigned)y * (unsigned)W;            // W, H <= 65535 (launcher): fits 32 bits
        if (ALIGNED) {
            const unsigned off = rowoff + (unsigned)mcu_x * 16u;
            const uint4 vr = *reinterpret_cast<const uint4*>(pr + off);
            const uint4 vg = *reinterpret_cast<const uint4*>(pg + off);
            const uint4 vb = *reinterpret_cast<const uint4*>(pb + off);
            R[0] = vr.x; R[1] = vr.y; R[2] = vr.z; R[3] = vr.w;
            G[0] = vg.x; G[1] = vg.y; G[2] = vg.z; G[3] = vg.w;
            B[0] = vb.x; B[1] = vb.y; B[2] = vb.z; B[3] = vb.w;
        } else {
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) {
                uint32_t ar = 0, ag = 0, ab = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int x = min(mcu_x * 16 + w4 * 4 + k, W - 1);   // ref :104
                    ar |= (uint32_t)pr[rowoff + x] << (8 * k);
                    ag |= (uint32_t)pg[rowoff + x] << (8 * k);
                    ab |= (uint32_t)pb[rowoff + x] << (8 * k);
                }
                R[w4] = ar; G[w4] = ag; B[w4] = ab;
            }
        }
    }

#ifdef JPEZY_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tr_t1 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- 2. luma + row pass of the left and right block, into the transpose tile.  The integer samples stay in
    //         registers as floats (ys: luma 16, cs: chroma 8) for the chroma row pass and the rare levels 2 and 3. ----
    float ys[16], cs[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    {
        float4* dst = reinterpret_cast<float4*>(ldsf + m * Y_MCU + row * Y_PITCH);
        float X[8];
        luma8(R, G, B, ys);
        fdct8f(ys, X);
        dst[0] = make_float4(X[0], X[1], X[2], X[3]);
        dst[1] = make_float4(X[4], X[5], X[6], X[7]);
        luma8(R + 2, G + 2, B + 2, ys + 8);
        fdct8f(ys + 8, X);
        dst[2] = make_float4(X[0], X[1], X[2], X[3]);
        dst[3] = make_float4(X[4], X[5], X[6], X[7]);
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the phases apart: the scheduler otherwise overlaps them and needs >80 VGPRs
    // ---- 2b. chroma samples (top-left pixel of every 2x2, ref :134-142): the odd-row lane takes its even neighbour's
    //         pixels (DPP row_shr:4) and computes Cr, the even-row lane Cb.  Only the samples survive, so the raw
    //         pixel registers die here. ----
    if (!GRAY) {
        const bool odd = (row & 1) != 0;
        uint32_t R2[4], G2[4], B2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            R2[k] = (uint32_t)__builtin_amdgcn_update_dpp((int)R[k], (int)R[k], 0x114, 0xF, 0xA, false);
            G2[k] = (uint32_t)__builtin_amdgcn_update_dpp((int)G[k], (int)G[k], 0x114, 0xF, 0xA, false);
            B2[k] = (uint32_t)__builtin_amdgcn_update_dpp((int)B[k], (int)B[k], 0x114, 0xF, 0xA, false);
        }
        // (Cb, Cr) = (-.1687 R - .3313 G + .5 B), (.5 R - .4187 G - .0813 B)   (ref :249-256)
        const float k1 = odd ? 0.5f : -0.1687f, k2 = odd ? -0.4187f : -0.3313f, k3 = odd ? -0.0813f : 0.5f;
        float cv[8], e[8];
        chroma_px2<0, 2>(R2[0], G2[0], B2[0], R2[0], G2[0], B2[0], k1, k2, k3, cv[0], cv[1], e[0], e[1]);
        chroma_px2<0, 2>(R2[1], G2[1], B2[1], R2[1], G2[1], B2[1], k1, k2, k3, cv[2], cv[3], e[2], e[3]);
        __builtin_amdgcn_sched_barrier(0);
        chroma_px2<0, 2>(R2[2], G2[2], B2[2], R2[2], G2[2], B2[2], k1, k2, k3, cv[4], cv[5], e[4], e[5]);
        chroma_px2<0, 2>(R2[3], G2[3], B2[3], R2[3], G2[3], B2[3], k1, k2, k3, cv[6], cv[7], e[6], e[7]);
        constexpr float TH = 0.5f - CHROMA_EPS;
#ifdef JPEZY_ABL_NOCFLAG
        if (false) {
#else
        if (wave_any(absmax8(e) > TH)) {
#endif
            bool f;
            f = __builtin_fabsf(e[0]) > TH; if (wave_any(f)) { if (f) cv[0] = chroma_px_ref<0>(R2[0], G2[0], B2[0], odd); }
            f = __builtin_fabsf(e[1]) > TH; if (wave_any(f)) { if (f) cv[1] = chroma_px_ref<2>(R2[0], G2[0], B2[0], odd); }
            f = __builtin_fabsf(e[2]) > TH; if (wave_any(f)) { if (f) cv[2] = chroma_px_ref<0>(R2[1], G2[1], B2[1], odd); }
            f = __builtin_fabsf(e[3]) > TH; if (wave_any(f)) { if (f) cv[3] = chroma_px_ref<2>(R2[1], G2[1], B2[1], odd); }
            f = __builtin_fabsf(e[4]) > TH; if (wave_any(f)) { if (f) cv[4] = chroma_px_ref<0>(R2[2], G2[2], B2[2], odd); }
            f = __builtin_fabsf(e[5]) > TH; if (wave_any(f)) { if (f) cv[5] = chroma_px_ref<2>(R2[2], G2[2], B2[2], odd); }
            f = __builtin_fabsf(e[6]) > TH; if (wave_any(f)) { if (f) cv[6] = chroma_px_ref<0>(R2[3], G2[3], B2[3], odd); }
            f = __builtin_fabsf(e[7]) > TH; if (wave_any(f)) { if (f) cv[7] = chroma_px_ref<2>(R2[3], G2[3], B2[3], odd); }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) cs[k] = cv[k];
    }
    __builtin_amdgcn_sched_barrier(0);
    wave_sync();

    // ---- 3+4. luma column pass, quantise + zig-zag into the staging area; top block first, then the bottom block
    //           (kept apart so that only one block column of coefficients is live at a time) ----
    const int cq = row, j = cq & 7;
    const unsigned ju = (unsigned)j;   // unsigned table indices: scalar base + 32-bit offset addressing
    float col[16];
    {
        const float* src = ldsf + m * Y_MCU + cq;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) col[rr] = src[rr * Y_PITCH];
    }
    wave_sync();   // tile consumed; the slice is reused (chroma tile | staging)
    char* stage = reinterpret_cast<char*>(lds) + CT_BYTES;
    const int bx = cq >> 3;
    char* sbase = stage + (m * BPM + bx) * STG_BLK;   // this lane's first block (m, bx); the others are immediates away
    const F32Column* lcol = &tab->f32col[0][ju];
    const uint32_t zz_lo = lcol->zz_lo, zz_hi = lcol->zz_hi;
    const signed char* dcq_l = p.dcq_luma;
    const signed char* dcq_c = p.dcq_chroma;
    {
        float ks[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ks[i] = lcol->ks[i];
        const float dl = lcol->delta1;
        const int dc_top = dc_lookup(col, dcq_l), dc_bot = dc_lookup(col + 8, dcq_l);
        {
            float F[8];
            fdct8f(col, F);
            quant_block_column(F, ks, dl, j, dc_top, live, sbase, zz_lo, zz_hi, 0, m * BPM + bx, queue, FORCE != 0 DUMP_ARG);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            float F[8];
            fdct8f(col + 8, F);
            quant_block_column(F, ks, dl, j, dc_bot, live, sbase, zz_lo, zz_hi, 2 * STG_BLK, m * BPM + 2 + bx, queue, FORCE != 0 DUMP_ARG);
        }
    }

    __builtin_amdgcn_sched_barrier(0);
    // ---- 5. chroma row pass, transpose, column pass ----
    if (!GRAY) {
        const bool odd = (row & 1) != 0;
        float cX[8];
        fdct8f(cs, cX);
        float4* dst = reinterpret_cast<float4*>(ldsf + m * C_MCU + (odd ? C_COMP : 0) + (row >> 1) * C_PITCH);
        dst[0] = make_float4(cX[0], cX[1], cX[2], cX[3]);
        dst[1] = make_float4(cX[4], cX[5], cX[6], cX[7]);
        wave_sync();

        float Fc[8];
        int dc_c;
        {
            float col[8];
            const float* src = ldsf + m * C_MCU + (cq >> 3) * C_COMP + j;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) col[rr] = src[rr * C_PITCH];
            dc_c = dc_lookup(col, dcq_c);
            fdct8f(col, Fc);
        }
        float ks[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ks[i] = lcol[8].ks[i];
        quant_block_column(Fc, ks, lcol[8].delta1, j, dc_c, live, sbase, zz_lo, zz_hi, 4 * STG_BLK, m * BPM + 4 + bx, queue, FORCE != 0 DUMP_ARG);
    }
    wave_sync();

    // ---- 5b. levels 2 and 3 for the queued coefficients (FORCE 1/2: every coefficient of the quad) ----
    const unsigned nq = queue[0];
#ifdef JPEZY_DEFER_PROBE
    // timing probe of "deferred resolves" (tools/experiments, WRONG RESULTS: nothing resolves the entries): the guard-band hits
    // of the wave are appended to a global list sharded 64 ways (one returning atomic per wave with hits), levels 2/3 are not run
    if (FORCE == 0) {
        if (nq) {
            unsigned long long* shard = p.fallback_count + 64 + (qidx & 63u) * 8u;          // probe: counters live in the fallback array
            unsigned base = 0;
            if (lane == 0) base = (unsigned)atomicAdd(shard, (unsigned long long)nq);
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            uint2* list = reinterpret_cast<uint2*>(p.defer_list) + (size_t)(qidx & 63u) * 4096u;
            for (unsigned e = lane; e < nq && e < (unsigned)QUEUE_CAP; e += 64)
                list[(base + e) & 4095u] = make_uint2(qidx, reinterpret_cast<const unsigned short*>(queue + 1)[e]);
        }
    } else
#endif
    if (FORCE == 3 || (FORCE == 0 && nq > (unsigned)QUEUE_CAP)) {
        // More guard-band hits than the queue holds (adversarial patterns; FORCE 3 exercises it): every lane evaluates
        // the 24 coefficients of its three block columns in the reference's order by itself.  The integer samples go
        // to LDS as bytes (1.5 KB in the dead chroma tile); a lane walks its block row by row, keeps the eight running
        // sums of its column (i = 0..7) and adds (pic * cos[j][x]) * cos[i][y] for x = 0..7 to each -- for every i
        // exactly the reference's sequence (ref :146-166).  ~3,500 FP64 operations per lane, 7 us per wave, against
        // ~1 ms for the cooperative path on all 1536 coefficients.
        signed char* smp = reinterpret_cast<signed char*>(lds);
        {
            uint32_t w4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                w4[k] = ((uint32_t)(int)ys[4 * k] & 0xFFu) | (((uint32_t)(int)ys[4 * k + 1] & 0xFFu) << 8) |
                        (((uint32_t)(int)ys[4 * k + 2] & 0xFFu) << 16) | (((uint32_t)(int)ys[4 * k + 3] & 0xFFu) << 24);
            const int by = row >> 3, y = row & 7;
            uint32_t* d0 = reinterpret_cast<uint32_t*>(smp + ((m * 4 + by * 2) * 64 + y * 8));
            d0[0] = w4[0]; d0[1] = w4[1];
            d0[16] = w4[2]; d0[17] = w4[3];                      // the right block, 64 bytes further
            if (!GRAY) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    w4[k] = ((uint32_t)(int)cs[4 * k] & 0xFFu) | (((uint32_t)(int)cs[4 * k + 1] & 0xFFu) << 8) |
                            (((uint32_t)(int)cs[4 * k + 2] & 0xFFu) << 16) | (((uint32_t)(int)cs[4 * k + 3] & 0xFFu) << 24);
                uint32_t* dc = reinterpret_cast<uint32_t*>(smp + 1024 + ((m * 2 + (row & 1)) * 64 + (row >> 1) * 8));
                dc[0] = w4[0]; dc[1] = w4[1];
            }
        }
        wave_sync();
        const double cu = j ? 1.0 : JPEZY_S;
#pragma unroll 1
        for (int bc = 0; bc < (GRAY ? 2 : 3); ++bc) {
            // block column bc of this lane: 0 top luma block, 1 bottom luma block, 2 chroma block (Cb / Cr by cq >> 3)
            const int blk = m * BPM + (bc < 2 ? bc * 2 + bx : 4 + bx);
            const signed char* src = bc < 2 ? smp + (m * 4 + bc * 2 + bx) * 64 : smp + 1024 + (m * 2 + bx) * 64;
            const int tbl = bc < 2 ? 0 : 1;
            double S[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll 1
            for (int y = 0; y < 8; ++y) {
                // x is not unrolled: this path must not raise the kernel's register count (it is never the hot one)
#pragma unroll 1
                for (int x = 0; x < 8; ++x) {
                    const double px = (double)(int)src[y * 8 + x] * c_cos[j * 8 + x];
#pragma unroll
                    for (int i = 0; i < 8; ++i) S[i] += px * c_cos[i * 8 + y];
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double cv = i ? 1.0 : JPEZY_S;
                const int dct = (int)(S[i] * cu * cv / 4);
                const int qv = dct / tab->qt[tbl][i * 8 + j];
                *reinterpret_cast<int16_t*>(stage + blk * STG_BLK + 2 * (int)c_zzinv[i * 8 + j]) = (int16_t)qv;
            }
        }
        if (lane == 0) atomicAdd(p.fallback_count + (qidx & (COUNTER_SHARDS - 1)), (unsigned long long)(4 * BPM * 64));
        wave_sync();
    } else {
        const bool all = FORCE != 0;
        const unsigned total = all ? (unsigned)(4 * BPM * 64) : nq;
        if (total) {
            const int valid_mcus = min(4, p.mcu_cols - quad_x * 4);
            unsigned done = 0;
#pragma unroll 1
            for (unsigned e = 0; e < total; ++e) {
                const unsigned code = all ? e : reinterpret_cast<const unsigned short*>(queue + 1)[e];
                const int blk = __builtin_amdgcn_readfirstlane((int)(code >> 6)), nat = __builtin_amdgcn_readfirstlane((int)(code & 63));
                const int em = blk / BPM, eb = blk - em * BPM;
                if (em >= valid_mcus) continue;
                const int ei = nat >> 3, ej = nat & 7;
                const int comp = eb < 4 ? 0 : eb - 3, tbl = comp ? 1 : 0;
                // the 8 lanes that hold the block's rows: luma block (by,bx): rows by*8+y, samples ys[8bx..]; chroma:
                // Cb on even-row lanes, Cr on odd-row lanes, samples cs[]
                const int by = (eb >> 1) & 1, bx = eb & 1;
                const int first = comp ? (comp == 2 ? 4 : 0) + em : by * 32 + em;
                const int stride = comp ? 8 : 4;
                const bool part = (m == em) && (comp ? ((row & 1) == (comp == 2)) : ((row >> 3) == by));
                const int yrow = comp ? (row >> 1) : (row & 7);
                float w[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) w[k] = comp ? cs[k] : (bx ? ys[8 + k] : ys[k]);
                const int qv = resolve_coef<FORCE>(w, part, yrow, first, stride, ei, ej, tab->qt[tbl][nat], tab->qinv[tbl][nat]);
                if (lane == 0) *reinterpret_cast<int16_t*>(stage + blk * STG_BLK + 2 * (int)c_zzinv[nat]) = (int16_t)qv;
                ++done;
            }
            if (lane == 0 && done) atomicAdd(p.fallback_count + (qidx & (COUNTER_SHARDS - 1)), (unsigned long long)done);
            wave_sync();
        }
    }

#ifdef JPEZY_TRACE
    const unsigned long long tr_t2 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- 6. coalesced store of the quad's coefficients ----
    {
        const int valid_chunks = min(4, p.mcu_cols - quad_x * 4) * BPM * 8;         // 16-byte chunks
        int16_t* gbase = p.coeffs + (size_t)frame * p.coeffs_per_frame +
                         ((size_t)mcu_y * p.mcu_cols + (size_t)quad_x * 4) * (BPM * 64);
        uint4* g4 = reinterpret_cast<uint4*>(gbase);
#pragma unroll
        for (int k = 0; k < BPM * 128 * 4 / 1024; ++k) {
            const int c = k * 64 + lane;
            if (c < valid_chunks) {
                // streamed out, never re-read by this kernel: a non-temporal store leaves less dirty data in the L2s
                // for the end-of-kernel write-back (measured: 2 us per 4096^2 frame)
                const uint4 v = *reinterpret_cast<const uint4*>(stage + (c >> 3) * STG_BLK + (c & 7) * 16);
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(v4u{v.x, v.y, v.z, v.w}, reinterpret_cast<v4u*>(g4 + c));
            }
        }
    }
#ifdef JPEZY_TRACE
    if (frame == 0 && qidx < 65536u) {
#if JPEZY_TRACE > 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        const unsigned long long tr_t3 = __builtin_amdgcn_s_memrealtime();
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        if (lane == 0) {
            p.trace[qidx * 4 + 0] = tr_t0;
            p.trace[qidx * 4 + 1] = ((tr_t1 - tr_t0) << 32) | (tr_t2 - tr_t0);
            p.trace[qidx * 4 + 2] = tr_t3 - tr_t0;
            p.trace[qidx * 4 + 3] = ((unsigned long long)xcc << 32) | hw;
        }
    }
#endif
}

// ======================================================================================================
// Encode, variant 2: the luma transforms of a quad as one matrix product on the matrix pipe (v_mfma_f32_16x16x32_f16, the
// constants as two f16 limbs), everything else as in variant 1 (colour conversion with its guard band, chroma butterflies on
// the VALU, levels 2 and 3 for guard-band hits, exact DC table).  OPT-IN (jpezy_ctx_set_variant(ctx, 2)): the accumulation
// inside an f16 MFMA is undocumented and is NOT a correctly rounded dot product (tools/ubench/mfma_f16_numerics.hip), so the
// level-1 guard band of this variant rests on a measured error model (40 x 2^-24 x sum |x G| per coefficient: twice the worst
// error seen per instruction, four instructions per chain, plus the limb truncation), not on a proof -- DESIGN.md section 4.
// ======================================================================================================
constexpr int WPB2 = 2;                           // waves per workgroup (as variant 1: the waves never talk to each other)
constexpr int XB_PITCH = 144, XM_PITCH = 4 * XB_PITCH + 16;      // sample exchange tile: bytes per block / per MCU (f16 samples)
constexpr int REGA_BYTES = (4 * XM_PITCH > CT_BYTES) ? 4 * XM_PITCH : CT_BYTES;     // exchange tile, later the chroma tile
constexpr int TILE2_BYTES = REGA_BYTES + STG_BYTES;
constexpr int QUEUE2_DWORDS = 64;
constexpr int QUEUE2_CAP = 2 * (QUEUE2_DWORDS - 1);
constexpr int WAVE2_LDS_DWORDS = TILE2_BYTES / 4 + QUEUE2_DWORDS;
__constant__ unsigned char c_zz[64] = JPEZY_ZZ_INIT;

template <bool GRAY, bool ALIGNED, int FORCE>
__global__ __launch_bounds__(64 * WPB2, JPEZY_F32_WAVES) void fdct_quant_mfma_kernel(EncParams p)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[WPB2][WAVE2_LDS_DWORDS];
    constexpr int BPM = GRAY ? 4 : 6;

    // WPB waves per workgroup; the wave index is made an SGPR so that everything derived from it (quad position, plane
    // and coefficient base addresses, the LDS slice) is computed once on the scalar unit
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // The constant operands (16 KB, the same for every wave of the launch) are read from global memory, i.e. out of the L1/L2:
    // a copy in LDS would have to be shared by ten waves to fit -- measured: workgroups of ten waves behind one barrier run in
    // lockstep phases and release their slots only together, 40.9 us per 4096^2 frame.
    const uint4* atab = reinterpret_cast<const uint4*>(p.tab->mfma_a);
    const unsigned qidx = blockIdx.x * (unsigned)WPB2 + (unsigned)wave;          // quad index inside the frame
    if (qidx >= (unsigned)(p.mcu_rows * p.quads_per_row)) return;     // wave-uniform
    const int frame = (int)blockIdx.y;
    const int mcu_y = (int)fast_div(qidx, p.qpr_magic, p.qpr_shift);
    const int quad_x = (int)qidx - mcu_y * p.quads_per_row;

#ifdef JPEZY_TRACE
    const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    uint32_t* lds = lds_all[wave];
    float* ldsf = reinterpret_cast<float*>(lds);
    unsigned* queue = lds + TILE2_BYTES / 4;                            // [0] = count, then 16-bit entries
    if (lane == 0) queue[0] = 0;
    const int row = lane >> 2, m = lane & 3;
    const int mcu_x_raw = quad_x * 4 + m;
    const bool live = mcu_x_raw < p.mcu_cols;
    const int mcu_x = live ? mcu_x_raw : p.mcu_cols - 1;
    const int W = p.W, H = p.H;
    const uint8_t* pr = p.r + (size_t)frame * p.plane_stride;
    const uint8_t* pg = p.g + (size_t)frame * p.plane_stride;
    const uint8_t* pb = p.b + (size_t)frame * p.plane_stride;
    const DeviceTables* tab = p.tab;
#ifdef JPEZY_DUMP_T
    float* dump_quad = p.dump_t ? p.dump_t + (size_t)frame * p.coeffs_per_frame + ((size_t)mcu_y * p.mcu_cols + (size_t)quad_x * 4) * (BPM * 64) : nullptr;
#define DUMP_ARG , dump_quad
#else
#define DUMP_ARG
#endif

    // ---- 1. this lane's 16-pixel row segment of the three planes ----
    uint32_t R[4], G[4], B[4];
    {
        const int y = min(mcu_y * 16 + row, H - 1);                   // edge replication, ref :101
        const unsigned rowoff = (unsigned)y * (unsigned)W;            // W, H <= 65535 (launcher): fits 32 bits
        if (ALIGNED) {
            const unsigned off = rowoff + (unsigned)mcu_x * 16u;
            const uint4 vr = *reinterpret_cast<const uint4*>(pr + off);
            const uint4 vg = *reinterpret_cast<const uint4*>(pg + off);
            const uint4 vb = *reinterpret_cast<const uint4*>(pb + off);
            R[0] = vr.x; R[1] = vr.y; R[2] = vr.z; R[3] = vr.w;
            G[0] = vg.x; G[1] = vg.y; G[2] = vg.z; G[3] = vg.w;
            B[0] = vb.x; B[1] = vb.y; B[2] = vb.z; B[3] = vb.w;
        } else {
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) {
                uint32_t ar = 0, ag = 0, ab = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int x = min(mcu_x * 16 + w4 * 4 + k, W - 1);   // ref :104
                    ar |= (uint32_t)pr[rowoff + x] << (8 * k);
                    ag |= (uint32_t)pg[rowoff + x] << (8 * k);
                    ab |= (uint32_t)pb[rowoff + x] << (8 * k);
                }
                R[w4] = ar; G[w4] = ag; B[w4] = ab;
            }
        }
    }

#ifdef JPEZY_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tr_t1 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- 2'. luma samples (integers held as floats in ys[], also for the rare levels 2 and 3); packed to f16 (exact:
    //          |Y| <= 128), scaled by 2^-12 (the constant operands carry 2^12 so that both of their f16 limbs are normal
    //          numbers) and exchanged through LDS into the B-operand layout of v_mfma_f32_16x16x32_f16 ----
    float ys[16], cs[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    {
        luma8(R, G, B, ys);
        luma8(R + 2, G + 2, B + 2, ys + 8);
        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
        uint32_t hp[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            h2_t v = __builtin_bit_cast(h2_t, __builtin_amdgcn_cvt_pkrtz(ys[2 * q], ys[2 * q + 1]));
            v = v * h2_t{ (_Float16)0x1p-12f, (_Float16)0x1p-12f };
            hp[q] = __builtin_bit_cast(uint32_t, v);
        }
        char* xt = reinterpret_cast<char*>(lds) + m * XM_PITCH + ((row >> 3) * 2) * XB_PITCH + (row & 7) * 16;
        *reinterpret_cast<uint4*>(xt) = make_uint4(hp[0], hp[1], hp[2], hp[3]);                 // row of the left block
        *reinterpret_cast<uint4*>(xt + XB_PITCH) = make_uint4(hp[4], hp[5], hp[6], hp[7]);      // row of the right block
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the phases apart: the scheduler otherwise overlaps them and needs >80 VGPRs
    // ---- 2b. chroma samples (top-left pixel of every 2x2, ref :134-142): the odd-row lane takes its even neighbour's
    //         pixels (DPP row_shr:4) and computes Cr, the even-row lane Cb.  Only the samples survive, so the raw
    //         pixel registers die here. ----
    if (!GRAY) {
        const bool odd = (row & 1) != 0;
        uint32_t R2[4], G2[4], B2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            R2[k] = (uint32_t)__builtin_amdgcn_update_dpp((int)R[k], (int)R[k], 0x114, 0xF, 0xA, false);
            G2[k] = (uint32_t)__builtin_amdgcn_update_dpp((int)G[k], (int)G[k], 0x114, 0xF, 0xA, false);
            B2[k] = (uint32_t)__builtin_amdgcn_update_dpp((int)B[k], (int)B[k], 0x114, 0xF, 0xA, false);
        }
        // (Cb, Cr) = (-.1687 R - .3313 G + .5 B), (.5 R - .4187 G - .0813 B)   (ref :249-256)
        const float k1 = odd ? 0.5f : -0.1687f, k2 = odd ? -0.4187f : -0.3313f, k3 = odd ? -0.0813f : 0.5f;
        float cv[8], e[8];
        chroma_px2<0, 2>(R2[0], G2[0], B2[0], R2[0], G2[0], B2[0], k1, k2, k3, cv[0], cv[1], e[0], e[1]);
        chroma_px2<0, 2>(R2[1], G2[1], B2[1], R2[1], G2[1], B2[1], k1, k2, k3, cv[2], cv[3], e[2], e[3]);
        __builtin_amdgcn_sched_barrier(0);
        chroma_px2<0, 2>(R2[2], G2[2], B2[2], R2[2], G2[2], B2[2], k1, k2, k3, cv[4], cv[5], e[4], e[5]);
        chroma_px2<0, 2>(R2[3], G2[3], B2[3], R2[3], G2[3], B2[3], k1, k2, k3, cv[6], cv[7], e[6], e[7]);
        constexpr float TH = 0.5f - CHROMA_EPS;
#ifdef JPEZY_ABL_NOCFLAG
        if (false) {
#else
        if (wave_any(absmax8(e) > TH)) {
#endif
            bool f;
            f = __builtin_fabsf(e[0]) > TH; if (wave_any(f)) { if (f) cv[0] = chroma_px_ref<0>(R2[0], G2[0], B2[0], odd); }
            f = __builtin_fabsf(e[1]) > TH; if (wave_any(f)) { if (f) cv[1] = chroma_px_ref<2>(R2[0], G2[0], B2[0], odd); }
            f = __builtin_fabsf(e[2]) > TH; if (wave_any(f)) { if (f) cv[2] = chroma_px_ref<0>(R2[1], G2[1], B2[1], odd); }
            f = __builtin_fabsf(e[3]) > TH; if (wave_any(f)) { if (f) cv[3] = chroma_px_ref<2>(R2[1], G2[1], B2[1], odd); }
            f = __builtin_fabsf(e[4]) > TH; if (wave_any(f)) { if (f) cv[4] = chroma_px_ref<0>(R2[2], G2[2], B2[2], odd); }
            f = __builtin_fabsf(e[5]) > TH; if (wave_any(f)) { if (f) cv[5] = chroma_px_ref<2>(R2[2], G2[2], B2[2], odd); }
            f = __builtin_fabsf(e[6]) > TH; if (wave_any(f)) { if (f) cv[6] = chroma_px_ref<0>(R2[3], G2[3], B2[3], odd); }
            f = __builtin_fabsf(e[7]) > TH; if (wave_any(f)) { if (f) cv[7] = chroma_px_ref<2>(R2[3], G2[3], B2[3], odd); }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) cs[k] = cv[k];
    }
    __builtin_amdgcn_sched_barrier(0);
    wave_sync();

    // ---- 3'+4'. the sixteen luma blocks of the quad as ONE matrix product on the matrix pipe:
    //   t[p][n] = sum_k G[p][k] * x[k][n],  p = zig-zag position, k = 8 y + x, n = 4 m + 2 by + bx the block,
    //   G[p][k] = cos_i(y) cos_j(x) cu cv / (4 Q) (DeviceTables::mfma_a: two f16 limbs, rows in zig-zag order, quantiser folded
    //   in; row 0 is all ones so that the DC comes out as the exact sample sum for the table lookup).
    // 4 row tiles x 2 K steps x 2 limbs = 16 v_mfma_f32_16x16x32_f16; lane (g = lane >> 4, n = lane & 15) receives
    // t[16 T + 4 g + r][n], r = 0..3: four consecutive zig-zag positions per tile -- one 8-byte store into the staging area.
    const int cq = row, j = cq & 7;
    const unsigned ju = (unsigned)j;
    const int n_blk = lane & 15, g4 = lane >> 4;
    typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
    typedef float f4_t __attribute__((ext_vector_type(4)));
    h8_t bfrag[2];
    {
        const char* xr = reinterpret_cast<const char*>(lds) + (n_blk >> 2) * XM_PITCH + (n_blk & 3) * XB_PITCH + g4 * 16;
        bfrag[0] = __builtin_bit_cast(h8_t, *reinterpret_cast<const uint4*>(xr));          // block row g (K step 0)
        bfrag[1] = __builtin_bit_cast(h8_t, *reinterpret_cast<const uint4*>(xr + 64));     // block row g + 4 (K step 1)
    }
    wave_sync();   // exchange tile consumed; the slice is reused (chroma tile | staging)
    char* stage = reinterpret_cast<char*>(lds) + REGA_BYTES;
    const int bx = cq >> 3;
    char* sbase = stage + (m * BPM + bx) * STG_BLK;
    const F32Column* lcol = &tab->f32col[0][ju];
    const uint32_t zz_lo = lcol->zz_lo, zz_hi = lcol->zz_hi;
    const signed char* dcq_l = p.dcq_luma;
    const signed char* dcq_c = p.dcq_chroma;
    {
        f4_t acc[4];
#pragma unroll
        for (int T = 0; T < 4; ++T) acc[T] = f4_t{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int hs = 0; hs < 4; ++hs) {            // limb (low first: the small terms are accumulated first), K step
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                const h8_t afrag = __builtin_bit_cast(h8_t, atab[(hs * 4 + T) * 64 + lane]);
                acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, bfrag[hs & 1], acc[T], 0, 0, 0);
            }
        }
        const int lblk = (n_blk >> 2) * BPM + (n_blk & 3);                       // block index inside the quad's staging area
        const bool live_n = quad_x * 4 + (n_blk >> 2) < p.mcu_cols;
        // the block's quantised DC from the exact table: the sample sum (row 0 of G is ones: exact) is this lane's acc[0][0]
        int dcv = 0;
        if (g4 == 0) {
            const unsigned si = (unsigned)(__builtin_fminf(__builtin_fmaxf(acc[0][0], -8192.f), 8192.f) + 8192.f);
            dcv = dcq_l[si];
        }
        bool cand = false;
        int q[16];
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            float d[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = acc[T][r];
                q[4 * T + r] = (int)t;
                d[r] = __builtin_fabsf(t - __builtin_rintf(t));
            }
            if (T == 0 && g4 == 0) d[0] = 1.f;                                  // the DC never uses the guard band
            const float dmin = __builtin_fminf(__builtin_fminf(d[0], d[1]), __builtin_fminf(d[2], d[3]));
            cand = cand || dmin < tab->mfma_delta[T][g4];
        }
        if (g4 == 0) q[0] = dcv;
#ifdef JPEZY_DUMP_T
        if (live_n && dump_quad)
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) dump_quad[lblk * 64 + (int)c_zz[16 * T + 4 * g4 + r]] = acc[T][r];
#endif
        if (FORCE == 0 && wave_any(cand)) {
            if (cand && live_n) {
#pragma unroll
                for (int T = 0; T < 4; ++T)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t = acc[T][r];
                        const int pz = 16 * T + 4 * g4 + r;
                        if (__builtin_fabsf(t - __builtin_rintf(t)) < tab->mfma_delta[T][g4] && __builtin_fabsf(t) > 0.5f && pz != 0) {
                            const unsigned slot = atomicAdd(&queue[0], 1u);
                            if (slot < (unsigned)QUEUE2_CAP)
                                reinterpret_cast<unsigned short*>(queue + 1)[slot] = (unsigned short)((lblk << 6) | (int)c_zz[pz]);
                        }
                    }
            }
        }
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            const uint32_t lo = ((uint32_t)q[4 * T] & 0xFFFFu) | ((uint32_t)q[4 * T + 1] << 16);
            const uint32_t hi = ((uint32_t)q[4 * T + 2] & 0xFFFFu) | ((uint32_t)q[4 * T + 3] << 16);
            *reinterpret_cast<uint2*>(stage + lblk * STG_BLK + 32 * T + 8 * g4) = make_uint2(lo, hi);
        }
    }

    __builtin_amdgcn_sched_barrier(0);
    // ---- 5. chroma row pass, transpose, column pass ----
    if (!GRAY) {
        const bool odd = (row & 1) != 0;
        float cX[8];
        fdct8f(cs, cX);
        float4* dst = reinterpret_cast<float4*>(ldsf + m * C_MCU + (odd ? C_COMP : 0) + (row >> 1) * C_PITCH);
        dst[0] = make_float4(cX[0], cX[1], cX[2], cX[3]);
        dst[1] = make_float4(cX[4], cX[5], cX[6], cX[7]);
        wave_sync();

        float Fc[8];
        int dc_c;
        {
            float col[8];
            const float* src = ldsf + m * C_MCU + (cq >> 3) * C_COMP + j;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) col[rr] = src[rr * C_PITCH];
            dc_c = dc_lookup(col, dcq_c);
            fdct8f(col, Fc);
        }
        float ks[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ks[i] = lcol[8].ks[i];
        quant_block_column(Fc, ks, lcol[8].delta1, j, dc_c, live, sbase, zz_lo, zz_hi, 4 * STG_BLK, m * BPM + 4 + bx, queue, FORCE != 0 DUMP_ARG, QUEUE2_CAP);
    }
    wave_sync();

    // ---- 5b. levels 2 and 3 for the queued coefficients (FORCE 1/2: every coefficient of the quad) ----
    const unsigned nq = queue[0];
    if (FORCE == 3 || (FORCE == 0 && nq > (unsigned)QUEUE2_CAP)) {
        // More guard-band hits than the queue holds (adversarial patterns; FORCE 3 exercises it): every lane evaluates
        // the 24 coefficients of its three block columns in the reference's order by itself.  The integer samples go
        // to LDS as bytes (1.5 KB in the dead chroma tile); a lane walks its block row by row, keeps the eight running
        // sums of its column (i = 0..7) and adds (pic * cos[j][x]) * cos[i][y] for x = 0..7 to each -- for every i
        // exactly the reference's sequence (ref :146-166).  ~3,500 FP64 operations per lane, 7 us per wave, against
        // ~1 ms for the cooperative path on all 1536 coefficients.
        signed char* smp = reinterpret_cast<signed char*>(lds);
        {
            uint32_t w4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                w4[k] = ((uint32_t)(int)ys[4 * k] & 0xFFu) | (((uint32_t)(int)ys[4 * k + 1] & 0xFFu) << 8) |
                        (((uint32_t)(int)ys[4 * k + 2] & 0xFFu) << 16) | (((uint32_t)(int)ys[4 * k + 3] & 0xFFu) << 24);
            const int by = row >> 3, y = row & 7;
            uint32_t* d0 = reinterpret_cast<uint32_t*>(smp + ((m * 4 + by * 2) * 64 + y * 8));
            d0[0] = w4[0]; d0[1] = w4[1];
            d0[16] = w4[2]; d0[17] = w4[3];                      // the right block, 64 bytes further
            if (!GRAY) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    w4[k] = ((uint32_t)(int)cs[4 * k] & 0xFFu) | (((uint32_t)(int)cs[4 * k + 1] & 0xFFu) << 8) |
                            (((uint32_t)(int)cs[4 * k + 2] & 0xFFu) << 16) | (((uint32_t)(int)cs[4 * k + 3] & 0xFFu) << 24);
                uint32_t* dc = reinterpret_cast<uint32_t*>(smp + 1024 + ((m * 2 + (row & 1)) * 64 + (row >> 1) * 8));
                dc[0] = w4[0]; dc[1] = w4[1];
            }
        }
        wave_sync();
        const double cu = j ? 1.0 : JPEZY_S;
#pragma unroll 1
        for (int bc = 0; bc < (GRAY ? 2 : 3); ++bc) {
            // block column bc of this lane: 0 top luma block, 1 bottom luma block, 2 chroma block (Cb / Cr by cq >> 3)
            const int blk = m * BPM + (bc < 2 ? bc * 2 + bx : 4 + bx);
            const signed char* src = bc < 2 ? smp + (m * 4 + bc * 2 + bx) * 64 : smp + 1024 + (m * 2 + bx) * 64;
            const int tbl = bc < 2 ? 0 : 1;
            double S[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll 1
            for (int y = 0; y < 8; ++y) {
                // x is not unrolled: this path must not raise the kernel's register count (it is never the hot one)
#pragma unroll 1
                for (int x = 0; x < 8; ++x) {
                    const double px = (double)(int)src[y * 8 + x] * c_cos[j * 8 + x];
#pragma unroll
                    for (int i = 0; i < 8; ++i) S[i] += px * c_cos[i * 8 + y];
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double cv = i ? 1.0 : JPEZY_S;
                const int dct = (int)(S[i] * cu * cv / 4);
                const int qv = dct / tab->qt[tbl][i * 8 + j];
                *reinterpret_cast<int16_t*>(stage + blk * STG_BLK + 2 * (int)c_zzinv[i * 8 + j]) = (int16_t)qv;
            }
        }
        if (lane == 0) atomicAdd(p.fallback_count + (qidx & (COUNTER_SHARDS - 1)), (unsigned long long)(4 * BPM * 64));
        wave_sync();
    } else {
        const bool all = FORCE != 0;
        const unsigned total = all ? (unsigned)(4 * BPM * 64) : nq;
        if (total) {
            const int valid_mcus = min(4, p.mcu_cols - quad_x * 4);
            unsigned done = 0;
#pragma unroll 1
            for (unsigned e = 0; e < total; ++e) {
                const unsigned code = all ? e : reinterpret_cast<const unsigned short*>(queue + 1)[e];
                const int blk = __builtin_amdgcn_readfirstlane((int)(code >> 6)), nat = __builtin_amdgcn_readfirstlane((int)(code & 63));
                const int em = blk / BPM, eb = blk - em * BPM;
                if (em >= valid_mcus) continue;
                const int ei = nat >> 3, ej = nat & 7;
                const int comp = eb < 4 ? 0 : eb - 3, tbl = comp ? 1 : 0;
                // the 8 lanes that hold the block's rows: luma block (by,bx): rows by*8+y, samples ys[8bx..]; chroma:
                // Cb on even-row lanes, Cr on odd-row lanes, samples cs[]
                const int by = (eb >> 1) & 1, bx = eb & 1;
                const int first = comp ? (comp == 2 ? 4 : 0) + em : by * 32 + em;
                const int stride = comp ? 8 : 4;
                const bool part = (m == em) && (comp ? ((row & 1) == (comp == 2)) : ((row >> 3) == by));
                const int yrow = comp ? (row >> 1) : (row & 7);
                float w[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) w[k] = comp ? cs[k] : (bx ? ys[8 + k] : ys[k]);
                const int qv = resolve_coef<FORCE>(w, part, yrow, first, stride, ei, ej, tab->qt[tbl][nat], tab->qinv[tbl][nat]);
                if (lane == 0) *reinterpret_cast<int16_t*>(stage + blk * STG_BLK + 2 * (int)c_zzinv[nat]) = (int16_t)qv;
                ++done;
            }
            if (lane == 0 && done) atomicAdd(p.fallback_count + (qidx & (COUNTER_SHARDS - 1)), (unsigned long long)done);
            wave_sync();
        }
    }

#ifdef JPEZY_TRACE
    const unsigned long long tr_t2 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- 6. coalesced store of the quad's coefficients ----
    {
        const int valid_chunks = min(4, p.mcu_cols - quad_x * 4) * BPM * 8;         // 16-byte chunks
        int16_t* gbase = p.coeffs + (size_t)frame * p.coeffs_per_frame +
                         ((size_t)mcu_y * p.mcu_cols + (size_t)quad_x * 4) * (BPM * 64);
        uint4* g4 = reinterpret_cast<uint4*>(gbase);
#pragma unroll
        for (int k = 0; k < BPM * 128 * 4 / 1024; ++k) {
            const int c = k * 64 + lane;
            if (c < valid_chunks) {
                // streamed out, never re-read by this kernel: a non-temporal store leaves less dirty data in the L2s
                // for the end-of-kernel write-back (measured: 2 us per 4096^2 frame)
                const uint4 v = *reinterpret_cast<const uint4*>(stage + (c >> 3) * STG_BLK + (c & 7) * 16);
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(v4u{v.x, v.y, v.z, v.w}, reinterpret_cast<v4u*>(g4 + c));
            }
        }
    }
#ifdef JPEZY_TRACE
    if (frame == 0 && qidx < 65536u) {
#if JPEZY_TRACE > 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        const unsigned long long tr_t3 = __builtin_amdgcn_s_memrealtime();
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        if (lane == 0) {
            p.trace[qidx * 4 + 0] = tr_t0;
            p.trace[qidx * 4 + 1] = ((tr_t1 - tr_t0) << 32) | (tr_t2 - tr_t0);
            p.trace[qidx * 4 + 2] = tr_t3 - tr_t0;
            p.trace[qidx * 4 + 3] = ((unsigned long long)xcc << 32) | hw;
        }
    }
#endif
}


}  // namespace f32

template <bool GRAY, bool ALIGNED>
static void enc_f32_launch2(const EncParams& p, int force, dim3 grid, hipStream_t s)
{
    if (force == 1)
        hipLaunchKernelGGL((f32::fdct_quant_f32_kernel<GRAY, ALIGNED, 1>), grid, dim3(64 * WPB), 0, s, p);
    else if (force == 2)
        hipLaunchKernelGGL((f32::fdct_quant_f32_kernel<GRAY, ALIGNED, 2>), grid, dim3(64 * WPB), 0, s, p);
    else if (force == 3)
        hipLaunchKernelGGL((f32::fdct_quant_f32_kernel<GRAY, ALIGNED, 3>), grid, dim3(64 * WPB), 0, s, p);
    else
        hipLaunchKernelGGL((f32::fdct_quant_f32_kernel<GRAY, ALIGNED, 0>), grid, dim3(64 * WPB), 0, s, p);
}

template <bool GRAY, bool ALIGNED>
static void enc_mfma_launch2(const EncParams& p, int force, dim3 grid, hipStream_t s)
{
    if (force == 1)
        hipLaunchKernelGGL((f32::fdct_quant_mfma_kernel<GRAY, ALIGNED, 1>), grid, dim3(64 * f32::WPB2), 0, s, p);
    else if (force == 2)
        hipLaunchKernelGGL((f32::fdct_quant_mfma_kernel<GRAY, ALIGNED, 2>), grid, dim3(64 * f32::WPB2), 0, s, p);
    else if (force == 3)
        hipLaunchKernelGGL((f32::fdct_quant_mfma_kernel<GRAY, ALIGNED, 3>), grid, dim3(64 * f32::WPB2), 0, s, p);
    else
        hipLaunchKernelGGL((f32::fdct_quant_mfma_kernel<GRAY, ALIGNED, 0>), grid, dim3(64 * f32::WPB2), 0, s, p);
}

hipError_t launch_fdct_quant_mfma(const EncParams& p, bool gray, int force, hipStream_t stream)
{
    const long quads = (long)p.mcu_rows * p.quads_per_row;
    if (quads <= 0 || p.n_frames <= 0) return hipSuccess;
    if (p.n_frames > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((quads + f32::WPB2 - 1) / f32::WPB2), (unsigned)p.n_frames);
    const bool al = (p.W % 16 == 0) && (p.plane_stride % 16 == 0) &&
                    (((uintptr_t)p.r | (uintptr_t)p.g | (uintptr_t)p.b) % 16 == 0);
    if (gray) { if (al) enc_mfma_launch2<true, true>(p, force, grid, stream); else enc_mfma_launch2<true, false>(p, force, grid, stream); }
    else      { if (al) enc_mfma_launch2<false, true>(p, force, grid, stream); else enc_mfma_launch2<false, false>(p, force, grid, stream); }
    return hipGetLastError();
}

hipError_t launch_fdct_quant_f32(const EncParams& p, bool gray, int force, hipStream_t stream)
{
    const long quads = (long)p.mcu_rows * p.quads_per_row;
    if (quads <= 0 || p.n_frames <= 0) return hipSuccess;
    if (p.n_frames > 65535) return hipErrorInvalidValue;               // grid.y limit; callers chunk larger batches
    const dim3 grid((unsigned)((quads + WPB - 1) / WPB), (unsigned)p.n_frames);
    const bool al = (p.W % 16 == 0) && (p.plane_stride % 16 == 0) &&
                    (((uintptr_t)p.r | (uintptr_t)p.g | (uintptr_t)p.b) % 16 == 0);
    if (gray) { if (al) enc_f32_launch2<true, true>(p, force, grid, stream); else enc_f32_launch2<true, false>(p, force, grid, stream); }
    else      { if (al) enc_f32_launch2<false, true>(p, force, grid, stream); else enc_f32_launch2<false, false>(p, force, grid, stream); }
    return hipGetLastError();
}

}  // namespace jpezy_dev
