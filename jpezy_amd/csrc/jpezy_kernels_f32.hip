// jpezy_kernels_f32.hip -- encode kernel, variant 1: three precision levels, same bits as the reference.
//
// Level 1 (every coefficient): colour conversion as an FP32 estimate with a guard band (below), separable 8-point
//   butterflies in FP32, written as PACKED FP32 instructions (v_pk_add/mul/fma_f32 with op_sel / neg modifiers: one
//   instruction produces the sum AND the difference of a butterfly, or one product term for two outputs): 17
//   instructions per 8-point transform instead of 34.  Every result is the same IEEE operation on the same operands as
//   in the scalar form, so the error bound is the scalar form's.  A quantised coefficient t = F*cu*cv/(4Q) is accepted
//   when it is further than delta1 from every non-zero integer; delta1 = 1.25 x the worst-case FP32 error of t over the
//   coefficients of the lane's block column (DeviceTables::delta1, at most 1.06e-4 luma / 5.8e-5 chroma with the
//   Annex-K tables; DESIGN.md "exactness").  The test is one-sided: the kernel forms t' = fma(F, ks, delta1) and looks
//   at fract(t') < 2 delta1 (t within delta1 of an integer on either side <=> t' in [n, n + 2 delta1)).
// Level 2 (guard-band hits, ~0.1 per quad): the 8 lanes holding the block's rows recompute that one coefficient in FP64
//   from the integer samples; accepted when further than 1e-6 from every boundary m*Q, m != 0.
// Level 3 (true boundary cases, ~0.2 per quad): the 64 terms are added in the reference's exact order.
// Colour conversion: Y = trunc(fma chain in FP32) is exact unless the exact value is an integer, which happens exactly
//   when 299R+587G+114B is a multiple of 1000 (1 pixel in 1000): there the reference's own FP64 rounding decides and the
//   FP64 formula is evaluated.  Same for Cb/Cr (multiples of 10000).  Also one-sided: the chain starts from a bias.
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
#define K5 0x1.1c73b39ae68c8p-1f
#define K6 0x1.87de2a6aea963p-2f
#define K7 0x1.8f8b83c69a60bp-3f
#ifndef JPEZY_F32_WAVES
#define JPEZY_F32_WAVES 5
#endif
// instruction-selection switches (same arithmetic, same results): packed FP32 forms of the transform / of the fused
// multiply-adds of the luma and chroma estimates / of the quantiser's product-and-bias
#ifndef JPEZY_PK_TRANSFORM
#define JPEZY_PK_TRANSFORM 1
#endif
#ifndef JPEZY_PK_LUMA
#define JPEZY_PK_LUMA 1
#endif
#ifndef JPEZY_PK_CHROMA
#define JPEZY_PK_CHROMA 1
#endif
#ifndef JPEZY_PK_QUANT
#define JPEZY_PK_QUANT 1
#endif
#ifndef JPEZY_PIN_CONSTANTS
#define JPEZY_PIN_CONSTANTS 1
#endif

typedef float f2 __attribute__((ext_vector_type(2)));   // an aligned VGPR (or SGPR) pair: the operand of v_pk_*_f32

// Workgroup = EWPB waves = EWPB horizontally adjacent quads (4: 256 pixels x 16 rows).  The waves share nothing but the
// pixel load: with JPEZY_COOP_LOAD the workgroup fetches its rows in whole 256-byte pieces (16 lanes per row, 4 rows per
// wave instruction, LDS-DMA) and every wave then picks its quad's 64-byte row segments out of LDS; without it each wave
// loads its own 64-byte segments of 16 rows per instruction straight into registers -- the shape the texture addresser
// handles worst (tools/ubench/mem_pattern.hip: the kernel's memory pattern alone, no arithmetic, 22.4 us per 4096^2
// frame in that shape against 19.4 us in whole 256-byte pieces).
// The launcher picks the 4-wave form where a row of quads divides into groups of four (4096 and 7680 wide frames: 64 and 120
// quads per row) and 2-wave workgroups with direct loads elsewhere (1920 wide: 30 quads -- groups of four would leave two of
// every 32 wave slots idle, measured +3.8 % on the 32 x 1080p batch).
#ifndef JPEZY_COOP_LOAD
#define JPEZY_COOP_LOAD 1
#endif

// Level-1 guard bands on t = v/Q: DeviceTables::delta1[table][j].  Norm-wise bound of the FP32 error of F[i][j]:
// gamma_13 * sum|cos_i| * sum|cos_j| * 128 (at most 13 roundings on any input->output path, u = 2^-24), times the
// coefficient's scale factor cu*cv/(4Q), plus the roundings of ks and of the fused product-and-bias.
// tests/test_f32_error_bound.py recomputes the table and measures errors 10x smaller on adversarial blocks.
constexpr double DELTA2 = 1e-6;             // level-2 guard band on v (FP64 tree-sum error < 1e-9)

// LDS geometry in dwords (floats).  Column reads are ds_read_b32 over 32-lane groups (32 banks): conflict free
// when the per-MCU stride == 8 (mod 32); row writes are 16-byte stores over 8-lane groups: pitch 20 keeps them apart.
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

// ---- the 8-point transform in packed FP32 ----------------------------------------------------------------------------
// VOP3P on 64-bit operands: op_sel[i] picks the dword of source i that feeds the LOW result, op_sel_hi[i] the one that
// feeds the HIGH result, neg_lo/neg_hi negate a source per half.  Written as inline asm: hipcc folds neither the negations
// nor the two-sided broadcasts into the modifiers (it builds the pairs with v_mov / v_xor instead).  The statements are
// not volatile: the scheduler interleaves them like any other instruction.
#define PK_ADD(d, a, b, MODS) asm("v_pk_add_f32 %0, %1, %2 " MODS : "=v"(d) : "v"(a), "v"(b))
#define PK_MULS(d, a, k, MODS) asm("v_pk_mul_f32 %0, %1, %2 " MODS : "=v"(d) : "v"(a), "s"(k))
#define PK_FMAS(d, a, k, c, MODS) asm("v_pk_fma_f32 %0, %1, %2, %3 " MODS : "=v"(d) : "v"(a), "s"(k), "v"(c))

struct PkCos {             // five SGPR pairs; every other constant pair of the transform is one of these with its halves
    f2 k13, k37, k51, k75, k26;   // swapped and / or negated by the modifiers
};
__device__ __forceinline__ PkCos pk_cos()
{
    return PkCos{ f2{ K1, K3 }, f2{ K3, K7 }, f2{ K5, K1 }, f2{ K7, K5 }, f2{ K2, K6 } };
}

// Output order of fdct8p as a sequence of eight values: position p holds coefficient pair_row(p).  Tiles, quantiser
// records and zig-zag offsets are laid out in this order, so no value is ever moved between registers to re-pair it.
__device__ __forceinline__ constexpr int pair_row(int p) { return (int)((0x75316240u >> (4 * p)) & 7u); }   // 0,4,2,6,1,3,5,7

// A[k] = (x[k], x[7-k]), k = 0..3.  X[0] = (X0, X4), X[1] = (X2, X6), X[2] = (X1, X3), X[3] = (X5, X7) with
// X[u] = sum_x x[x] * cos((2x+1)u*pi/16), except X4, which is left WITHOUT its factor cos(pi/4): both passes' factors
// are folded into the quantiser scale ks (F32Column::ks carries cos(pi/4) per index 4).  X0 is the plain sum (exact
// for integers below 2^24).  Operation for operation the scalar sequence
//   s_k = x_k + x_{7-k}, d_k = x_k - x_{7-k}; e0 = s0 + s3, e2 = s0 - s3, e1 = s1 + s2, e3 = s1 - s2;
//   X0 = e0 + e1, X4 = e0 - e1; X2 = fma(e3, K6, e2*K2), X6 = fma(-e3, K2, e2*K6);
//   X1 = fma(d3,K7, fma(d2,K5, fma(d1,K3, d0*K1))), X3 = fma(-d3,K5, fma(-d2,K1, fma(-d1,K7, d0*K3))),
//   X5 = fma(d3,K3, fma(d2,K7, fma(-d1,K1, d0*K5))), X7 = fma(-d3,K1, fma(d2,K3, fma(-d1,K5, d0*K7)))
// (tests/test_f32_error_bound.py emulates exactly this), two results per instruction: 17 instructions.
#define FMAF(a, b, c) __builtin_fmaf((a), (b), (c))
__device__ __forceinline__ void fdct8p(const f2* A, f2* X, const PkCos& c)
{
#if !JPEZY_PK_TRANSFORM
    const float s0 = A[0].x + A[0].y, s1 = A[1].x + A[1].y, s2 = A[2].x + A[2].y, s3 = A[3].x + A[3].y;
    const float d0 = A[0].x - A[0].y, d1 = A[1].x - A[1].y, d2 = A[2].x - A[2].y, d3 = A[3].x - A[3].y;
    const float e0 = s0 + s3, e1 = s1 + s2, e2 = s0 - s3, e3 = s1 - s2;
    X[0] = f2{ e0 + e1, e0 - e1 };
    X[1] = f2{ FMAF(e3, K6, e2 * K2), FMAF(-e3, K2, e2 * K6) };
    X[2] = f2{ FMAF(d3, K7, FMAF(d2, K5, FMAF(d1, K3, d0 * K1))), FMAF(-d3, K5, FMAF(-d2, K1, FMAF(-d1, K7, d0 * K3))) };
    X[3] = f2{ FMAF(d3, K3, FMAF(d2, K7, FMAF(-d1, K1, d0 * K5))), FMAF(-d3, K1, FMAF(d2, K3, FMAF(-d1, K5, d0 * K7))) };
    return;
#endif
    f2 P0, P1, P2, P3, Q0, Q1, t, u;
    PK_ADD(P0, A[0], A[0], "op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]");   // (s0, d0) = x0 +- x7
    PK_ADD(P1, A[1], A[1], "op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]");
    PK_ADD(P2, A[2], A[2], "op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]");
    PK_ADD(P3, A[3], A[3], "op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]");
    PK_ADD(Q0, P0, P3, "op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]");       // (e0, e2) = s0 +- s3
    PK_ADD(Q1, P1, P2, "op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]");       // (e1, e3) = s1 +- s2
    PK_ADD(X[0], Q0, Q1, "op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]");     // (X0, X4) = e0 +- e1
    PK_MULS(t, Q0, c.k26, "op_sel:[1,0] op_sel_hi:[1,1]");                                         // e2 * (K2, K6)
    PK_FMAS(X[1], Q1, c.k26, t, "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]");                // + e3 * (K6, -K2)
    PK_MULS(t, P0, c.k13, "op_sel:[1,0] op_sel_hi:[1,1]");                                         // d0 * (K1, K3)
    PK_FMAS(t, P1, c.k37, t, "op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_hi:[0,1,0]");                   // + d1 * (K3, -K7)
    PK_FMAS(t, P2, c.k51, t, "op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_hi:[0,1,0]");                   // + d2 * (K5, -K1)
    PK_FMAS(X[2], P3, c.k75, t, "op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_hi:[0,1,0]");                // + d3 * (K7, -K5)
    PK_MULS(u, P0, c.k75, "op_sel:[1,1] op_sel_hi:[1,0]");                                         // d0 * (K5, K7)
    PK_FMAS(u, P1, c.k51, u, "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]");    // + d1 * (-K1, -K5)
    PK_FMAS(u, P2, c.k37, u, "op_sel:[1,1,0] op_sel_hi:[1,0,1]");                                  // + d2 * (K7, K3)
    PK_FMAS(X[3], P3, c.k13, u, "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]");                // + d3 * (K3, -K1)
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
//   t' = fma(.114f, B, fma(.587f, G, fma(.299f, R, -128 + eps)))
// is within 2.4e-5 of y* + eps (three roundings of at most 2^-18 each, three constants rounded to FP32: 1.2e-5), and y* is
// either an integer or at least 1e-3 away from one.  With eps = LUMA_EPS = 2^-12 (2.4e-4): y* integral <=> fract(t')
// in [eps - 2.4e-5, eps + 2.4e-5], inside [0, 2 eps); y* not integral => fract(t') in [1e-3 + eps - 2.4e-5, 1 - 1e-3 +
// eps + 2.4e-5], above 2 eps and below 1, and t' lies strictly between the same two integers as y*: trunc(t') =
// trunc(y*).  So the test is ONE comparison of fract(t') (v_fract_f32: x - floor(x), exact, either sign) against 2 eps,
// reduced over a half row with v_min3; the flagged pixels -- exactly the 1-in-1000 with y* integral, where the
// reference's own FP64 rounding sequence decides -- evaluate the FP64 formula.  Chroma: c* = N / 10000 (N integer),
// FP32 error 1.7e-5, non-integral c* at least 1e-4 from an integer; bias 3 * 2^-16 (4.6e-5), threshold 3 * 2^-15: margins
// of 2.9e-5 / 3.7e-5 on the three inequalities.  tests/test_f32_error_bound.py checks all 2^24 RGB triples.
constexpr float LUMA_EPS = 0x1p-12f, LUMA_TH = 0x1p-11f;
constexpr float CHROMA_EPS = 0x1.8p-15f, CHROMA_TH = 0x1.8p-14f;

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// two luma samples (byte B0 of the first word triple, byte B1 of the second): y = trunc(t'), fr = fract(t')
template <int B0, int B1>
__device__ __forceinline__ void luma_px2(uint32_t r0, uint32_t g0, uint32_t b0, uint32_t r1, uint32_t g1, uint32_t b1, f2& y, f2& fr)
{
    const f2 r = { ubyte<B0>(r0), ubyte<B1>(r1) }, g = { ubyte<B0>(g0), ubyte<B1>(g1) }, b = { ubyte<B0>(b0), ubyte<B1>(b1) };
#if JPEZY_PK_LUMA
    const f2 c1 = { 0.299f, 0.299f }, c2 = { 0.587f, 0.587f }, c3 = { 0.114f, 0.114f }, c0 = { -128.f + LUMA_EPS, -128.f + LUMA_EPS };
    const f2 t = pk_fma(c3, b, pk_fma(c2, g, pk_fma(c1, r, c0)));
#else
    const f2 t = { FMAF(0.114f, b.x, FMAF(0.587f, g.x, FMAF(0.299f, r.x, -128.f + LUMA_EPS))),
                   FMAF(0.114f, b.y, FMAF(0.587f, g.y, FMAF(0.299f, r.y, -128.f + LUMA_EPS))) };
#endif
    y = f2{ __builtin_truncf(t.x), __builtin_truncf(t.y) };
    fr = f2{ __builtin_amdgcn_fractf(t.x), __builtin_amdgcn_fractf(t.y) };
}
template <int B>
__device__ __forceinline__ float luma_px_ref(uint32_t wr, uint32_t wg, uint32_t wb)
{
    return (float)ref_y((double)ubyte<B>(wr), (double)ubyte<B>(wg), (double)ubyte<B>(wb));
}
__device__ __forceinline__ float min8(const f2* e)
{
    return __builtin_fminf(__builtin_fminf(__builtin_fminf(e[0].x, e[0].y), __builtin_fminf(e[1].x, e[1].y)),
                           __builtin_fminf(__builtin_fminf(e[2].x, e[2].y), __builtin_fminf(e[3].x, e[3].y)));
}

// The 8 luma samples of one block row: words w[0..1] of the three planes (pixels 0..7) -> A[k] = (Y[k], Y[7-k]),
// the input form of fdct8p.  Pixel k is byte k of word 0, pixel 7-k byte 3-k of word 1.
__device__ __forceinline__ void luma8(const uint32_t* wr, const uint32_t* wg, const uint32_t* wb, f2* A)
{
    f2 e[4];
    luma_px2<0, 3>(wr[0], wg[0], wb[0], wr[1], wg[1], wb[1], A[0], e[0]);
    luma_px2<1, 2>(wr[0], wg[0], wb[0], wr[1], wg[1], wb[1], A[1], e[1]);
    __builtin_amdgcn_sched_barrier(0);   // 4 pixels at a time: more in flight only costs registers
    luma_px2<2, 1>(wr[0], wg[0], wb[0], wr[1], wg[1], wb[1], A[2], e[2]);
    luma_px2<3, 0>(wr[0], wg[0], wb[0], wr[1], wg[1], wb[1], A[3], e[3]);
#ifdef JPEZY_ABL_NOCFLAG    // timing probe (wrong results, tools/ab/ab_build.py): what the colour guard tests and their rare path cost
    if (false) {
#else
    if (wave_any(min8(e) < LUMA_TH)) {   // one pixel in 1000: the reference's FP64 rounding decides
#endif
        bool f;
        f = e[0].x < LUMA_TH; if (f) A[0].x = luma_px_ref<0>(wr[0], wg[0], wb[0]);
        f = e[1].x < LUMA_TH; if (f) A[1].x = luma_px_ref<1>(wr[0], wg[0], wb[0]);
        f = e[2].x < LUMA_TH; if (f) A[2].x = luma_px_ref<2>(wr[0], wg[0], wb[0]);
        f = e[3].x < LUMA_TH; if (f) A[3].x = luma_px_ref<3>(wr[0], wg[0], wb[0]);
        f = e[3].y < LUMA_TH; if (f) A[3].y = luma_px_ref<0>(wr[1], wg[1], wb[1]);
        f = e[2].y < LUMA_TH; if (f) A[2].y = luma_px_ref<1>(wr[1], wg[1], wb[1]);
        f = e[1].y < LUMA_TH; if (f) A[1].y = luma_px_ref<2>(wr[1], wg[1], wb[1]);
        f = e[0].y < LUMA_TH; if (f) A[0].y = luma_px_ref<3>(wr[1], wg[1], wb[1]);
    }
}

// two chroma samples (Cb on even-row lanes, Cr on odd-row lanes); k1..k3: this lane's three coefficients
template <int B0, int B1>
__device__ __forceinline__ void chroma_px2(uint32_t r0, uint32_t g0, uint32_t b0, uint32_t r1, uint32_t g1, uint32_t b1, float k1,
                                           float k2, float k3, f2& c, f2& fr)
{
    const f2 r = { ubyte<B0>(r0), ubyte<B1>(r1) }, g = { ubyte<B0>(g0), ubyte<B1>(g1) }, b = { ubyte<B0>(b0), ubyte<B1>(b1) };
#if JPEZY_PK_CHROMA
    const f2 q1 = { k1, k1 }, q2 = { k2, k2 }, q3 = { k3, k3 }, q0 = { CHROMA_EPS, CHROMA_EPS };
    const f2 t = pk_fma(q3, b, pk_fma(q2, g, pk_fma(q1, r, q0)));
#else
    const f2 t = { FMAF(k3, b.x, FMAF(k2, g.x, FMAF(k1, r.x, CHROMA_EPS))), FMAF(k3, b.y, FMAF(k2, g.y, FMAF(k1, r.y, CHROMA_EPS))) };
#endif
    c = f2{ __builtin_truncf(t.x), __builtin_truncf(t.y) };
    fr = f2{ __builtin_amdgcn_fractf(t.x), __builtin_amdgcn_fractf(t.y) };
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
// The small tables of levels 2 and 3 as the persistent kernel keeps them in LDS (its loop holds no vector-memory load: a wait
// for one -- vmcnt counts in issue order -- would also wait for the previous quad's coefficient stores).
struct PsTables {
    double cos[64];           // c_cos
    double qinv[2][64];       // DeviceTables::qinv
    int qt[2][64];            // DeviceTables::qt
    unsigned char zzinv[64];  // c_zzinv
};

template <int FORCE, bool PS>
__device__ __forceinline__ int resolve_coef(const float* w, bool part, int y, int first, int stride, int i, int j,
                                            int Q, double qinv, const PsTables* pst)
{
    double t[8];
    {
        double cy;
        if (PS) cy = pst->cos[i * 8 + y]; else cy = c_cos[i * 8 + y];
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

// Quantised DC of a block from the exact table (DeviceTables::dcq): sum = the lane's X0 of the column pass -- on the
// j == 0 lane that is the block's integer sample sum (exact in FP32).  Issued right after the first adds of the column
// pass, long before the value is needed, so the L2 latency never sits on a wave's critical path.
__device__ __forceinline__ int dc_lookup(float sum, const signed char* dcq)
{
    // index = sum + 8192, formed in FP32 (exact) and clamped there (non-DC lanes carry arbitrary values); an
    // unsigned index keeps the lookup a scalar-base + 32-bit-offset load
    const unsigned si = (unsigned)(__builtin_amdgcn_fmed3f(sum, -8192.f, 8192.f) + 8192.f);
    return dcq[si];
}

// Quantise one block column and stage it in zig-zag order.  F: the column pass' four output pairs (order pair_row);
// ks: the quantiser scales in the same order; dd = (delta1, delta1), th = 2 delta1; j: natural column; dc: the block's
// quantised DC (only the j == 0 lane uses it); base: LDS address of this lane's FIRST block, blk_off the byte offset of
// the block to write (an immediate after inlining); zz_lo/zz_hi: byte p = LDS byte offset of the coefficient at pair
// position p inside a block (2 * zig-zag index < 128), packed so that the eight addresses cost two registers.
__device__ __forceinline__ void quant_block_column(const f2* F, const f2* ks, f2 dd, float th, int j, int dc,
                                                   bool live, char* base, uint32_t zz_lo, uint32_t zz_hi, int blk_off, int blk,
                                                   unsigned* queue, bool force
#ifdef JPEZY_DUMP_T
                                                   , float* dump_quad
#endif
                                                   )
{
    // t' = F * ks + delta1 (one rounding); q = (int)t' truncates toward zero like the reference's integer division;
    // fr = fract(t') < 2 delta1 <=> the unbiased t is within delta1 of an integer (header comment)
    f2 t[4];
#pragma unroll
#if JPEZY_PK_QUANT
    for (int p = 0; p < 4; ++p) t[p] = pk_fma(F[p], ks[p], dd);
#else
    for (int p = 0; p < 4; ++p) t[p] = f2{ FMAF(F[p].x, ks[p].x, dd.x), FMAF(F[p].y, ks[p].y, dd.x) };
#endif
#ifdef JPEZY_DUMP_T   // diagnostic build: the level-1 values as the guard test sees them, bias removed
    if (live && dump_quad)
#pragma unroll
        for (int p = 0; p < 8; ++p) dump_quad[blk * 64 + pair_row(p) * 8 + j] = (p & 1 ? t[p >> 1].y : t[p >> 1].x) - dd.x;
#endif
    int q[8];
    float fr[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const float tp = p & 1 ? t[p >> 1].y : t[p >> 1].x;
        q[p] = (int)tp;                                                   // v_cvt_i32_f32 truncates toward zero
        fr[p] = __builtin_amdgcn_fractf(tp);
    }
    if (j == 0) { q[0] = dc; fr[0] = 1.f; }                               // the DC term: exact table, no guard band
    // v_min3_f32: 3.5 instructions for 8 values
    const float fmin = __builtin_fminf(__builtin_fminf(__builtin_fminf(fr[0], fr[1]), __builtin_fminf(fr[2], fr[3])),
                                       __builtin_fminf(__builtin_fminf(fr[4], fr[5]), __builtin_fminf(fr[6], fr[7])));
#ifdef JPEZY_ABL_NOGUARD    // timing probe (wrong results): what the coefficient guard tests and levels 2/3 cost
    const bool cand = false;
#else
    const bool cand = force || fmin < th;
#endif
    // rare on noisy content; flat content (exact zeros) enters and finds nothing to queue.  (Lanes that are not live
    // repeat the quad's last MCU, so leaving them in the vote changes nothing and keeps it a bare v_cmp + s_cmp.)
    if (wave_any(cand)) {
        if (cand && live) {       // Fully unrolled: a runtime index into the arrays would send them to scratch
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float tp = p & 1 ? t[p >> 1].y : t[p >> 1].x;
                // near an integer other than zero (a band around zero is not a truncation boundary)
                bool f = fr[p] < th && __builtin_fabsf(tp) > 0.5f && !(p == 0 && j == 0);
                if (force) f = true;
                if (f) {
                    const unsigned slot = atomicAdd(&queue[0], 1u);
                    if (slot < (unsigned)QUEUE_CAP)
                        reinterpret_cast<unsigned short*>(queue + 1)[slot] = (unsigned short)((blk << 6) | (pair_row(p) * 8 + j));
                }
            }
        }
    }
#pragma unroll
    for (int pp = 1; pp <= 8; ++pp) {      // p = 0 last: it waits for the DC lookup
        const int p = pp & 7;
        // (one SDWA add per store; the eight addresses formed once per lane and kept in registers measured 1 us slower)
        const uint32_t off = ((p < 4 ? zz_lo : zz_hi) >> (8 * (p & 3))) & 0xFFu;
        *reinterpret_cast<int16_t*>(base + off + blk_off) = (int16_t)q[p];
    }
}

// One block column out of a row-major LDS tile, paired as fdct8p wants it: A[k] = (tile[k], tile[7-k]) (rows PITCH dwords
// apart).  ds_read2_b32 takes two independent offsets, so every pair arrives in its register pair; left to itself hipcc
// merges the eight loads by ADJACENT rows and re-pairs them with eight v_mov.  The wait is part of the statement (the
// compiler does not count LDS loads issued by inline asm); "memory": not to be moved across the tile's barriers.
template <int PITCH>
__device__ __forceinline__ void lds_column(const float* src, f2* A)
{
    static_assert(7 * PITCH <= 255, "ds_read2_b32 offsets are 8-bit dword counts");
    const unsigned addr = (unsigned)(uintptr_t)src;
    asm volatile("ds_read2_b32 %0, %4 offset1:%8\n\t"
                 "ds_read2_b32 %1, %4 offset0:%5 offset1:%9\n\t"
                 "ds_read2_b32 %2, %4 offset0:%6 offset1:%10\n\t"
                 "ds_read2_b32 %3, %4 offset0:%7 offset1:%11\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(A[0]), "=&v"(A[1]), "=&v"(A[2]), "=&v"(A[3])
                 : "v"(addr), "n"(1 * PITCH), "n"(2 * PITCH), "n"(3 * PITCH), "n"(7 * PITCH), "n"(6 * PITCH), "n"(5 * PITCH), "n"(4 * PITCH)
                 : "memory");
}

// natural-order view of a row of samples held as fdct8p wants them: x-th sample of A[k] = (s[k], s[7-k])
__device__ __forceinline__ float pick(const f2* A, int x) { return x < 4 ? A[x].x : A[7 - x].y; }

// development builds (-DJPEZY_TRACE=3, tools/profile/wave_phases.py): the shader clock at the phase boundaries of every wave
#if defined(JPEZY_TRACE) && JPEZY_TRACE >= 3
#define PHASE_STAMP(k)                                                                                        \
    do {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph[k]) : : "memory");                       \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
#else
#define PHASE_STAMP(k) do { } while (0)
#endif

// The lane's quantiser records: F32Column of its block column j (luma, chroma).  The one-quad kernel loads them inside the quad
// (three 16-byte loads off one address each); the persistent kernel loads them once and keeps them in registers.
struct LaneConsts {
    f2 ks_l[4], ks_c[4];      // quantiser scales in pair order
    f2 dd_l, dd_c;            // (delta1, delta1)
    float th_l, th_c;         // 2 delta1
    uint32_t zz_lo, zz_hi;    // staged byte offsets of the column's eight coefficients
};
__device__ __forceinline__ LaneConsts load_lane_consts(const DeviceTables* tab, int lane)
{
    const unsigned ju = (0x75316240u >> (4 * ((lane >> 2) & 7))) & 7u;     // natural column j = pair_row(row & 7)
    const F32Column* lcol = &tab->f32col[0][ju];
    LaneConsts c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c.ks_l[k] = f2{ lcol->ks[2 * k], lcol->ks[2 * k + 1] };
        c.ks_c[k] = f2{ lcol[8].ks[2 * k], lcol[8].ks[2 * k + 1] };
    }
    c.dd_l = f2{ lcol->delta1[0], lcol->delta1[1] };
    c.dd_c = f2{ lcol[8].delta1[0], lcol[8].delta1[1] };
    c.th_l = lcol->th; c.th_c = lcol[8].th;
    c.zz_lo = lcol->zz_lo; c.zz_hi = lcol->zz_hi;
    return c;
}

// ---- steps 2-6 for ONE quad: everything between "the lane's 16-pixel row segments are in registers" and "the quad's 3 KB of
//      coefficients are on their way to HBM".  R, G, B: this lane's 16 bytes of each plane (lane = 4 * row + MCU); lds: the wave's
//      private slice (WAVE_LDS_DWORDS).  Shared by the one-quad-per-wave kernel and the persistent kernel below. ----
#ifdef JPEZY_TRACE
struct QuadTrace { unsigned long long t2; unsigned long long ph[8]; };
#define QUAD_TRACE_PARAM , QuadTrace* tr
#define QUAD_TRACE_ARG , &tr
#else
#define QUAD_TRACE_PARAM
#define QUAD_TRACE_ARG
#endif
// pre / dcq_lds / cos_lds (persistent kernel): the lane's records already in registers, the two quantised-DC tables
// ([2][16385] bytes) and the cosine table in LDS; null in the one-quad kernel, which reads all three from global memory.
// The scheduler fences between the phases of a quad keep the one-quad kernel at 79 VGPRs (6 waves per SIMD); the persistent kernel has
// 128 registers per lane anyway (16 waves per CU) and may let the scheduler overlap the phases (JPEZY_PS_FENCES=0).
#ifndef JPEZY_PS_DCQ_LDS
#define JPEZY_PS_DCQ_LDS 1     // 0: the quantised-DC tables stay in global memory (32 KB of LDS more for ring slots; the loop then holds three byte loads)
#endif
#ifndef JPEZY_PS_FENCES
#define JPEZY_PS_FENCES 1
#endif
#define PHASE_FENCE() do { if (!PS || JPEZY_PS_FENCES) __builtin_amdgcn_sched_barrier(0); } while (0)
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// after_pixels(): called once the raw pixel registers R, G, B are dead (behind step 2b) -- variant 3 requests the next quad's there
template <bool GRAY, int FORCE, bool PS, class AFTER_PIXELS>
__device__ __forceinline__ void encode_quad_compute(const EncParams& p, const uint32_t* R, const uint32_t* G, const uint32_t* B, uint32_t* lds,
                                            int lane, int mcu_y, int quad_x, int frame, unsigned qidx, const LaneConsts* pre,
                                            const signed char* dcq_lds, const PsTables* pst, AFTER_PIXELS after_pixels QUAD_TRACE_PARAM)
{
#if defined(JPEZY_TRACE) && JPEZY_TRACE >= 3
    unsigned long long* ph = tr->ph;
#endif
    constexpr int BPM = GRAY ? 4 : 6;
    float* ldsf = reinterpret_cast<float*>(lds);
    unsigned* queue = lds + TILE_BYTES / 4;                            // [0] = count, then 16-bit entries
    const int row = lane >> 2, m = lane & 3;
    const bool live = quad_x * 4 + m < p.mcu_cols;
    const DeviceTables* tab = p.tab;
#ifdef JPEZY_DUMP_T
    float* dump_quad = p.dump_t ? p.dump_t + (size_t)frame * p.coeffs_per_frame + ((size_t)mcu_y * p.mcu_cols + (size_t)quad_x * 4) * (BPM * 64) : nullptr;
#define DUMP_ARG , dump_quad
#else
#define DUMP_ARG
#endif
    if (lane == 0) queue[0] = 0;
#ifdef JPEZY_PROBE_SALU   // timing probe (results unchanged): JPEZY_PROBE_SALU extra scalar-ALU instructions per quad, in four places
#define PROBE_SALU() do { int d_ = lane; d_ = __builtin_amdgcn_readfirstlane(d_); _Pragma("unroll") for (int k_ = 0; k_ < JPEZY_PROBE_SALU / 4; ++k_) asm volatile("s_add_u32 %0, %0, 1" : "+s"(d_)); asm volatile("" :: "s"(d_)); } while (0)
#else
#define PROBE_SALU() do { } while (0)
#endif
#ifdef JPEZY_PROBE_VALU   // the same with full-rate vector instructions
#define PROBE_VALU() do { int d_ = lane; _Pragma("unroll") for (int k_ = 0; k_ < JPEZY_PROBE_VALU / 4; ++k_) asm volatile("v_add_u32 %0, %0, 1" : "+v"(d_)); asm volatile("" :: "v"(d_)); } while (0)
#else
#define PROBE_VALU() do { } while (0)
#endif
#ifdef JPEZY_PROBE_NOP    // s_nop 0
#define PROBE_NOP() do { _Pragma("unroll") for (int k_ = 0; k_ < JPEZY_PROBE_NOP / 4; ++k_) asm volatile("s_nop 0"); } while (0)
#else
#define PROBE_NOP() do { } while (0)
#endif
#ifdef JPEZY_PROBE_HALF   // a second-class vector instruction (v_cvt_f32_ubyte0)
#define PROBE_HALF() do { float d_ = __builtin_bit_cast(float, lane); _Pragma("unroll") for (int k_ = 0; k_ < JPEZY_PROBE_HALF / 4; ++k_) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(d_)); asm volatile("" :: "v"(d_)); } while (0)
#else
#define PROBE_HALF() do { } while (0)
#endif
#ifdef JPEZY_PROBE_PK     // a packed FP32 instruction
#define PROBE_PK() do { f2 d_ = { 1.f, 2.f }; _Pragma("unroll") for (int k_ = 0; k_ < JPEZY_PROBE_PK / 4; ++k_) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(d_)); asm volatile("" :: "v"(d_)); } while (0)
#else
#define PROBE_PK() do { } while (0)
#endif
#ifdef JPEZY_PROBE_LDS    // a 2-byte LDS store into the (still unused) queue area of the wave's slice
#define PROBE_LDS() do { _Pragma("unroll") for (int k_ = 0; k_ < JPEZY_PROBE_LDS / 4; ++k_) asm volatile("ds_write_b16 %0, %1 offset:%2" :: "v"((unsigned)(uintptr_t)(queue + 8) + 2u * (unsigned)lane), "v"(lane), "n"(0) : "memory"); } while (0)
#else
#define PROBE_LDS() do { } while (0)
#endif
#define PROBE_ALL() do { PROBE_SALU(); PROBE_VALU(); PROBE_NOP(); PROBE_HALF(); PROBE_PK(); PROBE_LDS(); } while (0)
    PROBE_ALL();
    PkCos kc = pk_cos();
#if JPEZY_PIN_CONSTANTS
    // ten SGPRs for the whole kernel: left alone, hipcc rebuilds every constant pair with s_mov_b32 in front of the packed
    // instruction that uses it (~100 scalar instructions per wave, which share the SIMD's issue with the vector ones)
    asm volatile("" : "+s"(kc.k13), "+s"(kc.k37), "+s"(kc.k51), "+s"(kc.k75), "+s"(kc.k26));
#endif
    // ---- 2. luma + row pass of the left and right block, into the transpose tile.  The integer samples stay in
    //         registers as floats, paired as the transform wants them -- YL/YR[k] = (Y[k], Y[7-k]) of the left / right
    //         block row, CS[k] likewise for the chroma row -- for the chroma row pass and the rare levels 2 and 3. ----
    f2 YL[4], YR[4], CS[4] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
    {
        f2* dst = reinterpret_cast<f2*>(ldsf + m * Y_MCU + row * Y_PITCH);
        f2 X[4];
        luma8(R, G, B, YL);
        fdct8p(YL, X, kc);
        dst[0] = X[0]; dst[1] = X[1]; dst[2] = X[2]; dst[3] = X[3];
        luma8(R + 2, G + 2, B + 2, YR);
        fdct8p(YR, X, kc);
        dst[4] = X[0]; dst[5] = X[1]; dst[6] = X[2]; dst[7] = X[3];
    }
    PHASE_FENCE();   // keep the phases apart: the scheduler otherwise overlaps them and needs more VGPRs
    PHASE_STAMP(2);
    // ---- 2b. chroma samples (top-left pixel of every 2x2, ref :134-142): the odd-row lane takes its even neighbour's
    //         pixels (DPP row_shr:4) and computes Cr, the even-row lane Cb.  Only the samples survive, so the raw
    //         pixel registers die here.  Sample s is pixel 2s: byte 2(s & 1) of word s >> 1. ----
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
        f2 e[4];
        chroma_px2<0, 2>(R2[0], G2[0], B2[0], R2[3], G2[3], B2[3], k1, k2, k3, CS[0], e[0]);    // samples 0, 7
        chroma_px2<2, 0>(R2[0], G2[0], B2[0], R2[3], G2[3], B2[3], k1, k2, k3, CS[1], e[1]);    // samples 1, 6
        PHASE_FENCE();
        chroma_px2<0, 2>(R2[1], G2[1], B2[1], R2[2], G2[2], B2[2], k1, k2, k3, CS[2], e[2]);    // samples 2, 5
        chroma_px2<2, 0>(R2[1], G2[1], B2[1], R2[2], G2[2], B2[2], k1, k2, k3, CS[3], e[3]);    // samples 3, 4
#ifdef JPEZY_ABL_NOCFLAG
        if (false) {
#else
        if (wave_any(min8(e) < CHROMA_TH)) {
#endif
            bool f;
            f = e[0].x < CHROMA_TH; if (f) CS[0].x = chroma_px_ref<0>(R2[0], G2[0], B2[0], odd);
            f = e[1].x < CHROMA_TH; if (f) CS[1].x = chroma_px_ref<2>(R2[0], G2[0], B2[0], odd);
            f = e[2].x < CHROMA_TH; if (f) CS[2].x = chroma_px_ref<0>(R2[1], G2[1], B2[1], odd);
            f = e[3].x < CHROMA_TH; if (f) CS[3].x = chroma_px_ref<2>(R2[1], G2[1], B2[1], odd);
            f = e[3].y < CHROMA_TH; if (f) CS[3].y = chroma_px_ref<0>(R2[2], G2[2], B2[2], odd);
            f = e[2].y < CHROMA_TH; if (f) CS[2].y = chroma_px_ref<2>(R2[2], G2[2], B2[2], odd);
            f = e[1].y < CHROMA_TH; if (f) CS[1].y = chroma_px_ref<0>(R2[3], G2[3], B2[3], odd);
            f = e[0].y < CHROMA_TH; if (f) CS[0].y = chroma_px_ref<2>(R2[3], G2[3], B2[3], odd);
        }
    }
    PHASE_FENCE();
    after_pixels();
    PROBE_ALL();
    PHASE_STAMP(3);
    wave_sync();

    // ---- 3+4. luma column pass, quantise + zig-zag into the staging area; top block first, then the bottom block.
    //           The row pass stored its outputs in pair order, so the lane at position c of a block row handles the
    //           natural column j = pair_row(c). ----
    const int cq = row, j = (int)((0x75316240u >> (4 * (cq & 7))) & 7u);
    [[maybe_unused]] const unsigned ju = (unsigned)j;   // unsigned table indices: scalar base + 32-bit offset addressing
    f2 TP[4], BT[4];
    lds_column<Y_PITCH>(ldsf + m * Y_MCU + cq, TP);
    lds_column<Y_PITCH>(ldsf + m * Y_MCU + cq + 8 * Y_PITCH, BT);
    PHASE_STAMP(4);
    wave_sync();   // tile consumed; the slice is reused (chroma tile | staging)
    char* stage = reinterpret_cast<char*>(lds) + CT_BYTES;
    const int bx = cq >> 3;
    char* sbase = stage + (m * BPM + bx) * STG_BLK;   // this lane's first block (m, bx); the others are immediates away
    const F32Column* lcol = &tab->f32col[0][ju];                       // (one-quad kernel: read where they are used, luma now, chroma later)
    const uint32_t zz_lo = PS ? pre->zz_lo : lcol->zz_lo, zz_hi = PS ? pre->zz_hi : lcol->zz_hi;
    const signed char* dcq_l = PS && JPEZY_PS_DCQ_LDS ? dcq_lds : p.dcq_luma;
    const signed char* dcq_c = PS && JPEZY_PS_DCQ_LDS ? dcq_lds + 16385 : p.dcq_chroma;
    {
        f2 ks[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ks[k] = PS ? pre->ks_l[k] : f2{ lcol->ks[2 * k], lcol->ks[2 * k + 1] };
        const f2 dd = PS ? pre->dd_l : f2{ lcol->delta1[0], lcol->delta1[1] };
        const float th = PS ? pre->th_l : lcol->th;
        {
            f2 F[4];
            fdct8p(TP, F, kc);
            const int dc_top = dc_lookup(F[0].x, dcq_l);
            quant_block_column(F, ks, dd, th, j, dc_top, live, sbase, zz_lo, zz_hi, 0, m * BPM + bx, queue, FORCE != 0 DUMP_ARG);
        }
        PHASE_FENCE();
        {
            f2 F[4];
            fdct8p(BT, F, kc);
            const int dc_bot = dc_lookup(F[0].x, dcq_l);
            quant_block_column(F, ks, dd, th, j, dc_bot, live, sbase, zz_lo, zz_hi, 2 * STG_BLK, m * BPM + 2 + bx, queue, FORCE != 0 DUMP_ARG);
        }
    }

    PHASE_FENCE();
    PROBE_ALL();
    PHASE_STAMP(5);
    // ---- 5. chroma row pass, transpose, column pass ----
    if (!GRAY) {
        const bool odd = (row & 1) != 0;
        f2 cX[4];
        fdct8p(CS, cX, kc);
        f2* dst = reinterpret_cast<f2*>(ldsf + m * C_MCU + (odd ? C_COMP : 0) + (row >> 1) * C_PITCH);
        dst[0] = cX[0]; dst[1] = cX[1]; dst[2] = cX[2]; dst[3] = cX[3];
        wave_sync();

        f2 Fc[4];
        int dc_c;
        {
            f2 col[4];
            lds_column<C_PITCH>(ldsf + m * C_MCU + (cq >> 3) * C_COMP + (cq & 7), col);
            fdct8p(col, Fc, kc);
            dc_c = dc_lookup(Fc[0].x, dcq_c);
        }
        f2 ks[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ks[k] = PS ? pre->ks_c[k] : f2{ lcol[8].ks[2 * k], lcol[8].ks[2 * k + 1] };
        const f2 dd = PS ? pre->dd_c : f2{ lcol[8].delta1[0], lcol[8].delta1[1] };
        quant_block_column(Fc, ks, dd, PS ? pre->th_c : lcol[8].th, j, dc_c, live, sbase, zz_lo, zz_hi, 4 * STG_BLK, m * BPM + 4 + bx, queue, FORCE != 0 DUMP_ARG);
    }
    wave_sync();
    PROBE_ALL();
    PHASE_STAMP(6);

    // ---- 5b. levels 2 and 3 for the queued coefficients (FORCE 1/2: every coefficient of the quad) ----
    const unsigned nq = queue[0];
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
            for (int k = 0; k < 4; ++k) {
                const f2* A = k < 2 ? YL : YR;
                const int x0 = 4 * (k & 1);
                w4[k] = ((uint32_t)(int)pick(A, x0) & 0xFFu) | (((uint32_t)(int)pick(A, x0 + 1) & 0xFFu) << 8) |
                        (((uint32_t)(int)pick(A, x0 + 2) & 0xFFu) << 16) | (((uint32_t)(int)pick(A, x0 + 3) & 0xFFu) << 24);
            }
            const int by = row >> 3, y = row & 7;
            uint32_t* d0 = reinterpret_cast<uint32_t*>(smp + ((m * 4 + by * 2) * 64 + y * 8));
            d0[0] = w4[0]; d0[1] = w4[1];
            d0[16] = w4[2]; d0[17] = w4[3];                      // the right block, 64 bytes further
            if (!GRAY) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    w4[k] = ((uint32_t)(int)pick(CS, 4 * k) & 0xFFu) | (((uint32_t)(int)pick(CS, 4 * k + 1) & 0xFFu) << 8) |
                            (((uint32_t)(int)pick(CS, 4 * k + 2) & 0xFFu) << 16) | (((uint32_t)(int)pick(CS, 4 * k + 3) & 0xFFu) << 24);
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
                // the 8 lanes that hold the block's rows: luma block (by, ebx): rows by*8+y, samples of the left / right
                // block row; chroma: Cb on even-row lanes, Cr on odd-row lanes
                const int by = (eb >> 1) & 1, ebx = eb & 1;
                const int first = comp ? (comp == 2 ? 4 : 0) + em : by * 32 + em;
                const int stride = comp ? 8 : 4;
                const bool part = (m == em) && (comp ? ((row & 1) == (comp == 2)) : ((row >> 3) == by));
                const int yrow = comp ? (row >> 1) : (row & 7);
                float w[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) w[k] = comp ? pick(CS, k) : (ebx ? pick(YR, k) : pick(YL, k));
                int Q, zpos;
                double qinv;
                if (PS) { Q = pst->qt[tbl][nat]; qinv = pst->qinv[tbl][nat]; zpos = pst->zzinv[nat]; }
                else { Q = tab->qt[tbl][nat]; qinv = tab->qinv[tbl][nat]; zpos = c_zzinv[nat]; }
                const int qv = resolve_coef<FORCE, PS>(w, part, yrow, first, stride, ei, ej, Q, qinv, pst);
                if (lane == 0) *reinterpret_cast<int16_t*>(stage + blk * STG_BLK + 2 * zpos) = (int16_t)qv;
                ++done;
            }
            if (lane == 0 && done) atomicAdd(p.fallback_count + (qidx & (COUNTER_SHARDS - 1)), (unsigned long long)done);
            wave_sync();
        }
    }

    PHASE_STAMP(7);
#ifdef JPEZY_TRACE
    tr->t2 = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---- 6. coalesced store of the quad's coefficients (staged by encode_quad_compute behind a wave_sync) ----
// ALL_LANES: every lane stores in every one of the BPM / 2 instructions -- a lane beyond the quad's valid chunks (a last quad with
// fewer than four MCUs) repeats the chunk valid_chunks below its own, same bytes to the same address -- so that the number of store
// instructions per quad is fixed and the compiler can wait for loads issued BEFORE them with an exact vmcnt.
template <bool GRAY, bool ALL_LANES = false>
__device__ __forceinline__ void encode_quad_store(const EncParams& p, uint32_t* lds, int lane, int mcu_y, int quad_x, int frame)
{
    constexpr int BPM = GRAY ? 4 : 6;
    const char* stage = reinterpret_cast<const char*>(lds) + CT_BYTES;
    {
#ifdef JPEZY_ABL_NOSTORE     // TIMING PROBE (wrong results): the coefficients are staged and read back but never stored (only lanes whose
        const int valid_chunks = (lds[0] == 0x12345678u) ? 1 : 0;                   // staged data match a value they never have would store)
#else
        const int valid_chunks = min(4, p.mcu_cols - quad_x * 4) * BPM * 8;         // 16-byte chunks
#endif
        int16_t* gbase = p.coeffs + (size_t)frame * p.coeffs_per_frame +
                         ((size_t)mcu_y * p.mcu_cols + (size_t)quad_x * 4) * (BPM * 64);
        uint4* g4 = reinterpret_cast<uint4*>(gbase);
#pragma unroll
        for (int k = 0; k < BPM * 128 * 4 / 1024; ++k) {
            int c = k * 64 + lane;
            if (ALL_LANES) {                   // valid_chunks >= 32: at most three subtractions' worth, done as two conditional ones + a clamp
                c = c < valid_chunks ? c : c - valid_chunks;
                c = c < valid_chunks ? c : c - valid_chunks;
                c = c < valid_chunks ? c : lane & 31;
            }
            if (ALL_LANES || c < valid_chunks) {
                // streamed out, never re-read by this kernel: a non-temporal store leaves less dirty data in the L2s
                // for the end-of-kernel write-back (measured: 2 us per 4096^2 frame)
                const uint4 v = *reinterpret_cast<const uint4*>(stage + (c >> 3) * STG_BLK + (c & 7) * 16);
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(v4u{v.x, v.y, v.z, v.w}, reinterpret_cast<v4u*>(g4 + c));
            }
        }
    }
}

template <bool GRAY, int FORCE, bool PS>
__device__ __forceinline__ void encode_quad(const EncParams& p, const uint32_t* R, const uint32_t* G, const uint32_t* B, uint32_t* lds,
                                            int lane, int mcu_y, int quad_x, int frame, unsigned qidx, const LaneConsts* pre,
                                            const signed char* dcq_lds, const PsTables* pst QUAD_TRACE_PARAM)
{
    encode_quad_compute<GRAY, FORCE, PS>(p, R, G, B, lds, lane, mcu_y, quad_x, frame, qidx, pre, dcq_lds, pst, NoHook()
#ifdef JPEZY_TRACE
                                         , tr
#endif
    );
    encode_quad_store<GRAY>(p, lds, lane, mcu_y, quad_x, frame);
}

template <bool GRAY, bool ALIGNED, int FORCE, int EWPB>
__global__ __launch_bounds__(64 * EWPB, JPEZY_F32_WAVES) void fdct_quant_f32_kernel(EncParams p)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[EWPB][WAVE_LDS_DWORDS];
    [[maybe_unused]] constexpr int BPM = GRAY ? 4 : 6;
    constexpr bool COOP = ALIGNED && EWPB == 4 && JPEZY_COOP_LOAD;     // (the cooperative load is written for 4 waves: 4 x 4 rows of 256 bytes)
    static_assert(!COOP || 3 * 4096 <= EWPB * WAVE_LDS_DWORDS * 4, "the pixel staging area lies over the waves' slices");

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
    // grid x = groups of EWPB quads: groups_per_row = ceil(quads_per_row / EWPB) per MCU row (a group never straddles rows)
    const int mcu_y = (int)fast_div(blockIdx.x, p.gpr_magic, p.gpr_shift);
    const int gx = (int)blockIdx.x - mcu_y * p.groups_per_row;
    const int quad_x = gx * EWPB + wave;
    const bool has_quad = quad_x < p.quads_per_row;                    // wave-uniform
    if (!COOP && !has_quad) return;
    const unsigned qidx = (unsigned)(mcu_y * p.quads_per_row + quad_x);   // quad index inside the frame
    const int frame = (int)blockIdx.y;
#ifdef JPEZY_TRACE
    const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime();
    QuadTrace tr;
#if JPEZY_TRACE >= 3
    unsigned long long* ph = tr.ph;
    for (int k = 0; k < 8; ++k) ph[k] = 0;
    PHASE_STAMP(0);
#endif
#endif
    uint32_t* lds = lds_all[wave];
    const int row = lane >> 2, m = lane & 3;
    const int mcu_x = min(quad_x * 4 + m, p.mcu_cols - 1);
    const int W = p.W, H = p.H;
    const uint8_t* pr = p.r + (size_t)frame * p.plane_stride;
    const uint8_t* pg = p.g + (size_t)frame * p.plane_stride;
    const uint8_t* pb = p.b + (size_t)frame * p.plane_stride;

    // ---- 1. this lane's 16-pixel row segment of the three planes ----
    uint32_t R[4], G[4], B[4];
    if (COOP) {
        // The workgroup's 256 x 16 pixels of each plane go to LDS as [plane][row][16 pieces of 16 bytes]: wave w fetches rows
        // 4w .. 4w+3, 16 lanes per row, with ONE LDS-DMA instruction per plane (the destination of an LDS-DMA is the wave's
        // base + 16 x lane, so the image is lane-linear: 4 rows of 256 bytes).  Piece c of row r lies at position
        // c ^ 4(r & 3) -- the swizzle is applied to the SOURCE address -- so that the 16-byte reads below, whose 16-lane
        // groups span the rows {0,3,5,6} / {1,2,4,7} of one quad, hit 16 different bank groups.  The area lies over the
        // waves' slices: a second barrier before anybody writes a slice.
        char* stg = reinterpret_cast<char*>(&lds_all[0][0]);
        {
            const int lr = lane >> 4, cp = lane & 15;
            const int y = min(mcu_y * 16 + wave * 4 + lr, H - 1);                     // edge replication, ref :101
            const int piece = min(gx * 16 + (cp ^ (4 * lr)), p.mcu_cols - 1);         // W % 16 == 0 here: a piece is an MCU column
            const unsigned off = (unsigned)y * (unsigned)W + (unsigned)piece * 16u;   // W, H <= 65535 (launcher): fits 32 bits
            typedef __attribute__((address_space(1))) const void* gptr;
            typedef __attribute__((address_space(3))) void* lptr;
            __builtin_amdgcn_global_load_lds((gptr)(pr + off), (lptr)(stg + wave * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr)(pg + off), (lptr)(stg + 4096 + wave * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr)(pb + off), (lptr)(stg + 8192 + wave * 1024), 16, 0, 0);
        }
        __syncthreads();                                   // waits for this wave's DMA (vmcnt) and for the other three
        {
            const char* src = stg + row * 256 + (((wave * 4 + m) ^ (4 * (row & 3))) * 16);
            const uint4 vr = *reinterpret_cast<const uint4*>(src);
            const uint4 vg = *reinterpret_cast<const uint4*>(src + 4096);
            const uint4 vb = *reinterpret_cast<const uint4*>(src + 8192);
            R[0] = vr.x; R[1] = vr.y; R[2] = vr.z; R[3] = vr.w;
            G[0] = vg.x; G[1] = vg.y; G[2] = vg.z; G[3] = vg.w;
            B[0] = vb.x; B[1] = vb.y; B[2] = vb.z; B[3] = vb.w;
        }
        __syncthreads();                                   // staging area consumed: the slices are private from here on
        if (!has_quad) return;
    } else {
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
    PHASE_STAMP(1);
#ifdef JPEZY_ABL_LIGHT_TAIL
    // TIMING PROBE (wrong results; jpezy_experiment.h): the workgroups of the launch's last JPEZY_ABL_LIGHT_TAIL groups do their loads and their
    // stores and nothing in between -- the shortest waves a tail of any finer-grained design (half quads) could have.
    // What the launch gains from that is the upper bound of what such a design can gain.
    if (blockIdx.x + (unsigned)JPEZY_ABL_LIGHT_TAIL >= gridDim.x) {
        char* st0 = reinterpret_cast<char*>(lds) + CT_BYTES;
        uint32_t* w = reinterpret_cast<uint32_t*>(st0) + lane * 12;
#pragma unroll
        for (int k = 0; k < 4; ++k) { w[k] = R[k]; w[4 + k] = G[k]; w[8 + k] = B[k]; }
        wave_sync();
        if (has_quad) {
            const int valid_chunks = min(4, p.mcu_cols - quad_x * 4) * BPM * 8;
            int16_t* gbase = p.coeffs + (size_t)frame * p.coeffs_per_frame + ((size_t)mcu_y * p.mcu_cols + (size_t)quad_x * 4) * (BPM * 64);
            uint4* g4 = reinterpret_cast<uint4*>(gbase);
#pragma unroll
            for (int k = 0; k < BPM * 128 * 4 / 1024; ++k) {
                const int c = k * 64 + lane;
                if (c < valid_chunks) {
                    const uint4 v = *reinterpret_cast<const uint4*>(st0 + (c >> 3) * STG_BLK + (c & 7) * 16);
                    typedef unsigned v4u __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(v4u{v.x, v.y, v.z, v.w}, reinterpret_cast<v4u*>(g4 + c));
                }
            }
        }
        return;
    }
#endif
    encode_quad<GRAY, FORCE, false>(p, R, G, B, lds, lane, mcu_y, quad_x, frame, qidx, nullptr, nullptr, nullptr QUAD_TRACE_ARG);
#ifdef JPEZY_TRACE
    if (frame == 0 && qidx < 65536u) {
#if JPEZY_TRACE > 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        const unsigned long long tr_t3 = __builtin_amdgcn_s_memrealtime();
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        if (lane == 0) {
            p.trace[qidx * 4 + 0] = tr_t0;
            p.trace[qidx * 4 + 1] = ((tr_t1 - tr_t0) << 32) | (tr.t2 - tr_t0);
            p.trace[qidx * 4 + 2] = tr_t3 - tr_t0;
            p.trace[qidx * 4 + 3] = ((unsigned long long)xcc << 32) | hw;
#if JPEZY_TRACE >= 3
            unsigned long long t_end;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end) : : "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k) p.trace[4 * 65536 + qidx * 9 + k] = ph[k];
            p.trace[4 * 65536 + qidx * 9 + 8] = t_end;
#endif
        }
    }
#endif
}


// ======================================================================================================================
// Persistent form (encode variant 2): the loads are decoupled from the computing waves.
//
// A workgroup = PS_NC compute waves + PS_NL loader waves, two workgroups per CU, grid = the resident workgroups; workgroup w
// owns the groups (256 x 16 pixels = four quads) w, w + gridDim.x, ... of the launch (all frames).  A loader wave does nothing
// but LDS-DMA: it waits until its ring slot has been released, fetches the next group's 16 rows of 256 bytes of the three
// planes (12 wave instructions, no VGPR traffic, same swizzled image as the cooperative load above), waits for them to land
// and publishes the slot.  A compute wave draws the workgroup's next quad from a counter in LDS, waits for that quad's slot,
// takes its three 16-byte row segments out of it, releases the slot and runs steps 2-6 (encode_quad) unchanged -- so a
// wave never waits for HBM with its registers and its LDS slice idle, the stores of quad n drain under the arithmetic of quad
// n + 1 instead of holding a finished wave's slot, and the waves of a workgroup never meet at a barrier (a wave that
// resolves guard-band hits delays nobody).  Hand-off words (LDS, monotonic): full[slot] = fills completed, freec[slot] = quads
// taken out.  Every spin is bounded: a wave that waits PS_SPIN_CAP polls raises `abort`, which every wave sees -- the launch
// always drains (the results are then wrong, and the exact-path counter's last shard says so).
#ifndef JPEZY_PS_NC
#define JPEZY_PS_NC 14
#endif
#ifndef JPEZY_PS_NSLOT
#define JPEZY_PS_NSLOT 2
#endif
#ifndef JPEZY_PS_WG_PER_CU
#define JPEZY_PS_WG_PER_CU 1
#endif
#ifndef JPEZY_PS_STAGGER
#define JPEZY_PS_STAGGER 0     // s_sleep units (64 cycles) by which compute wave c delays its first draw, times c
#endif
#ifndef JPEZY_PS_NL
#define JPEZY_PS_NL 2          // loader waves
#endif
#ifndef JPEZY_PS_LAG
#define JPEZY_PS_LAG 0         // groups a loader wave keeps in flight besides the one it has just issued
#endif
constexpr int PS_NC = JPEZY_PS_NC, PS_NSLOT = JPEZY_PS_NSLOT, PS_NL = JPEZY_PS_NL, PS_LAG = JPEZY_PS_LAG;
static_assert(PS_NL * (PS_LAG + 1) <= PS_NSLOT && 12 * PS_LAG <= 63, "every group in flight needs a ring slot of its own; vmcnt is a 6-bit count");
#ifndef JPEZY_PS_PAD_WAVES
#define JPEZY_PS_PAD_WAVES 0   // waves that leave at once: they round the workgroup up to a multiple of four waves, so that two workgroups
#endif                         // always spread evenly over a CU's four SIMDs (the register budget is per SIMD)
constexpr int PS_WAVES = PS_NC + PS_NL + JPEZY_PS_PAD_WAVES;
constexpr int PS_SLOT_BYTES = 3 * 4096;
constexpr int PS_DCQ_BYTES = JPEZY_PS_DCQ_LDS ? (2 * 16385 + 15) / 16 * 16 : 16;      // LDS copy of DeviceTables::dcq (the tail of the last 16 bytes is never indexed)
constexpr unsigned PS_SPIN_CAP = 1u << 20;     // x ~300 cycles per poll: > 100 ms
struct PsControl {
    unsigned full[4];
    unsigned freec[4];
    unsigned next;
    unsigned abort;
    unsigned pad[6];
};
static_assert(PS_NSLOT <= 4 && PS_WAVES <= 16, "ring of at most four slots, workgroup of at most 1024 threads");
static_assert(JPEZY_PS_WG_PER_CU * (PS_NSLOT * PS_SLOT_BYTES + PS_NC * WAVE_LDS_DWORDS * 4 + (int)sizeof(PsControl) + PS_DCQ_BYTES + (int)sizeof(PsTables)) <= 160 * 1024, "LDS per CU");
static_assert(offsetof(DeviceTables, dcq) % 16 == 0 && offsetof(DeviceTables, dcq) + PS_DCQ_BYTES <= sizeof(DeviceTables), "the 16-byte copy of dcq stays inside the tables");

__device__ __forceinline__ unsigned lds_peek(const unsigned* w)     // one ds_read_b32, never cached in a register
{
    return (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// waits until *w >= want (or the workgroup aborts); false on abort
__device__ __forceinline__ bool lds_wait_ge(const unsigned* w, unsigned want, PsControl* ctl)
{
    unsigned spins = 0;
    while (lds_peek(w) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins >= PS_SPIN_CAP || lds_peek(&ctl->abort)) {
            __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return false;
        }
    }
    asm volatile("" ::: "memory");     // nothing that follows is read before the word has been seen (LDS is in order per wave)
    return true;
}

template <bool GRAY, int FORCE>
__global__ __launch_bounds__(64 * PS_WAVES, (JPEZY_PS_WG_PER_CU * PS_WAVES + 3) / 4) void fdct_quant_f32_ps_kernel(EncParams p)
{
    __shared__ __attribute__((aligned(16))) char ring[PS_NSLOT][PS_SLOT_BYTES];
    __shared__ __attribute__((aligned(16))) uint32_t slices[PS_NC][WAVE_LDS_DWORDS];
    __shared__ PsControl ctl;
    // the loop of a compute wave holds no vector-memory LOAD: a wait for one (vmcnt counts in issue order) would also wait for
    // the previous quad's coefficient stores.  So the workgroup keeps its own copy of the two quantised-DC tables and of the
    // cosine table in LDS, and every compute lane its quantiser records in registers.
    __shared__ __attribute__((aligned(16))) signed char dcq_s[PS_DCQ_BYTES];
    __shared__ PsTables pst;
    const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned nwg = gridDim.x, w = blockIdx.x;
    const unsigned n = (p.ps_total_groups - w + nwg - 1) / nwg;        // groups of this workgroup (the launcher keeps nwg <= total)
    if (threadIdx.x < sizeof(PsControl) / 4) reinterpret_cast<unsigned*>(&ctl)[threadIdx.x] = 0;
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.dcq_luma);           // &tab->dcq[0][0]: 16-byte aligned (offset 1024 of DeviceTables), [2][16385] contiguous
        uint4* dst = reinterpret_cast<uint4*>(dcq_s);
        if (JPEZY_PS_DCQ_LDS)
            for (unsigned k = threadIdx.x; k < PS_DCQ_BYTES / 16; k += 64 * PS_WAVES) dst[k] = src[k];
        if (threadIdx.x < 64) {
            pst.cos[threadIdx.x] = c_cos[threadIdx.x];
            pst.zzinv[threadIdx.x] = c_zzinv[threadIdx.x];
        }
        if (threadIdx.x < 128) {
            (&pst.qinv[0][0])[threadIdx.x] = (&p.tab->qinv[0][0])[threadIdx.x];
            (&pst.qt[0][0])[threadIdx.x] = (&p.tab->qt[0][0])[threadIdx.x];
        }
    }
    __syncthreads();

    if (wave >= PS_NC + PS_NL) return;
    if (wave >= PS_NC) {
        // ---- loader wave l: groups l, l + PS_NL, ... of the workgroup; group i goes to ring slot i % PS_NSLOT.  It keeps up to
        //      PS_LAG + 1 groups in flight: after issuing group i it waits (vmcnt counts in issue order) for the group it issued
        //      PS_LAG rounds earlier and publishes that one. ----
        const int l = wave - PS_NC, lane = lane0;
        const int lr = lane >> 4, cp = lane & 15;
        unsigned issued = 0, i = (unsigned)l;
        for (; i < n; i += PS_NL, ++issued) {
            const unsigned slot = i % PS_NSLOT, round = i / PS_NSLOT;
            char* dst = ring[slot];
            const unsigned g = w + i * nwg;
            const unsigned frame = fast_div(g, p.gpf_magic, p.gpf_shift);
            const unsigned rem = g - frame * p.ps_groups_per_frame;
            const int mcu_y = (int)fast_div(rem, p.gpr_magic, p.gpr_shift);
            const int gx = (int)rem - mcu_y * p.groups_per_row;
            if (round && !lds_wait_ge(&ctl.freec[slot], 4u * round, &ctl)) return;
            const uint8_t* pr = p.r + (size_t)frame * p.plane_stride;
            const uint8_t* pg = p.g + (size_t)frame * p.plane_stride;
            const uint8_t* pb = p.b + (size_t)frame * p.plane_stride;
            const int piece = min(gx * 16 + (cp ^ (4 * lr)), p.mcu_cols - 1);         // W % 16 == 0 here: a piece is an MCU column
            typedef __attribute__((address_space(1))) const void* gptr;
            typedef __attribute__((address_space(3))) void* lptr;
#ifdef JPEZY_ABL_PS_NOLOAD   // TIMING PROBE (wrong results; jpezy_experiment.h): the ring is published without having been filled
            if (piece < 0)
#endif
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int y = min(mcu_y * 16 + rg * 4 + lr, p.H - 1);                 // edge replication, ref :101
                const unsigned off = (unsigned)y * (unsigned)p.W + (unsigned)piece * 16u;
                __builtin_amdgcn_global_load_lds((gptr)(pr + off), (lptr)(dst + rg * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr)(pg + off), (lptr)(dst + 4096 + rg * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr)(pb + off), (lptr)(dst + 8192 + rg * 1024), 16, 0, 0);
            }
            if (issued >= (unsigned)PS_LAG) {
                asm volatile("s_waitcnt vmcnt(%0)" : : "n"(12 * PS_LAG) : "memory");   // group i - PS_LAG * PS_NL has landed in LDS
                const unsigned j = i - PS_LAG * PS_NL;
                if (lane == 0) __hip_atomic_store(&ctl.full[j % PS_NSLOT], j / PS_NSLOT + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (unsigned k = issued < (unsigned)PS_LAG ? issued : (unsigned)PS_LAG; k >= 1; --k) {
            const unsigned j = i - k * PS_NL;
            if (lane == 0) __hip_atomic_store(&ctl.full[j % PS_NSLOT], j / PS_NSLOT + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }

    // ---- compute wave ----
    uint32_t* lds = slices[wave];
    LaneConsts lc = load_lane_consts(p.tab, lane0);
    // the records are IN the registers before the loop starts (the compiler would otherwise place its wait for them at their
    // first use, inside the loop, where it would wait for the previous quad's stores on every round)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(lc.ks_l[k]), "+v"(lc.ks_c[k]));
    asm volatile("" : "+v"(lc.dd_l), "+v"(lc.dd_c), "+v"(lc.th_l), "+v"(lc.th_c), "+v"(lc.zz_lo), "+v"(lc.zz_hi));
    if (JPEZY_PS_STAGGER)
        for (int k = 0; k < wave; ++k) __builtin_amdgcn_s_sleep(JPEZY_PS_STAGGER);
    // A wave's round: [pixels of quad q requested] -> wait for them, give the slot back -> steps 2-5b -> draw the NEXT quad, wait
    // for its slot and request its pixels -> step 6 (stores of quad q).  The next quad's three LDS reads, and the two LDS round
    // trips in front of them (draw, slot word), thus run under this quad's stores instead of in front of the next quad's arithmetic.
    uint32_t R[4], G[4], B[4];
    auto draw = [&]() -> unsigned {
        unsigned q = 0;
        if (lane0 == 0) q = __hip_atomic_fetch_add(&ctl.next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (unsigned)__builtin_amdgcn_readfirstlane((int)q);
    };
    auto request_pixels = [&](unsigned q, int lane) {       // the slot of quad q has been published
        const unsigned slot = (q >> 2) % PS_NSLOT, sub = q & 3u;
        const int row = lane >> 2, m = lane & 3;
        const char* src = ring[slot] + row * 256 + (((sub * 4 + m) ^ (4 * (row & 3))) * 16);
        const uint4 vr = *reinterpret_cast<const uint4*>(src);
        const uint4 vg = *reinterpret_cast<const uint4*>(src + 4096);
        const uint4 vb = *reinterpret_cast<const uint4*>(src + 8192);
        R[0] = vr.x; R[1] = vr.y; R[2] = vr.z; R[3] = vr.w;
        G[0] = vg.x; G[1] = vg.y; G[2] = vg.z; G[3] = vg.w;
        B[0] = vb.x; B[1] = vb.y; B[2] = vb.z; B[3] = vb.w;
    };
    unsigned q = draw();
    bool have = q < 4u * n;
    if (have) {
        if (!lds_wait_ge(&ctl.full[(q >> 2) % PS_NSLOT], (q >> 2) / PS_NSLOT + 1, &ctl)) have = false;
        else request_pixels(q, lane0);
    }
    while (have) {
        // everything derived from the lane index (LDS addresses, table pointers, per-lane constants) is formed anew for every
        // quad, as the one-quad kernel does: hoisted out of the loop it would hold ~20 registers for the whole launch
        int lane = lane0;
        asm volatile("" : "+v"(lane));
#ifdef JPEZY_TRACE
        const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime();
        QuadTrace tr;
#if JPEZY_TRACE >= 3
        unsigned long long* ph = tr.ph;
        PHASE_STAMP(0);
#endif
#endif
        const unsigned i = q >> 2, sub = q & 3u;
        // the segments are in registers (not merely requested) before the slot is given back; only LDS is waited for, the
        // previous quad's coefficient stores stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(G[0]), "+v"(G[1]), "+v"(G[2]), "+v"(G[3]),
                                              "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]) : : "memory");
        if (lane == 0) __hip_atomic_fetch_add(&ctl.freec[i % PS_NSLOT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned g = w + i * nwg;
        const unsigned frame = fast_div(g, p.gpf_magic, p.gpf_shift);
        const unsigned rem = g - frame * p.ps_groups_per_frame;
        const int mcu_y = (int)fast_div(rem, p.gpr_magic, p.gpr_shift);
        const int quad_x = ((int)rem - mcu_y * p.groups_per_row) * 4 + (int)sub;
        const bool has_quad = quad_x < p.quads_per_row;                // false: a clamped group's surplus quad
        const unsigned qidx = (unsigned)(mcu_y * p.quads_per_row + quad_x);
#ifdef JPEZY_TRACE
        const unsigned long long tr_t1 = __builtin_amdgcn_s_memrealtime();
#endif
        PHASE_STAMP(1);
        if (has_quad)
            encode_quad_compute<GRAY, FORCE, true>(p, R, G, B, lds, lane, mcu_y, quad_x, (int)frame, qidx, &lc, dcq_s, &pst, NoHook() QUAD_TRACE_ARG);
        const unsigned q2 = draw();
        bool have2 = q2 < 4u * n;
        if (have2) {
            if (!lds_wait_ge(&ctl.full[(q2 >> 2) % PS_NSLOT], (q2 >> 2) / PS_NSLOT + 1, &ctl)) have2 = false;
            else request_pixels(q2, lane);
        }
        if (has_quad) encode_quad_store<GRAY>(p, lds, lane, mcu_y, quad_x, (int)frame);
#ifdef JPEZY_TRACE
        if (has_quad && frame == 0 && qidx < 65536u) {      // (the stores are NOT waited for here: the next quad's arithmetic covers them)
            const unsigned long long tr_t3 = __builtin_amdgcn_s_memrealtime();
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            if (lane == 0) {
                p.trace[qidx * 4 + 0] = tr_t0;
                p.trace[qidx * 4 + 1] = ((tr_t1 - tr_t0) << 32) | (tr.t2 - tr_t0);
                p.trace[qidx * 4 + 2] = tr_t3 - tr_t0;
                p.trace[qidx * 4 + 3] = ((unsigned long long)xcc << 32) | hw;
#if JPEZY_TRACE >= 3
                unsigned long long t_end;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end) : : "memory");
#pragma unroll
                for (int k = 0; k < 8; ++k) p.trace[4 * 65536 + qidx * 9 + k] = ph[k];
                p.trace[4 * 65536 + qidx * 9 + 8] = t_end;
#endif
            }
        }
#endif
        q = q2; have = have2;
    }
    if (lds_peek(&ctl.abort) && lane0 == 0 && wave == 0)
        atomicAdd(p.fallback_count + (COUNTER_SHARDS - 1), 1ull << 40);
}


// ======================================================================================================================
// Persistent form, second shape (encode variant 3): no loader waves, no ring.  One workgroup of 16 waves per CU, every wave a
// compute wave with a fixed share of the quads (quad index = wave's global index + k x waves of the launch: the 16 waves of a
// workgroup work on 16 horizontally adjacent quads, 1 KB of every pixel row).  A wave requests the NEXT quad's three 16-byte
// row segments straight into the registers of the current one as soon as those are dead (after the chroma estimate, step 2b),
// so the HBM round trip runs under steps 3-6; its only wait for them stands behind the current quad's stores, as vmcnt(number of
// store instructions): loads and stores complete in issue order, so that wait covers the loads and leaves the stores in flight.
// For that count to be exact the stores are unconditional (encode_quad_store<ALL_LANES>) and the loop holds no other
// vector-memory instruction: quantiser records in registers, DC / cosine / quantiser tables in LDS as in variant 2.
#ifndef JPEZY_PS2_WAVES
#define JPEZY_PS2_WAVES 16
#endif
constexpr int PS2_WAVES = JPEZY_PS2_WAVES;      // waves per workgroup = quads per run (quad_of below)
static_assert(PS2_WAVES * WAVE_LDS_DWORDS * 4 + (2 * 16385 + 15) / 16 * 16 + (int)sizeof(PsTables) <= 160 * 1024, "LDS per CU");

template <bool GRAY, int FORCE>
__global__ __launch_bounds__(64 * PS2_WAVES) void fdct_quant_f32_ps2_kernel(EncParams p)
{
    constexpr int DCQ_BYTES = (2 * 16385 + 15) / 16 * 16;
    __shared__ __attribute__((aligned(16))) uint32_t slices[PS2_WAVES][WAVE_LDS_DWORDS];
    __shared__ __attribute__((aligned(16))) signed char dcq_s[DCQ_BYTES];
    __shared__ PsTables pst;
    const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // The workgroup owns the runs of 16 consecutive quads number w, w + gridDim.x, ... (w = blockIdx.x) and its waves DRAW quads from
    // them through a counter in LDS: draw j is quad ((j / 16) * gridDim.x + w) * 16 + j % 16.  A fixed share per wave ends badly:
    // the four waves of a SIMD are served oldest first, so the older ones run through their quads in 3 us each while the youngest
    // takes up to 18 us for its first -- measured with four quads per wave: the waves finished after 19 us on average, the last one
    // of a workgroup after 26 us, the launch after 30 (profiles/r05_ps2_timeline_static.txt).
    __shared__ unsigned next_draw;
    const unsigned nwg = gridDim.x, w = blockIdx.x, total = p.ps_total_quads;
    auto quad_of = [&](unsigned j) -> unsigned { return ((j / PS2_WAVES) * nwg + w) * PS2_WAVES + (j % PS2_WAVES); };   // increasing in j
    unsigned q = quad_of((unsigned)wave);
    if (threadIdx.x == 0) next_draw = PS2_WAVES;

    // quad q -> (frame, mcu_y, quad_x); scalar
    auto locate = [&](unsigned qq, unsigned& frame, int& mcu_y, int& quad_x) {
        frame = fast_div(qq, p.qpf_magic, p.qpf_shift);
        const unsigned rem = qq - frame * p.ps_quads_per_frame;
        mcu_y = (int)fast_div(rem, p.qpr_magic, p.qpr_shift);
        quad_x = (int)rem - mcu_y * p.quads_per_row;
    };
    uint32_t R[4], G[4], B[4];
    auto request_pixels = [&](unsigned qq, int lane) {
        unsigned frame; int mcu_y, quad_x;
        locate(qq, frame, mcu_y, quad_x);
        const int row = lane >> 2, m = lane & 3;
        const int y = min(mcu_y * 16 + row, p.H - 1);                              // edge replication, ref :101
        const int mcu_x = min(quad_x * 4 + m, p.mcu_cols - 1);
        const unsigned off = (unsigned)y * (unsigned)p.W + (unsigned)mcu_x * 16u;  // W, H <= 65535 (launcher): fits 32 bits
        const size_t fo = (size_t)frame * p.plane_stride;
        const uint4 vr = *reinterpret_cast<const uint4*>(p.r + fo + off);
        const uint4 vg = *reinterpret_cast<const uint4*>(p.g + fo + off);
        const uint4 vb = *reinterpret_cast<const uint4*>(p.b + fo + off);
        R[0] = vr.x; R[1] = vr.y; R[2] = vr.z; R[3] = vr.w;
        G[0] = vg.x; G[1] = vg.y; G[2] = vg.z; G[3] = vg.w;
        B[0] = vb.x; B[1] = vb.y; B[2] = vb.z; B[3] = vb.w;
    };
    if (q < total) request_pixels(q, lane0);        // in flight while the tables are copied
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.dcq_luma);           // &tab->dcq[0][0]: 16-byte aligned, [2][16385] contiguous
        uint4* dst = reinterpret_cast<uint4*>(dcq_s);
        for (unsigned k = threadIdx.x; k < DCQ_BYTES / 16; k += 64 * PS2_WAVES) dst[k] = src[k];
        if (threadIdx.x < 64) {
            pst.cos[threadIdx.x] = c_cos[threadIdx.x];
            pst.zzinv[threadIdx.x] = c_zzinv[threadIdx.x];
        }
        if (threadIdx.x < 128) {
            (&pst.qinv[0][0])[threadIdx.x] = (&p.tab->qinv[0][0])[threadIdx.x];
            (&pst.qt[0][0])[threadIdx.x] = (&p.tab->qt[0][0])[threadIdx.x];
        }
    }
    uint32_t* lds = slices[wave];
    LaneConsts lc = load_lane_consts(p.tab, lane0);
    __syncthreads();                                // (waits for every load above: vmcnt(0) in front of the barrier)
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(lc.ks_l[k]), "+v"(lc.ks_c[k]));
    asm volatile("" : "+v"(lc.dd_l), "+v"(lc.dd_c), "+v"(lc.th_l), "+v"(lc.th_c), "+v"(lc.zz_lo), "+v"(lc.zz_hi));

#ifndef JPEZY_PS2_STAGGER
#define JPEZY_PS2_STAGGER 0       // s_sleep units (64 cycles)
#endif
#ifndef JPEZY_PS2_STAGGER_BY
#define JPEZY_PS2_STAGGER_BY(w) ((w) >> 2)
#endif
    // waves that start together run the phases of a quad in lockstep -- all in the conversions, then all in the LDS transposes --
    // and use one unit of the CU at a time; a start offset spreads them over the phases
    if (JPEZY_PS2_STAGGER)
        for (int k = 0; k < JPEZY_PS2_STAGGER_BY(wave); ++k) __builtin_amdgcn_s_sleep(JPEZY_PS2_STAGGER);
    while (q < total) {
        int lane = lane0;                           // lane-derived values are formed anew for every quad (see variant 2)
        asm volatile("" : "+v"(lane));
        unsigned frame; int mcu_y, quad_x;
        locate(q, frame, mcu_y, quad_x);
        const unsigned qidx = (unsigned)(mcu_y * p.quads_per_row + quad_x);
        unsigned qn = 0;
#ifdef JPEZY_TRACE
        const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime();
        QuadTrace tr;
#if JPEZY_TRACE >= 3
        unsigned long long* ph = tr.ph;
        PHASE_STAMP(0);
#endif
        const unsigned long long tr_t1 = tr_t0;
#endif
        PHASE_STAMP(1);
        // steps 2-5b; between 2b and 3 the next quad's pixels are requested into R, G, B
        encode_quad_compute<GRAY, FORCE, true>(p, R, G, B, lds, lane, mcu_y, quad_x, (int)frame, qidx, &lc, dcq_s, &pst,
                                                // (unconditional: behind a branch the loaded values would have to be merged with the old ones
                                                // at once, and the wait for them would stand here; a wave's last round re-reads its last quad)
                                                [&]() {
                                                    unsigned j = 0;
                                                    if (lane == 0) j = __hip_atomic_fetch_add(&next_draw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                                    qn = quad_of((unsigned)__builtin_amdgcn_readfirstlane((int)j));
                                                    request_pixels(qn < total ? qn : q, lane);
                                                } QUAD_TRACE_ARG);
        encode_quad_store<GRAY, true>(p, lds, lane, mcu_y, quad_x, (int)frame);
        // the next quad's pixels: requested before the stores above, so this wait (placed by the compiler: vmcnt = the store
        // instructions issued since) does not include the stores
        asm volatile("" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(G[0]), "+v"(G[1]), "+v"(G[2]), "+v"(G[3]),
                          "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]));
#ifdef JPEZY_TRACE
        if (frame == 0 && qidx < 65536u) {      // "stores issued" here = stores + the wait for the next quad's pixels
            const unsigned long long tr_t3 = __builtin_amdgcn_s_memrealtime();
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            if (lane == 0) {
                p.trace[qidx * 4 + 0] = tr_t0;
                p.trace[qidx * 4 + 1] = ((tr_t1 - tr_t0) << 32) | (tr.t2 - tr_t0);
                p.trace[qidx * 4 + 2] = tr_t3 - tr_t0;
                p.trace[qidx * 4 + 3] = ((unsigned long long)xcc << 32) | hw;
#if JPEZY_TRACE >= 3
                unsigned long long t_end;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end) : : "memory");
#pragma unroll
                for (int k = 0; k < 8; ++k) p.trace[4 * 65536 + qidx * 9 + k] = ph[k];
                p.trace[4 * 65536 + qidx * 9 + 8] = t_end;
#endif
            }
        }
#endif
        q = qn;
    }
}

}  // namespace f32

template <bool GRAY, bool ALIGNED, int EW>
static void enc_f32_launch2(const EncParams& p, int force, dim3 grid, hipStream_t s)
{
    if (force == 1)
        hipLaunchKernelGGL((f32::fdct_quant_f32_kernel<GRAY, ALIGNED, 1, EW>), grid, dim3(64 * EW), 0, s, p);
    else if (force == 2)
        hipLaunchKernelGGL((f32::fdct_quant_f32_kernel<GRAY, ALIGNED, 2, EW>), grid, dim3(64 * EW), 0, s, p);
    else if (force == 3)
        hipLaunchKernelGGL((f32::fdct_quant_f32_kernel<GRAY, ALIGNED, 3, EW>), grid, dim3(64 * EW), 0, s, p);
    else
        hipLaunchKernelGGL((f32::fdct_quant_f32_kernel<GRAY, ALIGNED, 0, EW>), grid, dim3(64 * EW), 0, s, p);
}

hipError_t launch_fdct_quant_f32(const EncParams& p0, bool gray, int force, hipStream_t stream)
{
    EncParams p = p0;
    const bool al = (p.W % 16 == 0) && (p.plane_stride % 16 == 0) &&
                    (((uintptr_t)p.r | (uintptr_t)p.g | (uintptr_t)p.b) % 16 == 0);
    // four quads per workgroup with the cooperative load where the rows divide evenly, two with direct loads elsewhere
    const int ew = (al && JPEZY_COOP_LOAD && p.quads_per_row % 4 == 0) ? 4 : 2;
    p.groups_per_row = (p.quads_per_row + ew - 1) / ew;
    const long groups = (long)p.mcu_rows * p.groups_per_row;
    if (groups <= 0 || p.n_frames <= 0) return hipSuccess;
    if (p.n_frames > 65535) return hipErrorInvalidValue;               // grid.y limit; callers chunk larger batches
    fast_div_setup((unsigned)p.groups_per_row, &p.gpr_magic, &p.gpr_shift);
    const dim3 grid((unsigned)groups, (unsigned)p.n_frames);
    if (ew == 4) {
        if (gray) enc_f32_launch2<true, true, 4>(p, force, grid, stream); else enc_f32_launch2<false, true, 4>(p, force, grid, stream);
    } else if (gray) {
        if (al) enc_f32_launch2<true, true, 2>(p, force, grid, stream); else enc_f32_launch2<true, false, 2>(p, force, grid, stream);
    } else {
        if (al) enc_f32_launch2<false, true, 2>(p, force, grid, stream); else enc_f32_launch2<false, false, 2>(p, force, grid, stream);
    }
    return hipGetLastError();
}


template <bool GRAY>
static void enc_f32_ps_launch2(const EncParams& p, int force, unsigned nwg, hipStream_t s)
{
    const dim3 grid(nwg), block(64 * f32::PS_WAVES);
    if (force == 1)
        hipLaunchKernelGGL((f32::fdct_quant_f32_ps_kernel<GRAY, 1>), grid, block, 0, s, p);
    else if (force == 2)
        hipLaunchKernelGGL((f32::fdct_quant_f32_ps_kernel<GRAY, 2>), grid, block, 0, s, p);
    else if (force == 3)
        hipLaunchKernelGGL((f32::fdct_quant_f32_ps_kernel<GRAY, 3>), grid, block, 0, s, p);
    else
        hipLaunchKernelGGL((f32::fdct_quant_f32_ps_kernel<GRAY, 0>), grid, block, 0, s, p);
}

bool fdct_quant_f32_ps_applies(const EncParams& p)
{
    const bool al = (p.W % 16 == 0) && (p.plane_stride % 16 == 0) && (((uintptr_t)p.r | (uintptr_t)p.g | (uintptr_t)p.b) % 16 == 0);
    const unsigned long long total = (unsigned long long)p.mcu_rows * (unsigned long long)(p.quads_per_row / 4) * (unsigned long long)p.n_frames;
    return al && p.quads_per_row % 4 == 0 && total > 0 && total < (1ull << 31);
}

hipError_t launch_fdct_quant_f32_ps(const EncParams& p0, bool gray, int force, int n_cus, hipStream_t stream)
{
    if (!fdct_quant_f32_ps_applies(p0)) return launch_fdct_quant_f32(p0, gray, force, stream);
    EncParams p = p0;
    p.groups_per_row = p.quads_per_row / 4;
    p.ps_groups_per_frame = (unsigned)p.mcu_rows * (unsigned)p.groups_per_row;
    p.ps_total_groups = p.ps_groups_per_frame * (unsigned)p.n_frames;
    fast_div_setup((unsigned)p.groups_per_row, &p.gpr_magic, &p.gpr_shift);
    fast_div_setup(p.ps_groups_per_frame, &p.gpf_magic, &p.gpf_shift);
    const unsigned resident = (unsigned)(n_cus > 0 ? n_cus : 256) * JPEZY_PS_WG_PER_CU;
    const unsigned nwg = p.ps_total_groups < resident ? p.ps_total_groups : resident;
    if (gray) enc_f32_ps_launch2<true>(p, force, nwg, stream); else enc_f32_ps_launch2<false>(p, force, nwg, stream);
    return hipGetLastError();
}

template <bool GRAY>
static void enc_f32_ps2_launch2(const EncParams& p, int force, unsigned nwg, hipStream_t s)
{
    const dim3 grid(nwg), block(64 * f32::PS2_WAVES);
    if (force == 1)
        hipLaunchKernelGGL((f32::fdct_quant_f32_ps2_kernel<GRAY, 1>), grid, block, 0, s, p);
    else if (force == 2)
        hipLaunchKernelGGL((f32::fdct_quant_f32_ps2_kernel<GRAY, 2>), grid, block, 0, s, p);
    else if (force == 3)
        hipLaunchKernelGGL((f32::fdct_quant_f32_ps2_kernel<GRAY, 3>), grid, block, 0, s, p);
    else
        hipLaunchKernelGGL((f32::fdct_quant_f32_ps2_kernel<GRAY, 0>), grid, block, 0, s, p);
}

hipError_t launch_fdct_quant_f32_ps2(const EncParams& p0, bool gray, int force, int n_cus, hipStream_t stream)
{
    const bool al = (p0.W % 16 == 0) && (p0.plane_stride % 16 == 0) && (((uintptr_t)p0.r | (uintptr_t)p0.g | (uintptr_t)p0.b) % 16 == 0);
    const unsigned long long total = (unsigned long long)p0.mcu_rows * (unsigned long long)p0.quads_per_row * (unsigned long long)p0.n_frames;
    if (!al || total == 0 || total >= (1ull << 31)) return launch_fdct_quant_f32(p0, gray, force, stream);
    EncParams p = p0;
    p.ps_quads_per_frame = (unsigned)p.mcu_rows * (unsigned)p.quads_per_row;
    p.ps_total_quads = (unsigned)total;
    fast_div_setup(p.ps_quads_per_frame, &p.qpf_magic, &p.qpf_shift);
    const unsigned cus = (unsigned)(n_cus > 0 ? n_cus : 256);
    const unsigned need = (p.ps_total_quads + f32::PS2_WAVES - 1) / f32::PS2_WAVES;
    const unsigned nwg = need < cus ? need : cus;
    if (gray) enc_f32_ps2_launch2<true>(p, force, nwg, stream); else enc_f32_ps2_launch2<false>(p, force, nwg, stream);
    return hipGetLastError();
}

}  // namespace jpezy_dev
