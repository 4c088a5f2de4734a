// jpezy_f32_quad.h -- device code of ONE quad (64 x 16 pixels, 24 blocks) of the f32 encode path, shared by the one-quad-per-wave kernel
// (jpezy_kernels_f32.hip, encode variant 1) and the persistent kernels (jpezy_kernels_f32_ps.hip, variants 2 and 3): the packed 8-point
// transform, the colour estimates with their guard tests, the quantiser, levels 2 and 3, and encode_quad_compute / encode_quad_store.
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
#pragma once
#include "jpezy_device.h"
#include "../../include/jpezy_constants.h"

#ifdef JPEZY_WITH_LAB
#include "jpezy_lab.h"         // the laboratory: timing probes (PROBE_*, JPEZY_ABL_*), the persistent kernels' build knobs (JPEZY_PS_*)
#else
// the shipped build: no probes; the persistent kernels (PS = true) are not instantiated, their knobs are constants
#define PROBE_ALL() do { } while (0)
#define JPEZY_LAB_COLOUR_VOTE(x) (x)
#define JPEZY_LAB_GUARD_CAND(x) (x)
#define JPEZY_LAB_VALID_CHUNKS(x) (x)
#define JPEZY_PS_CONSTS_LDS 0
#define JPEZY_PS_DC_FORMULA 1
#define JPEZY_PS_DCQ_LDS 0
#define JPEZY_PS_HOOK_LATE 0
#define JPEZY_PS_FENCES 1
#endif

namespace jpezy_dev {
namespace f32 {

static __constant__ double c_cos[64] = JPEZY_COS_INIT;
static __constant__ unsigned char c_zzinv[64] = JPEZY_ZZ_INV_INIT;

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
    if (JPEZY_LAB_COLOUR_VOTE(wave_any(min8(e) < LUMA_TH))) {   // one pixel in 1000: the reference's FP64 rounding decides
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
    F32Column f32col[2][8];   // DeviceTables::f32col (JPEZY_PS_CONSTS_LDS: the lane's quantiser records are read from here per quad)
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

// The same value without a table (persistent kernel, JPEZY_PS_DC_FORMULA): with the shipped constants S * S = 0.4999999999999999 puts
// int(((sum * S) * S) / 4) at (|sum| - 1) >> 3 for every sum != 0 (0 for sum = 0), and the C division by Q truncates toward zero:
// q = sign(sum) * (((|sum| - 1) >> 3) / Q).  In FP32, all exact but the division: d = trunc(|sum| / 8 - 1 / 8) (|sum| = 0 gives -0.125,
// truncated to 0), q = trunc(d * rq + bias) with rq = fl(1 / Q), bias = 1 / (2 Q): d <= 1023 puts d * rq within 6e-5 of d / Q, whose
// fractional part is a multiple of 1 / Q.  jpezy_capi.hip evaluates exactly these operations for every sum in [-8192, 8192] against
// DeviceTables::dcq at context creation and only then lets the kernel use them (EncParams::dc_rq[t] != 0).
// Only the j == 0 lane's result is used (quant_block_column: `if (j == 0) q[0] = dc`); the other lanes feed it their X0 of some other
// column -- an arbitrary float -- and throw the result away: no clamp is needed for them (an out-of-range v_cvt_i32_f32 saturates).
__device__ __forceinline__ int dc_formula(float sum, float rq, float bias)
{
    const float d = __builtin_truncf(FMAF(__builtin_fabsf(sum), 0.125f, -0.125f));
    const float q = __builtin_truncf(FMAF(d, rq, bias));
    return (int)__builtin_copysignf(q, sum);
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
    const bool cand = JPEZY_LAB_GUARD_CAND(force || fmin < th);
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
#ifndef JPEZY_DC_FORMULA_ONEQUAD
#define JPEZY_DC_FORMULA_ONEQUAD 2   // the one-quad kernel's quantised DC: 0 = table lookup (three byte loads per quad), 1 = dc_formula unchecked (A/B only),
                                     // 2 = dc_formula where the host check allowed it (EncParams::dc_rq != 0), else the table.  26.45 against 26.79 us (five rounds)
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
        if (JPEZY_LAB_COLOUR_VOTE(wave_any(min8(e) < CHROMA_TH))) {
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
    if (!JPEZY_PS_HOOK_LATE) after_pixels();
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
    constexpr bool PRE = PS && !JPEZY_PS_CONSTS_LDS;                   // the records are in registers (pre); else read where they are used, luma now, chroma later:
    const F32Column* lcol = PS && JPEZY_PS_CONSTS_LDS ? &pst->f32col[0][ju] : &tab->f32col[0][ju];   // from LDS / from global memory (one-quad kernel)
    const uint32_t zz_lo = PRE ? pre->zz_lo : lcol->zz_lo, zz_hi = PRE ? pre->zz_hi : lcol->zz_hi;
    // persistent kernels: the formula by build switch (their launchers check its validity); one-quad kernel: by the kernel argument
    // (dc_rq == 0: the formula does not hold for this build's constants -- the table lookup stays)
    const bool DCF = PS ? (bool)JPEZY_PS_DC_FORMULA : (JPEZY_DC_FORMULA_ONEQUAD == 1 || (JPEZY_DC_FORMULA_ONEQUAD == 2 && p.dc_rq[0] != 0.f && p.dc_rq[1] != 0.f));
    const signed char* dcq_l = PS && JPEZY_PS_DCQ_LDS ? dcq_lds : p.dcq_luma;
    const signed char* dcq_c = PS && JPEZY_PS_DCQ_LDS ? dcq_lds + 16385 : p.dcq_chroma;
    {
        f2 ks[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ks[k] = PRE ? pre->ks_l[k] : f2{ lcol->ks[2 * k], lcol->ks[2 * k + 1] };
        const f2 dd = PRE ? pre->dd_l : f2{ lcol->delta1[0], lcol->delta1[1] };
        const float th = PRE ? pre->th_l : lcol->th;
        {
            f2 F[4];
            fdct8p(TP, F, kc);
            const int dc_top = DCF ? dc_formula(F[0].x, p.dc_rq[0], p.dc_bias[0]) : dc_lookup(F[0].x, dcq_l);
            quant_block_column(F, ks, dd, th, j, dc_top, live, sbase, zz_lo, zz_hi, 0, m * BPM + bx, queue, FORCE != 0 DUMP_ARG);
        }
        PHASE_FENCE();
        {
            f2 F[4];
            fdct8p(BT, F, kc);
            const int dc_bot = DCF ? dc_formula(F[0].x, p.dc_rq[0], p.dc_bias[0]) : dc_lookup(F[0].x, dcq_l);
            quant_block_column(F, ks, dd, th, j, dc_bot, live, sbase, zz_lo, zz_hi, 2 * STG_BLK, m * BPM + 2 + bx, queue, FORCE != 0 DUMP_ARG);
        }
    }

    PHASE_FENCE();
    if (JPEZY_PS_HOOK_LATE) after_pixels();
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
            dc_c = DCF ? dc_formula(Fc[0].x, p.dc_rq[1], p.dc_bias[1]) : dc_lookup(Fc[0].x, dcq_c);
        }
        f2 ks[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ks[k] = PRE ? pre->ks_c[k] : f2{ lcol[8].ks[2 * k], lcol[8].ks[2 * k + 1] };
        const f2 dd = PRE ? pre->dd_c : f2{ lcol[8].delta1[0], lcol[8].delta1[1] };
        quant_block_column(Fc, ks, dd, PRE ? pre->th_c : lcol[8].th, j, dc_c, live, sbase, zz_lo, zz_hi, 4 * STG_BLK, m * BPM + 4 + bx, queue, FORCE != 0 DUMP_ARG);
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
        const int valid_chunks = JPEZY_LAB_VALID_CHUNKS(min(4, p.mcu_cols - quad_x * 4) * BPM * 8);         // 16-byte chunks
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

}  // namespace f32
}  // namespace jpezy_dev
