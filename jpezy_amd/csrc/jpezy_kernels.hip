// jpezy_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels for the jpezy hot path.
//
//   fdct_quant_kernel   : RGB->YCbCr + 4:2:0 decimation + 8x8 FDCT + Annex-K quantise + zig-zag
//                         (ref encoder/jpezy_encoder.hpp:90-172, 244-256; jpezy.hpp:36-45,131-152)
//   dequant_idct_kernel : de-zig-zag + dequantise + 8x8 IDCT + nearest upsample + YCbCr->RGB + clamp
//                         (ref decoder/jpezy_decoder.hpp:504-578, 645-676)
//
// Work decomposition (both kernels): one 64-lane wavefront owns a "quad" = 4 horizontally adjacent
// 16x16 MCUs (64x16 pixels, 24 blocks); a 256-thread workgroup is 4 independent waves (no s_barrier --
// each wave has a private LDS slice and synchronises with itself only).  Lane = (row, m): row = lane>>2
// is a pixel row of the MCU, m = lane&3 the MCU of the quad, so one lane streams a 16-pixel row segment
// of each plane as a single 16-byte access (4 lanes = 64 contiguous bytes) and the 3 KB of coefficients
// of a quad move as 3 coalesced 1 KB wave accesses.  The two separable 1-D passes run in registers (one
// 8-point transform per lane-row/column, 2 independent transforms per lane for ILP) with a padded,
// bank-conflict-free LDS transpose between them.  No MFMA: FP64 8x8 is VALU work.
//
// Exactness (DESIGN.md "exactness"): the reference truncates FP64 results, so the output depends on the
// exact rounding sequence only where the true value sits on a quantiser / integer boundary.  The fast
// separable transform (FMA allowed, error < 1e-9) is accepted when the fixed-point value is >= 2 units
// of 2^-24 (2^-18 for samples) away from every boundary; otherwise the coefficient is re-evaluated by
// exact_*() in the reference's exact operation order (plain IEEE mul/add, no contraction).  DC terms are
// sums of integers and are always evaluated exactly.  Colour conversion is evaluated in the reference's
// exact FP64 order everywhere.  This file must be compiled with -ffp-contract=off; every fused
// multiply-add below is an explicit __builtin_fma in a fast-path estimate.
#include "jpezy_device.h"
#include "../../include/jpezy_constants.h"

namespace jpezy_dev {

__constant__ double c_cos[64] = JPEZY_COS_INIT;            // [u*8+x] = cos((2x+1)u*pi/16)
__constant__ unsigned char c_zzinv[64] = JPEZY_ZZ_INV_INIT;  // natural index -> zig-zag position
// the same table by COLUMN: c_zzcol[u] = the eight zig-zag positions of natural column u (rows v = 0..7), one byte each -- the decode kernel's
// lane (column u) fetches its eight staging offsets with ONE 8-byte load instead of eight byte loads (JPEZY_DEC_ZZCOL)
struct ZzCol { uint32_t lo, hi; };
constexpr unsigned char kZzInvH[64] = JPEZY_ZZ_INV_INIT;
constexpr ZzCol zz_col(int u)
{
    uint32_t lo = 0, hi = 0;
    for (int v = 0; v < 4; ++v) { lo |= (uint32_t)kZzInvH[v * 8 + u] << (8 * v); hi |= (uint32_t)kZzInvH[(v + 4) * 8 + u] << (8 * v); }
    return ZzCol{ lo, hi };
}
__constant__ ZzCol c_zzcol[8] = { zz_col(0), zz_col(1), zz_col(2), zz_col(3), zz_col(4), zz_col(5), zz_col(6), zz_col(7) };
#ifndef JPEZY_DEC_ZZCOL
#define JPEZY_DEC_ZZCOL 1
#endif

__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned magic, unsigned shift)   // see fast_div_setup
{
    const unsigned q = __umulhi(n, magic);
    return magic ? (((n - q) >> 1) + q) >> shift : n;
}

// wave-uniform "some lane": one v_cmp into an SGPR pair + s_cmp (HIP's __any goes through a 0/1 VGPR)
__device__ __forceinline__ bool wave_any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0ull; }

// Outputs are streamed out and never re-read by the kernel: a non-temporal store leaves less dirty data in the eight
// L2s for the end-of-kernel write-back (measured on the f32 encode kernel: 2 us per 4096^2 frame).
#ifndef JPEZY_NO_NT
__device__ __forceinline__ void nt_store16(uint4* dst, uint4 v)
{
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(v4u{v.x, v.y, v.z, v.w}, reinterpret_cast<v4u*>(dst));
}
#else
__device__ __forceinline__ void nt_store16(uint4* dst, uint4 v) { *dst = v; }
#endif

#define JPEZY_S JPEZY_INV_SQRT2

// cos(k*pi/16) -- the same correctly rounded doubles as the cos table rows (fast path only)
#define C1 0x1.f6297cff75cb0p-1
#define C2 0x1.d906bcf328d46p-1
#define C3 0x1.a9b66290ea1a3p-1
#define C4 0x1.6a09e667f3bcdp-1
#define C5 0x1.1c73b39ae68c8p-1
#define C6 0x1.87de2a6aea963p-2
#define C7 0x1.8f8b83c69a60bp-3

#define FMA(a, b, c) __builtin_fma((a), (b), (c))

// LDS geometry (dwords), chosen so that the column reads (ds_read_b64, 32-lane groups, 64 banks) are
// conflict free: per-MCU stride == 16 (mod 64) dwords.  Row pitch 36 dwords keeps 16-byte alignment and
// limits the ds_write_b128 conflicts to 2-way.
constexpr int Y_PITCH = 36;                 // 16 doubles + 2 pad
constexpr int Y_MCU = 16 * Y_PITCH + 16;    // 592
constexpr int C_PITCH = 20;                 // 8 doubles + 2 pad
constexpr int C_COMP = 8 * C_PITCH;         // 160
constexpr int C_MCU = 2 * C_COMP + 16;      // 336
constexpr int TILE_DWORDS = 4 * Y_MCU;      // 2368 dwords = 9472 B: transpose tiles / staging
// Behind the tiles: the wave's queue of guard-band hits (count + entries), never overlapped by a tile.
constexpr int QUEUE_CAP = 126;
constexpr int WAVE_LDS_DWORDS = TILE_DWORDS + 64;   // 9728 B per wave, 38912 B per workgroup (4 per CU)

// Decode kernel LDS slice (dwords): the luma tile is exchanged in two halves (left blocks, then right blocks) so that
// the slice is 5.4 KB instead of 9.5 KB and LDS no longer caps the kernel at 4 waves per SIMD.
// Geometry from tools/profile/lds_bank_model.py (the lane groups and bank functions of MI355X_MICROARCH.md): the coefficient
// staging area has a 144-byte block pitch (the zig-zag-indexed 2-byte column reads of the four MCUs then start 24 banks
// apart: 108 LDS cycles per wave instead of 192), the luma half tile an MCU stride == 8 (mod 32) dwords (conflict-free
// ds_write_b64 columns; the ds_read_b128 rows become 2-way: 128 cycles for both instead of 160 -- no pitch makes both
// directions conflict-free), the chroma tile unpadded rows (64 instead of 80).  Model: 456 -> 324 cycles per wave;
// counters (profiles/r02e_ab_decode.txt): SQ_LDS_BANK_CONFLICT 219 -> 104, SQ_LDS_IDX_ACTIVE 487 -> 371 per wave -- and
// 39.5 -> 39.3 us: the kernel is not bound by its LDS traffic.
#ifdef JPEZY_DEC_LDS_R01     // round-1 geometry, kept for A/B counters (tools/ab/ab_build.py)
constexpr int DH_PITCH = 20, DH_MCU = 16 * DH_PITCH + 16;                    // 336
constexpr int DC_PITCH = 20, DC_COMP = 8 * DC_PITCH, DC_MCU = 2 * DC_COMP + 16;
constexpr int DSTG_PITCH = 128;               // bytes per staged block
#else
constexpr int DH_PITCH = 20;                  // 8 doubles + 2 pad
constexpr int DH_MCU = 16 * DH_PITCH + 8;     // 328
constexpr int DC_PITCH = 16, DC_COMP = 8 * DC_PITCH, DC_MCU = 2 * DC_COMP + 8;    // 16 / 128 / 264
constexpr int DSTG_PITCH = 144;
#endif
constexpr int TF_PITCH = 16, TF_MCU = 16 * TF_PITCH + 8;   // tolerance mode: the luma tile as floats, 4 x 264 dwords
constexpr int DEC_TILE_DWORDS = 4 * 336;      // 1344 dwords = 5376 B
constexpr int DEC_LDS_DWORDS = DEC_TILE_DWORDS + 16;
static_assert(4 * DC_MCU <= DEC_TILE_DWORDS && 4 * DH_MCU <= DEC_TILE_DWORDS && 24 * DSTG_PITCH <= DEC_TILE_DWORDS * 4 &&
              1024 + 128 <= DEC_TILE_DWORDS && 4 * TF_MCU <= DEC_TILE_DWORDS, "decode slice too small");

__device__ __forceinline__ void wave_sync()
{
    // LDS traffic of one wave is executed in order; this only stops the compiler from moving LDS
    // accesses of different lanes across the phase boundary.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// X[u] = sum_x x[x] * cos((2x+1)u*pi/16), u = 0..7 ; X[0] is the plain (exact, for integers) sum.
__device__ __forceinline__ void fdct8(const double* x, double* X)
{
    const double s0 = x[0] + x[7], s1 = x[1] + x[6], s2 = x[2] + x[5], s3 = x[3] + x[4];
    const double d0 = x[0] - x[7], d1 = x[1] - x[6], d2 = x[2] - x[5], d3 = x[3] - x[4];
    const double e0 = s0 + s3, e1 = s1 + s2, e2 = s0 - s3, e3 = s1 - s2;
    X[0] = e0 + e1;
    X[4] = (e0 - e1) * C4;
    X[2] = FMA(e3, C6, e2 * C2);
    X[6] = FMA(-e3, C2, e2 * C6);
    X[1] = FMA(d3, C7, FMA(d2, C5, FMA(d1, C3, d0 * C1)));
    X[3] = FMA(-d3, C5, FMA(-d2, C1, FMA(-d1, C7, d0 * C3)));
    X[5] = FMA(d3, C3, FMA(d2, C7, FMA(-d1, C1, d0 * C5)));
    X[7] = FMA(-d3, C1, FMA(d2, C3, FMA(-d1, C5, d0 * C7)));
}

// x[y] = sum_v X[v] * cos((2y+1)v*pi/16)   (X[0] already carries its 1/sqrt2)
__device__ __forceinline__ void idct8(const double* X, double* x)
{
    const double t0 = FMA(X[4], C4, X[0]), t1 = FMA(-X[4], C4, X[0]);
    const double t2 = FMA(X[6], C6, X[2] * C2), t3 = FMA(-X[6], C2, X[2] * C6);
    const double E0 = t0 + t2, E3 = t0 - t2, E1 = t1 + t3, E2 = t1 - t3;
    const double O0 = FMA(X[7], C7, FMA(X[5], C5, FMA(X[3], C3, X[1] * C1)));
    const double O1 = FMA(-X[7], C5, FMA(-X[5], C1, FMA(-X[3], C7, X[1] * C3)));
    const double O2 = FMA(X[7], C3, FMA(X[5], C7, FMA(-X[3], C1, X[1] * C5)));
    const double O3 = FMA(-X[7], C1, FMA(X[5], C3, FMA(-X[3], C5, X[1] * C7)));
    x[0] = E0 + O0; x[7] = E0 - O0;
    x[1] = E1 + O1; x[6] = E1 - O1;
    x[2] = E2 + O2; x[5] = E2 - O2;
    x[3] = E3 + O3; x[4] = E3 - O3;
}

// the same butterflies in FP32: luma of the opt-in tolerance mode of the decode kernel (jpezy_ctx_set_decode_tolerance)
#define FMAF(a, b, c) __builtin_fmaf((a), (b), (c))
__device__ __forceinline__ void idct8f(const float* X, float* x)
{
    const float t0 = FMAF(X[4], (float)C4, X[0]), t1 = FMAF(-X[4], (float)C4, X[0]);
    const float t2 = FMAF(X[6], (float)C6, X[2] * (float)C2), t3 = FMAF(-X[6], (float)C2, X[2] * (float)C6);
    const float E0 = t0 + t2, E3 = t0 - t2, E1 = t1 + t3, E2 = t1 - t3;
    const float O0 = FMAF(X[7], (float)C7, FMAF(X[5], (float)C5, FMAF(X[3], (float)C3, X[1] * (float)C1)));
    const float O1 = FMAF(-X[7], (float)C5, FMAF(-X[5], (float)C1, FMAF(-X[3], (float)C7, X[1] * (float)C3)));
    const float O2 = FMAF(X[7], (float)C3, FMAF(X[5], (float)C7, FMAF(-X[3], (float)C1, X[1] * (float)C5)));
    const float O3 = FMAF(-X[7], (float)C1, FMAF(X[5], (float)C3, FMAF(-X[3], (float)C5, X[1] * (float)C7)));
    x[0] = E0 + O0; x[7] = E0 - O0;
    x[1] = E1 + O1; x[6] = E1 - O1;
    x[2] = E2 + O2; x[5] = E2 - O2;
    x[3] = E3 + O3; x[4] = E3 - O3;
}

// ---- colour conversion in the reference's exact order (ref jpezy_encoder.hpp:244-256) ----
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

// In-order sum of one double per lane, lane 0 first: sum = (((0 + t0) + t1) + ...) + t63, every add
// rounded -- the reference's accumulation order.  Wave-uniform result.
__device__ __forceinline__ double ordered_wave_sum(double t)
{
    const int lo = (int)(unsigned)(__builtin_bit_cast(unsigned long long, t) & 0xFFFFFFFFull);
    const int hi = (int)(unsigned)(__builtin_bit_cast(unsigned long long, t) >> 32);
    double sum = 0;
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        const unsigned long long bits = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, k) << 32) |
                                        (unsigned)__builtin_amdgcn_readlane(lo, k);
        sum += __builtin_bit_cast(double, bits);
    }
    return sum;
}

struct BlockRef {   // the frame's planes: what the exact path needs to find a block again
    const uint8_t* r;
    const uint8_t* g;
    const uint8_t* b;
    int W, H;
};

// ---- exact-order FDCT + quantise of ONE coefficient by the whole wave (ref jpezy_encoder.hpp:146-172) ----
// All arguments are wave-uniform.  Lane k owns term k = y*8+x of the reference's double loop: it re-reads its
// pixel, converts it, forms (pic*cos[j][x])*cos[i][y]; the 64 terms are then added in the reference's order.
// comp 0: luma block with top-left pixel (px0,py0), step 1.  comp 1/2: Cb/Cr of the MCU at (px0,py0), step 2
// (top-left sample of each 2x2, ref :134-142).  Coordinates clamp to the image (ref :101,104).
__device__ __forceinline__ int exact_fdct_coef_wave(const BlockRef& img, int px0, int py0, int comp, int i, int j,
                                                    int Q, int lane)
{
    const int step = comp ? 2 : 1;
    const int y = lane >> 3, x = lane & 7;
    const int yy = min(py0 + y * step, img.H - 1);
    const int xx = min(px0 + x * step, img.W - 1);
    const size_t idx = (size_t)yy * img.W + xx;
    const double rf = (double)img.r[idx], gf = (double)img.g[idx], bf = (double)img.b[idx];
    const double pic = comp == 0 ? ref_y(rf, gf, bf) : comp == 1 ? ref_cb(rf, gf, bf) : ref_cr(rf, gf, bf);
    const double sum = ordered_wave_sum(pic * c_cos[j * 8 + x] * c_cos[i * 8 + y]);
    const double cu = j ? 1.0 : JPEZY_S, cv = i ? 1.0 : JPEZY_S;
    const int dct = (int)(sum * cu * cv / 4);
    return dct / Q;
}

// Quantise the 8 coefficients F[i] (vertical frequency i, this lane's horizontal frequency j).
// ks[i] = cu*cv/(4Q) * 2^24.  n[i] = trunc(v/Q * 2^24); q[i] = trunc-toward-zero(n / 2^24).
// Returns true when some coefficient lies within 1 unit of a multiple of 2^24 (candidate for the exact path).
__device__ __forceinline__ bool quant8(const double* F, const double* ks, bool dc_lane, double rq_dc, int* n, int* q)
{
    constexpr int MASK = (1 << QFRAC_BITS) - 1;
    unsigned m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        n[i] = (int)(F[i] * ks[i]);                          // v_cvt_i32_f64 truncates toward zero
        q[i] = (n[i] + ((n[i] >> 31) & MASK)) >> QFRAC_BITS;  // trunc-toward-zero division by 2^24
        m[i] = (unsigned)(n[i] + 1) & MASK;                  // 0,1,2 <=> within one unit of a boundary
    }
    // DC: F[0] of the j == 0 lane is the exact integer sum of the block, so the reference value
    // int(sum*S*S/4) is reproduced bit for bit; (|iv|+0.5)/Q is never within 0.5/Q of an integer.
    {
        const double iv = __builtin_trunc(F[0] * JPEZY_S * JPEZY_S / 4);
        int nq = (int)((__builtin_fabs(iv) + 0.5) * rq_dc);
        nq = iv < 0 ? -nq : nq;
        if (dc_lane) {
            q[0] = nq;
            m[0] = MASK;
        }
    }
    const unsigned a = min(min(m[0], m[1]), m[2]), b = min(min(m[3], m[4]), m[5]), c = min(m[6], m[7]);
    return min(min(a, b), c) <= 2u;
}

// byte offsets of the staging area: blocks padded to 144 B so that the 8 blocks written by one
// ds_write_b16 wave-instruction fall on different banks
constexpr int STG_BLK = 144;

// Quantise one block column, write it (zig-zag) to the staging area and queue the guard-band hits.
// blk = index of the block inside the quad (m*BPM + b).  Queue entry = blk << 6 | natural index.
__device__ __forceinline__ void quant_block_column(const double* F, const double* ks, int j, double rq_dc,
                                                   bool live, const int* zoff, char* stage_blk, int blk,
                                                   unsigned* queue)
{
    constexpr int MASK = (1 << QFRAC_BITS) - 1;
    int n[8], q[8];
    const bool cand = quant8(F, ks, j == 0, rq_dc, n, q);
    if (cand && live) {   // rare.  Fully unrolled: a runtime index into n[] would send the array to scratch
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // within one unit of a multiple of 2^24 -- except around 0, which is not a truncation boundary;
            // the DC term of the j == 0 lane is already exact
            const bool f = ((unsigned)(n[i] + 1) & MASK) <= 2u && (unsigned)(n[i] + 1) > 2u && !(i == 0 && j == 0);
            if (f) {
                const unsigned slot = atomicAdd(&queue[0], 1u);
                if (slot < (unsigned)QUEUE_CAP)
                    reinterpret_cast<unsigned short*>(queue + 1)[slot] = (unsigned short)((blk << 6) | (i * 8 + j));
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<int16_t*>(stage_blk + zoff[i]) = (int16_t)q[i];
}

__device__ __forceinline__ double byte_of(const uint32_t* w, int k)
{
    return (double)((w[k >> 2] >> ((k & 3) * 8)) & 0xFFu);
}

// ======================================================================================================
// ENCODE
// ======================================================================================================
template <bool GRAY, bool ALIGNED, bool FORCE_EXACT>
__global__ __launch_bounds__(64 * WPB, 4) void fdct_quant_kernel(EncParams p)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[WPB][WAVE_LDS_DWORDS];
    constexpr int BPM = GRAY ? 4 : 6;
    constexpr int STG_BASE = 4 * C_MCU * 4;                   // bytes: staging sits behind the chroma tile
    static_assert(STG_BASE + 4 * 6 * STG_BLK <= TILE_DWORDS * 4, "staging does not fit the tile area");

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;   // WPB waves per workgroup
    const long quad = (long)blockIdx.x * WPB + wave;
    const long quads_per_frame = (long)p.mcu_rows * p.quads_per_row;
    if (quad >= quads_per_frame * p.n_frames) return;   // wave-uniform
    const int frame = (int)(quad / quads_per_frame);
    const int qrem = (int)(quad - (long)frame * quads_per_frame);
    const int mcu_y = qrem / p.quads_per_row;
    const int quad_x = qrem - mcu_y * p.quads_per_row;

    uint32_t* lds = lds_all[wave];
    unsigned* queue = lds + TILE_DWORDS;                        // [0] = count, then 16-bit entries
    if (lane == 0) queue[0] = 0;
    const int row = lane >> 2, m = lane & 3;
    const int mcu_x_raw = quad_x * 4 + m;
    const bool live = mcu_x_raw < p.mcu_cols;
    const int mcu_x = live ? mcu_x_raw : p.mcu_cols - 1;
    const int W = p.W, H = p.H;
    const uint8_t* pr = p.r + (size_t)frame * p.plane_stride;
    const uint8_t* pg = p.g + (size_t)frame * p.plane_stride;
    const uint8_t* pb = p.b + (size_t)frame * p.plane_stride;
    const BlockRef img = { pr, pg, pb, W, H };

    // ---- 1. stream this lane's 16-pixel row segment of the three planes ----
    uint32_t R[4], G[4], B[4];
    {
        const int y = min(mcu_y * 16 + row, H - 1);             // edge replication, ref :101
        const size_t rowoff = (size_t)y * W;
        if (ALIGNED) {
            const size_t off = rowoff + (size_t)mcu_x * 16;
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

    // ---- 2. luma of the 16 pixels, row pass of the left / right block, into the transpose tile ----
    {
        double yv[16], X[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) yv[k] = ref_y(byte_of(R, k), byte_of(G, k), byte_of(B, k));
        fdct8(yv, X);
        fdct8(yv + 8, X + 8);
        double2* dst = reinterpret_cast<double2*>(lds + m * Y_MCU + row * Y_PITCH);
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = make_double2(X[2 * k], X[2 * k + 1]);
    }
    wave_sync();

    // ---- 3. luma column pass: lane (cq, m) owns column cq of the 16x16 tile = column j of two blocks ----
    const int cq = row;                 // 0..15
    const int j = cq & 7;
    const DeviceTables* tab = p.tab;
    char* stage = reinterpret_cast<char*>(lds) + STG_BASE;
    int zoff[8];                        // byte offset of natural coefficient (i, j) inside a staged block
#pragma unroll
    for (int i = 0; i < 8; ++i) zoff[i] = 2 * (int)c_zzinv[i * 8 + j];
    {
        double Ftop[8], Fbot[8];
        {
            double col[16];
            const uint32_t* src = lds + m * Y_MCU + cq * 2;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) col[rr] = *reinterpret_cast<const double*>(src + rr * Y_PITCH);
            fdct8(col, Ftop);
            fdct8(col + 8, Fbot);
        }
        wave_sync();   // every lane has read the luma tile: the slice is reused from here on

        // ---- 4. quantise + zig-zag the two luma block columns into the staging area ----
        double ks[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ks[i] = tab->qscale[0][j][i];
        const double rq = tab->rq_dc[0];
        const int bx = cq >> 3;   // 0: left blocks (Y0,Y2), 1: right blocks (Y1,Y3)
        quant_block_column(Ftop, ks, j, rq, live, zoff, stage + (m * BPM + bx) * STG_BLK, m * BPM + bx, queue);
        quant_block_column(Fbot, ks, j, rq, live, zoff, stage + (m * BPM + 2 + bx) * STG_BLK, m * BPM + 2 + bx, queue);
    }

    // ---- 5. chroma: top-left pixel of every 2x2 (ref :134-142) = even pixel rows, even columns.  The odd-row
    //         lane fetches its even neighbour's pixels (DPP row_shr:4) and computes Cr while the even-row lane
    //         computes Cb, so all 64 lanes carry one chroma row each. ----
    if (!GRAY) {
        const bool odd = (row & 1) != 0;
        uint32_t R2[4], G2[4], B2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // row_shr:4 within each 16-lane DPP row, written only to lanes 4-7 and 12-15 (bank_mask 0b1010)
            R2[k] = (uint32_t)__builtin_amdgcn_update_dpp((int)R[k], (int)R[k], 0x114, 0xF, 0xA, false);
            G2[k] = (uint32_t)__builtin_amdgcn_update_dpp((int)G[k], (int)G[k], 0x114, 0xF, 0xA, false);
            B2[k] = (uint32_t)__builtin_amdgcn_update_dpp((int)B[k], (int)B[k], 0x114, 0xF, 0xA, false);
        }
        // Cb = (-(0.1687 r) - 0.3313 g) + 0.5 b ; Cr = (0.5 r - 0.4187 g) - 0.0813 b  (ref :249-256), both as
        // trunc((k1*r - k2*g) + k3*b): (-a)*r == -(a*r) and x - y == x + (-y) hold bit for bit in IEEE-754.
        const double k1 = odd ? 0.5000 : -0.1687, k2 = odd ? 0.4187 : 0.3313, k3 = odd ? -0.0813 : 0.5000;
        double cv[8], cX[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            cv[k] = __builtin_trunc((k1 * byte_of(R2, 2 * k) - k2 * byte_of(G2, 2 * k)) + k3 * byte_of(B2, 2 * k));
        fdct8(cv, cX);
        double2* dst = reinterpret_cast<double2*>(lds + m * C_MCU + (odd ? C_COMP : 0) + (row >> 1) * C_PITCH);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = make_double2(cX[2 * k], cX[2 * k + 1]);
        wave_sync();

        double Fc[8];
        {
            double col[8];
            const uint32_t* src = lds + m * C_MCU + (cq >> 3) * C_COMP + j * 2;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) col[rr] = *reinterpret_cast<const double*>(src + rr * C_PITCH);
            fdct8(col, Fc);
        }
        double ks[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ks[i] = tab->qscale[1][j][i];
        const int comp = 1 + (cq >> 3);
        quant_block_column(Fc, ks, j, tab->rq_dc[1], live, zoff, stage + (m * BPM + 3 + comp) * STG_BLK,
                           m * BPM + 3 + comp, queue);
    }
    wave_sync();

    // ---- 5b. guard-band hits: re-evaluate in the reference's exact operation order, one coefficient at a
    //          time, all 64 lanes cooperating (rare: ~0.7 % of blocks on random pixels).  FORCE_EXACT (test
    //          hook) and a queue overflow send EVERY coefficient of the quad through this path. ----
    {
        const unsigned nq = queue[0];
        const bool all = FORCE_EXACT || nq > (unsigned)QUEUE_CAP;
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
                const int emx = quad_x * 4 + em;
                int px0 = emx * 16, py0 = mcu_y * 16, comp = 0;
                if (eb < 4) { px0 += (eb & 1) * 8; py0 += (eb >> 1) * 8; } else { comp = eb - 3; }
                const int Q = tab->qt[comp ? 1 : 0][nat];
                const int qv = exact_fdct_coef_wave(img, px0, py0, comp, ei, ej, Q, lane);
                if (lane == 0) *reinterpret_cast<int16_t*>(stage + blk * STG_BLK + 2 * (int)c_zzinv[nat]) = (int16_t)qv;
                ++done;
            }
            if (lane == 0 && done) atomicAdd(p.fallback_count + ((blockIdx.x * (unsigned)WPB + wave) & (COUNTER_SHARDS - 1)), (unsigned long long)done);
            wave_sync();
        }
    }

    // ---- 6. coalesced store of the quad's coefficients (BPM*128 bytes per MCU, contiguous) ----
    {
        const int valid_mcus = min(4, p.mcu_cols - quad_x * 4);
        const int valid_chunks = valid_mcus * BPM * 8;         // 16-byte chunks
        int16_t* gbase = p.coeffs + (size_t)frame * p.coeffs_per_frame +
                         ((size_t)mcu_y * p.mcu_cols + (size_t)quad_x * 4) * (BPM * 64);
        uint4* g4 = reinterpret_cast<uint4*>(gbase);
#pragma unroll
        for (int k = 0; k < BPM * 128 * 4 / 1024; ++k) {
            const int c = k * 64 + lane;
            if (c < valid_chunks) nt_store16(g4 + c, *reinterpret_cast<const uint4*>(stage + (c >> 3) * STG_BLK + (c & 7) * 16));
        }
    }
}

// ======================================================================================================
// DECODE
// ======================================================================================================

// exact-order IDCT of ONE sample (ref jpezy_decoder.hpp:645-670) from the block's list of NON-ZERO dequantised
// coefficients, kept in the reference's summation order (v outer, u inner).  Skipping the zero coefficients
// leaves the rounded running sum unchanged (x + (+-0) == x), so this is the reference's sum bit for bit.
// list[t] = {natural index, coefficient * quantiser}; K, list wave-uniform; x, y per lane.
__device__ __forceinline__ int exact_idct_sample_sparse(const int2* __restrict__ list, int K, int x, int y)
{
    double sum = 0;
#pragma unroll 1
    for (int t = 0; t < K; ++t) {
        const int2 e = list[t];
        const int v = e.x >> 3, u = e.x & 7;
        const double cv = (!v) ? JPEZY_S : 1.0, cu = (!u) ? JPEZY_S : 1.0;
        sum += cu * cv * e.y * c_cos[u * 8 + x] * c_cos[v * 8 + y];
    }
    return (int)(sum / 4 + 128);
}

// Sample of the reference, int(sum / 4 + 128) (ref :667), from the fast row sum, plus the guard key of the fast path.  v is
// within eps of an integer  <=>  fract(v) < eps or fract(v) > 1 - eps; fract(v) lies in [0, 1), where the high word of a double
// orders like the value, so the test is  hi(fract(v)) < hi(eps)  or  hi(fract(v)) >= hi(1 - eps)  (eps = 2^-18 and 1 - eps have
// zero low words, so the high words decide exactly -- fract == 1 - eps itself is flagged too, harmlessly).  A wild or non-finite v has fract 0 or NaN: flagged.  v_cvt_i32_f64
// truncates toward zero like the reference's int() and saturates.  The fast sum is within 2e-10 of the reference's sum
// (DESIGN.md); eps = 2^-18 is far above it.  The per-sample flag bits are only formed when a reduction over the eight
// keys says that some lane of the wave has a sample in the band.  (Round 1 converted fract(v) to FP32 and
// compared |e - 1/2|: one conversion per sample more.)
constexpr uint32_t SAMPLE_KEY_LO = 0x3ED00000u;                        // high word of 2^-18: fract below it <=> hi(fract) below it
constexpr uint32_t SAMPLE_KEY_HI = 0x3FEFFFF8u;                        // high word of 1 - 2^-18
// v = row sum / 4 + 128 arrives ready-made: the / 4 is folded into the dequantiser constants (an exact scaling of every
// intermediate value) and the level shift enters the row pass as one addition to its DC input.  fl(s/4 + 128) is what
// the reference forms (ref :667); a DC-only block still reproduces it bit for bit (its term reaches the addition
// unrounded, scaled by an exact 1/4).
__device__ __forceinline__ int sample_of(double v, uint32_t& key)
{
    key = (uint32_t)__double2hiint(__builtin_amdgcn_fract(v));
    return (int)v;
}
// flag bits (bit k: sample k is inside the guard band) of eight samples; dc_only: the block's samples are exact by
// construction (step 2 of the kernel) and exempt.  The keys are reduced two-sided (v_min3_u32 / v_max3_u32 chains: 4 + 4
// instructions for eight samples, no per-sample arithmetic); the bits are only spelled out when some lane needs them.
__device__ __forceinline__ unsigned guard_bits8(const uint32_t* e, bool dc_only)
{
    const uint32_t emax = max(max(max(max(max(max(max(e[0], e[1]), e[2]), e[3]), e[4]), e[5]), e[6]), e[7]);
    const uint32_t emin = min(min(min(min(min(min(min(e[0], e[1]), e[2]), e[3]), e[4]), e[5]), e[6]), e[7]);
    unsigned bits = 0;
    if (wave_any(!dc_only && (emin < SAMPLE_KEY_LO || emax >= SAMPLE_KEY_HI))) {
#pragma unroll
        for (int k = 0; k < 8; ++k) bits |= ((e[k] < SAMPLE_KEY_LO || e[k] >= SAMPLE_KEY_HI) ? 1u : 0u) << k;
        if (dc_only) bits = 0;
    }
    return bits;
}

// revise_value (ref :672-676) of four values and their packing into one word: saturate to int16 pairs
// (v_cvt_pk_i16_i32), saturate those to bytes (v_sat_pk_u8_i16), join the halves (v_perm_b32): 5 instructions for 4 values.
__device__ __forceinline__ uint32_t clamp_pack4(const int* v)
{
    const uint32_t p01 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(v[0], v[1]));
    const uint32_t p23 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(v[2], v[3]));
    uint32_t b01, b23;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b01) : "v"(p01));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b23) : "v"(p23));
    return __builtin_amdgcn_perm(b23, b01, 0x05040100u);     // bytes 0,1 of b01 then bytes 0,1 of b23
}

__device__ __forceinline__ uint32_t clamp_byte(double v)   // revise_value, ref :672-676
{
    const int iv = (int)v;                 // truncates; (-1,0) -> 0 like the reference's v < 0 -> 0
    return (uint32_t)min(max(iv, 0), 255);
}

// one coefficient from the LDS staging area as a sign-extended int (ds_read_i16).  The empty asm keeps the compiler from
// splitting the value into a zero-extended load for the "any non-zero AC" test and a v_bfe_i32 sign extension for the
// arithmetic (21 extra instructions per quad).
__device__ __forceinline__ int ld_coef(const int16_t* p)
{
    int v = *p;
    asm("" : "+v"(v));          // not volatile: the loads may still be reordered
    return v;
}

// floor(x) / floor(-x) as an int in one instruction (v_cvt_flr_i32_f32; asm: the compiler only forms it under fast-math flags)
__device__ __forceinline__ int cvt_floor(float x)
{
    int r;
    asm("v_cvt_flr_i32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ int cvt_floor_neg(float x)
{
    int r;
    asm("v_cvt_flr_i32_f32_e64 %0, -%1" : "=v"(r) : "v"(x));
    return r;
}
#ifndef JPEZY_DEC_INT_COLOUR
#define JPEZY_DEC_INT_COLOUR 1        // integer colour offsets (step 5 of dequant_idct_kernel); 0: doubles for every wave, as until round 3
#endif
constexpr float DEC_CHROMA_BAND = 5e-5f, DEC_CHROMA_GATE = 512.f;      // tests/test_colour_offsets.py
#ifndef JPEZY_DEC_FULLLINE
#define JPEZY_DEC_FULLLINE 1          // colour planes leave as whole 128-byte lines (see the store section of dequant_idct_kernel)
#endif
// waves per SIMD the register budget is set for (colour: 87 VGPRs since the chroma column pass runs after the luma halves)
#ifndef JPEZY_DEC_WAVES
#define JPEZY_DEC_WAVES 5
#endif
#ifndef JPEZY_DEC_WAVES_GRAY
#define JPEZY_DEC_WAVES_GRAY 5
#endif
// TOL (opt-in, jpezy_ctx_set_decode_tolerance): the LUMA transforms run as FP32 butterflies without guard keys or exact path
// and the luma tile is exchanged in one piece as floats; chroma, colour conversion and clamping stay as in the exact kernel.
// A luma sample then equals the reference's or differs from it by one (bound below), and R, G, B inherit exactly that
// difference through the 1-Lipschitz truncate-and-clamp: max |difference| <= 1 per channel, what BASELINE.json's north_star
// asks of the decoder.  Chroma is NOT relaxed: one unit on Cb is 1.77 on B.
// Bound: a wave whose raw coefficients all satisfy |c| <= coef_limit (|c * Q| <= 2^15; otherwise the wave takes the exact path
// for every sample, as the exact kernel does) has |in[u][v]| <= 2^13, so sum |in| <= 2^19 over a block and the FP32 result of the
// two butterfly passes (at most 13 roundings on any input->output path, dequantiser constants and level shift rounded once
// each) is within (13 + 2) * 2^-24 * 2^19 = 0.47 of the exact sample value; the butterflies' cosine constants are FP32
// roundings too (relative 2^-24 each on up to two factors per path: another 0.25 ... 0.4 by the same norm-wise count), so the
// rigorous figure is 0.7 ... 0.9 -- still below 1: the truncated samples differ by at most one.
// MODE: 0 = exact, with the coefficient range test; 1 = exact, no range test (8-bit quantiser tables: see launch_dequant_idct);
// 2 = tolerance mode (always with the range test: the FP32 bound above needs it)
template <bool GRAY, bool ALIGNED, bool FORCE_EXACT, int MODE>
__global__ __launch_bounds__(64 * WPB, GRAY ? JPEZY_DEC_WAVES_GRAY : JPEZY_DEC_WAVES) void dequant_idct_kernel(DecParams p)
{
    constexpr bool TOL = MODE == 2, RANGE = MODE != 1;
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[WPB][DEC_LDS_DWORDS];
    constexpr int BPM = 6;

    // WPB waves per workgroup; grid: x = quads of one frame / WPB, y = frame.  The wave index is made an SGPR so that the
    // quad position and every base address are computed once on the scalar unit.
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned qrem = blockIdx.x * (unsigned)WPB + (unsigned)wave;
    if (qrem >= (unsigned)(p.mcu_rows * p.quads_per_row)) return;      // wave-uniform
    const int frame = (int)blockIdx.y;
    const int mcu_y = (int)fast_div(qrem, p.qpr_magic, p.qpr_shift);
    const int quad_x = (int)qrem - mcu_y * p.quads_per_row;

    uint32_t* lds = lds_all[wave];
    if (lane == 0) lds[DEC_TILE_DWORDS] = 0;                    // exact-path block mask
    const int row = lane >> 2, m = lane & 3;
    const int mcu_x = quad_x * 4 + m;
    const bool live = mcu_x < p.mcu_cols;
    const int W = p.W, H = p.H;

    // The luma column's dequantiser constants and zig-zag offsets are requested before the coefficients: issued where the
    // column pass needs them -- behind the staging wait -- their L2 round trip is paid once per wave, in series with the
    // coefficient loads' (39.35 -> 38.55 us).  The chroma column's constants, fetched in the middle of the kernel, are
    // covered by the other waves: parking them in LDS at the top measured no gain (38.9 us).
    const int cq = row, u = cq & 7;
    double dq[TOL ? 1 : 8];
    float dqf[TOL ? 8 : 1];
    unsigned char zp[8];
#if JPEZY_DEC_ZZCOL
    const ZzCol zc = c_zzcol[u];
#endif
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        if (TOL) dqf[v] = p.dqscale_f[u * 8 + v]; else dq[v] = p.dqscale[(0 * 8 + u) * 8 + v];
#if JPEZY_DEC_ZZCOL
        zp[v] = (unsigned char)(((v < 4 ? zc.lo : zc.hi) >> (8 * (v & 3))) & 0xFFu);
#else
        zp[v] = c_zzinv[v * 8 + u];
#endif
    }

    // ---- 1. coalesced load of the quad's 3 KB of coefficients into the staging area ----
    const int16_t* gbase = p.coeffs + (size_t)frame * p.coeffs_per_frame +
                           ((size_t)mcu_y * p.mcu_cols + (size_t)quad_x * 4) * (BPM * 64);
    {
        const int valid_mcus = min(4, p.mcu_cols - quad_x * 4);
        const int valid_bytes = valid_mcus * BPM * 128;
        const uint4* g4 = reinterpret_cast<const uint4*>(gbase);
        // all three loads are issued before the first one is waited for (one memory latency per wave, not three)
        uint4 v[3];
        // --gray never looks at the chroma blocks (blocks 4 and 5 of every MCU: a third of the coefficients): their 16-byte pieces are
        // neither fetched nor staged (7680 x 4320: 33 MB of 199 MB less traffic per frame)
        bool wanted[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) wanted[k] = !GRAY || (unsigned)((k * 8 + (lane >> 3)) % 6) < 4u;
        if (valid_mcus == 4) {                       // wave-uniform; every quad but the last of a ragged row
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                v[k] = make_uint4(0, 0, 0, 0);
                if (wanted[k]) v[k] = g4[k * 64 + lane];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                v[k] = make_uint4(0, 0, 0, 0);
                if (wanted[k] && (k * 64 + lane) * 16 < valid_bytes) v[k] = g4[k * 64 + lane];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int c = k * 64 + lane;              // 16-byte chunk: block c >> 3, part c & 7
            if (wanted[k]) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(lds) + (c >> 3) * DSTG_PITCH + (c & 7) * 16) = v[k];
        }
    }
    wave_sync();

    // ---- 2. column pass (over v) for column u = cq&7 of two luma blocks and one chroma block.
    //         The DC term is dequantised in the reference's own order, ((cu*cv) * (coef*Q)), so that a block whose
    //         AC coefficients are all zero (flat / DC-only blocks: the common case in real images, where EVERY
    //         sample sits exactly on an integer) comes out of the fast path bit-identical to the reference: its only
    //         non-zero term passes through the butterflies and the transposes unchanged.  Such blocks are found
    //         with three ballots and their samples are exempt from the guard band below. ----
    const int16_t* stage = reinterpret_cast<const int16_t*>(lds);
    double gtop[TOL ? 1 : 8], gbot[TOL ? 1 : 8];
    float ftop[TOL ? 8 : 1], fbot[TOL ? 8 : 1];
    unsigned cpk[4] = { 0, 0, 0, 0 };                // the chroma column's raw coefficients, two per word
    int cmx = 0, cmn = 0;               // largest / smallest raw coefficient this lane touches
    unsigned long long ac_top = 0, ac_bot = 0, ac_chr = 0;   // lanes whose block column holds a non-zero AC coefficient
    {
        const double cucv_dc = JPEZY_S * JPEZY_S;                  // the reference's cu * cv for (0,0): 0.4999999999999999
        const int bx = cq >> 3;
        const int16_t* bt = stage + (m * BPM + bx) * (DSTG_PITCH / 2);
        const int16_t* bb = stage + (m * BPM + 2 + bx) * (DSTG_PITCH / 2);
        int c[8], acor;
        if constexpr (TOL) {
            float in[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) { c[v] = ld_coef(bt + zp[v]); in[v] = (float)c[v] * dqf[v]; cmx = max(cmx, c[v]); cmn = min(cmn, c[v]); }
            idct8f(in, ftop);
#pragma unroll
            for (int v = 0; v < 8; ++v) { c[v] = ld_coef(bb + zp[v]); in[v] = (float)c[v] * dqf[v]; cmx = max(cmx, c[v]); cmn = min(cmn, c[v]); }
            idct8f(in, fbot);
        } else {
            double in[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) { c[v] = ld_coef(bt + zp[v]); in[v] = (double)c[v] * dq[v]; cmx = max(cmx, c[v]); cmn = min(cmn, c[v]); }
            if (u == 0) in[0] = cucv_dc * (double)(c[0] * p.dqt[0]) * 0.25;
            acor = (u ? c[0] : 0) | c[1] | c[2] | c[3] | c[4] | c[5] | c[6] | c[7];
            ac_top = __ballot(acor != 0);
            idct8(in, gtop);
#pragma unroll
            for (int v = 0; v < 8; ++v) { c[v] = ld_coef(bb + zp[v]); in[v] = (double)c[v] * dq[v]; cmx = max(cmx, c[v]); cmn = min(cmn, c[v]); }
            if (u == 0) in[0] = cucv_dc * (double)(c[0] * p.dqt[0]) * 0.25;
            acor = (u ? c[0] : 0) | c[1] | c[2] | c[3] | c[4] | c[5] | c[6] | c[7];
            ac_bot = __ballot(acor != 0);
            idct8(in, gbot);
        }
        if (!GRAY) {
            const int comp = 1 + (cq >> 3);
            const int16_t* bc = stage + (m * BPM + 3 + comp) * (DSTG_PITCH / 2);
            // only the eight raw coefficients of the chroma column stay in registers (4 VGPRs); its column pass runs after
            // the luma halves, when the 32 VGPRs of the two luma columns are free again
#pragma unroll
            for (int v = 0; v < 8; ++v) { c[v] = ld_coef(bc + zp[v]); cmx = max(cmx, c[v]); cmn = min(cmn, c[v]); }
            acor = (u ? c[0] : 0) | c[1] | c[2] | c[3] | c[4] | c[5] | c[6] | c[7];
            ac_chr = __ballot(acor != 0);
#pragma unroll
            for (int v = 0; v < 4; ++v) cpk[v] = ((unsigned)c[2 * v] & 0xFFFFu) | ((unsigned)c[2 * v + 1] << 16);
        }
    }
    // The block columns of block (m, side) sit in lanes (8*side + k)*4 + m, k = 0..7: bits 0x11111111 << m of the low
    // (side 0: left luma blocks / Cb) or high (side 1: right luma blocks / Cr) half of a ballot.
    const unsigned colbits = 0x11111111u << m;
    const unsigned long long ac_luma = (row >> 3) ? ac_bot : ac_top;                 // this lane's pixel row: top or bottom blocks
    const bool dc_only_l = ((unsigned)ac_luma & colbits) == 0, dc_only_r = ((unsigned)(ac_luma >> 32) & colbits) == 0;
    const bool dc_only_cb = ((unsigned)ac_chr & colbits) == 0, dc_only_cr = ((unsigned)(ac_chr >> 32) & colbits) == 0;
    // fast path is only trusted for sane magnitudes (jpezy_capi.hip upload_dequant): |coef| <= coef_limit keeps every
    // dequantised input below 2^21 (exact mode) / 2^13 (tolerance mode); wave-uniform decision
    // RANGE false (the launcher: coef_limit >= 32768, i.e. every 8-bit quantiser table): no int16 coefficient leaves the trusted
    // range and the tracking above is dead code
    const bool force = FORCE_EXACT || (RANGE && wave_any(max(cmx, -cmn) > p.coef_limit));
    wave_sync();   // staging consumed (the exact path re-reads coefficients from global memory)

    // ---- 3. transpose in two halves: the lanes holding the LEFT block columns (cq < 8) publish them, every lane
    //         runs the row pass of its 8 left pixels; then the same for the right blocks ----
    int Y[16];
    unsigned yflags = 0;
#ifdef JPEZY_DEC_INTERLEAVE
    double gc_early[8];
#endif
    if constexpr (TOL) {
        // the whole luma tile as floats (pitch 16, MCU stride 264 dwords: the column stores are conflict-free, the 16-byte row
        // reads 2-way -- tools/profile/lds_bank_model.py); one exchange instead of two, no guard keys
        float* ldsf = reinterpret_cast<float*>(lds);
        {
            float* dst = ldsf + m * TF_MCU + cq;
#pragma unroll
            for (int y = 0; y < 8; ++y) {
                dst[y * TF_PITCH] = ftop[y];
                dst[(8 + y) * TF_PITCH] = fbot[y];
            }
        }
        wave_sync();
        {
            float in[16], out[8];
            const float4* src = reinterpret_cast<const float4*>(ldsf + m * TF_MCU + row * TF_PITCH);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float4 t = src[k]; in[4 * k] = t.x; in[4 * k + 1] = t.y; in[4 * k + 2] = t.z; in[4 * k + 3] = t.w; }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                in[8 * half] += 128.f;
                idct8f(in + 8 * half, out);
#pragma unroll
                for (int k = 0; k < 8; ++k) Y[half * 8 + k] = (int)out[k];        // v_cvt_i32_f32 truncates like the reference's int()
            }
        }
        wave_sync();   // the tile was read; the chroma tile is written next
    } else
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if ((cq >> 3) == half) {
            uint32_t* dst = lds + m * DH_MCU + u * 2;
#pragma unroll
            for (int y = 0; y < 8; ++y) {
                *reinterpret_cast<double*>(dst + y * DH_PITCH) = gtop[y];
                *reinterpret_cast<double*>(dst + (8 + y) * DH_PITCH) = gbot[y];
            }
        }
        wave_sync();
#ifdef JPEZY_DEC_INTERLEAVE
        if (!GRAY && half == 0) {
            const int comp = 1 + (cq >> 3);
            double cin[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const int cv_ = (int)(short)(cpk[v >> 1] >> ((v & 1) * 16));
                cin[v] = (double)cv_ * p.dqscale[(comp * 8 + u) * 8 + v];
            }
            if (u == 0) cin[0] = (JPEZY_S * JPEZY_S) * (double)((int)(short)(cpk[0] & 0xFFFFu) * p.dqt[comp * 64]) * 0.25;
            idct8(cin, gc_early);
        }
#endif
        {
            double in[8], out[8];
            const double2* src = reinterpret_cast<const double2*>(lds + m * DH_MCU + row * DH_PITCH);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const double2 t = src[k]; in[2 * k] = t.x; in[2 * k + 1] = t.y; }
            in[0] += 128.0;          // level shift, see sample_of
            idct8(in, out);
            uint32_t e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) Y[half * 8 + k] = sample_of(out[k], e[k]);
            yflags |= guard_bits8(e, half ? dc_only_r : dc_only_l) << (half * 8);
        }
        wave_sync();   // the half tile was read; it is rewritten next
    }
    if (force) yflags = 0xFFFFu;
    int Cb[8], Cr[8];
    unsigned cflags = 0;
    if (!GRAY) {
        {
            double gc[8];
#ifdef JPEZY_DEC_INTERLEAVE
            if (!TOL) {
#pragma unroll
                for (int y = 0; y < 8; ++y) gc[y] = gc_early[y];
            } else
#endif
            {
                const int comp = 1 + (cq >> 3);
                double cin[8];
#pragma unroll
                for (int v = 0; v < 8; ++v) {
                    const int cv_ = (int)(short)(cpk[v >> 1] >> ((v & 1) * 16));
                    cin[v] = (double)cv_ * p.dqscale[(comp * 8 + u) * 8 + v];
                }
                if (u == 0) cin[0] = (JPEZY_S * JPEZY_S) * (double)((int)(short)(cpk[0] & 0xFFFFu) * p.dqt[comp * 64]) * 0.25;
                idct8(cin, gc);
            }
            uint32_t* dst = lds + m * DC_MCU + (cq >> 3) * DC_COMP + u * 2;
#pragma unroll
            for (int y = 0; y < 8; ++y) *reinterpret_cast<double*>(dst + y * DC_PITCH) = gc[y];
        }
        wave_sync();
        // A chroma row serves two pixel rows, i.e. two lanes (l and l ^ 4).  The even-row lane runs the row pass of the Cb
        // row, the odd-row lane that of the Cr row, and they swap the eight samples and the guard bits (DPP row_shr/shl:4)
        // -- instead of both lanes computing both rows.
        const bool odd = (row & 1) != 0;
        double in[8], out[8];
        const double2* src = reinterpret_cast<const double2*>(lds + m * DC_MCU + (odd ? DC_COMP : 0) + (row >> 1) * DC_PITCH);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double2 a = src[k];
            in[2 * k] = a.x; in[2 * k + 1] = a.y;
        }
        in[0] += 128.0;
        idct8(in, out);
        {
            uint32_t e[8];
            int mine[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) mine[k] = sample_of(out[k], e[k]) - 128;     // the chroma samples travel as u' = Cb - 128 /
            const unsigned myf = guard_bits8(e, odd ? dc_only_cr : dc_only_cb);         // v' = Cr - 128 (exact integers: ref :567-568)
            // DPP inside each 16-lane row: banks 0,2 hold even pixel rows (Cb), banks 1,3 odd ones (Cr)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                Cb[k] = __builtin_amdgcn_update_dpp(mine[k], mine[k], 0x114, 0xF, 0xA, false);   // odd lanes take lane - 4's
                Cr[k] = __builtin_amdgcn_update_dpp(mine[k], mine[k], 0x104, 0xF, 0x5, false);   // even lanes take lane + 4's
            }
            const unsigned fcb = (unsigned)__builtin_amdgcn_update_dpp((int)myf, (int)myf, 0x114, 0xF, 0xA, false);
            const unsigned fcr = (unsigned)__builtin_amdgcn_update_dpp((int)myf, (int)myf, 0x104, 0xF, 0x5, false);
            cflags = fcb | (fcr << 8);
        }
        if (force) cflags = 0xFFFFu;
    }
    if (!live) { yflags = 0; cflags = 0; }

    // ---- 4. exact path for flagged samples.  Lanes OR the blocks they need into a mask; for each such block the
    //         wave builds the list of its non-zero dequantised coefficients (in the reference's summation order),
    //         then every lane re-evaluates its own flagged samples of that block in exact order and leaves them in
    //         a patch table [lane][16] it reads back.  Two rounds -- luma blocks, then chroma blocks -- share the
    //         table.  Cost grows with the number of non-zero coefficients: DC-only blocks (every sample sits on an
    //         integer) cost one term per sample. ----
    {
        unsigned* need_blocks = lds + DEC_TILE_DWORDS;             // zeroed at kernel start
        int* patch = reinterpret_cast<int*>(lds);                  // tiles are dead by now: 64 x 16 x 4 B = 4 KB
        int2* list = reinterpret_cast<int2*>(lds + 1024);          // 64 entries x 8 B behind the patch table
        const unsigned myflags = force ? (GRAY ? 0xFFFFu : 0xFFFFFFFFu) : (yflags | (cflags << 16));
        const int by = row >> 3;
        unsigned myblocks = 0;
        if (live) {
            if (myflags & 0x000000FFu) myblocks |= 1u << (m * BPM + by * 2);
            if (myflags & 0x0000FF00u) myblocks |= 1u << (m * BPM + by * 2 + 1);
            if (myflags & 0x00FF0000u) myblocks |= 1u << (m * BPM + 4);
            if (myflags & 0xFF000000u) myblocks |= 1u << (m * BPM + 5);
        }
        // common case: no lane of the wave has a flagged sample -- one vote, no LDS traffic
        unsigned all_todo = 0;
        if (wave_any(myblocks != 0)) {
            wave_sync();                                           // all tile reads are done
            if (myblocks) atomicOr(need_blocks, myblocks);
            wave_sync();
            all_todo = __builtin_amdgcn_readfirstlane((int)need_blocks[0]);
        }
        if (all_todo) {
            constexpr unsigned LUMA_BLOCKS = 0x0F | (0x0F << 6) | (0x0F << 12) | (0x0F << 18);   // blocks 0-3 of the 4 MCUs
            unsigned done = 0;
#pragma unroll 1
            for (int round = 0; round < 2; ++round) {
                unsigned todo = all_todo & (round ? ~LUMA_BLOCKS : LUMA_BLOCKS);
                if (!todo) continue;
#pragma unroll 1
                while (todo) {
                    const int blk = __builtin_ctz(todo);
                    todo &= todo - 1;
                    const int em = blk / BPM, eb = blk - em * BPM;
                    const int comp = eb < 4 ? 0 : eb - 3;
                    // build the non-zero list: lane k looks at natural index k
                    const int16_t* gblk = gbase + (size_t)blk * 64;
                    const int d = (int)gblk[c_zzinv[lane]] * p.dqt[comp * 64 + lane];      // inverse_quantization :645-650
                    const unsigned long long nzmask = __ballot(d != 0);
                    const int K = __builtin_popcountll(nzmask);
                    if (d != 0) list[__builtin_popcountll(nzmask & ((1ull << lane) - 1ull))] = make_int2(lane, d);
                    wave_sync();
                    if (m == em && ((myblocks >> blk) & 1u)) {
                        // my flagged samples inside this block: 8 consecutive k (bits of myflags), 8 consecutive patch slots
                        const int k0 = eb < 4 ? (eb & 1) * 8 : 16 + (eb - 4) * 8;
                        const int yy = eb < 4 ? (row & 7) : (row >> 1);
#pragma unroll 1
                        for (int k = k0; k < k0 + 8; ++k) {
                            if ((myflags >> k) & 1u) {
                                patch[lane * 16 + (k & 15)] = exact_idct_sample_sparse(list, K, k & 7, yy);
                                ++done;
                            }
                        }
                    }
                    wave_sync();
                }
                if (myflags && live) {
                    if (round == 0) {
#pragma unroll
                        for (int k = 0; k < 16; ++k)
                            if ((myflags >> k) & 1u) Y[k] = patch[lane * 16 + k];
                    } else if (!GRAY) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            if ((myflags >> (16 + k)) & 1u) Cb[k] = patch[lane * 16 + k] - 128;
                            if ((myflags >> (24 + k)) & 1u) Cr[k] = patch[lane * 16 + 8 + k] - 128;
                        }
                    }
                }
                wave_sync();
            }
            if (done) atomicAdd(p.fallback_count + (qrem & (COUNTER_SHARDS - 1)), (unsigned long long)done);
        }
    }

    // ---- 5. YCbCr -> RGB (ref :567-578), clamp, pack, store ----
    // Y, U = Cb - 128, V = Cr - 128 are integers, so each of r = Y + 1.402 V, g = Y - 0.3441 U - 0.7139 V, b = Y + 1.7718 U is Y plus a
    // term of the chroma sample alone, and revise_value(trunc(Y + t)) == clamp(Y + floor(t)) unless t is an integer whose double
    // evaluation may fall on either side (negative values clamp to 0 under both roundings).  1.402 V is a multiple of 0.002 and
    // 1.7718 U of 0.0002: inside the gate |U|, |V| <= 512 they are integral at zero, where the products are exact, and at
    // V = +-500 (1.402 x 500 = 701), where BOTH the FP32 product 1.402f x 500 and the reference's double product 500 x 1.4020
    // round to exactly 701.0 (asserted in tests/test_colour_offsets.py: a change of constant or gate that breaks either equality
    // fails there), so floor and trunc agree on them; 1.7718 U is integral only for multiples of 5000.  c = 0.3441 U + 0.7139 V
    // is a multiple of 0.0001: integral for one pair in 10,000.  Round 4: three integer offsets per chroma sample from FP32 arithmetic (floor-
    // converting v_cvt_flr_i32_f32: the FP32 error is below 3.4e-5 inside |U|, |V| <= 512, a third of the lattice spacing), three
    // integer additions per pixel; a chroma sample whose c is within 5e-5 of a NON-ZERO integer is one of those pairs (U = V = 0,
    // every gray pixel, has c = 0 exactly and the reference subtracts two zeros), and a wave that holds one -- or a chroma sample
    // outside the gate, or forced samples (their range is not bounded) -- converts in doubles, the reference's own sequence, as
    // every wave did before.  tests/test_colour_offsets.py enumerates every pair inside the gate: offsets equal to the integer
    // floors, flags equal to the integral pairs, results equal to the reference's formula.  (The conversion was 236 of the kernel's
    // 767 vector instructions, all of the slow class: 34.2 -> see DESIGN.md 4.2.)
    uint32_t Rw[4], Gw[4], Bw[4];
    bool int_colour = false;
#if JPEZY_DEC_INT_COLOUR
    if (!GRAY) {
        int oR[8], oG[8], oB[8];
        float kmin = 2.f, amax = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float uf = (float)Cb[c], vf = (float)Cr[c];
            const float cc = FMAF(0.3441f, uf, 0.7139f * vf);
            oR[c] = cvt_floor(1.402f * vf);
            oB[c] = cvt_floor(1.7718f * uf);
            oG[c] = cvt_floor_neg(cc);
            const float n = __builtin_rintf(cc);                                           // v_rndne_f32
            kmin = __builtin_fminf(kmin, __builtin_fmaxf(__builtin_fabsf(cc - n), 1.f - __builtin_fabsf(n)));
            amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(uf)), __builtin_fabsf(vf));
        }
        int_colour = !force && !wave_any(kmin <= DEC_CHROMA_BAND || !(amax <= DEC_CHROMA_GATE));
        if (int_colour) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int ri[4], gi[4], bi[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = 2 * q + (k >> 1);
                    ri[k] = Y[4 * q + k] + oR[c];
                    gi[k] = Y[4 * q + k] + oG[c];
                    bi[k] = Y[4 * q + k] + oB[c];
                }
                Rw[q] = clamp_pack4(ri);
                Gw[q] = clamp_pack4(gi);
                Bw[q] = clamp_pack4(bi);
            }
        }
    }
#endif
    if (GRAY) {
#pragma unroll
        for (int q = 0; q < 4; ++q) Rw[q] = Gw[q] = Bw[q] = clamp_pack4(Y + 4 * q);
    } else if (!int_colour) {              // the reference's exact order in doubles
#pragma unroll
        for (int q = 0; q < 4; ++q) {              // four pixels = two chroma samples -> one word of each plane
            int ri[4], gi[4], bi[4];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int c = 2 * q + cc;
                const double up = (double)Cb[c], vp = (double)Cr[c];       // (sample - 128, subtracted as integers above)
                const double pr_ = vp * 1.4020, pg1 = up * 0.3441, pg2 = vp * 0.7139, pb_ = up * 1.7718;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const double yp = (double)Y[2 * c + h];
                    ri[2 * cc + h] = (int)(yp + pr_);            // truncates; revise_value clamps below (ref :672-676)
                    gi[2 * cc + h] = (int)(yp - pg1 - pg2);
                    bi[2 * cc + h] = (int)(yp + pb_);
                }
            }
            Rw[q] = clamp_pack4(ri);
            Gw[q] = clamp_pack4(gi);
            Bw[q] = clamp_pack4(bi);
        }
    }
#if JPEZY_DEC_FULLLINE
    // The two waves of a workgroup hold the two 64-byte halves of every 128-byte line of their 16 pixel rows.  They swap through
    // LDS so that wave 0 stores rows 0..7 and wave 1 rows 8..15 of BOTH quads: eight lanes = one whole line.  The launch takes the
    // same time either way (34.2 us per 4096^2 frame, three interleaved rounds), but half-line non-temporal stores are counted --
    // and moved -- as 63.8 MB of writes for 50.3 MB of planes; whole lines bring WRITE_SIZE to 49.2 MB (profiles/r04_dec_traffic.txt).
    if (ALIGNED && !GRAY && WPB == 2) {
        const unsigned q0 = blockIdx.x * 2u;                                         // the workgroup's first quad
        const int qx0 = (int)q0 - (int)fast_div(q0, p.qpr_magic, p.qpr_shift) * p.quads_per_row;
        const bool pair_ok = (p.quads_per_row & 1) == 0 && (qx0 + 2) * 64 <= W && q0 + 1u < (unsigned)(p.mcu_rows * p.quads_per_row);   // workgroup-uniform
        if (pair_ok) {
            wave_sync();                                                             // (the slice's earlier uses are over)
            uint4* slot = reinterpret_cast<uint4*>(reinterpret_cast<char*>(lds_all[wave]) + lane * 48);
            slot[0] = make_uint4(Rw[0], Rw[1], Rw[2], Rw[3]);
            slot[1] = make_uint4(Gw[0], Gw[1], Gw[2], Gw[3]);
            slot[2] = make_uint4(Bw[0], Bw[1], Bw[2], Bw[3]);
            __syncthreads();
            const int r8 = wave * 8 + (lane >> 3), seg = lane & 7;
            const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(lds_all[seg >> 2]) + ((r8 << 2) | (seg & 3)) * 48);
            const uint4 vR = src[0], vG = src[1], vB = src[2];
            const int pyy = mcu_y * 16 + r8;
            if (pyy < H) {
                const size_t off = (size_t)frame * p.plane_stride + (size_t)pyy * W + (size_t)qx0 * 64 + (size_t)seg * 16;
                nt_store16(reinterpret_cast<uint4*>(p.r + off), vR);
                nt_store16(reinterpret_cast<uint4*>(p.g + off), vG);
                nt_store16(reinterpret_cast<uint4*>(p.b + off), vB);
            }
            return;
        }
    }
#endif
    const int py = mcu_y * 16 + row;
    if (live && py < H) {
        uint8_t* orp = p.r + (size_t)frame * p.plane_stride + (size_t)py * W;
        uint8_t* ogp = p.g + (size_t)frame * p.plane_stride + (size_t)py * W;
        uint8_t* obp = p.b + (size_t)frame * p.plane_stride + (size_t)py * W;
        if (ALIGNED) {
            const size_t off = (size_t)mcu_x * 16;
            // Colour: non-temporal -- the planes are never re-read by the kernel, and 50 MB of dirty lines are not left in the eight
            // L2s for the end-of-kernel write-back.  A quad covers 64 bytes of a pixel row, so the stores go out as half lines
            // (WRITE_SIZE counts +30 % requests) -- and the launch is still 1.5-2.4 us shorter (profiles/r03a_ab_decode.txt:
            // 37.4 -> 35.1-35.9 us; round 2 judged this by the request count alone and kept plain stores).  Gray (three planes of
            // the same bytes, twice the pixels per coefficient read): plain stores measure better, 46.9 against 50.5 us at 8K.
            if (!GRAY) {
                nt_store16(reinterpret_cast<uint4*>(orp + off), make_uint4(Rw[0], Rw[1], Rw[2], Rw[3]));
                nt_store16(reinterpret_cast<uint4*>(ogp + off), make_uint4(Gw[0], Gw[1], Gw[2], Gw[3]));
                nt_store16(reinterpret_cast<uint4*>(obp + off), make_uint4(Bw[0], Bw[1], Bw[2], Bw[3]));
            } else {
                *reinterpret_cast<uint4*>(orp + off) = make_uint4(Rw[0], Rw[1], Rw[2], Rw[3]);
                *reinterpret_cast<uint4*>(ogp + off) = make_uint4(Gw[0], Gw[1], Gw[2], Gw[3]);
                *reinterpret_cast<uint4*>(obp + off) = make_uint4(Bw[0], Bw[1], Bw[2], Bw[3]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int x = mcu_x * 16 + k;
                if (x < W) {                                   // ref :546-551
                    orp[x] = (uint8_t)(Rw[k >> 2] >> ((k & 3) * 8));
                    ogp[x] = (uint8_t)(Gw[k >> 2] >> ((k & 3) * 8));
                    obp[x] = (uint8_t)(Bw[k >> 2] >> ((k & 3) * 8));
                }
            }
        }
    }
}

// ======================================================================================================
// launchers
// ======================================================================================================
template <typename P>
static bool is_aligned16(const P& p, const void* a, const void* b, const void* c)
{
    return (p.W % 16 == 0) && (p.plane_stride % 16 == 0) && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) % 16 == 0);
}

template <bool GRAY, bool ALIGNED>
static void enc_launch2(const EncParams& p, bool force, dim3 grid, hipStream_t s)
{
    if (force)
        hipLaunchKernelGGL((fdct_quant_kernel<GRAY, ALIGNED, true>), grid, dim3(64 * WPB), 0, s, p);
    else
        hipLaunchKernelGGL((fdct_quant_kernel<GRAY, ALIGNED, false>), grid, dim3(64 * WPB), 0, s, p);
}

hipError_t launch_fdct_quant(const EncParams& p, bool gray, bool force_exact, hipStream_t stream)
{
    const long quads = (long)p.n_frames * p.mcu_rows * p.quads_per_row;
    if (quads <= 0) return hipSuccess;
    const dim3 grid((unsigned)((quads + WPB - 1) / WPB));
    const bool al = is_aligned16(p, p.r, p.g, p.b);
    if (gray) { if (al) enc_launch2<true, true>(p, force_exact, grid, stream); else enc_launch2<true, false>(p, force_exact, grid, stream); }
    else      { if (al) enc_launch2<false, true>(p, force_exact, grid, stream); else enc_launch2<false, false>(p, force_exact, grid, stream); }
    return hipGetLastError();
}

template <bool GRAY, bool ALIGNED>
static void dec_launch2(const DecParams& p, bool force, bool tol, dim3 grid, hipStream_t s)
{
    if (force)           // every sample through the reference-order path: the tolerance switch has nothing left to relax
        hipLaunchKernelGGL((dequant_idct_kernel<GRAY, ALIGNED, true, 0>), grid, dim3(64 * WPB), 0, s, p);
    else if (tol)
        hipLaunchKernelGGL((dequant_idct_kernel<GRAY, ALIGNED, false, 2>), grid, dim3(64 * WPB), 0, s, p);
    else if (p.coef_limit >= 32768)      // no int16 coefficient can exceed it: the kernel without the range test
        hipLaunchKernelGGL((dequant_idct_kernel<GRAY, ALIGNED, false, 1>), grid, dim3(64 * WPB), 0, s, p);
    else
        hipLaunchKernelGGL((dequant_idct_kernel<GRAY, ALIGNED, false, 0>), grid, dim3(64 * WPB), 0, s, p);
}

hipError_t launch_dequant_idct(const DecParams& p0, bool gray, bool force_exact, bool tolerant, hipStream_t stream)
{
    const long quads = (long)p0.mcu_rows * p0.quads_per_row;
    if (quads <= 0 || p0.n_frames <= 0) return hipSuccess;
    if (p0.n_frames > 65535) return hipErrorInvalidValue;              // grid.y limit; callers chunk larger batches
    DecParams p = p0;
    fast_div_setup((unsigned)p.quads_per_row, &p.qpr_magic, &p.qpr_shift);
    const dim3 grid((unsigned)((quads + WPB - 1) / WPB), (unsigned)p.n_frames);
    const bool al = is_aligned16(p, p.r, p.g, p.b);
    if (gray) { if (al) dec_launch2<true, true>(p, force_exact, tolerant, grid, stream); else dec_launch2<true, false>(p, force_exact, tolerant, grid, stream); }
    else      { if (al) dec_launch2<false, true>(p, force_exact, tolerant, grid, stream); else dec_launch2<false, false>(p, force_exact, tolerant, grid, stream); }
    return hipGetLastError();
}

}  // namespace jpezy_dev
