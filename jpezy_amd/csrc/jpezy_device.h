// jpezy_device.h -- shared declarations of the gfx950 kernels and their launchers (internal).
#pragma once
#include "jpezy_experiment.h"
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace jpezy_dev {

// Fixed-point scale of the quantiser guard band: a coefficient is v/Q * 2^QFRAC_BITS truncated to int32.
// |v/Q| <= 1024/10 with the Annex-K tables, so 2^24 keeps |n| < 2^31.
constexpr int QFRAC_BITS = 24;

// Per (table, block column j) record of the f32 encode kernel: one 64-byte line per lane, three loads off one address.
// The kernel's packed 8-point transform delivers its outputs as the pairs (0,4) (2,6) (1,3) (5,7): everything indexed by
// the coefficient row is stored in that order, position p <-> row kPairRow[p].
constexpr int kPairRow[8] = { 0, 4, 2, 6, 1, 3, 5, 7 };
struct F32Column {
    float ks[8];              // ks[p] = cu(j) * cv(i) / (4 * Q_t[i*8+j]) * cos(pi/4)^[i == 4] * cos(pi/4)^[j == 4], i = kPairRow[p]
    // level-1 guard band: 1.25 x max over i of the worst-case FP32 error of t[i][j] = F[i][j] * ks
    // (jpezy_capi.hip; tests/test_f32_error_bound.py re-derives it); twice, as the addend pair of a v_pk_fma_f32
    float delta1[2];
    float th;                 // 2 * delta1: the kernel's test is fract(F * ks + delta1) < th
    // byte p of zz_lo (p < 4) / zz_hi (p - 4) = 2 * (zig-zag position of natural coefficient (kPairRow[p], j)): the byte
    // offsets of one block column inside a staged block, packed so the kernel spends two registers on them instead of eight
    uint32_t zz_lo, zz_hi;
    uint32_t pad[3];
};
static_assert(sizeof(F32Column) == 64, "one record per 64-byte line");

// Device-resident tables built by the host at context creation (jpezy_capi.hip).
struct DeviceTables {
    F32Column f32col[2][8];   // first: every field at an immediate offset from the table pointer
    // quantised DC as a function of the block's integer sample sum S in [-8192, 8192] (index S + 8192):
    // dcq[t][.] = int(((S * s) * s) / 4) / Q_t[0] with s = 1/sqrt(2), evaluated on the host in the reference's
    // exact FP64 order (ref encoder/jpezy_encoder.hpp:163,171)
    signed char dcq[2][16385];
    // encode variant 0: qscale[t][j][i] = cu(j) * cv(i) / (4 * Q_t[i*8+j]) * 2^QFRAC_BITS   (t: 0 luma, 1 chroma)
    double qscale[2][8][8];
    double rq_dc[2];          // 1 / Q_t[0]
    int qt[2][64];            // natural order
    double qinv[2][64];       // 1.0 / Q_t[k] (levels 2/3 of the f32 kernel)
};

// The exact-path counter is sharded over COUNTER_SHARDS words: thousands of waves adding to ONE word serialise
// at ~12 ns per atomic (88 per us, MI355X_MICROARCH.md 'dequeue') and held every wave slot until they drained.
constexpr int COUNTER_SHARDS = 1024;

// Waves per workgroup.  Waves never talk to each other (no s_barrier); the only effect of the grouping is which
// quads share a CU.  Measured on 4096^2 / 32x1080p encode: 2 waves (two adjacent quads = one 128-byte line of every
// pixel row) is ~4 % faster than 1 or 4.
#ifndef JPEZY_WPB
#define JPEZY_WPB 2
#endif
constexpr int WPB = JPEZY_WPB;

struct EncParams {
    const uint8_t* r;
    const uint8_t* g;
    const uint8_t* b;
    size_t plane_stride;      // bytes between frames of one plane
    int16_t* coeffs;
    size_t coeffs_per_frame;  // int16 elements
    const DeviceTables* tab;
    const signed char* dcq_luma;     // &tab->dcq[0][0], &tab->dcq[1][0]: separate kernel arguments so that the f32 kernel's
    const signed char* dcq_chroma;   // DC lookups are scalar-base + 32-bit-offset loads
    unsigned long long* fallback_count;
    int W, H, mcu_cols, mcu_rows, quads_per_row, n_frames;
    unsigned qpr_magic, qpr_shift;   // fast_div by quads_per_row
    int groups_per_row;              // f32 kernel: workgroups per MCU row, and fast_div by it (set by its launcher)
    unsigned gpr_magic, gpr_shift;
    // persistent f32 kernel (variant 2): groups of four quads over all frames of the launch, fast_div by the groups of one frame
    unsigned ps_total_groups, ps_groups_per_frame, gpf_magic, gpf_shift;
    // variant 3: quads over all frames of the launch, fast_div by the quads of one frame
    unsigned ps_total_quads, ps_quads_per_frame, qpf_magic, qpf_shift;
    // quantised DC without the table (f32::dc_formula): fl(1 / Q_t[0]) and 1 / (2 Q_t[0]); 0: the formula did not reproduce DeviceTables::dcq
    // for these constants (checked for every sum at context creation) and must not be used
    float dc_rq[2], dc_bias[2];
#ifdef JPEZY_TRACE
    unsigned long long* trace;       // development builds only (tools/profile/wave_trace.py): 4 words per wave
#endif
#ifdef JPEZY_DUMP_T
    float* dump_t;                   // development builds only (tools/measure/check_level1_bound.py): the f32 kernel's level-1
                                     // t = F * ks of every coefficient, coefficient-buffer layout, NATURAL order in a block
#endif
};

struct DecParams {
    const int16_t* coeffs;
    size_t coeffs_per_frame;
    uint8_t* r;
    uint8_t* g;
    uint8_t* b;
    size_t plane_stride;
    const double* dqscale;    // [3 comps][8 (u=lane col)][8 (v)] : cu*cv*Q[v*8+u] / 4   (dequant and the final / 4 folded in)
    const float* dqscale_f;   // [8 (u)][8 (v)]: the luma constants rounded to FP32 (tolerance mode)
    const int* dqt;           // [3 comps][64] natural order quant values (exact path)
    int coef_limit;           // raw coefficients above it send the wave to the exact path: 2^23 / largest quantiser (exact mode;
                              // >= 32768 = no test needed) or 2^15 / largest quantiser (tolerance mode)
    unsigned long long* fallback_count;
    int W, H, mcu_cols, mcu_rows, quads_per_row, n_frames;
    unsigned qpr_magic, qpr_shift;   // fast_div by quads_per_row (set by the launcher)
};

// (magic, shift) such that n / d == (((n - mulhi(n, magic)) >> 1) + mulhi(n, magic)) >> shift for all 32-bit n;
// magic == 0 encodes d == 1
inline void fast_div_setup(unsigned d, unsigned* magic, unsigned* shift)
{
    unsigned l = 0;
    while ((1ull << l) < d) ++l;                       // ceil(log2 d)
    *magic = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    *shift = l ? l - 1 : 0;
    if (d == 1) { *magic = 0; *shift = 0; }
}

hipError_t launch_fdct_quant(const EncParams& p, bool gray, bool force_exact, hipStream_t stream);
// variant 1: FP32 first level, FP64 second level, reference-order third level.  force: 0 normal, 1 every
// coefficient through the reference-order chain, 2 every coefficient through the FP64 second level, 3 every quad
// through the per-lane evaluator of the queue-overflow case.
hipError_t launch_fdct_quant_f32(const EncParams& p, bool gray, int force, hipStream_t stream);
// variant 2: the same arithmetic in persistent workgroups with LDS-DMA loader waves (jpezy_kernels_f32_ps.hip); frames whose rows do
// not divide into groups of four quads, or unaligned planes, go to variant 1's launch.  n_cus: compute units of the device.
hipError_t launch_fdct_quant_f32_ps(const EncParams& p, bool gray, int force, int n_cus, hipStream_t stream);
bool fdct_quant_f32_ps_applies(const EncParams& p);
// variant 3: persistent workgroups of 16 compute waves, every wave prefetching its next quad's pixels into registers; any frame
// with W % 16 == 0 and 16-byte aligned planes, anything else goes to variant 1's launch
hipError_t launch_fdct_quant_f32_ps2(const EncParams& p, bool gray, int force, int n_cus, hipStream_t stream);
// tolerant: luma in FP32 without guard band (samples within one of the reference's, jpezy_kernels.hip); chroma stays exact
hipError_t launch_dequant_idct(const DecParams& p, bool gray, bool force_exact, bool tolerant, hipStream_t stream);

// any-layout decode (jpezy_kernels_generic.hip)
struct GenericDecParams {
    const int16_t* coeffs;
    int* samples;             // [block][64] natural order, scratch
    const int* qt;            // [3 comps][64] natural order
    uint8_t* r;
    uint8_t* g;
    uint8_t* b;
    int W, H, ncomp, gray;
    int ch[3], cv[3], hmax, vmax, mcu_cols, mcu_rows, blocks_per_mcu;
    int blk_start[3];         // first block of each component inside an MCU
    int level;                // level shift of inverse_dct: 128, or 2048 when SOF0 says precision != 8 (ref :654)
    unsigned mw_magic, mw_shift, mh_magic, mh_shift;        // fast_div by hmax*8, vmax*8 (set by the launcher)
    unsigned dx_magic[3], dx_shift[3], dy_magic[3], dy_shift[3];   // fast_div by hmax/H, vmax/V of each component
    const double* dqscale;    // [3 comps][8 (u)][8 (v)] cu*cv*Q/4 as for the fused kernel (fast path)
    int coef_limit;           // raw coefficients above it send their block to the reference-order path
    int force_exact;          // test hook: every block through the reference-order path
    unsigned long long* fallback_count;   // samples evaluated in reference order (sharded, COUNTER_SHARDS)
    // batch form: n_frames frames of ONE layout, size and set of quantiser tables; frame f's coefficients at coeffs + f * blocks * 64,
    // its samples at samples + f * blocks * 64, its planes at r/g/b + f * plane_stride (a multiple of 4)
    int n_frames = 1;
    size_t plane_stride = 0;
};
hipError_t launch_dequant_idct_generic(const GenericDecParams& p, hipStream_t stream);

}  // namespace jpezy_dev
