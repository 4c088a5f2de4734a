// jpezy_device.h -- shared declarations of the gfx950 kernels and their launchers (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace jpezy_dev {

// Fixed-point scale of the quantiser guard band: a coefficient is v/Q * 2^QFRAC_BITS truncated to int32.
// |v/Q| <= 1024/10 with the Annex-K tables, so 2^24 keeps |n| < 2^31.
constexpr int QFRAC_BITS = 24;
// IDCT samples: (sum/4+128) * 2^SFRAC_BITS; samples outside +-2^(30-SFRAC_BITS) take the exact path.
constexpr int SFRAC_BITS = 18;

// Device-resident tables built by the host at context creation (jpezy_capi.hip).
struct DeviceTables {
    // encode: qscale[t][j][i] = cu(j) * cv(i) / (4 * Q_t[i*8+j]) * 2^QFRAC_BITS   (t: 0 luma, 1 chroma)
    double qscale[2][8][8];
    double rq_dc[2];          // 1 / Q_t[0]
    int qt[2][64];            // natural order
};

struct EncParams {
    const uint8_t* r;
    const uint8_t* g;
    const uint8_t* b;
    size_t plane_stride;      // bytes between frames of one plane
    int16_t* coeffs;
    size_t coeffs_per_frame;  // int16 elements
    const DeviceTables* tab;
    unsigned long long* fallback_count;
    int W, H, mcu_cols, mcu_rows, quads_per_row, n_frames;
};

struct DecParams {
    const int16_t* coeffs;
    size_t coeffs_per_frame;
    uint8_t* r;
    uint8_t* g;
    uint8_t* b;
    size_t plane_stride;
    const double* dqscale;    // [3 comps][8 (u=lane col)][8 (v)] : cu*cv*Q[v*8+u]   (dequant folded in)
    const int* dqt;           // [3 comps][64] natural order quant values (exact path)
    unsigned long long* fallback_count;
    int W, H, mcu_cols, mcu_rows, quads_per_row, n_frames;
};

hipError_t launch_fdct_quant(const EncParams& p, bool gray, bool force_exact, hipStream_t stream);
hipError_t launch_dequant_idct(const DecParams& p, bool gray, bool force_exact, hipStream_t stream);

}  // namespace jpezy_dev
