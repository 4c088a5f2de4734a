// jpezy_kernels_generic.hip -- decode for ANY baseline layout the reference's decoder accepts (1 or 3 components,
// sampling factors 1..4): the reference's arithmetic itself, one sample per lane.
//   generic_idct_kernel : one wavefront per 8x8 block; lane (y, x) accumulates the 64 terms
//                         ((cu*cv) * (coef*Q)) * cos[u][x] * cos[v][y] in the reference's order (v outer, u inner) and
//                         stores int(sum/4 + sl), sl = 128 (2048 if precision != 8)  (ref decoder/jpezy_decoder.hpp:645-670)
//   generic_rgb_kernel  : one lane per pixel; nearest-neighbour replication of each component (ref :504-528), then
//                         make_rgb / revise_value in the reference's FP64 order (ref :531-578, 672-676)
// Exact by construction (plain IEEE mul/add, -ffp-contract=off), no guard bands.  ~10x slower than the fused kernel of
// jpezy_kernels.hip, which covers jpezy_encode's own 2x2,1x1,1x1 layout; this one exists so that every file the reference
// decodes also decodes here.
#include "jpezy_device.h"
#include "../../include/jpezy_constants.h"

namespace jpezy_dev {
namespace generic {

__constant__ double c_cos[64] = JPEZY_COS_INIT;
__constant__ unsigned char c_zzinv[64] = JPEZY_ZZ_INV_INIT;
#define JPEZY_S JPEZY_INV_SQRT2

__global__ __launch_bounds__(64) void generic_idct_kernel(GenericDecParams p)
{
    __shared__ int dct[64];
    const long blk = (long)blockIdx.x;                       // global block index: mcu * blocks_per_mcu + k
    const int lane = threadIdx.x;
    const int k = (int)(blk % p.blocks_per_mcu);
    int comp = 0;
    if (k >= p.blk_start[1]) comp = 1;
    if (k >= p.blk_start[2]) comp = 2;
    const int16_t* z = p.coeffs + blk * 64;
    // natural index `lane`: coefficient at zig-zag position zzinv[lane], times its quantiser (ref :645-650)
    dct[lane] = (int)z[c_zzinv[lane]] * p.qt[comp * 64 + lane];
    __syncthreads();
    const int y = lane >> 3, x = lane & 7;
    double sum = 0;
    for (int v = 0; v < 8; ++v) {
        const double cv = (!v) ? JPEZY_S : 1.0;
        for (int u = 0; u < 8; ++u) {
            const double cu = (!u) ? JPEZY_S : 1.0;
            sum += cu * cv * dct[v * 8 + u] * c_cos[u * 8 + x] * c_cos[v * 8 + y];
        }
    }
    p.samples[blk * 64 + lane] = (int)(sum / 4 + p.level);
}

__device__ __forceinline__ uint8_t revise(double v) { return (v < 0.0) ? 0 : (v > 255.0) ? 255 : (uint8_t)v; }

__global__ __launch_bounds__(256) void generic_rgb_kernel(GenericDecParams p)
{
    const long px = (long)blockIdx.x * 256 + threadIdx.x;
    if (px >= (long)p.W * p.H) return;
    const int y = (int)(px / p.W), x = (int)(px - (long)y * p.W);
    const int mw = p.hmax * 8, mh = p.vmax * 8;
    const int ux = x / mw, uy = y / mh, ix = x - ux * mw, iy = y - uy * mh;
    const long mcu = (long)uy * p.mcu_cols + ux;
    int s[3] = { 0, 0x80, 0x80 };                              // missing components read 0x80 (ref :104-105)
    for (int c = 0; c < p.ncomp; ++c) {
        // decode_mcu (ref :504-528) writes block (kx, ky) of the component at plane offset (kx*8, ky*8) -- not scaled by the
        // replication factor -- as a rectangle of 8*dupx x 8*dupy samples, ky outer, kx inner; the last write to a position
        // stays.  For H == hmax or H == 1 that is ordinary nearest-neighbour upsampling.  For the other legal factors
        // (H = 2 or 3 under hmax = 3 or 4) later blocks overwrite part of earlier ones and the right/bottom end of the
        // plane is never written: it keeps the initial value of comp[] (0 / 0x80, ref :104-105) in every MCU.
        const int dupx = p.hmax / p.ch[c], dupy = p.vmax / p.cv[c];
        const int kx = min(p.ch[c] - 1, ix >> 3), ky = min(p.cv[c] - 1, iy >> 3);      // last block written over (ix, iy)
        const int xu = ix - kx * 8, yu = iy - ky * 8;
        if (xu >= 8 * dupx || yu >= 8 * dupy) continue;                                // never written
        const int sx = xu / dupx, sy = yu / dupy;
        const long blk = mcu * p.blocks_per_mcu + p.blk_start[c] + ky * p.ch[c] + kx;
        s[c] = p.samples[blk * 64 + sy * 8 + sx];
    }
    const double yp = s[0], up = s[1], vp = s[2];
    if (!p.gray) {
        p.r[px] = revise(yp + (vp - 0x80) * 1.4020);
        p.g[px] = revise(yp - (up - 0x80) * 0.3441 - (vp - 0x80) * 0.7139);
        p.b[px] = revise(yp + (up - 0x80) * 1.7718);
    } else {
        p.r[px] = p.g[px] = p.b[px] = revise(yp);
    }
}

}  // namespace generic

hipError_t launch_dequant_idct_generic(const GenericDecParams& p, hipStream_t s)
{
    const long nblk = (long)p.mcu_cols * p.mcu_rows * p.blocks_per_mcu;
    if (nblk <= 0) return hipSuccess;
    if (nblk > 0x7FFFFFFFL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(generic::generic_idct_kernel, dim3((unsigned)nblk), dim3(64), 0, s, p);
    const long npx = (long)p.W * p.H;
    hipLaunchKernelGGL(generic::generic_rgb_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace jpezy_dev
