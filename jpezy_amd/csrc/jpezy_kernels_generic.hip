// jpezy_kernels_generic.hip -- decode for ANY baseline layout the reference's decoder accepts (1 or 3 components,
// sampling factors 1..4).  Blocks are independent 8x8 inverse transforms whatever the layout:
//   generic_idct_kernel : one wavefront per EIGHT consecutive blocks, lane = (block, column).  Fast path as in the fused
//                         kernel of jpezy_kernels.hip: FP64 separable butterflies (column pass, transpose through LDS, row
//                         pass), sample = int(v) with a guard band of 2^-18 around every integer, DC-only blocks exact by
//                         construction.  A block with a sample inside the band (5e-4 of the blocks), with a coefficient
//                         above the magnitude guard, or under the force_exact test hook is recomputed by all 64 lanes in
//                         the reference's own order: lane (y, x) accumulates the 64 terms
//                         ((cu*cv) * (coef*Q)) * cos[u][x] * cos[v][y], v outer, u inner, and stores int(sum/4 + sl),
//                         sl = 128 (2048 if precision != 8)                 (ref decoder/jpezy_decoder.hpp:645-670)
//   generic_rgb_kernel  : one lane per pixel; block placement of each component exactly as decode_mcu does it (ref
//                         :504-528), then make_rgb / revise_value in the reference's FP64 order (ref :531-578, 672-676)
// The fused kernel covers jpezy_encode's own 2x2,1x1,1x1 layout in one pass; this pair exists so that every file the
// reference decodes also decodes here, at about a third of the fused kernel's speed.
#include "jpezy_device.h"
#include "../../include/jpezy_constants.h"

namespace jpezy_dev {
namespace generic {

__constant__ double c_cos[64] = JPEZY_COS_INIT;
__constant__ unsigned char c_zzinv[64] = JPEZY_ZZ_INV_INIT;
#define JPEZY_S JPEZY_INV_SQRT2
#define GFMA(a, b, c) __builtin_fma((a), (b), (c))

__device__ __forceinline__ void gsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// x[y] = sum_v X[v] * cos((2y+1)v*pi/16)   (X[0] already carries its 1/sqrt2); an estimate: FMAs allowed
__device__ __forceinline__ void idct8_est(const double* X, double* x)
{
    const double C1 = 0.98078528040323044913, C2 = 0.92387953251128675613, C3 = 0.83146961230254523708, C4 = 0.70710678118654752440,
                 C5 = 0.55557023301960222474, C6 = 0.38268343236508977173, C7 = 0.19509032201612826785;
    const double t0 = GFMA(X[4], C4, X[0]), t1 = GFMA(-X[4], C4, X[0]);
    const double t2 = GFMA(X[6], C6, X[2] * C2), t3 = GFMA(-X[6], C2, X[2] * C6);
    const double E0 = t0 + t2, E3 = t0 - t2, E1 = t1 + t3, E2 = t1 - t3;
    const double O0 = GFMA(X[7], C7, GFMA(X[5], C5, GFMA(X[3], C3, X[1] * C1)));
    const double O1 = GFMA(-X[7], C5, GFMA(-X[5], C1, GFMA(-X[3], C7, X[1] * C3)));
    const double O2 = GFMA(X[7], C3, GFMA(X[5], C7, GFMA(-X[3], C1, X[1] * C5)));
    const double O3 = GFMA(-X[7], C1, GFMA(X[5], C3, GFMA(-X[3], C5, X[1] * C7)));
    x[0] = E0 + O0; x[7] = E0 - O0;
    x[1] = E1 + O1; x[6] = E1 - O1;
    x[2] = E2 + O2; x[5] = E2 - O2;
    x[3] = E3 + O3; x[4] = E3 - O3;
}

constexpr int G_BLOCKS = 8;                 // blocks per wavefront
constexpr int G_STG = 72;                   // int16 elements between staged blocks (144 bytes: the eight blocks' zig-zag reads spread over the banks)
constexpr int G_PITCH = 18;                 // dwords per transposed row: 8 doubles + 1 pad
constexpr int G_TILE = 8 * G_PITCH + 2;     // dwords per block of the transpose tile
constexpr float G_EPS = 0x1p-18f;

__global__ __launch_bounds__(64) void generic_idct_kernel(GenericDecParams p, long nblk)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[G_BLOCKS * G_TILE];     // 1168 dwords; the coefficient staging (576 B.. 1152 B) shares it
    __shared__ int dct[64];
    const int lane = threadIdx.x, b = lane >> 3, u = lane & 7;
    const long g0 = (long)blockIdx.x * G_BLOCKS, g = g0 + b;
    const bool live = g < nblk;

    // ---- coefficients of the eight blocks: 1 KB contiguous -> LDS ----
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.coeffs + g0 * 64);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g0 + (lane >> 3) < nblk) v = src[lane];                              // lane: block lane>>3, 16-byte piece lane&7
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(lds) + (lane >> 3) * (G_STG * 2) + (lane & 7) * 16) = v;
    }
    gsync();
    const int k = live ? (int)(g % p.blocks_per_mcu) : 0;
    int comp = 0;
    if (k >= p.blk_start[1]) comp = 1;
    if (k >= p.blk_start[2]) comp = 2;

    // ---- column pass for column u of block b ----
    double col[8];
    int amax = 0, acor = 0;
    {
        const int16_t* blk = reinterpret_cast<const int16_t*>(lds) + b * G_STG;
        double in[8];
        int c[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            c[v] = blk[c_zzinv[v * 8 + u]];
            in[v] = (double)c[v] * p.dqscale[(comp * 8 + u) * 8 + v];
            amax = max(amax, max(c[v], -c[v]));
        }
        // the DC term in the reference's own order, so that a block with no other coefficient comes out exact
        if (u == 0) in[0] = (JPEZY_S * JPEZY_S) * (double)(c[0] * p.qt[comp * 64]) * 0.25;
        acor = (u ? c[0] : 0) | c[1] | c[2] | c[3] | c[4] | c[5] | c[6] | c[7];
        idct8_est(in, col);
    }
    const unsigned long long ac_any = __ballot(acor != 0);
    const bool dc_only = ((ac_any >> (8 * b)) & 0xFFull) == 0;
    const unsigned long long big = __ballot(amax > p.coef_limit);
    gsync();                                                                     // staging consumed: the tile overwrites it

    // ---- transpose, row pass ----
#pragma unroll
    for (int y = 0; y < 8; ++y) *reinterpret_cast<double*>(lds + b * G_TILE + y * G_PITCH + u * 2) = col[y];
    gsync();
    int smp[8];
    bool flagged = false;
    {
        const int y = u;                                                         // this lane now owns row y of block b
        double in[8], out[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) in[q] = *reinterpret_cast<const double*>(lds + b * G_TILE + y * G_PITCH + q * 2);
        in[0] += (double)p.level;
        idct8_est(in, out);
        float em = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            smp[q] = (int)out[q];
            em = __builtin_fmaxf(em, __builtin_fabsf((float)__builtin_amdgcn_fract(out[q]) - 0.5f));
        }
        flagged = !dc_only && em > 0.5f - G_EPS;
    }
    // blocks that go to the reference-order path: any sample in the band, any coefficient above the magnitude guard
    unsigned long long fl = __ballot(flagged) | big;
    unsigned need = 0;
#pragma unroll
    for (int q = 0; q < G_BLOCKS; ++q) need |= ((fl >> (8 * q)) & 0xFFull) ? (1u << q) : 0u;
    if (p.force_exact) need = 0xFFu;
    if (live && !((need >> b) & 1u)) {
        int4* dst = reinterpret_cast<int4*>(p.samples + g * 64 + u * 8);
        dst[0] = make_int4(smp[0], smp[1], smp[2], smp[3]);
        dst[1] = make_int4(smp[4], smp[5], smp[6], smp[7]);
    }
    // ---- reference order, one block at a time, lane = sample (y, x) ----
    unsigned done = 0;
    while (need) {
        const int q = __builtin_ctz(need);
        need &= need - 1;
        const long gq = g0 + q;
        if (gq >= nblk) continue;
        const int kq = (int)(gq % p.blocks_per_mcu);
        int cq = 0;
        if (kq >= p.blk_start[1]) cq = 1;
        if (kq >= p.blk_start[2]) cq = 2;
        gsync();
        // natural index `lane`: coefficient at zig-zag position zzinv[lane], times its quantiser (ref :645-650)
        dct[lane] = (int)p.coeffs[gq * 64 + c_zzinv[lane]] * p.qt[cq * 64 + lane];
        gsync();
        const int y = lane >> 3, x = lane & 7;
        double sum = 0;
        for (int v = 0; v < 8; ++v) {
            const double cv = (!v) ? JPEZY_S : 1.0;
            for (int uu = 0; uu < 8; ++uu) {
                const double cu = (!uu) ? JPEZY_S : 1.0;
                sum += cu * cv * dct[v * 8 + uu] * c_cos[uu * 8 + x] * c_cos[v * 8 + y];
            }
        }
        p.samples[gq * 64 + lane] = (int)(sum / 4 + p.level);
        done += 64;
    }
    if (done && lane == 0) atomicAdd(p.fallback_count + (blockIdx.x & (COUNTER_SHARDS - 1)), (unsigned long long)done);
}

__device__ __forceinline__ uint32_t revise(double v) { return (v < 0.0) ? 0u : (v > 255.0) ? 255u : (uint32_t)v; }

__device__ __forceinline__ unsigned gdiv(unsigned n, unsigned magic, unsigned shift)      // fast_div_setup, jpezy_device.h
{
    const unsigned q = __umulhi(n, magic);
    return magic ? (((n - q) >> 1) + q) >> shift : n;
}

// One thread: four consecutive pixels of a row (one 4-byte store per plane when the row allows it); a workgroup: 256 pixels
// of four rows.  Everything that depends on the row only (MCU row, the component's block row, whether decode_mcu ever
// writes that row of the component's plane) is computed once per thread.
__global__ __launch_bounds__(256) void generic_rgb_kernel(GenericDecParams p)
{
    const unsigned x0 = (blockIdx.x * 64u + (threadIdx.x & 63u)) * 4u;
    const unsigned y = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (y >= (unsigned)p.H || x0 >= (unsigned)p.W) return;
    {   // batch form: blockIdx.z is the frame
        const size_t f = blockIdx.z;
        p.samples += f * ((size_t)p.mcu_cols * p.mcu_rows * p.blocks_per_mcu * 64);
        p.r += f * p.plane_stride; p.g += f * p.plane_stride; p.b += f * p.plane_stride;
    }
    const unsigned mw = (unsigned)p.hmax * 8u, mh = (unsigned)p.vmax * 8u;
    const unsigned uy = gdiv(y, p.mh_magic, p.mh_shift), iy = y - uy * mh;
    // decode_mcu (ref :504-528) writes block (kx, ky) of a component at plane offset (kx*8, ky*8) -- not scaled by the
    // replication factor -- as a rectangle of 8*dupx x 8*dupy samples, ky outer, kx inner; the last write to a position
    // stays.  For H == hmax or H == 1 that is ordinary nearest-neighbour upsampling.  For the other legal factors
    // (H = 2 or 3 under hmax = 3 or 4) later blocks overwrite part of earlier ones and the right/bottom end of the
    // plane is never written: it keeps the initial value of comp[] (0 / 0x80, ref :104-105) in every MCU.
    unsigned rowblk[3], rowsmp[3];
    bool rowok[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned cvv = (unsigned)p.cv[c], dupy = (unsigned)p.vmax / cvv;
        const unsigned ky = min(cvv - 1u, iy >> 3), yu = iy - ky * 8u;                   // last block row written over iy
        rowok[c] = c < p.ncomp && yu < 8u * dupy;
        rowblk[c] = (unsigned)p.blk_start[c] + ky * (unsigned)p.ch[c];
        rowsmp[c] = gdiv(yu, p.dy_magic[c], p.dy_shift[c]) * 8u;
    }
    const unsigned npx = min(4u, (unsigned)p.W - x0);
    int smp[3][4];
    {
        // the four pixels sit in one MCU (x0 is a multiple of 4, MCUs are multiples of 8 wide)
        const unsigned ux = gdiv(x0, p.mw_magic, p.mw_shift), ix0 = x0 - ux * mw;
        const size_t mcu_blk = ((size_t)uy * p.mcu_cols + ux) * (size_t)p.blocks_per_mcu;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int init = c ? 0x80 : 0;                                              // missing components read 0x80 (ref :104-105)
            smp[c][0] = smp[c][1] = smp[c][2] = smp[c][3] = init;
            if (!rowok[c]) continue;
            const unsigned chh = (unsigned)p.ch[c], dupx = (unsigned)p.hmax / chh;
            if (chh == (unsigned)p.hmax) {              // one sample per pixel: four neighbours of one block row, one 16-byte load
                const int4 v = *reinterpret_cast<const int4*>(p.samples + (mcu_blk + rowblk[c] + (ix0 >> 3)) * 64 + rowsmp[c] + (ix0 & 7u));
                smp[c][0] = v.x; smp[c][1] = v.y; smp[c][2] = v.z; smp[c][3] = v.w;
            } else if (chh == 1u && dupx == 2u) {       // every sample twice
                const int2 v = *reinterpret_cast<const int2*>(p.samples + (mcu_blk + rowblk[c]) * 64 + rowsmp[c] + (ix0 >> 1));
                smp[c][0] = smp[c][1] = v.x; smp[c][2] = smp[c][3] = v.y;
            } else if (chh == 1u && dupx == 4u) {       // one sample for all four
                smp[c][0] = smp[c][1] = smp[c][2] = smp[c][3] = p.samples[(mcu_blk + rowblk[c]) * 64 + rowsmp[c] + (ix0 >> 2)];
            } else {                                    // the other legal factors: the last block written over each pixel
                for (unsigned j = 0; j < 4; ++j) {
                    const unsigned ix = ix0 + j;
                    const unsigned kx = min(chh - 1u, ix >> 3), xu = ix - kx * 8u;
                    if (xu < 8u * dupx) smp[c][j] = p.samples[(mcu_blk + rowblk[c] + kx) * 64 + rowsmp[c] + gdiv(xu, p.dx_magic[c], p.dx_shift[c])];
                }
            }
        }
    }
    uint32_t rw = 0, gw = 0, bw = 0;
#pragma unroll
    for (unsigned j = 0; j < 4; ++j) {
        const double yp = smp[0][j], up = smp[1][j], vp = smp[2][j];
        uint32_t r, g, b;
        if (!p.gray) {
            r = revise(yp + (vp - 0x80) * 1.4020);
            g = revise(yp - (up - 0x80) * 0.3441 - (vp - 0x80) * 0.7139);
            b = revise(yp + (up - 0x80) * 1.7718);
        } else {
            r = g = b = revise(yp);
        }
        rw |= r << (8 * j); gw |= g << (8 * j); bw |= b << (8 * j);
    }
    const size_t off = (size_t)y * p.W + x0;
    if (npx == 4 && (p.W & 3) == 0) {                           // rows start 4-byte aligned (the planes are device allocations)
        *reinterpret_cast<uint32_t*>(p.r + off) = rw;
        *reinterpret_cast<uint32_t*>(p.g + off) = gw;
        *reinterpret_cast<uint32_t*>(p.b + off) = bw;
    } else {
        for (unsigned j = 0; j < npx; ++j) {
            p.r[off + j] = (uint8_t)(rw >> (8 * j)); p.g[off + j] = (uint8_t)(gw >> (8 * j)); p.b[off + j] = (uint8_t)(bw >> (8 * j));
        }
    }
}

}  // namespace generic

hipError_t launch_dequant_idct_generic(const GenericDecParams& p_in, hipStream_t s)
{
    GenericDecParams p = p_in;
    const int nfr = p.n_frames < 1 ? 1 : p.n_frames;
    const long nblk = (long)p.mcu_cols * p.mcu_rows * p.blocks_per_mcu * nfr;       // the block loop does not care where a frame ends
    if (nblk <= 0) return hipSuccess;
    if (nblk > 0x7FFFFFFFL || nfr > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(generic::generic_idct_kernel, dim3((unsigned)((nblk + generic::G_BLOCKS - 1) / generic::G_BLOCKS)), dim3(64), 0, s, p, nblk);
    fast_div_setup((unsigned)p.hmax * 8u, &p.mw_magic, &p.mw_shift);
    fast_div_setup((unsigned)p.vmax * 8u, &p.mh_magic, &p.mh_shift);
    for (int c = 0; c < 3; ++c) {
        fast_div_setup((unsigned)(p.hmax / p.ch[c]), &p.dx_magic[c], &p.dx_shift[c]);
        fast_div_setup((unsigned)(p.vmax / p.cv[c]), &p.dy_magic[c], &p.dy_shift[c]);
    }
    const unsigned gx = ((unsigned)p.W + 255u) / 256u, gy = ((unsigned)p.H + 3u) / 4u;
    hipLaunchKernelGGL(generic::generic_rgb_kernel, dim3(gx, gy, (unsigned)nfr), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace jpezy_dev
