// jpezy_kernels_f32.hip -- encode kernel, variant 1 (the default): one quad per wave, one launch of exactly as many waves as quads.
// Three precision levels, same bits as the reference: the arithmetic of a quad is jpezy_f32_quad.h.
#include "jpezy_f32_quad.h"

namespace jpezy_dev {
namespace f32 {

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


}  // namespace jpezy_dev
