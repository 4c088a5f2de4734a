// jpezy_kernels_f32_ps.hip -- the f32 encode path in PERSISTENT workgroups (encode variants 2 and 3): the loads decoupled
// from the computing waves.  Same arithmetic, same bits (jpezy_f32_quad.h); parity: tests/test_gpu_persistent.py.  Neither is the
// default: on 4096^2 random pixels variant 3 equals the one-quad kernel and variant 2 is 2 us behind it, and both lose 2-3 us per step
// in short replays -- the kernel is bound by instruction issue, not by its loads (profiles/r05_persistent_ab.txt).
#include "jpezy_f32_quad.h"

namespace jpezy_dev {
namespace f32 {

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
        if (JPEZY_PS_CONSTS_LDS && threadIdx.x < sizeof(pst.f32col) / 16)
            reinterpret_cast<uint4*>(&pst.f32col[0][0])[threadIdx.x] = reinterpret_cast<const uint4*>(&p.tab->f32col[0][0])[threadIdx.x];
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
// workgroups per CU: two when they are small enough (12 waves: six always-computing waves per SIMD -- which takes the DC formula instead of
// the 32 KB table, the quantiser records read from LDS instead of kept in 22 registers, and a kernel within 80 VGPRs)
constexpr int PS2_WG_PER_CU = PS2_WAVES <= 12 ? 2 : 1;
static_assert(PS2_WG_PER_CU * (PS2_WAVES * WAVE_LDS_DWORDS * 4 + (JPEZY_PS_DC_FORMULA ? 16 : (2 * 16385 + 15) / 16 * 16) + (int)sizeof(PsTables) + 16) <= 160 * 1024, "LDS per CU");

template <bool GRAY, int FORCE>
__global__ __launch_bounds__(64 * PS2_WAVES, (PS2_WG_PER_CU * PS2_WAVES + 3) / 4) void fdct_quant_f32_ps2_kernel(EncParams p)
{
    constexpr int DCQ_BYTES = JPEZY_PS_DC_FORMULA ? 16 : (2 * 16385 + 15) / 16 * 16;
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
        if (!JPEZY_PS_DC_FORMULA)
            for (unsigned k = threadIdx.x; k < DCQ_BYTES / 16; k += 64 * PS2_WAVES) dst[k] = src[k];
        if (threadIdx.x < 64) {
            pst.cos[threadIdx.x] = c_cos[threadIdx.x];
            pst.zzinv[threadIdx.x] = c_zzinv[threadIdx.x];
        }
        if (threadIdx.x < 128) {
            (&pst.qinv[0][0])[threadIdx.x] = (&p.tab->qinv[0][0])[threadIdx.x];
            (&pst.qt[0][0])[threadIdx.x] = (&p.tab->qt[0][0])[threadIdx.x];
        }
        if (JPEZY_PS_CONSTS_LDS && threadIdx.x < sizeof(pst.f32col) / 16)
            reinterpret_cast<uint4*>(&pst.f32col[0][0])[threadIdx.x] = reinterpret_cast<const uint4*>(&p.tab->f32col[0][0])[threadIdx.x];
    }
    uint32_t* lds = slices[wave];
    LaneConsts lc = {};
    if (!JPEZY_PS_CONSTS_LDS) lc = load_lane_consts(p.tab, lane0);
    __syncthreads();                                // (waits for every load above: vmcnt(0) in front of the barrier)
    if (!JPEZY_PS_CONSTS_LDS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(lc.ks_l[k]), "+v"(lc.ks_c[k]));
        asm volatile("" : "+v"(lc.dd_l), "+v"(lc.dd_c), "+v"(lc.th_l), "+v"(lc.th_c), "+v"(lc.zz_lo), "+v"(lc.zz_hi));
    }

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
    if (JPEZY_PS_DC_FORMULA && (p0.dc_rq[0] == 0.f || p0.dc_rq[1] == 0.f)) return launch_fdct_quant_f32(p0, gray, force, stream);   // formula not valid for these constants
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
    if (JPEZY_PS_DC_FORMULA && (p.dc_rq[0] == 0.f || p.dc_rq[1] == 0.f)) return launch_fdct_quant_f32(p0, gray, force, stream);   // formula not valid for these constants
    const unsigned cus = (unsigned)(n_cus > 0 ? n_cus : 256) * f32::PS2_WG_PER_CU;
    const unsigned need = (p.ps_total_quads + f32::PS2_WAVES - 1) / f32::PS2_WAVES;
    const unsigned nwg = need < cus ? need : cus;
    if (gray) enc_f32_ps2_launch2<true>(p, force, nwg, stream); else enc_f32_ps2_launch2<false>(p, force, nwg, stream);
    return hipGetLastError();
}

}  // namespace jpezy_dev
