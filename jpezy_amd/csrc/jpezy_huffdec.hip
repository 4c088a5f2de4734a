// jpezy_huffdec.hip -- the decoder's serial head on the GPU (decode side): Huffman decoding of baseline
// scans into zig-zagged int16 coefficients, same results as jpezy_host::read_jpeg / the reference's
// decoder::decode_huffman (ref decoder/jpezy_decoder.hpp:583-642).  Three forms: one scan (the kernels right below), many
// independent streams per launch (batch form: the scans of many files, or the restart intervals of one scan), and a lane per
// stream for short restart intervals.
//
// A Huffman stream has no entry points, but it is self-synchronising: a decoder started at a wrong place falls into
// step with the true one after a few symbols.  The scan (after removing the 0xFF00 stuffing) is cut into subsequences
// of SUBSEQ_BITS bits, one lane each.  Lane i decodes from the state its predecessor left it -- (bit offset of the
// first code word inside subsequence i, block index inside the MCU, zig-zag index inside the block) -- to the end of
// its subsequence and publishes the state it leaves to lane i+1.  Lane 0's entry state is known (0, 0, 0); everybody
// else starts from a guess and the passes are repeated until no exit state changes: a fixed point in which every lane
// decodes from its predecessor's true exit state, i.e. the sequential decode (Klein & Wiseman's observation, the
// scheme of Weissenberger & Schmidt's GPU JPEG decoder).  Lanes whose entry state did not change skip the pass.
// Then: prefix sum of the blocks every lane completes -> global block index of every lane; one more pass that writes
// the coefficients (DC still as differences); per component a prefix sum over the DC differences (pre_DC, ref :611).
//
// Anything irregular -- an invalid code, a run past the end of a block, a scan that ends early, restart markers that are
// not exactly where they belong -- makes the caller (jpezy_capi.hip) fall back to the host decoder, whose verdict is
// authoritative.  Round 3: lookup tables and a branch-free decode step (jpezy_huffdec_core.h), one confirmation + refinement
// launch with a fixed-point test, the end of the entropy-coded segment found by the stuffing count.  Round 4, single scan: the chain
// runs without the host looking in between (ScanState, guarded launches), speculation runs backwards (a lane's last walk is its
// confirmation walk), marks at every quarter of a subsequence (shorter coefficient walks, re-walks that stop where they rejoin),
// two independent table lookups per symbol, unstuffing with wide loads and LDS-staged stores, DC differences in an array of their
// own.  The batch form keeps round 3's schedule (forward speculation with proposals): its calls are bound by PCIe.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

#include "jpezy_huffdec.h"

namespace jpezy_dev {
namespace huffdec {

// 1,024 bits.  512 was measured once more on the final round-4 decoder (profiles/r04_huffdec_s512.txt): the same 0.61 ms on 4096^2 noise,
// 10-20 % less on everything shorter or smoother (smooth 4096^2 0.39 -> 0.32 ms, 1080p 0.47 -> 0.43, small files 0.32 -> 0.26, 256-file
// batches -4 %) -- but dense high-quality noise (blocks longer than a subsequence) leaves fewer lanes in step after the speculation, the
// "more than half of the lanes wrong at the first step" cut hands such files to the host decoder (one of the test suite's libjpeg files,
// and about twice as many files of the fuzz corpus), and relaxing the cut costs flat images milliseconds of futile refinement.
#ifndef JPEZY_SUBSEQ_BITS
#define JPEZY_SUBSEQ_BITS 1024
#endif
constexpr int SUBSEQ_BITS = JPEZY_SUBSEQ_BITS;
constexpr int WGS = 256;

// packed lane state: bit 31 valid, bit 30 error, p (bit offset of the next code word past the subsequence end) << 16,
// b (block inside the MCU) << 8, k (zig-zag index, 0 = a DC symbol comes next)
__device__ __forceinline__ uint32_t pack_state(unsigned p, unsigned b, unsigned k) { return 0x80000000u | (p << 16) | (b << 8) | k; }
constexpr uint32_t STATE_ERR = 0xC0000000u;

// A workgroup's window of the unstuffed scan lives in LDS (coalesced copy, then ~100 ns per access instead of a
// dependent global load per symbol): the 256 subsequences of the workgroup plus a tail for the last lane's overrun.
constexpr int OVERFLOW = 12 * 1024 / SUBSEQ_BITS;      // most subsequences a speculating lane may decode beyond its own (window size)
constexpr int OVERFLOW_DEFAULT = 3 * 1024 / SUBSEQ_BITS;   // in 1024-bit units: measured on 4096^2 random pixels: 1: 92 % of the proposals true, 3: 99.5 %, 8: all; 2-4 is the fastest overall
constexpr int WINDOW_LINEAR = (WGS + OVERFLOW) * SUBSEQ_BITS / 32 + 16;
// A subsequence is 32 words, so lane l's cursor sits at word 32 l + k: without padding all 64 lanes of a wave hit the
// same LDS bank on every read.  One pad word per 32 spreads them over the banks (word i lives at i + i / 32).
constexpr int WINDOW_WORDS = WINDOW_LINEAR + WINDOW_LINEAR / 32 + 1;
__device__ __forceinline__ unsigned pad_index(unsigned i) { return i + (i >> 5); }

// The cursor: a bit position relative to the window's first bit and the three window words around it in registers -- w0 holds the bit
// at pos, w1 the word after it, w2 one more, fetched a symbol ahead of its use so that no LDS latency sits between two symbols except the
// two table lookups.  The window words are stored most-significant-bit first (load_window swaps the bytes once).
struct Cursor {
    const uint32_t* w;
    unsigned pos;
    unsigned long long ww;        // the word that holds the bit at pos (high half) and the one after it
    uint32_t w2;
    unsigned nxt;                 // window index of the word after w2
    __device__ __forceinline__ void init(const uint32_t* win, unsigned rel)
    {
        w = win;
        pos = rel;
        const unsigned i = rel >> 5;
        ww = ((unsigned long long)w[pad_index(i)] << 32) | w[pad_index(i + 1)];
        w2 = w[pad_index(i + 2)];
        nxt = i + 3;
    }
    // the 32 bits at pos: one 64-bit shift (round 3 formed them from two 32-bit words in four instructions)
    __device__ __forceinline__ uint32_t peek32() const { return (uint32_t)((ww << (pos & 31u)) >> 32); }
    __device__ __forceinline__ uint32_t prefetch() const { return w[pad_index(nxt)]; }     // issued at the top of a step, used at its bottom
    __device__ __forceinline__ void advance(unsigned np, uint32_t ahead)               // np - pos <= 32: at most one word further
    {
        const bool adv = ((np ^ pos) >> 5) != 0u;
        pos = np;
        ww = adv ? (ww << 32) | w2 : ww;
        w2 = adv ? ahead : w2;
        nxt += adv ? 1u : 0u;
    }
};

// The same cursor over a stream in GLOBAL memory (the lane-per-stream kernel of short restart intervals: 256 streams per workgroup do not
// fit LDS).  Four words in registers; the next one is requested when a word boundary is crossed and used at the boundary after it -- some
// five symbols later --, so the load's latency is not in the symbol loop.  U holds bytes; words beyond the stream's buffer read as zero.
struct GlobalCursor {
    const uint32_t* g;
    unsigned n_words;
    unsigned pos;
    uint32_t w0, w1, w2, w3;
    unsigned nxt;
    __device__ __forceinline__ uint32_t word(unsigned i) const { return i < n_words ? __builtin_bswap32(g[i]) : 0u; }
    __device__ __forceinline__ void init(const uint32_t* U, unsigned u_words)
    {
        g = U; n_words = u_words; pos = 0;
        w0 = word(0); w1 = word(1); w2 = word(2); w3 = word(3);
        nxt = 4;
    }
    __device__ __forceinline__ uint32_t peek32() const
    {
        const unsigned sh = pos & 31u;
        return (w0 << sh) | ((w1 >> 1) >> (31u - sh));
    }
    __device__ __forceinline__ uint32_t prefetch() const { return 0u; }
    __device__ __forceinline__ void advance(unsigned np, uint32_t)
    {
        if (((np ^ pos) >> 5) != 0u) {
            w0 = w1; w1 = w2; w2 = w3;
            w3 = word(nxt);
            ++nxt;
        }
        pos = np;
    }
};

__device__ __forceinline__ void load_window(uint32_t* win, const uint32_t* U, unsigned first_sub, size_t u_words)
{
    const size_t w0 = (size_t)first_sub * (SUBSEQ_BITS / 32);
    for (unsigned i = threadIdx.x; i < (unsigned)WINDOW_LINEAR; i += WGS) win[pad_index(i)] = w0 + i < u_words ? __builtin_bswap32(U[w0 + i]) : 0u;
}

__device__ __forceinline__ void load_setup(Setup& S, const Setup* g)
{
    const uint4* src = reinterpret_cast<const uint4*>(g);
    uint4* dst = reinterpret_cast<uint4*>(&S);
    for (unsigned i = threadIdx.x; i < sizeof(Setup) / 16; i += WGS) dst[i] = src[i];
}
static_assert(sizeof(Setup) % 16 == 0, "Setup is copied in 16-byte pieces");

// decode (without emitting) until pos >= end
__device__ __forceinline__ void run_subsequence(const uint16_t* tabs, unsigned bpm, unsigned tdmask, Cursor& c, Walk& s, unsigned end)
{
    while (c.pos < end) decode_step<false>(tabs, bpm, tdmask, c, s, 0, 0, nullptr);
}

// The same walk, leaving the state at every EMIT_PARTS-th of the subsequence (state at the first code word at or behind the mark, and the blocks
// completed before it): the coefficient pass starts a lane at each mark, so its serial walks are EMIT_PARTS times shorter (round 4; a lane
// decodes ~24 bits per microsecond whatever else the chip is doing, and a single file fills a quarter of its lanes).
constexpr int EMIT_PARTS = 4, PART_BITS = SUBSEQ_BITS / EMIT_PARTS;
// have_old: mark / mark_blocks hold the lane's previous walk of the same subsequence (from another entry state).  A walk that reaches a mark in
// the state the previous walk had there is the previous walk from then on: it stops, and the caller keeps the old exit state (returns the
// number of blocks to add to the old count: new blocks in front of the mark minus old ones; the marks behind it move by as much).  A
// corrected entry state usually falls into step within a few hundred bits, so most re-walks are a quarter or two long.
__device__ __forceinline__ bool run_subsequence_marks(const uint16_t* tabs, unsigned bpm, unsigned tdmask, Cursor& c, Walk& s, unsigned base,
                                                      uint32_t* mark, unsigned* mark_blocks, bool have_old, int* rejoin_delta)
{
    bool rejoined = false;
#pragma unroll
    for (int q = 1; q <= EMIT_PARTS; ++q) {               // (unrolled: mark[] stays in registers)
        const unsigned lim = base + (unsigned)q * PART_BITS;
        if (!rejoined) {
            while (c.pos < lim) decode_step<false>(tabs, bpm, tdmask, c, s, 0, 0, nullptr);
            if (q < EMIT_PARTS) {
                const uint32_t now = pack_state(c.pos - lim, s.b(), s.k);
                if (have_old && now == mark[q - 1]) {
                    rejoined = true;
                    *rejoin_delta = (int)s.nblocks - (int)mark_blocks[q - 1];
                }
                mark[q - 1] = now;
                mark_blocks[q - 1] = s.nblocks;
            }
        } else if (q < EMIT_PARTS) {
            mark_blocks[q - 1] += (unsigned)*rejoin_delta;
        }
    }
    return rejoined;
}

// Speculation, round 4 form: lane i starts `overflow` subsequences BEFORE its own, from the guess (0, 0, 0) -- from the true state when that is
// the start of the scan --, walks up to its own subsequence and through it.  A decoder started at a wrong place is in step with the true one
// after a few hundred bits -- bit position, zig-zag index and, after some more MCUs, the block phase --, so the state it holds at the start of
// its own subsequence is very likely the true one (92 % after one subsequence of warm-up, 99.5 % after three), and its walk through its own
// subsequence -- entry state, exit state, blocks, marks -- is then already what the synchronisation phase would compute from its
// predecessor's exit state: the confirmation walk of every lane (round 3: speculation forwards from the lane's own start, proposals by
// atomicMax, then every lane once more from the adopted proposal) costs nothing any more.  The synchronisation launches only re-walk where a
// lane's entry state is not its predecessor's exit state.
__global__ __launch_bounds__(WGS) void spec_kernel(const Setup* gS, const uint32_t* U, size_t u_words, const ScanState* st, uint32_t* exit_state,
                                                   uint32_t* last_entry, unsigned* nblocks_out, uint32_t* marks, unsigned* mark_blocks,
                                                   unsigned overflow)
{
    __shared__ Setup S;
    __shared__ uint32_t win[WINDOW_WORDS];
    const unsigned n_sub = st->n_sub;
    const unsigned i0 = blockIdx.x * WGS, i = i0 + threadIdx.x;
    if (i0 >= n_sub) return;                                         // (workgroup-uniform: the launch is sized before the stuffing is counted)
    const unsigned wbase = i0 >= overflow ? i0 - overflow : 0u;       // the window starts `overflow` subsequences in front of the workgroup's first
    load_setup(S, gS);
    load_window(win, U, wbase, u_words);
    __syncthreads();
    if (i >= n_sub) return;
    const uint16_t* tabs = reinterpret_cast<const uint16_t*>(&S);
    const unsigned bpm = (unsigned)S.bpm, tdmask = S.tdmask;
    const unsigned start = i >= overflow ? i - overflow : 0u;         // (start == 0: the scan's own start state, no guess)
    Cursor c;
    c.init(win, (start - wbase) * SUBSEQ_BITS);
    Walk wk;
    wk.init(0, 0, tdmask);
    const unsigned own = (i - wbase) * SUBSEQ_BITS;                   // relative to the window's first bit, like c.pos
    run_subsequence(tabs, bpm, tdmask, c, wk, own);                   // the warm-up: up to the start of the lane's own subsequence
    const uint32_t entry = pack_state(c.pos - own, wk.b(), wk.k);
    wk.nblocks = 0;
    uint32_t mk[EMIT_PARTS - 1];
    unsigned mkb[EMIT_PARTS - 1];
    int delta = 0;
    (void)run_subsequence_marks(tabs, bpm, tdmask, c, wk, own, mk, mkb, false, &delta);
    const uint32_t exit_now = pack_state(c.pos - (own + SUBSEQ_BITS), wk.b(), wk.k);
    {   // lanes that leave in the state they entered (a stream with a period that divides the subsequence; one vote per wave)
        const unsigned long long same = __builtin_amdgcn_ballot_w64(entry == exit_now);
        if (same && (threadIdx.x & 63u) == (unsigned)__builtin_ctzll(same)) atomicAdd(const_cast<unsigned*>(&st->periodic), (unsigned)__builtin_popcountll(same));
    }
    last_entry[i] = entry;
    exit_state[i] = exit_now;
    nblocks_out[i] = wk.nblocks;
#pragma unroll
    for (int q = 0; q < EMIT_PARTS - 1; ++q) {
        marks[(size_t)i * (EMIT_PARTS - 1) + q] = mk[q];
        mark_blocks[(size_t)i * (EMIT_PARTS - 1) + q] = mkb[q];
    }
}

// (batch form) the proposals that won become the initial exit states
__global__ void adopt_proposals_kernel(const unsigned long long* proposal, unsigned n_sub, uint32_t* exit_state)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_sub) exit_state[i] = (uint32_t)proposal[i];
}

// One launch: every workgroup iterates over its own 256 subsequences until nothing changes inside it (a lane re-decodes
// only when its entry state changed), so a true entry state at the workgroup's first lane -- or a lane that falls into
// step with the true decode by itself -- propagates through the whole workgroup within the launch.  Launches are
// repeated until no exit state changes anywhere (jpezy_capi.hip): two or three in practice.
__global__ __launch_bounds__(WGS) void sync_kernel(const Setup* gS, const uint32_t* U, size_t u_words, const ScanState* st, uint32_t* exit_state,
                                                   uint32_t* last_entry, unsigned* nblocks_out, uint32_t* marks, unsigned* mark_blocks,
                                                   unsigned* changed, const unsigned* prev, int max_inner)
{
    __shared__ Setup S;
    __shared__ uint32_t win[WINDOW_WORDS];
    __shared__ uint32_t sh_exit[WGS + 1];
    const unsigned n_sub = st->n_sub;
    const unsigned i0 = blockIdx.x * WGS, t = threadIdx.x, i = i0 + t;
    if (i0 >= n_sub) return;
    // enqueued blindly behind another launch: nothing left to do, or a stream that does not synchronise (RefineBudget in jpezy_capi_huffdec.hip)
    if (prev && ((prev[1] == 0u && prev[2] == 0u) || scan_hopeless(prev, n_sub))) return;
    const bool live = i < n_sub;
    uint32_t my_last = live ? last_entry[i] : 0u, my_exit = live ? exit_state[i] : 0u;
    const uint32_t exit_before = my_exit;
    unsigned nb = live ? nblocks_out[i] : 0u;
    sh_exit[t + 1] = my_exit;
    if (t == 0) sh_exit[0] = i0 ? exit_state[i0 - 1] : pack_state(0, 0, 0);
    __syncthreads();
    if (!__syncthreads_or(live && sh_exit[t] != my_last)) return;      // a quiet workgroup: nobody's entry state has changed
    load_setup(S, gS);
    load_window(win, U, i0, u_words);
    __syncthreads();
    const uint16_t* tabs = reinterpret_cast<const uint16_t*>(&S);
    const unsigned bpm = (unsigned)S.bpm, tdmask = S.tdmask;
    bool first_moved = false;                              // this lane's first decode of the launch left another state than it had
    bool walked = false;
    uint32_t mk[EMIT_PARTS - 1];                           // the marks of the lane's last walk (the one that counts), stored once at the end
    unsigned mkb[EMIT_PARTS - 1];
    for (int inner = 0; inner < max_inner; ++inner) {
        const uint32_t entry = sh_exit[t];
        bool redo = live && entry != my_last;
        __syncthreads();                                   // everybody has read its entry before anybody publishes
        if (redo) {
            // a lane that has walked before (this launch or an earlier one) and did not end in the error state has marks to fall back into
            bool have_old = my_last != 0xFFFFFFFFu && !(my_exit & 0x40000000u) && !(my_last & 0x40000000u);
            if (have_old && !walked) {
#pragma unroll
                for (int q = 0; q < EMIT_PARTS - 1; ++q) {
                    mk[q] = marks[(size_t)i * (EMIT_PARTS - 1) + q];
                    mkb[q] = mark_blocks[(size_t)i * (EMIT_PARTS - 1) + q];
                }
            }
            const unsigned nb_old = nb;
            my_last = entry;
            nb = 0;
            walked = true;
            if (entry & 0x40000000u) {
                my_exit = STATE_ERR;
#pragma unroll
                for (int q = 0; q < EMIT_PARTS - 1; ++q) mk[q] = STATE_ERR;
            } else {
                Cursor c;
                c.init(win, t * SUBSEQ_BITS + ((entry >> 16) & 0x3FFFu));
                Walk wk;
                wk.init((entry >> 8) & 0xFFu, entry & 0xFFu, tdmask);
                const unsigned end = (t + 1) * SUBSEQ_BITS;
                int delta = 0;
                if (run_subsequence_marks(tabs, bpm, tdmask, c, wk, t * SUBSEQ_BITS, mk, mkb, have_old, &delta)) {
                    nb = nb_old + (unsigned)delta;         // the previous walk from the mark on: same exit state
                } else {
                    nb = wk.nblocks;
                    my_exit = pack_state(c.pos - end, wk.b(), wk.k);
                }
            }
            redo = sh_exit[t + 1] != my_exit;
            sh_exit[t + 1] = my_exit;
            first_moved = first_moved || (inner == 0 && redo);
        }
        if (!__syncthreads_or(redo)) break;                // nothing changed inside the workgroup
    }
    if (live) {
        last_entry[i] = my_last;
        nblocks_out[i] = nb;
        if (walked) {
#pragma unroll
            for (int q = 0; q < EMIT_PARTS - 1; ++q) {
                marks[(size_t)i * (EMIT_PARTS - 1) + q] = mk[q];
                mark_blocks[(size_t)i * (EMIT_PARTS - 1) + q] = mkb[q];
            }
        }
        // a launch cut off by max_inner can leave a lane whose predecessor's exit moved after the lane last decoded
        // (A -> B -> A inside the launch leaves exit_before == my_exit): such a lane is still pending and must keep
        // the host iterating, or a stale nblocks/last_entry would pass for the fixed point
        const bool pending = sh_exit[t] != my_last;
        if (my_exit != exit_before) {
            exit_state[i] = my_exit;
            atomicAdd(changed, 1u);                        // changed[0]: lanes whose exit state moved in this launch
            // changed[2]: ... among them a workgroup's last lane: the next workgroup may have read the old value at its start, so the
            // launch cannot vouch for the fixed point across that boundary
            if (t == WGS - 1 && i + 1 < n_sub) atomicAdd(changed + 2, 1u);
        }
        if (pending) atomicAdd(changed + 1, 1u);           // changed[1]: lanes that still have to re-decode (cut off by max_inner)
        if (first_moved) atomicAdd(changed + 3, 1u);       // changed[3]: lanes whose state moved at the launch's first step (first launch: wrong proposals)
    }
}

// A lane per EMIT_PARTS-th of a subsequence: part 0 starts from the predecessor's exit state, the others from the marks the last
// synchronisation walk left (sync_kernel).  A workgroup takes WGS / EMIT_PARTS subsequences.
constexpr int EMIT_SUBS = WGS / EMIT_PARTS;
constexpr int EMIT_WINDOW_LINEAR = (EMIT_SUBS + 1) * SUBSEQ_BITS / 32 + 16;        // its subsequences plus the overrun of the last symbol
static_assert(EMIT_WINDOW_LINEAR <= WINDOW_LINEAR && WGS % EMIT_PARTS == 0 && SUBSEQ_BITS % EMIT_PARTS == 0, "emit geometry");
__global__ __launch_bounds__(WGS) void emit_kernel(const Setup* gS, const uint32_t* U, size_t u_words, ScanState* st, const uint32_t* exit_state,
                                                   const uint32_t* marks, const unsigned* mark_blocks, const unsigned long long* blocks_before,
                                                   int16_t* out, int16_t* dc_out, int guarded)
{
    __shared__ Setup S;
    __shared__ uint32_t win[EMIT_WINDOW_LINEAR + EMIT_WINDOW_LINEAR / 32 + 1];
    const unsigned n_sub = st->n_sub;
    if (guarded && !scan_settled(st->changed, st->changed2, n_sub)) return;     // (workgroup-uniform)
    unsigned* const error = &st->error;
    unsigned long long* const last_bit = &st->last_bit;
    const unsigned i0 = blockIdx.x * EMIT_SUBS, ts = threadIdx.x / EMIT_PARTS, q = threadIdx.x % EMIT_PARTS, i = i0 + ts;
    if (i0 >= n_sub) return;
    load_setup(S, gS);
    {
        const size_t w0 = (size_t)i0 * (SUBSEQ_BITS / 32);
        for (unsigned k = threadIdx.x; k < (unsigned)EMIT_WINDOW_LINEAR; k += WGS) win[pad_index(k)] = w0 + k < u_words ? __builtin_bswap32(U[w0 + k]) : 0u;
    }
    __syncthreads();
    if (i >= n_sub) return;
    const unsigned long long g0 = blocks_before[i] + (q ? mark_blocks[(size_t)i * (EMIT_PARTS - 1) + q - 1] : 0u);
    if (g0 >= S.total_blocks) return;                  // everything this lane sees lies behind the last block
    const uint32_t entry = q ? marks[(size_t)i * (EMIT_PARTS - 1) + q - 1] : i ? exit_state[i - 1] : pack_state(0, 0, 0);
    if (entry & 0x40000000u) { *error = 1u; return; }
    const uint16_t* tabs = reinterpret_cast<const uint16_t*>(&S);
    const unsigned bpm = (unsigned)S.bpm, tdmask = S.tdmask, total = S.total_blocks;
    const unsigned base = ts * SUBSEQ_BITS + q * PART_BITS;
    Cursor c;
    c.init(win, base + ((entry >> 16) & 0x3FFFu));
    Walk wk;
    wk.init((entry >> 8) & 0xFFu, entry & 0xFFu, tdmask);
    const unsigned end = base + PART_BITS;
    // stop at the end of the last block: what follows are pad bits, not symbols
    while (c.pos < end && g0 + wk.nblocks < total) {
        if (!decode_step<true>(tabs, bpm, tdmask, c, wk, g0, total, out, dc_out)) { *error = 1u; return; }
    }
    // exactly one lane completes the last block; the bit behind it, counted from the start of the stream
    if (wk.nblocks && g0 + wk.nblocks >= total) *last_bit = (unsigned long long)i0 * SUBSEQ_BITS + c.pos;
}

// ---- DC differences -> absolute values, per component (pre_DC, ref :611-614); component q owns blocks [cstart[q], cstart[q] + ccount[q]) of
// every MCU.  All components in THREE launches (grid.y = component; round 2 took four per component: gather, two-launch scan, scatter): (1) every workgroup turns its 2,048
// DC differences into prefix sums inside the workgroup, in place, and leaves its total; (2) one workgroup per component scans the totals;
// (3) every workgroup adds what came before it.  Sums wrap in 32 bits and are stored as int16: the low 16 bits are those of the true sum.
struct DcGeom { unsigned bpm, ncomp, cstart[3], ccount[3]; unsigned long long nmcu; };
constexpr int DC_PER_WG = 2048;
__device__ __forceinline__ size_t dc_slot(const DcGeom& g, unsigned comp, size_t j)
{
    const unsigned count = g.ccount[comp];
    const size_t mcu = j / count, t = j - mcu * count;
    return (mcu * g.bpm + g.cstart[comp] + t) * 64;
}
// dc (may be null; round 4): the DC differences lie in an array of their own, dc[block] (the coefficient launch put them there): the first launch
// sums in place THERE -- 2 bytes per block instead of a 128-byte line per block -- and the second one writes the values to the coefficients.
__global__ __launch_bounds__(256) void dc_local_kernel(int16_t* coeffs, int16_t* dc, DcGeom g, int* totals, unsigned wg_per_comp, const ScanState* guard)
{
    __shared__ int wsum[4];
    if (guard && !scan_settled(guard->changed, guard->changed2, guard->n_sub)) return;
    const unsigned comp = blockIdx.y;
    const size_t nd = (size_t)g.nmcu * g.ccount[comp], j0 = (size_t)blockIdx.x * DC_PER_WG + (size_t)threadIdx.x * 8;
    if ((size_t)blockIdx.x * DC_PER_WG >= nd) return;                        // workgroup-uniform
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int v[8], sum = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        v[q] = j0 + q < nd ? (int)(dc ? dc[dc_slot(g, comp, j0 + q) >> 6] : coeffs[dc_slot(g, comp, j0 + q)]) : 0;
        sum += v[q];
    }
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < wv) woff += wsum[q];
        tot += wsum[q];
    }
    int run = woff + inc - sum;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        run += v[q];
        if (j0 + q < nd) {
            if (dc) dc[dc_slot(g, comp, j0 + q) >> 6] = (int16_t)run;
            else coeffs[dc_slot(g, comp, j0 + q)] = (int16_t)run;
        }
    }
    if (threadIdx.x == 0) totals[(size_t)comp * wg_per_comp + blockIdx.x] = tot;
}
// totals -> what came before each workgroup (exclusive), one workgroup per component, 256 totals per step with a running carry
__global__ __launch_bounds__(256) void dc_totals_kernel(int* totals, DcGeom g, unsigned wg_per_comp, const ScanState* guard)
{
    __shared__ int wsum[4];
    if (guard && !scan_settled(guard->changed, guard->changed2, guard->n_sub)) return;
    const unsigned comp = blockIdx.x;
    const size_t nd = (size_t)g.nmcu * g.ccount[comp];
    const unsigned n = (unsigned)((nd + DC_PER_WG - 1) / DC_PER_WG);
    int* t = totals + (size_t)comp * wg_per_comp;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int carry = 0;
    for (unsigned i0 = 0; i0 < n; i0 += 256) {
        const unsigned i = i0 + threadIdx.x;
        const int v = i < n ? t[i] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < wv) woff += wsum[q];
            tot += wsum[q];
        }
        if (i < n) t[i] = carry + woff + inc - v;
        carry += tot;
        __syncthreads();
    }
}
// SELF_SUM (round 4, up to DC_SELF_SUM_MAX workgroups per component): the workgroup adds up the raw totals in front of it itself -- a few hundred
// values out of the L2 -- and the launch that scanned them is gone (a dependent launch costs ~5 us)
constexpr unsigned DC_SELF_SUM_MAX = 1024;
template <bool SELF_SUM>
__global__ __launch_bounds__(256) void dc_add_kernel(int16_t* coeffs, const int16_t* dc, DcGeom g, const int* totals, unsigned wg_per_comp,
                                                     const ScanState* guard)
{
    __shared__ int red[4];
    if (guard && !scan_settled(guard->changed, guard->changed2, guard->n_sub)) return;
    const unsigned comp = blockIdx.y;
    const size_t nd = (size_t)g.nmcu * g.ccount[comp], j0 = (size_t)blockIdx.x * DC_PER_WG + (size_t)threadIdx.x * 8;
    if ((size_t)blockIdx.x * DC_PER_WG >= nd) return;
    if (blockIdx.x == 0 && !dc) return;                                     // (nothing comes before the first workgroup; with dc it still has to deliver)
    int before;
    if (SELF_SUM) {
        int sum = 0;
        for (unsigned k = threadIdx.x; k < blockIdx.x; k += 256) sum += totals[(size_t)comp * wg_per_comp + k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
        __syncthreads();
        before = red[0] + red[1] + red[2] + red[3];
    } else {
        before = totals[(size_t)comp * wg_per_comp + blockIdx.x];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (j0 + q < nd) {
            const size_t slot = dc_slot(g, comp, j0 + q);
            coeffs[slot] = (int16_t)((int)(dc ? dc[slot >> 6] : coeffs[slot]) + before);
        }
}

// ---- 0xFF00 -> 0xFF ----
constexpr int CHUNK = 64;
// S holds everything from the first byte of the scan to the end of the file (n_max bytes).  The entropy-coded segment ends before the
// first marker -- a 0xFF followed by anything but 0x00, or a 0xFF that is the file's last byte -- and the count kernel finds it on the
// way (first_marker: atomicMin, initialised to all ones), so the host never walks the scan: a file goes up as it is.
__global__ void scan_state_init_kernel(ScanState* st)
{
    if (threadIdx.x == 0) {
        ScanState z = {};
        z.last_bit = ~0ull;
        z.first_marker = ~0ull;
        *st = z;
    }
}
// Round 4: a thread fetches its 64 bytes with four 16-byte loads and walks them in registers (round 3: 64 single-byte loads per thread --
// 17 us for a 5 MB scan), and the copy launch compacts into LDS and stores whole words (round 3: a byte store per byte, 35 us).  The
// buffers are 64 bytes longer than the data (jpezy_read_jpeg_gpu), so the last chunk's loads stay inside them.
__device__ __forceinline__ void load_chunk(const uint8_t* S, size_t b0, uint32_t* w)
{
    const uint4* p = reinterpret_cast<const uint4*>(S + b0);
#pragma unroll
    for (int k = 0; k < CHUNK / 16; ++k) {
        const uint4 v = p[k];
        w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
    }
}
__global__ __launch_bounds__(256) void unstuff_count_kernel(const uint8_t* S, size_t n_max, uint32_t* counts, unsigned long long* first_marker)
{
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b0 = c * CHUNK;
    if (b0 >= n_max) return;
    uint32_t w[CHUNK / 4];
    load_chunk(S, b0, w);
    unsigned cnt = 0;
    uint8_t prev = b0 ? S[b0 - 1] : 0;
    const unsigned e = (unsigned)(n_max - b0 < (size_t)CHUNK ? n_max - b0 : (size_t)CHUNK);
    unsigned marker = ~0u;                           // offset inside the chunk's neighbourhood: b0 + marker (may be -1: the byte before the chunk)
#pragma unroll
    for (unsigned k = 0; k < (unsigned)CHUNK; ++k) {
        const uint8_t v = (uint8_t)(w[k >> 2] >> ((k & 3u) * 8u));
        if (k < e) {
            if (v == 0x00 && prev == 0xFF) { ++cnt; prev = 0x01; }                  // FF 00 00: only the first zero is stuffing
            else {
                if (prev == 0xFF && marker == ~0u) marker = k;                       // (k: one more than the 0xFF's offset)
                prev = v;
            }
        }
    }
    unsigned long long mk = marker != ~0u ? (unsigned long long)b0 + marker - 1ull : ~0ull;
    if (b0 + e == n_max && prev == 0xFF && marker == ~0u) mk = n_max - 1;
    // (counts of chunks behind the marker are never used: only prefix sums up to the marker's chunk are)
    counts[c] = cnt;
    if (mk != ~0ull) atomicMin(first_marker, mk);
}
// A workgroup takes 256 consecutive chunks: their output is one contiguous byte range of U, assembled in LDS at the position it has there
// (shifted so that LDS words line up with the words of U) and stored as whole words; the partial first and last word go out as bytes (the
// neighbouring workgroups complete them).  The chunk that holds the segment's last byte leaves the totals in st.
__global__ __launch_bounds__(256) void unstuff_copy_kernel(const uint8_t* S, size_t n_max, const unsigned long long* removed_before, uint8_t* U,
                                                           ScanState* st)
{
    __shared__ uint32_t buf[256 * CHUNK / 4 + 4];
    __shared__ unsigned end_sh;
    uint8_t* const lb = reinterpret_cast<uint8_t*>(buf);
    const size_t c0 = (size_t)blockIdx.x * 256, c = c0 + threadIdx.x;
    const size_t b0 = c * CHUNK;
    const unsigned long long fm = st->first_marker;
    const size_t n = fm < n_max ? (size_t)fm : n_max;
    if (c0 * CHUNK >= n) return;                                                     // workgroup-uniform
    const unsigned long long rb0 = removed_before[c0];
    uint8_t* const P = U + c0 * CHUNK - rb0;                                         // first output byte of the workgroup
    const unsigned shift = (unsigned)(reinterpret_cast<uintptr_t>(P) & 3u);
    if (b0 < n) {
        const unsigned long long rb = removed_before[c];
        uint32_t w[CHUNK / 4];
        load_chunk(S, b0, w);
        uint8_t prev = b0 ? S[b0 - 1] : 0;
        const unsigned e = (unsigned)(n - b0 < (size_t)CHUNK ? n - b0 : (size_t)CHUNK);
        uint8_t* dst = lb + shift + (b0 - c0 * CHUNK) - (size_t)(rb - rb0);
        unsigned cnt = 0;
#pragma unroll
        for (unsigned k = 0; k < (unsigned)CHUNK; ++k) {
            const uint8_t v = (uint8_t)(w[k >> 2] >> ((k & 3u) * 8u));
            if (k < e) {
                if (v == 0x00 && prev == 0xFF) { prev = 0x01; ++cnt; }
                else { *dst++ = v; prev = v; }
            }
        }
        const bool last = b0 + e == n;
        if (last) {                                  // the chunk that holds the segment's last byte
            st->removed = rb + cnt;
            st->n_sub = (unsigned)((((unsigned long long)n - (rb + cnt)) * 8 + SUBSEQ_BITS - 1) / SUBSEQ_BITS);
        }
        if (last || threadIdx.x == 255) end_sh = (unsigned)(dst - lb);
    }
    __syncthreads();
    const unsigned end = end_sh, nwords = (end + 3) / 4;
    uint32_t* const A = reinterpret_cast<uint32_t*>(P - shift);                      // 4-byte aligned
    for (unsigned wd = threadIdx.x; wd < nwords; wd += 256) {
        const unsigned lo = wd * 4, hi = lo + 4;
        if (lo >= shift && hi <= end) {
            A[wd] = buf[wd];
        } else {
            for (unsigned k = lo < shift ? shift : lo; k < (hi < end ? hi : end); ++k) reinterpret_cast<uint8_t*>(A)[k] = lb[k];
        }
    }
}

// ======================================================================================================
// Batch form: the same kernels with the file as an index (jpezy_huffdec.h).  A single 1080p file keeps 80 waves busy
// for a chain of launches that is pure latency; 256 of them fill the chip for the same chain.
// ======================================================================================================
__device__ __forceinline__ unsigned file_of_chunk(const BatchFile* F, unsigned n_files, unsigned c)
{
    unsigned lo = 0, hi = n_files;                  // F[lo].chunk0 <= c < F[hi].chunk0
    while (hi - lo > 1) {
        const unsigned mid = (lo + hi) >> 1;
        if (F[mid].chunk0 <= c) lo = mid; else hi = mid;
    }
    return lo;
}

// F[f].n_bytes may be an upper bound (the file from its first scan byte to its end): as in the single-scan kernels the count launch finds
// where the entropy-coded segment ends (F[f].first_marker, atomicMin, set to all ones by the host) and the copy launch stops there.
__global__ __launch_bounds__(256) void unstuff_count_batch_kernel(const uint8_t* S, BatchFile* F, unsigned n_files, unsigned total_chunks,
                                                                  uint32_t* counts)
{
    const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= total_chunks) return;
    const unsigned f = file_of_chunk(F, n_files, c);
    const size_t base = (size_t)F[f].chunk0 * CHUNK, n = F[f].n_bytes;
    const size_t b0 = (size_t)(c - F[f].chunk0) * CHUNK;
    unsigned cnt = 0, marker = ~0u;
    if (b0 < n) {
        uint8_t prev = b0 ? S[base + b0 - 1] : 0;
        const size_t e = b0 + CHUNK < n ? b0 + CHUNK : n;
        for (size_t i = b0; i < e; ++i) {
            const uint8_t v = S[base + i];
            if (v == 0x00 && prev == 0xFF) { ++cnt; prev = 0x01; }
            else {
                if (prev == 0xFF && marker == ~0u) marker = (unsigned)(i - 1);
                prev = v;
            }
        }
        if (e == n && prev == 0xFF && marker == ~0u) marker = (unsigned)(n - 1);
    }
    counts[c] = cnt;
    if (marker != ~0u) atomicMin(&F[f].first_marker, marker);
}

__global__ __launch_bounds__(256) void unstuff_copy_batch_kernel(const uint8_t* S, BatchFile* F, unsigned n_files, unsigned total_chunks,
                                                                 const unsigned long long* removed_before, uint8_t* U)
{
    const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= total_chunks) return;
    const unsigned f = file_of_chunk(F, n_files, c);
    const size_t base = (size_t)F[f].chunk0 * CHUNK;
    const size_t n = F[f].first_marker < F[f].n_bytes ? F[f].first_marker : F[f].n_bytes;       // the segment ends at its first marker
    const size_t b0 = (size_t)(c - F[f].chunk0) * CHUNK;
    const unsigned long long rb0 = removed_before[F[f].chunk0];
    if (c == F[f].chunk0) {                          // the file's first chunk resets its flags (n_sub and removed stay 0 for an empty segment)
        F[f].changed[0] = F[f].changed[1] = F[f].changed[2] = F[f].changed[3] = 0;
        F[f].error = 0;
        F[f].last_bit = ~0ull;
    }
    if (b0 >= n) return;
    const unsigned long long rb = removed_before[c] - rb0;
    uint8_t* dst = U + F[f].u_off + b0 - rb;
    uint8_t prev = b0 ? S[base + b0 - 1] : 0;
    const size_t e = b0 + CHUNK < n ? b0 + CHUNK : n;
    unsigned cnt = 0;
    for (size_t i = b0; i < e; ++i) {
        const uint8_t v = S[base + i];
        if (v == 0x00 && prev == 0xFF) { prev = 0x01; ++cnt; continue; }
        *dst++ = v;
        prev = v;
    }
    if (e == n) {                                    // the chunk that holds the segment's last byte publishes the file's totals
        const unsigned removed = (unsigned)rb + cnt;
        F[f].removed = removed;
        F[f].n_sub = (unsigned)((((unsigned long long)n - removed) * 8 + SUBSEQ_BITS - 1) / SUBSEQ_BITS);
    }
}


__global__ __launch_bounds__(WGS) void spec_batch_kernel(const Setup* setups, const uint32_t* U, const BatchFile* F, const unsigned* wg_file,
                                                         const unsigned* wg_first, unsigned long long* proposal, unsigned overflow)
{
    __shared__ Setup S;
    __shared__ uint32_t win[WINDOW_WORDS];
    const unsigned f = wg_file[blockIdx.x], i0 = wg_first[blockIdx.x], n_sub = F[f].n_sub;
    if (i0 >= n_sub) return;                                            // (workgroup-uniform: the slots were sized before unstuffing)
    load_setup(S, setups + F[f].setup);
    load_window(win, U + F[f].u_off / 4, i0, F[f].u_words);
    __syncthreads();
    const unsigned i = i0 + threadIdx.x;
    if (i >= n_sub) return;
    unsigned long long* prop = proposal + F[f].sub0;
    const uint16_t* tabs = reinterpret_cast<const uint16_t*>(&S);
    const unsigned bpm = (unsigned)S.bpm, tdmask = S.tdmask;
    Cursor c;
    c.init(win, threadIdx.x * SUBSEQ_BITS);
    Walk wk;
    wk.init(0, 0, tdmask);
    for (unsigned r = 0; r <= overflow && i + r < n_sub; ++r) {
        const unsigned end = (threadIdx.x + r + 1) * SUBSEQ_BITS;
        run_subsequence(tabs, bpm, tdmask, c, wk, end);
        const unsigned long long rank = i == 0 ? 0xFFFFull : r;
        atomicMax(prop + i + r, (rank << 32) | pack_state(c.pos - end, wk.b(), wk.k));
    }
}

__global__ __launch_bounds__(WGS) void sync_batch_kernel(const Setup* setups, const uint32_t* U, BatchFile* F, const unsigned* wg_file,
                                                         const unsigned* wg_first, const unsigned* active, uint32_t* exit_state_all,
                                                         uint32_t* last_entry_all, unsigned* nblocks_all, int max_inner)
{
    __shared__ Setup S;
    __shared__ uint32_t win[WINDOW_WORDS];
    __shared__ uint32_t sh_exit[WGS + 1];
    const unsigned f = wg_file[blockIdx.x], i0 = wg_first[blockIdx.x], n_sub = F[f].n_sub;
    if (i0 >= n_sub || !active[f]) return;                              // workgroup-uniform
    uint32_t* exit_state = exit_state_all + F[f].sub0;
    uint32_t* last_entry = last_entry_all + F[f].sub0;
    unsigned* nblocks_out = nblocks_all + F[f].sub0;
    const unsigned t = threadIdx.x, i = i0 + t;
    const bool live = i < n_sub;
    uint32_t my_last = live ? last_entry[i] : 0u, my_exit = live ? exit_state[i] : 0u;
    const uint32_t exit_before = my_exit;
    unsigned nb = live ? nblocks_out[i] : 0u;
    sh_exit[t + 1] = my_exit;
    if (t == 0) sh_exit[0] = i0 ? exit_state[i0 - 1] : pack_state(0, 0, 0);
    __syncthreads();
    if (!__syncthreads_or(live && sh_exit[t] != my_last)) return;      // a quiet workgroup
    load_setup(S, setups + F[f].setup);
    load_window(win, U + F[f].u_off / 4, i0, F[f].u_words);
    __syncthreads();
    const uint16_t* tabs = reinterpret_cast<const uint16_t*>(&S);
    const unsigned bpm = (unsigned)S.bpm, tdmask = S.tdmask;
    bool first_moved = false;                              // this lane's first decode of the launch left another state than it had
    bool same_state = false;
    for (int inner = 0; inner < max_inner; ++inner) {
        const uint32_t entry = sh_exit[t];
        bool redo = live && entry != my_last;
        __syncthreads();
        if (redo) {
            const bool first_walk = my_last == 0xFFFFFFFFu;
            my_last = entry;
            nb = 0;
            if (entry & 0x40000000u) {
                my_exit = STATE_ERR;
            } else {
                Cursor c;
                c.init(win, t * SUBSEQ_BITS + ((entry >> 16) & 0x3FFFu));
                Walk wk;
                wk.init((entry >> 8) & 0xFFu, entry & 0xFFu, tdmask);
                const unsigned end = (t + 1) * SUBSEQ_BITS;
                run_subsequence(tabs, bpm, tdmask, c, wk, end);
                nb = wk.nblocks;
                my_exit = pack_state(c.pos - end, wk.b(), wk.k);
                same_state = same_state || (first_walk && my_exit == entry);
            }
            redo = sh_exit[t + 1] != my_exit;
            sh_exit[t + 1] = my_exit;
            first_moved = first_moved || (inner == 0 && redo);
        }
        if (!__syncthreads_or(redo)) break;
    }
    {   // lanes whose very first walk left as it entered: a stream with a period that divides the subsequence (see RefineBudget)
        const unsigned long long same = __builtin_amdgcn_ballot_w64(same_state);
        if (same && (t & 63u) == (unsigned)__builtin_ctzll(same)) atomicAdd(&F[f].periodic, (unsigned)__builtin_popcountll(same));
    }
    if (live) {
        last_entry[i] = my_last;
        nblocks_out[i] = nb;
        const bool pending = sh_exit[t] != my_last;
        if (my_exit != exit_before) {
            exit_state[i] = my_exit;
            atomicAdd(&F[f].changed[0], 1u);
            if (t == WGS - 1 && i + 1 < n_sub) atomicAdd(&F[f].changed[2], 1u);
        }
        if (pending) atomicAdd(&F[f].changed[1], 1u);
        if (first_moved) atomicAdd(&F[f].changed[3], 1u);
    }
}

__global__ __launch_bounds__(WGS) void emit_batch_kernel(const Setup* setups, const uint32_t* U, BatchFile* F, const unsigned* wg_file,
                                                         const unsigned* wg_first, const unsigned* active, const uint32_t* exit_state_all,
                                                         const unsigned long long* blocks_before_all, int16_t* coeffs)
{
    __shared__ Setup S;
    __shared__ uint32_t win[WINDOW_WORDS];
    const unsigned f = wg_file[blockIdx.x], i0 = wg_first[blockIdx.x], n_sub = F[f].n_sub;
    if (i0 >= n_sub || !active[f]) return;                              // active: here "the file converged" (set by the host)
    load_setup(S, setups + F[f].setup);
    load_window(win, U + F[f].u_off / 4, i0, F[f].u_words);
    __syncthreads();
    const unsigned i = i0 + threadIdx.x;
    if (i >= n_sub) return;
    const uint32_t* exit_state = exit_state_all + F[f].sub0;
    const unsigned long long g0 = blocks_before_all[F[f].sub0 + i] - blocks_before_all[F[f].sub0];
    if (g0 >= F[f].total_blocks) return;              // (the stream's own count: restart intervals share a Setup)
    const uint32_t entry = i ? exit_state[i - 1] : pack_state(0, 0, 0);
    if (entry & 0x40000000u) { F[f].error = 1u; return; }
    int16_t* out = coeffs + F[f].coeff_off;
    const uint16_t* tabs = reinterpret_cast<const uint16_t*>(&S);
    const unsigned bpm = (unsigned)S.bpm, tdmask = S.tdmask, total = F[f].total_blocks;
    Cursor c;
    c.init(win, threadIdx.x * SUBSEQ_BITS + ((entry >> 16) & 0x3FFFu));
    Walk wk;
    wk.init((entry >> 8) & 0xFFu, entry & 0xFFu, tdmask);
    const unsigned end = (threadIdx.x + 1) * SUBSEQ_BITS;
    while (c.pos < end && g0 + wk.nblocks < total) {
        if (!decode_step<true>(tabs, bpm, tdmask, c, wk, g0, total, out)) { F[f].error = 1u; return; }
    }
    if (g0 + wk.nblocks >= total) F[f].last_bit = (unsigned long long)i0 * SUBSEQ_BITS + c.pos;
}

// DC differences -> absolute values (pre_DC, ref :611-614): one workgroup per (file, component) walks the component's DC terms in
// scan order, 2048 per step with a running carry
__global__ __launch_bounds__(256) void dc_prefix_batch_kernel(int16_t* coeffs, const BatchFile* F, const unsigned* active)
{
    __shared__ long long wsum[4];
    const unsigned f = blockIdx.x, comp = blockIdx.y;
    if (comp >= F[f].ncomp || !active[f]) return;
    const unsigned count = F[f].ccount[comp], start = F[f].cstart[comp], bpm = F[f].bpm;
    const size_t nd = (size_t)F[f].nmcu * count;
    int16_t* co = coeffs + F[f].coeff_off;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long carry = 0;
    for (size_t j0 = 0; j0 < nd; j0 += 2048) {
        long long v[8], s = 0;
        const size_t jb = j0 + (size_t)threadIdx.x * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const size_t j = jb + q;
            v[q] = 0;
            if (j < nd) {
                const size_t mcu = j / count, t = j - mcu * count;
                v[q] = co[(mcu * bpm + start + t) * 64];
            }
            s += v[q];
        }
        long long inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        long long woff = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < wv) woff += wsum[q];
            tot += wsum[q];
        }
        long long run = carry + woff + inc - s;        // sum of everything before this thread's first element
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const size_t j = jb + q;
            run += v[q];
            if (j < nd) {
                const size_t mcu = j / count, t = j - mcu * count;
                co[(mcu * bpm + start + t) * 64] = (int16_t)run;
            }
        }
        carry += tot;
        __syncthreads();
    }
}

// Lane-per-stream form for SHORT streams (restart intervals of a few MCUs: hardware encoders, libjpeg -restart N B): a stream that starts in
// the known state needs no speculation when one lane walks all of it -- symbols, coefficients and DC predictors in one pass, straight from
// the unstuffed stream in global memory, the tables in LDS.  256 streams per workgroup; worth it while a stream is a few KB (a lane decodes
// ~3 MB/s), beyond that the subsequence form above wins.  F[f].error / last_bit as in the emit kernel.
__global__ __launch_bounds__(WGS) void stream_per_lane_kernel(const Setup* setups, const uint32_t* U, BatchFile* F, unsigned n_files, int16_t* coeffs)
{
    __shared__ Setup S;
    load_setup(S, setups);                                               // (one set of tables for the whole launch: the intervals of one scan)
    __syncthreads();
    const unsigned f = blockIdx.x * WGS + threadIdx.x;
    if (f >= n_files) return;
    const uint16_t* tabs = reinterpret_cast<const uint16_t*>(&S);
    const unsigned bpm_period = (unsigned)S.bpm, tdmask = S.tdmask, total = F[f].total_blocks;
    int16_t* out = coeffs + F[f].coeff_off;
    GlobalCursor c;
    c.init(U + F[f].u_off / 4, F[f].u_words);
    Walk wk;
    wk.init(0, 0, tdmask);
    while (wk.nblocks < total) {
        if (!decode_step<true>(tabs, bpm_period, tdmask, c, wk, 0ull, total, out)) { F[f].error = 1u; return; }
        if (c.pos > F[f].u_words * 32u) { F[f].error = 1u; return; }     // (ran off the buffer on zeros: cannot be a complete stream)
    }
    F[f].last_bit = c.pos;
    // DC differences -> values (pre_DC, ref :611-614), the predictors at zero at the start of the stream
    const unsigned bpm = F[f].bpm, c1 = F[f].cstart[1], c2 = F[f].cstart[2], ncomp = F[f].ncomp;
    int pred[3] = { 0, 0, 0 };
    for (unsigned blk = 0, b = 0; blk < total; ++blk) {
        const unsigned comp = (ncomp > 2 && b >= c2) ? 2u : (ncomp > 1 && b >= c1) ? 1u : 0u;
        const int v = pred[comp] + out[(size_t)blk * 64];
        pred[comp] = v;
        out[(size_t)blk * 64] = (int16_t)v;
        b = b + 1 == bpm ? 0 : b + 1;
    }
}

// ---- launchers ----
hipError_t launch_stream_per_lane(const Setup* setups, const uint32_t* U, BatchFile* F, unsigned n_files, int16_t* coeffs, hipStream_t s)
{
    if (!n_files) return hipSuccess;
    hipLaunchKernelGGL(stream_per_lane_kernel, dim3((n_files + WGS - 1) / WGS), dim3(WGS), 0, s, setups, U, F, n_files, coeffs);
    return hipGetLastError();
}
unsigned subseq_bits() { return SUBSEQ_BITS; }
size_t chunk_bytes() { return CHUNK; }

hipError_t launch_scan_state_init(ScanState* st, hipStream_t s)
{
    hipLaunchKernelGGL(scan_state_init_kernel, dim3(1), dim3(64), 0, s, st);
    return hipGetLastError();
}
hipError_t launch_unstuff_count(const uint8_t* S, size_t n_max, uint32_t* counts, ScanState* st, hipStream_t s)
{
    const size_t nc = (n_max + CHUNK - 1) / CHUNK;
    if (!nc) return hipSuccess;
    hipLaunchKernelGGL(unstuff_count_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, S, n_max, counts, &st->first_marker);
    return hipGetLastError();
}
hipError_t launch_unstuff_copy(const uint8_t* S, size_t n_max, const unsigned long long* removed_before, uint8_t* U, ScanState* st, hipStream_t s)
{
    const size_t nc = (n_max + CHUNK - 1) / CHUNK;
    if (!nc) return hipSuccess;
    hipLaunchKernelGGL(unstuff_copy_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, S, n_max, removed_before, U, st);   // (a workgroup per 256 chunks, as before)
    return hipGetLastError();
}
static unsigned spec_overflow();
hipError_t launch_speculate(const Setup* S, const uint32_t* U, size_t u_words, unsigned n_sub, const ScanState* st, uint32_t* exit_state,
                            uint32_t* last_entry, unsigned* nblocks, uint32_t* marks, unsigned* mark_blocks, hipStream_t s)
{
    hipLaunchKernelGGL(spec_kernel, dim3((n_sub + WGS - 1) / WGS), dim3(WGS), 0, s, S, U, u_words, st, exit_state, last_entry, nblocks, marks, mark_blocks,
                       spec_overflow());
    return hipGetLastError();
}
unsigned emit_parts() { return EMIT_PARTS; }
hipError_t launch_sync(const Setup* S, const uint32_t* U, size_t u_words, unsigned n_sub, const ScanState* st, uint32_t* exit_state, uint32_t* last_entry,
                       unsigned* nblocks, uint32_t* marks, unsigned* mark_blocks, unsigned* changed, const unsigned* prev, int max_inner, hipStream_t s)
{
    hipLaunchKernelGGL(sync_kernel, dim3((n_sub + WGS - 1) / WGS), dim3(WGS), 0, s, S, U, u_words, st, exit_state, last_entry, nblocks, marks, mark_blocks,
                       changed, prev,
                       max_inner < 1 ? 1 : max_inner > WGS + 1 ? WGS + 1 : max_inner);
    return hipGetLastError();
}
hipError_t launch_emit(const Setup* S, const uint32_t* U, size_t u_words, unsigned n_sub, ScanState* st, const uint32_t* exit_state,
                       const uint32_t* marks, const unsigned* mark_blocks, const unsigned long long* blocks_before, int16_t* out, int16_t* dc_out,
                       bool guarded, hipStream_t s)
{
    hipLaunchKernelGGL(emit_kernel, dim3((n_sub + EMIT_SUBS - 1) / EMIT_SUBS), dim3(WGS), 0, s, S, U, u_words, st, exit_state, marks, mark_blocks,
                       blocks_before, out, dc_out, guarded ? 1 : 0);
    return hipGetLastError();
}
size_t dc_prefix_scratch_ints(size_t nmcu, unsigned max_count) { return 3 * ((nmcu * max_count + DC_PER_WG - 1) / DC_PER_WG + 1); }
hipError_t launch_dc_prefix(int16_t* coeffs, int16_t* dc, unsigned bpm, unsigned ncomp, const unsigned cstart[3], const unsigned ccount[3], size_t nmcu,
                            int* scratch, const ScanState* guard, hipStream_t s)
{
    DcGeom g;
    g.bpm = bpm; g.ncomp = ncomp; g.nmcu = nmcu;
    unsigned maxc = 1;
    for (unsigned q = 0; q < 3; ++q) { g.cstart[q] = q < ncomp ? cstart[q] : 0; g.ccount[q] = q < ncomp ? ccount[q] : 1; maxc = g.ccount[q] > maxc ? g.ccount[q] : maxc; }
    const size_t wgs = (nmcu * maxc + DC_PER_WG - 1) / DC_PER_WG;
    if (!wgs || !ncomp) return hipSuccess;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dc_local_kernel, dim3((unsigned)wgs, ncomp), dim3(256), 0, s, coeffs, dc, g, scratch, (unsigned)wgs + 1, guard);
    static const size_t self_sum_max = [] {
        const char* e = std::getenv("JPEZY_DC_SELF_SUM_MAX");        // development / test knob: 0 forces the three-launch form
        return e ? (size_t)std::atoll(e) : (size_t)DC_SELF_SUM_MAX;
    }();
    if (wgs <= self_sum_max) {
        hipLaunchKernelGGL(dc_add_kernel<true>, dim3((unsigned)wgs, ncomp), dim3(256), 0, s, coeffs, (const int16_t*)dc, g, (const int*)scratch, (unsigned)wgs + 1, guard);
    } else {
        hipLaunchKernelGGL(dc_totals_kernel, dim3(ncomp), dim3(256), 0, s, scratch, g, (unsigned)wgs + 1, guard);
        hipLaunchKernelGGL(dc_add_kernel<false>, dim3((unsigned)wgs, ncomp), dim3(256), 0, s, coeffs, (const int16_t*)dc, g, (const int*)scratch, (unsigned)wgs + 1, guard);
    }
    return hipGetLastError();
}


static unsigned spec_overflow()      // speculation distance in subsequences: one place for the single-scan and the batch form
{
    static const unsigned overflow = [] {
        const char* e = std::getenv("JPEZY_HUFFDEC_OVERFLOW");      // development knob; the default covers what was measured
        const int v = e ? std::atoi(e) : OVERFLOW_DEFAULT;
        return (unsigned)(v < 0 ? 0 : v > OVERFLOW ? OVERFLOW : v);
    }();
    return overflow;
}

hipError_t launch_unstuff_count_batch(const uint8_t* S, BatchFile* F, unsigned n_files, unsigned total_chunks, uint32_t* counts, hipStream_t s)
{
    if (!total_chunks) return hipSuccess;
    hipLaunchKernelGGL(unstuff_count_batch_kernel, dim3((total_chunks + 255) / 256), dim3(256), 0, s, S, F, n_files, total_chunks, counts);
    return hipGetLastError();
}
hipError_t launch_unstuff_copy_batch(const uint8_t* S, BatchFile* F, unsigned n_files, unsigned total_chunks, const unsigned long long* removed_before,
                                     uint8_t* U, hipStream_t s)
{
    if (!total_chunks) return hipSuccess;
    hipLaunchKernelGGL(unstuff_copy_batch_kernel, dim3((total_chunks + 255) / 256), dim3(256), 0, s, S, F, n_files, total_chunks, removed_before, U);
    return hipGetLastError();
}
hipError_t launch_speculate_batch(const Setup* setups, const uint32_t* U, const BatchFile* F, const unsigned* wg_file, const unsigned* wg_first,
                                  unsigned n_wg, unsigned n_slots, unsigned long long* proposal, uint32_t* exit_state, hipStream_t s)
{
    if (!n_wg) return hipSuccess;
    hipError_t e = hipMemsetAsync(proposal, 0, (size_t)n_slots * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(spec_batch_kernel, dim3(n_wg), dim3(WGS), 0, s, setups, U, F, wg_file, wg_first, proposal, spec_overflow());
    hipLaunchKernelGGL(adopt_proposals_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, s, proposal, n_slots, exit_state);
    return hipGetLastError();
}
__global__ void reset_changed_batch_kernel(BatchFile* F, const unsigned* active, unsigned n_files)
{
    const unsigned f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n_files && active[f]) F[f].changed[0] = F[f].changed[1] = F[f].changed[2] = F[f].changed[3] = 0;
}
hipError_t launch_reset_changed_batch(BatchFile* F, const unsigned* active, unsigned n_files, hipStream_t s)
{
    if (!n_files) return hipSuccess;
    hipLaunchKernelGGL(reset_changed_batch_kernel, dim3((n_files + 255) / 256), dim3(256), 0, s, F, active, n_files);
    return hipGetLastError();
}
hipError_t launch_sync_batch(const Setup* setups, const uint32_t* U, BatchFile* F, const unsigned* wg_file, const unsigned* wg_first, unsigned n_wg,
                             const unsigned* active, uint32_t* exit_state, uint32_t* last_entry, unsigned* nblocks, int max_inner, hipStream_t s)
{
    if (!n_wg) return hipSuccess;
    hipLaunchKernelGGL(sync_batch_kernel, dim3(n_wg), dim3(WGS), 0, s, setups, U, F, wg_file, wg_first, active, exit_state, last_entry, nblocks,
                       max_inner < 1 ? 1 : max_inner > WGS + 1 ? WGS + 1 : max_inner);
    return hipGetLastError();
}
hipError_t launch_emit_batch(const Setup* setups, const uint32_t* U, BatchFile* F, const unsigned* wg_file, const unsigned* wg_first, unsigned n_wg,
                             const unsigned* active, const uint32_t* exit_state, const unsigned long long* blocks_before, int16_t* coeffs, hipStream_t s)
{
    if (!n_wg) return hipSuccess;
    hipLaunchKernelGGL(emit_batch_kernel, dim3(n_wg), dim3(WGS), 0, s, setups, U, F, wg_file, wg_first, active, exit_state, blocks_before, coeffs);
    return hipGetLastError();
}
hipError_t launch_dc_prefix_batch(int16_t* coeffs, const BatchFile* F, const unsigned* active, unsigned n_files, hipStream_t s)
{
    if (!n_files) return hipSuccess;
    hipLaunchKernelGGL(dc_prefix_batch_kernel, dim3(n_files, 3), dim3(256), 0, s, coeffs, F, active);
    return hipGetLastError();
}

}  // namespace huffdec
}  // namespace jpezy_dev
