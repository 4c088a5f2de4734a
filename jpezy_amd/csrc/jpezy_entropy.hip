// jpezy_entropy.hip -- the encoder's serial tail on the GPU: Annex-K Huffman coding of the
// zig-zagged coefficients, bit packing, 0xFF00 byte stuffing.  Same bytes as jpezy_host::write_jpeg / the reference's
// encoder::encode_huffman + bofstream (ref encoder/jpezy_encoder.hpp:174-242): MSB-first bits, DC predictors per
// component never reset, EOB only when a block ends in zeros, ZRL for runs over 15, zero pad bits before EOI.
//
// What is serial in the reference is the bit cursor and pre_DC[3].  Neither is a true dependency:
//   * pre_DC of a block is the DC of the previous block of the same component, which is simply read;
//   * the bit cursor is an exclusive prefix sum of the blocks' code lengths.
// Pipeline for a chunk of frames (all launches on one stream; "tile" = the 256 coded blocks of one workgroup):
//   1. code_tiles_kernel   one lane per coded block, every block coded ONCE: private stream in the lane's LDS row, scan of the
//                          256 lengths, shift-copy into the tile's stream (global scratch S, MSB-first words)
//   2. tile_bases_kernel   one workgroup per frame: prefix sums of the tile totals (bit offsets), stream length, the first
//                          tile of every 16 KB of output; latches and clears the frame's error flag
//   3. assemble_kernel     one thread per 64-byte chunk of the unstuffed stream U: funnel shifts out of one or two tile
//                          streams, and the chunk's 0xFF count (tile-local offsets + tile totals)
//   4. scan                of the 0xFF tile totals (one workgroup for a few thousand)
//   5. stuff_kernel        copies U to the output inserting 0x00 after every 0xFF; in the device-resident form it also
//                          decides fit / size, writes EOI and copies the JFIF header (workgroup 0 of each frame)
// (Round 1 and the first half of round 2 coded every block twice -- lengths, scan, then bits at the scanned offset with
// atomicOr into a zeroed buffer -- in 22, then 9 launches: 174 / 100 us per 4096x4096 frame; see DESIGN.md.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jpezy_entropy.h"
#include "../../include/jpezy_constants.h"   // JPEZY_PAD_BIT

namespace jpezy_dev {
namespace entropy {

// code tables: entry = (code << 8) | length; dc[t][category], ac[t][(run << 4) | size]   (t: 0 luma, 1 chroma)
struct alignas(16) LdsTables {
    uint32_t dc[2][16];
    uint32_t ac[2][256];
#ifdef JPEZY_ENT_NOFAST
    uint32_t fast[2][4];
#else
    uint32_t fast[2][1024];         // CodeTables::fast
#endif
};
static_assert(sizeof(LdsTables) <= sizeof(CodeTables) && sizeof(LdsTables) % 16 == 0, "table images must match");

// (no barrier: the caller's own barrier after staging its tile covers the tables)
__device__ __forceinline__ void load_tables(LdsTables& L, const CodeTables* T)
{
    const uint4* src = reinterpret_cast<const uint4*>(T);
    uint4* dst = reinterpret_cast<uint4*>(&L);
    for (unsigned i = threadIdx.x; i < sizeof(LdsTables) / 16; i += blockDim.x) dst[i] = src[i];
}

// exclusive prefix sum of one value per thread over a 256-thread workgroup; *total = the workgroup's sum (all threads)
__device__ __forceinline__ uint32_t wg256_exclusive_scan(uint32_t v, uint32_t* total)
{
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < wv) woff += wsum[k];
        tot += wsum[k];
    }
    *total = tot;
    return woff + inc - v;
}

// codes of one block (ref :174-225).  z: its 64 zig-zag coefficients (nullptr: an all-zero block), pred: DC of the previous
// block of the component; the codes go to the writer w, which also knows how many bits it has taken.  Returns false for
// values outside the Annex-K tables (the reference throws, the host writer returns JPEZY_E_FORMAT); such a value is coded as
// the largest size, so the stream stays well formed and the frame is flagged.
// The reference walks all 63 AC positions and counts zeros; here the lane first forms the 63-bit mask of its block's
// non-zero AC coefficients (MSB = zig-zag position 1 ... so that the next coefficient is a count-leading-zeros away) and
// then visits only those: the run before a coefficient is the gap between two set bits.  A wave's loop runs as long as
// its fullest block has non-zero coefficients and every iteration does the same work in every lane.  Branch-light on
// purpose: no early exit, the only branch is the rare run over 15 (ZRL codes), and the next coefficient is requested before
// the current one is coded -- an iteration waits for one LDS round trip (the code-table lookup), not two.  The kernel is bound
// by VALU issue (2,100 instructions per wave on noise, ~50 per non-zero coefficient): whatever can be derived after the
// loop (the length from the writer's cursor, the range check from a running maximum) is not tracked inside it.
// AC coefficients whose positions are the set bits of m (bit 31 - k: position base + k), in order.
template <class W>
__device__ __forceinline__ void code_ac(uint32_t m, int base, const int16_t* z, int& prev, const uint32_t* ac, const uint32_t* fast,
                                        uint32_t zrl, W& w, unsigned& amax)
{
    int lz = __builtin_clz(m | 1u);
    int vnext = z[base + lz];
#pragma unroll 1
    while (m) {
        const int n = base + lz, v = vnext;
        m &= 0x7FFFFFFFu >> lz;
        lz = __builtin_clz(m | 1u);             // 31 when nothing is left: a harmless in-bounds read
        vnext = z[base + lz];
        w.read_up_to(n);                         // (the in-place writer's licence: positions up to n are dead)
        int run = n - prev - 1;
        prev = n;
        // Round 3: the code of (run, size(v)) with the value bits appended comes ready-made out of ONE table, for |v| < 32 and
        // runs up to 15 -- every lane looks it up unconditionally (an index outside the table reads some other LDS word or, beyond
        // the workgroup's allocation, zero: never used) and appends once.  Larger values and runs over 15 (ZRL) are rare on
        // dense content: the wave enters the general path only when one of its lanes needs it (a wave-uniform branch; the
        // branchy form -- fast path or general path per lane -- issued more instructions than the code it replaced).
        const unsigned v32 = (unsigned)(v + 32);
        uint32_t e = fast[((unsigned)run << 6) + v32];
#ifdef JPEZY_ENT_NOFAST
        const bool slow = true;
#else
        const bool slow = v32 > 63u || run > 15;
#endif
        {
            if (slow) {       // (a divergent branch is skipped by the whole wave when no lane takes it: s_cbranch_execz)
                const unsigned a = (unsigned)(v < 0 ? -v : v);
                amax = a > amax ? a : amax;
                int sz = 32 - __builtin_clz(a);
                sz = sz > 10 ? 10 : sz;
                if (run > 15) {                          // ZRL codes in front of this coefficient (ref :198-206)
                    for (int r = run >> 4; r > 0; --r) w.put(zrl >> 8, (int)(zrl & 0xFF));
                    run &= 15;
                }
                const uint32_t g = ac[(run << 4) | sz];
                // code and value bits: at most 16 + 10 bits, and 5 bits of length
                e = ((((g >> 8) << sz) | ((uint32_t)(v + (v >> 31)) & ((1u << sz) - 1u))) << 5) | ((g & 0xFF) + (uint32_t)sz);
            }
        }
        w.put(e >> 5, (int)(e & 31u));
    }
}

template <class W>
__device__ __forceinline__ bool code_block(const int16_t* z, int pred, const uint32_t* dc, const uint32_t* ac, const uint32_t* fast, W& w)
{
    unsigned amax = 0;
    bool ok = true;
    uint32_t mhi = 0, mlo = 0;          // bit (31 - n) of mhi: position n in 0..31 is non-zero; mlo likewise for 32..63
    int dcv = 0;
    if (z) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint4 c = reinterpret_cast<const uint4*>(z)[k];
            const uint32_t wd[4] = { c.x, c.y, c.z, c.w };
            uint32_t m8 = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m8 |= ((wd[j] & 0xFFFFu) ? 1u : 0u) << (7 - 2 * j);
                m8 |= ((wd[j] >> 16) ? 1u : 0u) << (6 - 2 * j);
            }
            if (k < 4) mhi |= m8 << (24 - 8 * k); else mlo |= m8 << (24 - 8 * (k - 4));
            if (k == 0) dcv = (int)(short)(c.x & 0xFFFFu);
        }
        mhi &= 0x7FFFFFFFu;             // position 0 is the DC
    }
    {
        const int diff = dcv - pred;
        const unsigned a = (unsigned)(diff < 0 ? -diff : diff);
        int di = a ? 32 - __builtin_clz(a) : 0;
        ok = di <= 11;
        di = di > 11 ? 11 : di;
        const uint32_t e = dc[di];
        // code and value bits in one append: at most 11 + 11 bits
        w.put(((e >> 8) << di) | ((uint32_t)(diff + (diff >> 31)) & ((1u << di) - 1u)), (int)(e & 0xFF) + di);
    }
    int prev = 0;                        // position of the previous non-zero coefficient (0: the DC)
    if (z) {
        const uint32_t zrl = ac[0xF0];
        code_ac(mhi, 0, z, prev, ac, fast, zrl, w, amax);
        code_ac(mlo, 32, z, prev, ac, fast, zrl, w, amax);
    }
    w.read_up_to(63);
    if (prev != 63) {                    // the block ends in zeros (or has no AC coefficient at all): EOB
        const uint32_t e = ac[0x00];
        w.put(e >> 8, (int)(e & 0xFF));
    }
    return ok && amax <= 1023u;
}

constexpr int CHUNK = 64;   // bytes of the unstuffed stream U per thread of the 0xFF counting / stuffing kernels

__device__ __forceinline__ unsigned count_ff(uint32_t x)
{
    const uint32_t z = ~x;                                             // 0xFF bytes of x are zero bytes of z
    uint32_t y = (z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    y = ~(y | z | 0x7F7F7F7Fu);                                         // 0x80 exactly in the zero bytes of z
    return (unsigned)__builtin_popcount(y);
}

// ---- one coding pass (round 2): code_tiles -> tile_bases -> assemble ----
// The two-pass form above codes every block twice (lengths, then bits at the scanned offset).  Here a lane codes its block
// ONCE, into its own LDS row, in place: the row holds the block's 64 coefficients from byte 16 on, the private stream grows
// from byte 0, and a word is only written where every coefficient under it has already been read (RowWriter::limit, fed by
// code_ac's read_up_to) -- on ordinary content the codes are far shorter than the coefficients they replace.  A block whose
// stream would overtake its unread coefficients or exceed 140 bytes is counted only and re-coded afterwards straight from
// global memory (DirectWriter; high-quality tables on noise).  After the workgroup's scan over the 256 lengths each lane
// shift-copies its words to the tile's stream in global memory.  No atomics and no zeroed buffer: a word belongs to the lane
// whose segment contains the word's first bit; complete words are plain stores; the owner of a partial last word pulls
// the missing bits from the first word of the following lanes' rows (blocks shorter than 32 bits are consumed whole).
// Tile streams are MSB-first uint32 words, zero padded to a word.  assemble_kernel then forms the frame's unstuffed stream
// U output-driven (one thread per 64-byte chunk: funnel shifts across tile borders, byte order swapped on the way out) and
// counts the 0xFF bytes of its chunk while it has them.
constexpr int WG = 256;                                          // coded blocks per workgroup ("tile")
constexpr int ROW = 144, ROW_DATA = 16, ROW_LAST_WORD = 34;      // private stream: words 0..34; word 35 (bytes 140..143): its length
constexpr unsigned TILE_STREAM_WORDS = 256 * 208 / 4;            // worst case of 208 bytes per block

struct RowWriter {
    unsigned long long acc;         // the youngest bit is bit 0; nacc < 32 valid bits between two appends
    int nacc, wj2, npos;            // wj2 = 2 * (words stored or skipped) - 6; npos = position of the coefficient being coded
    unsigned long long bad;         // wave mask (scalar registers): lanes with a due word beyond the licence (or beyond the row) --
                                    // their blocks are re-coded (DirectWriter)
    char* row12;                    // the row's base + 12: word (wj2 + 6) / 2 sits at row12 + 2 * wj2
    __device__ __forceinline__ void init(uint32_t* r)
    {
        row12 = reinterpret_cast<char*>(r) + 12; acc = 0; nacc = 0; bad = 0;
        wj2 = -6; npos = 0;             // bytes 0..15 (words 0..3) are free from the start: 2 * 3 - 6 <= 0
    }
    // positions below n have been read (n itself counts as unread).  Word j (bytes 4j..4j+3) covers coefficients below 2j - 6,
    // so it may be written while 2j - 6 <= n: the licence test compares two registers the loop has anyway (round 2 formed
    // limit = (n + 6) >> 1 per coefficient); n <= 63 gives word 34, the last one of the row's stream
    __device__ __forceinline__ void read_up_to(int n) { npos = n; }
    // one predicated store; everything else is arithmetic (nested branches here cost more than the coding itself)
    __device__ __forceinline__ void word(uint32_t v, bool due)
    {
        const unsigned long long dm = __builtin_amdgcn_ballot_w64(due), lm = __builtin_amdgcn_ballot_w64(wj2 <= npos);
        if (due && wj2 <= npos) *reinterpret_cast<uint32_t*>(row12 + 2 * wj2) = v;
        bad |= dm & ~lm;
        wj2 += due ? 2 : 0;
    }
    __device__ __forceinline__ void put(uint32_t bits, int n)   // n <= 31
    {
        acc = (acc << n) | bits;
        const int t = nacc + n;          // < 64
        nacc = t & 31;
        word((uint32_t)(acc >> nacc), t >= 32);
    }
    __device__ __forceinline__ unsigned bits() const { return 16u * (unsigned)(wj2 + 6) + (unsigned)nacc; }   // before finish()
    __device__ __forceinline__ void finish()
    {
        word((uint32_t)(acc << (32 - nacc)), nacc > 0);          // left aligned, zero padded (nacc < 32)
    }
    __device__ __forceinline__ bool overflowed() const { return (bad >> (threadIdx.x & 63u)) & 1ull; }   // after finish()
};

// second coding pass of the rare block that did not fit its row: bits straight to the tile stream at their final place
struct DirectWriter {
    unsigned long long acc;
    int nacc;
    unsigned w;
    uint32_t* S;
    bool skip;                           // the first word starts in an earlier lane's segment: not ours to store
    __device__ __forceinline__ void init(uint32_t* tile_stream, unsigned bitoff)
    {
        S = tile_stream; w = bitoff >> 5; nacc = (int)(bitoff & 31u); acc = 0; skip = nacc != 0;
    }
    __device__ __forceinline__ void read_up_to(int) {}
    __device__ __forceinline__ void put(uint32_t bits, int n)
    {
        acc = (acc << n) | bits;
        nacc += n;
        if (nacc >= 32) {
            nacc -= 32;
            if (!skip) S[w] = (uint32_t)(acc >> nacc);
            skip = false;
            ++w;
        }
    }
};

__global__ __launch_bounds__(WG) void code_tiles_kernel(Job job, uint32_t* S, uint32_t* tile_total, unsigned* status)
{
    __shared__ LdsTables L;
    __shared__ __attribute__((aligned(16))) char tile[WG * ROW];
    load_tables(L, job.tables);
    const unsigned tid = threadIdx.x, frame = blockIdx.y;
    const unsigned nblk = job.blocks_per_frame, g0 = blockIdx.x * (unsigned)WG;
    const unsigned nb = nblk - g0 < (unsigned)WG ? nblk - g0 : (unsigned)WG;          // coded blocks of this tile
    const int16_t* fc = job.coeffs + (size_t)frame * job.coeffs_per_frame;
    // this lane's block: its place in global memory (nullptr: a zero chroma block of gray mode) and its DC predictor -- the
    // previous block of the same component in scan order, read from global memory (it may be another tile's), requested
    // before the tile is staged so that its latency hides behind the staging loads
    const bool valid = tid < nb;
    int pred = 0, table = 0;
    const int16_t* zg = nullptr;
    if (valid) {
        const unsigned g = g0 + tid, mcu = g / 6u, i = g - mcu * 6u;
        table = i < 4 ? 0 : 1;
        if (!(i >= 4 && job.bpm == 4)) {
            zg = fc + ((size_t)mcu * job.bpm + i) * 64;
            if (i >= 1 && i <= 3) pred = zg[-64];
            else if (mcu != 0) pred = i == 0 ? zg[-(job.bpm - 3) * 64] : zg[-job.bpm * 64];
        }
    }
    // the tile's stored blocks are contiguous in memory; row = the lane that codes the block
    if (job.bpm == 6) {
        const uint4* src = reinterpret_cast<const uint4*>(fc + (size_t)g0 * 64);
        for (unsigned c = tid; c < nb * 8u; c += WG)
            *reinterpret_cast<uint4*>(tile + (c >> 3) * ROW + ROW_DATA + (c & 7u) * 16u) = src[c];
    } else {    // gray: 6 coded blocks per MCU, 4 stored ones
        const unsigned g1 = g0 + nb - 1, m0 = g0 / 6u, i0 = g0 - m0 * 6u, m1 = g1 / 6u, i1 = g1 - m1 * 6u;
        const unsigned s0 = i0 < 4 ? m0 * 4u + i0 : (m0 + 1) * 4u, s1 = i1 < 4 ? m1 * 4u + i1 : m1 * 4u + 3u;   // s1 + 1 >= s0
        const unsigned nchunks = (s1 + 1 - s0) * 8u;
        const uint4* src = reinterpret_cast<const uint4*>(fc + (size_t)s0 * 64);
        for (unsigned c = tid; c < nchunks; c += WG) {
            const unsigned sb = s0 + (c >> 3), r = (sb >> 2) * 6u + (sb & 3u) - g0;
            *reinterpret_cast<uint4*>(tile + r * ROW + ROW_DATA + (c & 7u) * 16u) = src[c];
        }
    }
    __syncthreads();

    uint32_t* const row = reinterpret_cast<uint32_t*>(tile + tid * ROW);
    unsigned n = 0;
    bool ovf = false;
    if (valid) {
        RowWriter w;
        w.init(row);
        const bool ok = code_block(zg ? reinterpret_cast<const int16_t*>(tile + tid * ROW + ROW_DATA) : nullptr, pred, L.dc[table], L.ac[table],
                                   L.fast[table], w);
        n = w.bits();
        w.finish();
        ovf = w.overflowed();
        if (!ok) atomicOr(status + frame, 1u);
    }
    row[ROW_LAST_WORD + 1] = n;            // where the owner of a partial word finds the length of the lanes after it
    uint32_t total;
    const uint32_t o = wg256_exclusive_scan(n, &total);      // (barrier inside: rows and lengths are visible)
    const size_t t_index = (size_t)frame * gridDim.x + blockIdx.x;
    if (tid == 0) tile_total[t_index] = total;
    if (!valid) return;

    uint32_t* const Sg = S + t_index * TILE_STREAM_WORDS;
    const unsigned sh = o & 31u, w0 = o >> 5, end = o + n;
    uint32_t tail = 0;
    unsigned tail_word = 0, tail_fill = 0;                   // tail_fill != 0: this lane owns a partial last word
    if (!ovf) {
        const unsigned nwp = (n + 31u) >> 5;
        unsigned k = sh ? 1u : 0u;
        uint32_t prev = sh ? row[0] : 0u;
        for (; ((w0 + k + 1u) << 5) <= end; ++k) {
            const uint32_t cur = k < nwp ? row[k] : 0u;
            Sg[w0 + k] = __builtin_amdgcn_alignbit(prev, cur, sh);
            prev = cur;
        }
        if (((w0 + k) << 5) < end) {
            const uint32_t cur = k < nwp ? row[k] : 0u;
            tail_word = w0 + k;
            tail_fill = end - (tail_word << 5);
            tail = __builtin_amdgcn_alignbit(prev, cur, sh) & ~(0xFFFFFFFFu >> tail_fill);
        }
    } else {
        DirectWriter w;
        w.init(Sg, o);
        (void)code_block(zg, pred, L.dc[table], L.ac[table], L.fast[table], w);
        if (w.nacc > 0 && !w.skip) {
            tail_word = w.w;
            tail_fill = (unsigned)w.nacc;
            tail = (uint32_t)(w.acc << (32 - w.nacc));
        }
    }
    if (tail_fill) {
        unsigned fill = tail_fill;
        for (unsigned j = tid + 1; fill < 32u && j < nb; ++j) {
            const uint32_t* rj = reinterpret_cast<const uint32_t*>(tile + j * ROW);
            tail |= rj[0] >> fill;           // a stream shorter than a word is zero padded: nothing but its own bits
            fill += rj[ROW_LAST_WORD + 1];
        }
        Sg[tail_word] = tail;
    }
}

// One workgroup per frame: exclusive prefix sums of the frame's tile totals (bits, frame relative, uint64), the frame's
// stream length in bytes, and for every 16 KB piece of the unstuffed stream the tile its first bit lies in.
constexpr unsigned ASM_BITS = 256u * (unsigned)CHUNK * 8u;          // bits of U one assemble workgroup writes (131072)
__global__ __launch_bounds__(256) void tile_bases_kernel(const uint32_t* tile_total, unsigned tpf, unsigned long long* base,
                                                        unsigned long long* bytes, uint32_t* first_tile, unsigned ft_stride,
                                                        unsigned* status, unsigned* latched)
{
    const unsigned f = blockIdx.x, tid = threadIdx.x;
    const uint32_t* tt = tile_total + (size_t)f * tpf;
    unsigned long long* B = base + (size_t)f * (tpf + 1);
    uint32_t* ft = first_tile + (size_t)f * ft_stride;
    unsigned long long carry = 0;
    for (unsigned b0 = 0; b0 < tpf; b0 += 2048u) {
        uint32_t v[8], s = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned t = b0 + tid * 8u + k;
            v[k] = t < tpf ? tt[t] : 0u;
            s += v[k];
        }
        uint32_t total;
        unsigned long long run = carry + wg256_exclusive_scan(s, &total);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned t = b0 + tid * 8u + k;
            if (t < tpf) {
                B[t] = run;
                // pieces whose first bit lies in [run, run + v[k])
                for (unsigned long long w = (run + ASM_BITS - 1) / ASM_BITS; w * ASM_BITS < run + v[k] && w < ft_stride; ++w) ft[w] = t;
            }
            run += v[k];
        }
        carry += total;
        __syncthreads();                 // the scan's LDS is reused by the next batch
    }
    if (tid == 0) {
        B[tpf] = carry;
        bytes[f] = (carry + 7) >> 3;
        // device-resident form: the frame's error flag (set by code_tiles_kernel) is consumed here -- after every writer, before
        // every reader -- and left clear for the next call: no launch is spent on zeroing it
        if (latched) {
            latched[f] = status[f];
            status[f] = 0;
        }
    }
}

// Tiles a workgroup's ASM_BITS can touch.  The shortest block is a flat chroma block (DC category 0: 2 bits, EOB: 2 bits -- also
// every zero chroma block of gray mode), a flat luma block takes 2 + 4, so a flat MCU is 32 bits and 256 consecutive blocks
// hold at least 42 MCUs + the cheapest four consecutive blocks (4 + 4 + 6 + 6) = 1364 bits: a piece that starts inside a tile
// touches at most ASM_BITS / 1364 + 2 = 98 tiles (only a frame's LAST tile may be shorter, and nothing follows it).
// (Round 2 assumed 256 x 6 bits and a window of 96: flat frames of more than 2048 tiles lost the last chunks of a piece.)
constexpr unsigned MIN_TILE_BITS = 42u * 32u + 20u;
constexpr int ASM_WIN = 128;
static_assert((unsigned)(ASM_WIN - 2) * MIN_TILE_BITS >= ASM_BITS, "assemble window too small for flat content");
constexpr unsigned ASM_SELF_TILES = 2048;   // frames of up to this many tiles: every assembling workgroup scans the totals itself

// the 64 bytes of U that start at frame bit p: tile A (bits [bA, eA) of the frame, stream srcA) holds p; tile B (LB bits,
// stream srcB; LB = 0: there is none) follows it.  All 17 + 16 source words are requested at once, words past the end of a
// tile's stream read as zero (the streams are zero padded to a word), every output word is one funnel shift per tile.
__device__ __forceinline__ unsigned assemble_chunk(unsigned long long p, unsigned long long bA, unsigned long long eA, const uint32_t* srcA,
                                                   unsigned LB, const uint32_t* srcB, unsigned long long T, uint4* dst)
{
    const unsigned long long qA = p - bA;
    const unsigned LA = (unsigned)(eA - bA), aA = (unsigned)(qA >> 5), sA = (unsigned)(qA & 31u);
    uint32_t wa[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) wa[k] = qA < LA && ((aA + k) << 5) < LA ? srcA[aA + k] : 0u;
    // the next tile starts inside this chunk, at chunk bit dB = 32 * kb + rb
    const bool two = LB != 0 && p + CHUNK * 8 > eA;
    const unsigned dB = two ? (unsigned)(eA - p) : 0u, kb = dB >> 5, rb = dB & 31u;
    uint32_t wn[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) wn[k] = two && (unsigned)k >= kb && (((unsigned)k - kb) << 5) < LB ? srcB[(unsigned)k - kb] : 0u;
    uint32_t out[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        uint32_t v = sA ? __builtin_amdgcn_alignbit(wa[k], wa[k + 1], 32u - sA) : wa[k];
        v |= __builtin_amdgcn_alignbit(k > 0 ? wn[k - 1] : 0u, wn[k], rb);
        out[k] = v;
    }
#if JPEZY_PAD_BIT   // alternative frozen choice (include/jpezy_constants.h): one pad bits in the frame's last byte; a padded
                    // 0xFF is then stuffed like any other
    {
        const unsigned pad = (unsigned)((8 - (T & 7)) & 7);
        if (pad && T > p && T < p + CHUNK * 8) {
            const unsigned kk = (unsigned)((T - p) >> 5), keep = (unsigned)((T - p) & 31u);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if ((unsigned)k == kk) out[k] |= ((1u << pad) - 1u) << (32u - keep - pad);
        }
    }
#else
    (void)T;
#endif
    unsigned n_ff = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) n_ff += count_ff(out[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        dst[k] = make_uint4(__builtin_bswap32(out[4 * k]), __builtin_bswap32(out[4 * k + 1]), __builtin_bswap32(out[4 * k + 2]),
                            __builtin_bswap32(out[4 * k + 3]));
    return n_ff;
}

// One thread per 64-byte chunk of U, a workgroup per 16 KB piece (grid-stride over the pieces: the grid is sized for a typical
// stream, U for the worst case).
// SELF (frames of at most ASM_SELF_TILES tiles -- 4096x4096 has 1536): no launch between the coder and this kernel.  Every
// workgroup reads the frame's tile totals and scans them in LDS (8 values per thread), a thread finds its chunk's tile by
// binary search there; workgroup 0 of the frame publishes the stream length and latches + clears the frame's error flag
// (what tile_bases_kernel does for larger frames).
// !SELF: tile offsets, stream length and the first tile of every piece come from tile_bases_kernel; a workgroup loads the
// window of at most ASM_WIN tile offsets its piece can touch.
template <bool SELF>
__global__ __launch_bounds__(256) void assemble_kernel(const uint32_t* S, const uint32_t* tile_total, const unsigned long long* base,
                                                      unsigned long long* bytes, const uint32_t* first_tile, unsigned tpf,
                                                      unsigned ft_stride, uint32_t* U, size_t u_stride_words, uint32_t* loc,
                                                      uint32_t* ff_tile_total, unsigned* status, unsigned* latched)
{
    __shared__ unsigned long long wb[SELF ? 1 : ASM_WIN + 2];
    __shared__ uint32_t pre[SELF ? ASM_SELF_TILES + 2 : 1];
    const unsigned frame = blockIdx.y, tid = threadIdx.x;
    const size_t chunks_per_frame = u_stride_words * 4 / CHUNK, pieces = chunks_per_frame / 256;   // a multiple of 256 (launcher)
    unsigned long long T, nbytes;
    if constexpr (SELF) {
        const uint32_t* tt = tile_total + (size_t)frame * tpf;
        uint32_t v[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned t = tid * 8u + k;
            v[k] = t < tpf ? tt[t] : 0u;
            sum += v[k];
        }
        uint32_t total;
        uint32_t run = wg256_exclusive_scan(sum, &total);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            pre[tid * 8u + k] = run;           // entries past tpf: the total
            run += v[k];
        }
        if (tid == 255) { pre[ASM_SELF_TILES] = total; pre[ASM_SELF_TILES + 1] = total; }
        __syncthreads();
        T = total;
        nbytes = (T + 7) >> 3;
        if (blockIdx.x == 0 && tid == 0) {
            bytes[frame] = nbytes;
            if (latched) {                      // consumed after every writer (the coder), before every reader (the stuffing kernel)
                latched[frame] = status[frame];
                status[frame] = 0;
            }
        }
    } else {
        nbytes = bytes[frame];
        T = 0;
#if JPEZY_PAD_BIT
        T = base[(size_t)frame * (tpf + 1) + tpf];
#endif
    }
    for (size_t x = blockIdx.x; x < pieces; x += gridDim.x) {
        const unsigned long long c0 = (unsigned long long)x * 256u, c = c0 + tid;
        // the first piece of a frame is always written, and so is the one that holds the one-past-the-end chunk (the stuffing
        // kernel reads that chunk's offset); the pieces behind it are never read
        if (x != 0 && c0 * CHUNK > nbytes) break;
        unsigned n_ff = 0;
        const unsigned long long p = c * (CHUNK * 8ull);
        uint4* dst = reinterpret_cast<uint4*>(U + (size_t)frame * u_stride_words) + c * (CHUNK / 16);
        if constexpr (SELF) {
            if (c * CHUNK < nbytes) {
                unsigned lo = 0, hi = tpf;                              // pre[lo] <= p < pre[hi] = T
                while (hi - lo > 1) {
                    const unsigned mid = (lo + hi) >> 1;
                    if (pre[mid] <= p) lo = mid; else hi = mid;
                }
                const unsigned t = lo;
                const uint32_t* srcA = S + ((size_t)frame * tpf + t) * TILE_STREAM_WORDS;
                const unsigned LB = t + 1 < tpf ? pre[t + 2] - pre[t + 1] : 0u;
                n_ff = assemble_chunk(p, pre[t], pre[t + 1], srcA, LB, srcA + TILE_STREAM_WORDS, T, dst);
            }
        } else {
            const unsigned t0 = first_tile[(size_t)frame * ft_stride + x];     // (garbage past the stream's end: clamped, unused)
            const unsigned long long* B = base + (size_t)frame * (tpf + 1);
            __syncthreads();                                                    // the previous piece's window is no longer read
            for (unsigned i = tid; i < (unsigned)ASM_WIN + 2u; i += 256u) wb[i] = B[t0 + i < tpf ? t0 + i : tpf];
            __syncthreads();
            if (c * CHUNK < nbytes) {
                int lo = 0, hi = ASM_WIN;                               // wb[lo] <= p; first wb[hi] > p or the window's end
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (wb[mid] <= p) lo = mid; else hi = mid;
                }
                const int t = lo;
                const unsigned tA = t0 + (unsigned)t < tpf ? t0 + (unsigned)t : tpf - 1;
                const uint32_t* srcA = S + ((size_t)frame * tpf + tA) * TILE_STREAM_WORDS;
                const unsigned LB = tA + 1 < tpf ? (unsigned)(wb[t + 2] - wb[t + 1]) : 0u;
                n_ff = assemble_chunk(p, wb[t], wb[t + 1], srcA, LB, srcA + TILE_STREAM_WORDS, T, dst);
            }
        }
        uint32_t total;
        const uint32_t off = wg256_exclusive_scan(n_ff, &total);
        loc[(size_t)frame * chunks_per_frame + c] = off;
        if (tid == 0) ff_tile_total[(size_t)frame * pieces + x] = total;
        __syncthreads();                                                        // the scan's LDS is reused by the next piece
    }
}

// ---- exclusive prefix sums: 2048 elements per workgroup, recursive over the workgroup totals ----
constexpr int SCAN_T = 256, SCAN_E = 8, SCAN_N = SCAN_T * SCAN_E;

template <typename TIn>
__global__ __launch_bounds__(SCAN_T) void scan_local_kernel(const TIn* in, unsigned long long* out, unsigned long long* totals,
                                                           size_t n)
{
    __shared__ unsigned long long wsum[SCAN_T / 64];
    const size_t base = (size_t)blockIdx.x * SCAN_N + (size_t)threadIdx.x * SCAN_E;
    unsigned long long v[SCAN_E], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_E; ++k) {
        v[k] = base + k < n ? (unsigned long long)in[base + k] : 0ull;
        s += v[k];
    }
    // inclusive scan of the per-thread sums: within the wave by shuffles, across the four waves through LDS
    unsigned long long inc = s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    unsigned long long woff = 0;
    for (int k = 0; k < wv; ++k) woff += wsum[k];
    unsigned long long run = woff + inc - s;
#pragma unroll
    for (int k = 0; k < SCAN_E; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == SCAN_T - 1) {
        totals[blockIdx.x] = run;
        if (gridDim.x == 1) out[n] = run;          // a single workgroup: the scan is complete, no second launch
    }
}

__global__ __launch_bounds__(SCAN_T) void scan_add_kernel(unsigned long long* out, const unsigned long long* offs, size_t n,
                                                         unsigned long long* total_slot)
{
    const size_t base = (size_t)blockIdx.x * SCAN_N + (size_t)threadIdx.x * SCAN_E;
    const unsigned long long o = offs[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_E; ++k)
        if (base + k < n) out[base + k] += o;
    (void)total_slot;
}

// out[0..n) = exclusive prefix sums of in[0..n), out[n] = total.  tmp: scratch of scan_tmp_elems(n) uint64.
size_t scan_tmp_elems(size_t n)
{
    size_t t = 0;
    while (n > 1) {
        n = (n + SCAN_N - 1) / SCAN_N;
        t += n + 1 + n;      // scanned totals (n+1) and raw totals (n)
        if (n == 1) break;
    }
    return t + 4;
}

// second (and last) launch of a scan whose workgroup totals are few: every workgroup sums the totals in front of it
// itself (nb * 8 bytes out of the L2) and adds that base to its 2048 elements; the last one also writes the grand total.
// Replaces the recursive scan of the totals, the add pass and two device-to-device copies: 2 launches instead of 5.
__global__ __launch_bounds__(SCAN_T) void scan_finish_kernel(unsigned long long* out, const unsigned long long* totals, size_t n, size_t nb)
{
    __shared__ unsigned long long wsum[SCAN_T / 64];
    unsigned long long sum = 0;
    for (size_t k = threadIdx.x; k < (size_t)blockIdx.x; k += SCAN_T) sum += totals[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = sum;
    __syncthreads();
    unsigned long long o = 0;
#pragma unroll
    for (int k = 0; k < SCAN_T / 64; ++k) o += wsum[k];
    const size_t base = (size_t)blockIdx.x * SCAN_N + (size_t)threadIdx.x * SCAN_E;
#pragma unroll
    for (int k = 0; k < SCAN_E; ++k)
        if (base + k < n) out[base + k] += o;
    if (blockIdx.x == nb - 1 && threadIdx.x == 0) out[n] = o + totals[nb - 1];
}

// a few thousand elements: one workgroup walks them in batches of 2048 with a running carry -- one launch instead of two
// (a dependent launch costs ~4.7 us on this chip, a batch well under one)
constexpr size_t SCAN_ONE_WG_MAX = 16 * SCAN_N;
template <typename TIn>
__global__ __launch_bounds__(SCAN_T) void scan_one_wg_kernel(const TIn* in, unsigned long long* out, size_t n)
{
    __shared__ unsigned long long wsum[SCAN_T / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long carry = 0;
    for (size_t b0 = 0; b0 < n; b0 += SCAN_N) {
        const size_t base = b0 + (size_t)threadIdx.x * SCAN_E;
        unsigned long long v[SCAN_E], s = 0;
#pragma unroll
        for (int k = 0; k < SCAN_E; ++k) {
            v[k] = base + k < n ? (unsigned long long)in[base + k] : 0ull;
            s += v[k];
        }
        unsigned long long inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        unsigned long long woff = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < SCAN_T / 64; ++k) {
            if (k < wv) woff += wsum[k];
            tot += wsum[k];
        }
        unsigned long long run = carry + woff + inc - s;
#pragma unroll
        for (int k = 0; k < SCAN_E; ++k) {
            if (base + k < n) out[base + k] = run;
            run += v[k];
        }
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry;
}

constexpr size_t SCAN_FINISH_MAX_WGS = 4096;   // beyond it (n > 8.4 M) the totals are scanned recursively as before

template <typename TIn>
static hipError_t scan_exclusive(const TIn* in, unsigned long long* out, size_t n, unsigned long long* tmp, hipStream_t s)
{
    if (n == 0) return hipMemsetAsync(out, 0, sizeof(unsigned long long), s);
    if (n <= SCAN_ONE_WG_MAX) {
        hipLaunchKernelGGL((scan_one_wg_kernel<TIn>), dim3(1), dim3(SCAN_T), 0, s, in, out, n);
        return hipGetLastError();
    }
    const size_t nb = (n + SCAN_N - 1) / SCAN_N;
    unsigned long long* raw = tmp;             // [nb] workgroup totals
    unsigned long long* scanned = tmp + nb;    // [nb + 1]
    hipLaunchKernelGGL((scan_local_kernel<TIn>), dim3((unsigned)nb), dim3(SCAN_T), 0, s, in, out, raw, n);
    if (nb == 1) return hipGetLastError();
    if (nb <= SCAN_FINISH_MAX_WGS) {
        hipLaunchKernelGGL(scan_finish_kernel, dim3((unsigned)nb), dim3(SCAN_T), 0, s, out, raw, n, nb);
        return hipGetLastError();
    }
    hipError_t e = scan_exclusive<unsigned long long>(raw, scanned, nb, scanned + nb + 1, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(SCAN_T), 0, s, out, scanned, n, (unsigned long long*)nullptr);
    return hipMemcpyAsync(out + n, scanned + nb, sizeof(unsigned long long), hipMemcpyDeviceToDevice, s);
}

// ---- byte stuffing ----

// Copy U to the output inserting 0x00 after every 0xFF.  A workgroup takes 256 consecutive chunks of one frame: its output
// is one contiguous byte range (start = chunk offset + 0xFF bytes before it).  Every thread expands its 64 bytes into LDS
// at the position they have in that range, shifted so that LDS words line up with the 4-byte words of the destination;
// the range then goes out as whole words, coalesced (byte stores only for the partial first and last word, which the
// neighbouring workgroups complete).  Byte stores straight to global memory -- 64 lanes, 64 different cache lines per
// instruction -- cost 31 us per 4096x4096 frame.
constexpr int STUFF_WG = 256;
// plan.hdr != nullptr (device-resident form): the frame's file is header + stuffed stream + EOI at out + frame * out_stride;
// every workgroup works out whether the frame failed (a coefficient outside the tables: JPEZY_E_FORMAT = -5) or does not fit
// (JPEZY_E_NOSPACE = -6) and leaves at once if so; workgroup 0 of the frame reports the size, writes EOI and copies the header.
__global__ __launch_bounds__(STUFF_WG) void stuff_kernel(const uint32_t* U, size_t u_stride_words, const unsigned long long* frame_bytes,
                                                        const uint32_t* ff_loc, const uint32_t* ff_tile_total, uint8_t* out, size_t out_stride,
                                                        FilePlan plan)
{
    __shared__ uint32_t buf[STUFF_WG * CHUNK * 2 / 4 + 4];
    __shared__ unsigned long long red[2][STUFF_WG / 64];
    uint8_t* const lb = reinterpret_cast<uint8_t*>(buf);
    const size_t chunks_per_frame = u_stride_words * 4 / CHUNK, pieces = chunks_per_frame / STUFF_WG;
    const size_t frame = blockIdx.y;
    const unsigned long long nbytes = frame_bytes[frame];
    // grid-stride over the 16 KB pieces: the grid is sized for a typical stream, the buffer for the worst case
    for (size_t px = blockIdx.x; (unsigned long long)px * STUFF_WG * CHUNK < nbytes; px += gridDim.x) {
        const size_t c0 = px * STUFF_WG, c = c0 + threadIdx.x;
        // this thread's 64 bytes and their offset inside the piece: requested now, together with the tile totals below -- the
        // kernel is a chain of memory round trips (stream length -> {totals, offsets, bytes} -> stores), not arithmetic
        const uint32_t* loc = ff_loc + frame * chunks_per_frame;                      // 0xFF bytes before a chunk inside its 256-chunk tile
        const bool has_data = (unsigned long long)c * CHUNK < nbytes;
        uint4 data[CHUNK / 16];
        uint32_t my_loc = 0;
        if (has_data) {
            const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(U + frame * u_stride_words) + c * CHUNK);
#pragma unroll
            for (int k = 0; k < CHUNK / 16; ++k) data[k] = src[k];
            my_loc = loc[c];
        }
        // 0xFF bytes of the frame before this workgroup's chunks, and in the whole frame: the workgroup adds up the tile totals
        // itself (a few hundred 4-byte values out of the L2) -- a scan launched in between costs more than all of these sums
        const uint32_t* ft = ff_tile_total + frame * pieces;
        const size_t last = (size_t)(nbytes / ((unsigned long long)STUFF_WG * CHUNK)), used = last + 1 < pieces ? last + 1 : pieces;
        unsigned long long pre = 0, all = 0;
        for (size_t x = threadIdx.x; x < used; x += STUFF_WG) {
            const uint32_t v = ft[x];
            all += v;
            pre += x < px ? v : 0u;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            pre += __shfl_xor(pre, d, 64);
            all += __shfl_xor(all, d, 64);
        }
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = pre; red[1][threadIdx.x >> 6] = all; }
        __syncthreads();
        pre = all = 0;
#pragma unroll
        for (int k = 0; k < STUFF_WG / 64; ++k) { pre += red[0][k]; all += red[1][k]; }
        if (plan.hdr) {
            const unsigned long long body = nbytes + all;
            const unsigned long long total = plan.hdr_len + body + 2;
            const unsigned st = plan.latched[frame];
            const bool failed = st != 0 || total > out_stride;
            if (px == 0) {
                uint8_t* file = out + frame * out_stride;
                if (threadIdx.x == 0) {
                    plan.sizes[frame] = st ? -5 : failed ? -6 : (long long)total;
                    if (!failed) {
                        file[plan.hdr_len + body] = 0xFF;
                        file[plan.hdr_len + body + 1] = 0xD9;
                    }
                }
                if (!failed)
                    for (size_t i = threadIdx.x; i < plan.hdr_len; i += STUFF_WG) file[i] = plan.hdr[i];
            }
            if (failed) return;                                                      // workgroup-uniform
        }
        uint8_t* const P = out + (plan.hdr ? plan.hdr_len : 0) + frame * out_stride + c0 * CHUNK + pre;              // first output byte of the workgroup
        const unsigned shift = (unsigned)(reinterpret_cast<uintptr_t>(P) & 3u);
        if (has_data) {
            const int n = (int)((nbytes - (unsigned long long)c * CHUNK) < (unsigned long long)CHUNK ? (nbytes - (unsigned long long)c * CHUNK) : CHUNK);
            uint8_t* dst = lb + shift + threadIdx.x * CHUNK + my_loc;
#pragma unroll
            for (int k = 0; k < CHUNK / 16; ++k) {
                const uint32_t wd[4] = { data[k].x, data[k].y, data[k].z, data[k].w };
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (k * 16 + j < n) {
                        const uint8_t b = (uint8_t)(wd[j >> 2] >> ((j & 3) * 8));
                        *dst++ = b;
                        if (b == 0xFF) *dst++ = 0x00;
                    }
                }
            }
        }
        __syncthreads();
        // bytes of the workgroup's range: its chunks' bytes plus the 0xFF bytes among them
        const unsigned long long last_chunk = (nbytes + CHUNK - 1) / CHUNK;          // chunks of the frame that hold data
        const size_t ce = c0 + STUFF_WG < last_chunk ? c0 + STUFF_WG : (size_t)last_chunk;
        const unsigned long long src_end = (unsigned long long)ce * CHUNK < nbytes ? (unsigned long long)ce * CHUNK : nbytes;
        const unsigned total = (unsigned)(src_end - (unsigned long long)c0 * CHUNK) + (ce < c0 + STUFF_WG ? loc[ce] : ft[px]);
        uint32_t* const A = reinterpret_cast<uint32_t*>(P - shift);                  // 4-byte aligned
        const unsigned end = shift + total, nwords = (end + 3) / 4;
        for (unsigned w = threadIdx.x; w < nwords; w += STUFF_WG) {
            const unsigned lo = w * 4, hi = lo + 4;
            if (lo >= shift && hi <= end) {
                A[w] = buf[w];
            } else {
                for (unsigned k = lo < shift ? shift : lo; k < (hi < end ? hi : end); ++k) reinterpret_cast<uint8_t*>(A)[k] = lb[k];
            }
        }
        __syncthreads();             // the staging buffer and the partial sums are reused by the next piece
    }
}

// ---- host-side driver ----
hipError_t launch_scan_u32(const uint32_t* in, unsigned long long* out, size_t n, unsigned long long* tmp, hipStream_t s)
{
    hipError_t e = scan_exclusive<uint32_t>(in, out, n, tmp, s);
    return e != hipSuccess ? e : hipGetLastError();
}

hipError_t launch_scan_u64(const unsigned long long* in, unsigned long long* out, size_t n, unsigned long long* tmp, hipStream_t s)
{
    hipError_t e = scan_exclusive<unsigned long long>(in, out, n, tmp, s);
    return e != hipSuccess ? e : hipGetLastError();
}

hipError_t launch_stuff(const uint32_t* U, size_t u_stride_words, const unsigned long long* frame_bytes, int n_frames,
                        const uint32_t* ff_loc, const uint32_t* ff_tile_total, uint8_t* out, size_t out_stride, FilePlan plan, hipStream_t s)
{
    const size_t chunks = u_stride_words * 4 / CHUNK;
    if (!chunks || n_frames <= 0) return hipSuccess;
    if (n_frames > 65535 || chunks % STUFF_WG) return hipErrorInvalidValue;      // the frame index is a grid dimension
    size_t gx = chunks / STUFF_WG < 1024 ? chunks / STUFF_WG : 1024;
    if (n_frames > 1 && gx > 128) gx = 128;
    hipLaunchKernelGGL(stuff_kernel, dim3((unsigned)gx, (unsigned)n_frames), dim3(STUFF_WG), 0, s, U, u_stride_words,
                       frame_bytes, ff_loc, ff_tile_total, out, out_stride, plan);
    return hipGetLastError();
}

// dst[f] = 0xFF bytes of frame f's unstuffed stream (the host-delivered form sizes its output from it): the totals of the
// pieces the assembling kernel wrote -- up to the one that holds the one-past-the-end chunk
__global__ __launch_bounds__(256) void ff_frame_totals_kernel(const uint32_t* ff_tile_total, const unsigned long long* bytes, size_t pieces,
                                                             unsigned long long* dst)
{
    __shared__ unsigned long long red[4];
    const size_t piece_bytes = (size_t)256 * CHUNK;
    const size_t used = (size_t)(bytes[blockIdx.x] / piece_bytes) + 1 < pieces ? (size_t)(bytes[blockIdx.x] / piece_bytes) + 1 : pieces;
    unsigned long long sum = 0;
    for (size_t x = threadIdx.x; x < used; x += 256) sum += ff_tile_total[blockIdx.x * pieces + x];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) dst[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

hipError_t launch_ff_frame_totals(const uint32_t* ff_tile_total, const unsigned long long* bytes, size_t u_stride_words, int n_frames,
                                  unsigned long long* dst, hipStream_t s)
{
    if (n_frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(ff_frame_totals_kernel, dim3((unsigned)n_frames), dim3(256), 0, s, ff_tile_total, bytes,
                       u_stride_words * 4 / assemble_piece_bytes(), dst);
    return hipGetLastError();
}

size_t chunk_bytes() { return CHUNK; }

size_t tile_stream_bytes() { return (size_t)TILE_STREAM_WORDS * 4; }
size_t assemble_piece_bytes() { return (size_t)256 * CHUNK; }

hipError_t launch_code_tiles(const Job& job, uint32_t* S, uint32_t* tile_total, unsigned* status, hipStream_t s)
{
    if (!job.blocks_per_frame || job.n_frames <= 0) return hipSuccess;
    if (job.n_frames > 65535) return hipErrorInvalidValue;                       // the frame index is a grid dimension
    hipLaunchKernelGGL(code_tiles_kernel, dim3((unsigned)tiles256(job.blocks_per_frame), (unsigned)job.n_frames), dim3(WG), 0, s, job, S,
                       tile_total, status);
    return hipGetLastError();
}

hipError_t launch_tile_bases(const uint32_t* tile_total, unsigned tiles_per_frame, int n_frames, unsigned long long* base,
                             unsigned long long* bytes, uint32_t* first_tile, unsigned ft_stride, unsigned* status, unsigned* latched,
                             hipStream_t s)
{
    if (!tiles_per_frame || n_frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(tile_bases_kernel, dim3((unsigned)n_frames), dim3(256), 0, s, tile_total, tiles_per_frame, base, bytes, first_tile,
                       ft_stride, status, latched);
    return hipGetLastError();
}

bool assemble_scans_tiles_itself(size_t tiles_per_frame) { return tiles_per_frame <= ASM_SELF_TILES; }

hipError_t launch_assemble(const uint32_t* S, const uint32_t* tile_total, const unsigned long long* base, unsigned long long* bytes,
                           const uint32_t* first_tile, unsigned ft_stride, unsigned tiles_per_frame, int n_frames, uint32_t* U,
                           size_t u_stride_words, uint32_t* loc, uint32_t* ff_tile_total, unsigned* status, unsigned* latched, hipStream_t s)
{
    const size_t pieces = u_stride_words * 4 / assemble_piece_bytes();
    if (!pieces || n_frames <= 0) return hipSuccess;
    if (u_stride_words * 4 % assemble_piece_bytes() || n_frames > 65535) return hipErrorInvalidValue;
    // a grid for a typical stream (1024 pieces = 16 MB per frame), fewer per frame when the frames fill the chip
    size_t gx = pieces < 1024 ? pieces : 1024;
    if (n_frames > 1 && gx > 128) gx = 128;
    if (assemble_scans_tiles_itself(tiles_per_frame)) {
        hipLaunchKernelGGL(assemble_kernel<true>, dim3((unsigned)gx, (unsigned)n_frames), dim3(256), 0, s, S, tile_total, base, bytes, first_tile,
                           tiles_per_frame, ft_stride, U, u_stride_words, loc, ff_tile_total, status, latched);
    } else {
        if (pieces > ft_stride) return hipErrorInvalidValue;
        hipLaunchKernelGGL(assemble_kernel<false>, dim3((unsigned)gx, (unsigned)n_frames), dim3(256), 0, s, S, tile_total, base, bytes,
                           first_tile, tiles_per_frame, ft_stride, U, u_stride_words, loc, ff_tile_total, status, latched);
    }
    return hipGetLastError();
}

}  // namespace entropy
}  // namespace jpezy_dev
