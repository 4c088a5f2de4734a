// jpezy_huffdec_core.h -- the part of the GPU Huffman decoder that is plain arithmetic: the table format, the host-side
// table builder and the one-symbol decode step.  Included by the kernels (jpezy_huffdec.hip, through jpezy_huffdec.h), by the
// context code that builds the tables (jpezy_capi_huffdec.hip) and -- compiled by g++ with AddressSanitizer and UBSan, no HIP involved --
// by tests/fuzz/huffdec_core_fuzz.cpp, which walks whole scans with the same step and compares with the host decoder
// (the CPU test suite's view of the round-3 decoder: tests/test_host_codec.py).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define JPEZY_HD __host__ __device__ __forceinline__
#else
#define JPEZY_HD inline
#endif

namespace jpezy_dev {
namespace huffdec {

// Decoding tables: two 16-bit lookups, no loop, no compare chain -- and since round 4 the two are INDEPENDENT, so a step waits for one LDS round
// trip, not two.  l1 is indexed by the next L1_BITS bits of the stream and decodes every code of at most L1_BITS bits.  Canonical codes ascend
// with their length, so the longer ones sit at the top of the code space: hi is indexed by the low HI_BITS of the next 16 bits and is the
// complete decode for every 16-bit window whose first 16 - HI_BITS bits are all ones (long codes, and the short codes that reach up there,
// repeated).  A table with a code longer than L1_BITS bits below that region is left to the host decoder (the same capacity as round 3's
// sixteen 64-entry second-level tables, which were reached through the first lookup's result: a dependent read).
// An entry says everything the decoder's state machine needs, laid out so that a walk takes it apart with two instructions (round 4):
//   bits 4..0   len + s - 1   bits to move on, minus one (len: code length, s: value bits that follow -- AC: low nibble of the symbol; DC: the
//                             symbol, a category above 16 has no entry)
//   bits 9..5   s             (31: not a code -- only the coefficient pass looks)
//   bits 15..10 run           zig-zag positions skipped in front of the coefficient; 63 for the end-of-block symbol and for bits that are
//                             no code, which therefore ends the block at any position: what a synchronisation walk does with garbage
//                             (abandon the block, move one bit on) needs no test of its own
constexpr int L1_BITS = 10, HI_BITS = 10;
constexpr unsigned E_RUN_END = 63u, E_S_NONE = 31u;
constexpr unsigned E_NOT_A_CODE = (E_RUN_END << 10) | (E_S_NONE << 5);       // one bit on, block ended
constexpr unsigned make_entry(unsigned len, unsigned s, unsigned run) { return (run << 10) | (s << 5) | (len + s - 1u); }
constexpr unsigned HI_FIRST = 0x10000u - (1u << HI_BITS);        // the first 16-bit window the hi table answers for
struct alignas(16) Table {
    uint16_t l1[1 << L1_BITS];
    uint16_t hi[1 << HI_BITS];
};
constexpr unsigned TABLE_U16 = sizeof(Table) / 2;

struct Setup {
    Table dc[3], ac[3];       // indexed by the scan component's table selector Td (the reference uses Td for both)
    int bpm;                  // period of the table sequence over the blocks of an MCU (divides the blocks per MCU; at most MAX_PERIOD)
    unsigned tdmask;          // Td of block b of a period in bits [2 b + 1 : 2 b]
    unsigned total_blocks;
    unsigned pad;
};
constexpr int MAX_PERIOD = 16;
static_assert(sizeof(Table) == 4096 && sizeof(Setup) % 16 == 0, "table layout the kernels index by hand");


// ---- host: building the tables ----
// The device decoder's tables of one Huffman table (jpezy_huffdec.h).  dc: the symbol is the number of value bits.
// false: the counts do not describe a prefix code (more codes of some length than the code space has left), or a code longer than
// L1_BITS bits lies below the region the hi table covers -- such a table is left to the host decoder, whose canonical loop defines
// what it means.
inline bool build_dev_table(Table& t, const uint8_t bits[16], const uint8_t* vals, int n, bool dc)
{
    for (uint16_t& e : t.l1) e = (uint16_t)E_NOT_A_CODE;
    for (uint16_t& e : t.hi) e = (uint16_t)E_NOT_A_CODE;
    (void)n;
    unsigned code = 0;
    int p = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int c = 0; c < bits[l - 1]; ++c, ++p, ++code) {
            if (code >= (1u << l)) return false;                     // more codes of this length than the code space has left
            const unsigned sym = vals[p];
            unsigned e = E_NOT_A_CODE;
            if (dc) {
                if (sym <= 16) e = make_entry((unsigned)l, sym, 0u);                   // a category above 16 is no symbol
            } else {
                e = make_entry((unsigned)l, sym & 15u, sym == 0 ? E_RUN_END : sym >> 4);
            }
            const unsigned first = code << (16 - l), last = first + (1u << (16 - l)) - 1u;     // the 16-bit windows that start with this code
            if (l <= L1_BITS) {
                const unsigned lo = code << (L1_BITS - l);
                for (unsigned f = 0; f < (1u << (L1_BITS - l)); ++f) t.l1[lo + f] = (uint16_t)e;
            } else if (first < HI_FIRST) {
                return false;                                        // a long code outside the hi table's region
            }
            for (unsigned w = first < HI_FIRST ? HI_FIRST : first; w <= last; ++w) t.hi[w - HI_FIRST] = (uint16_t)e;
        }
        if (code > (1u << l)) return false;
        code <<= 1;
    }
    return true;
}

// the table selectors of the blocks of one period of the MCU's table sequence, two bits each (Setup::tdmask); false: period too long
inline bool pack_td_sequence(const int* seq, int period, unsigned* mask)
{
    if (period > MAX_PERIOD) return false;
    unsigned m = 0;
    for (int i = 0; i < period; ++i) m |= (unsigned)(seq[i] & 3) << (2 * i);
    *mask = m;
    return true;
}


// ---- the decode step (device and host) ----
JPEZY_HD int extend(int v, int cat) { return (v & (1 << (cat - 1))) ? v : v - ((1 << cat) - 1); }

// The decoder's state besides the cursor: block inside the table period (doubled: the shift that finds its table selector), zig-zag
// index (0: a DC symbol comes next), the offsets (uint16 units) of the block's table pair and of the AC half inside it.
struct Walk {
    unsigned b2, k, tdoff, kb, nblocks;
    unsigned long long tdm;              // table offset of block b of a period in bits [2 b + 12 : 2 b + 11]
    JPEZY_HD void init(unsigned b_, unsigned k_, unsigned tdmask)
    {
        b2 = 2u * b_; k = k_; nblocks = 0;
        kb = k_ ? 3u * TABLE_U16 : 0u;
        tdm = 0;
        for (unsigned q = 0; q < (unsigned)MAX_PERIOD; ++q) tdm |= (unsigned long long)((tdmask >> (2u * q)) & 3u) << (2u * q + 11u);
        tdoff = (unsigned)(tdm >> b2) & 0x1800u;
    }
    JPEZY_HD unsigned b() const { return b2 >> 1; }
};
static_assert(TABLE_U16 == 2048u, "Walk::tdm places the table selector at the bits of TABLE_U16");

// One symbol.  DC and AC symbols run through the same instructions (the lanes of a wave are at unrelated places of their blocks): a DC
// symbol is a (run 0, size = category) symbol at k = 0 from the DC table.  Straight-line code: every decision is a select; the two table
// reads do not depend on each other.  Bits that are no code, or a run past the end of the block: EMIT returns false (the true decode hit
// it: the stream is bad); a synchronisation pass -- which may well be decoding from a wrong guess, i.e. garbage -- abandons the block
// (one bit on for a non-code) and carries on, so that it can still fall into step further down.
// EMIT: coefficients of blocks gidx + nblocks < total go to out (DC: the difference, made absolute by the DC pass; to dc_out[block] if given).
template <bool EMIT, class CursorT>
JPEZY_HD bool decode_step(const uint16_t* tabs, unsigned bpm, unsigned tdmask, CursorT& c, Walk& s,
                                            unsigned long long gidx, unsigned total, int16_t* out, int16_t* dc_out = nullptr)
{
    (void)tdmask;
    const uint32_t ahead = c.prefetch();                               // used only when this symbol crosses a word boundary
    const uint32_t bits = c.peek32();
    const unsigned tb = s.tdoff + s.kb;                                // dc[td] or ac[td]
    const unsigned e1 = tabs[tb + (bits >> (32 - L1_BITS))];
    const unsigned e2 = tabs[tb + (1u << L1_BITS) + ((bits >> 16) & ((1u << HI_BITS) - 1u))];      // (independent of e1: both reads are in flight together)
    const unsigned e = bits >= (HI_FIRST << 16) ? e2 : e1;
    const unsigned skipm1 = e & 31u, run = e >> 10;
    const unsigned kk = s.k + run + 1u;                               // zig-zag index after this symbol (beyond 63: the block is over)
    if (EMIT) {
        const unsigned sz = (e >> 5) & 31u;
        if (sz == E_S_NONE || (run != E_RUN_END && kk > 64u)) return false;
        const bool dc_sym = dc_out && s.k == 0u;                      // (a DC difference of zero has no value bits but is written all the same:
        if ((sz || dc_sym) && gidx + s.nblocks < total) {             //  dc_out is not zeroed beforehand)
            const unsigned len = skipm1 + 1u - sz;
            const int16_t v = sz ? (int16_t)extend((int)((bits << len) >> (32u - sz)), (int)sz) : (int16_t)0;
            // dc_out (may be null): the DC differences go to an array of their own, one per block -- the prefix sums then run over 2 bytes per
            // block instead of one 128-byte line per block
            if (dc_sym) dc_out[gidx + s.nblocks] = v;
            else out[(gidx + s.nblocks) * 64 + kk - 1u] = v;
        }
    }
    const bool endb = kk > 63u;
    s.k = endb ? 0u : kk;
    s.kb = endb ? 0u : 3u * TABLE_U16;
    s.nblocks += endb ? 1u : 0u;
    const unsigned n2 = s.b2 + 2u == 2u * bpm ? 0u : s.b2 + 2u;
    s.b2 = endb ? n2 : s.b2;
    s.tdoff = (unsigned)(s.tdm >> s.b2) & 0x1800u;
    c.advance(c.pos + skipm1 + 1u, ahead);
    return true;
}


}  // namespace huffdec
}  // namespace jpezy_dev
