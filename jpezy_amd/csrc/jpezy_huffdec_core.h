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
// An entry says everything the decoder's state machine needs: E_VALID | E_EOB (AC symbol 0x00) | run << 9 | s << 4 | (length - 1), with s
// the number of value bits that follow the code (AC: low nibble of the symbol; DC: the symbol, a category above 16 has no entry).  An unused
// slot is E_NOT_A_CODE: no E_VALID, and otherwise an end-of-block of one bit -- what a synchronisation walk does with bits that are no
// code (abandon the block, move one bit on) then needs no test of its own (round 4); the coefficient pass tests E_VALID.
constexpr int L1_BITS = 10, HI_BITS = 10;
constexpr unsigned E_VALID = 0x8000u, E_EOB = 0x2000u, E_NOT_A_CODE = E_EOB;
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
                if (sym <= 16) e = E_VALID | (sym << 4) | (unsigned)(l - 1);           // a category above 16 is no symbol
            } else {
                e = E_VALID | (sym == 0 ? E_EOB : 0u) | ((sym >> 4) << 9) | ((sym & 15u) << 4) | (unsigned)(l - 1);
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

// The decoder's state besides the cursor: block inside the table period, zig-zag index (0: a DC symbol comes next), and the
// offset of the block's table pair inside the Setup (in uint16 units).
struct Walk {
    unsigned b, k, tdoff, nblocks;
    JPEZY_HD void init(unsigned b_, unsigned k_, unsigned tdmask)
    {
        b = b_; k = k_; nblocks = 0;
        tdoff = ((tdmask >> (2u * b)) & 3u) * TABLE_U16;
    }
};

// One symbol.  DC and AC symbols run through the same instructions (the lanes of a wave are at unrelated places of their blocks): a DC
// symbol is a (run 0, size = category) symbol at k = 0 from the DC table.  Straight-line code: every decision is a select.
// An invalid code, or a run past the end of the block: EMIT returns false (the true decode hit it: the stream is bad); a
// synchronisation pass -- which may well be decoding from a wrong guess, i.e. garbage -- abandons the block, moves one bit on and
// carries on, so that it can still fall into step further down.
// EMIT: coefficients of blocks gidx + nblocks < total go to out (DC: the difference, made absolute by the DC pass).
template <bool EMIT, class CursorT>
JPEZY_HD bool decode_step(const uint16_t* tabs, unsigned bpm, unsigned tdmask, CursorT& c, Walk& s,
                                            unsigned long long gidx, unsigned total, int16_t* out)
{
    const uint32_t ahead = c.prefetch();                               // used only when this symbol crosses a word boundary
    const uint32_t bits = c.peek32();
    const unsigned tb = s.tdoff + (s.k ? 3u * TABLE_U16 : 0u);         // dc[td] or ac[td]
    const unsigned e1 = tabs[tb + (bits >> (32 - L1_BITS))];
    const unsigned e2 = tabs[tb + (1u << L1_BITS) + ((bits >> 16) & ((1u << HI_BITS) - 1u))];      // (independent of e1: both reads are in flight together)
    const unsigned e = bits >= (HI_FIRST << 16) ? e2 : e1;
    const unsigned len = (e & 15u) + 1u, sz = (e >> 4) & 31u, run = (e >> 9) & 15u;
    const unsigned kk = s.k + run + 1u;                               // zig-zag index after this symbol
    if (EMIT) {
        if (!(e & E_VALID) || kk > 64u) return false;
        if (sz && gidx + s.nblocks < total)
            out[(gidx + s.nblocks) * 64 + kk - 1u] = (int16_t)extend((int)((bits << len) >> (32u - sz)), (int)sz);
    }
    // bits that are no code come out of the table as a one-bit end-of-block (E_NOT_A_CODE), a run past the end of the block ends it too
    const unsigned skip = len + sz;
    const unsigned kn = (e & E_EOB) ? 64u : kk;
    const bool endb = kn >= 64u;
    s.k = endb ? 0u : kn;
    s.nblocks += endb ? 1u : 0u;
    const unsigned b1 = s.b + 1u == bpm ? 0u : s.b + 1u;
    s.b = endb ? b1 : s.b;
    s.tdoff = ((tdmask >> (2u * s.b)) & 3u) * TABLE_U16;
    c.advance(c.pos + skip, ahead);
    return true;
}


}  // namespace huffdec
}  // namespace jpezy_dev
