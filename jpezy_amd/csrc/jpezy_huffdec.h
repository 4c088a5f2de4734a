// jpezy_huffdec.h -- GPU Huffman decoder of a baseline scan without restart markers (internal; see jpezy_huffdec.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace jpezy_dev {
namespace huffdec {

struct alignas(16) Table {
    uint16_t look[512];   // 9-bit lookup: (code length << 8) | symbol, 0 when the code is longer than 9 bits
    // canonical decoding of the longer codes without a loop: limit[l - 9] = first 16-bit left-aligned code word that is NOT
    // a code of length <= l (non-decreasing in l), so the length of a long code is 10 + #{l in 10..15: word >= limit};
    // off[l] = index of the first symbol of length l minus the first code of length l
    uint32_t limit[8];
    int off[17];
    uint8_t val[256];
};

struct Setup {
    Table dc[3], ac[3];       // indexed by the scan component's table selector Td (the reference uses Td for both)
    int bpm;                  // period of the table sequence over the blocks of an MCU (divides the blocks per MCU)
    int btd[48];              // Td of block b of a period
    unsigned total_blocks;
    unsigned pad[2];
};

unsigned subseq_bits();
size_t chunk_bytes();

// ---- batch form (round 3): many files per launch.  Every kernel above exists once more with the file as an index: a file owns a
// range of 64-byte chunks of the concatenated scans (unstuffing) and a range of subsequence slots (decoding); a workgroup never
// straddles two files (wg_file / wg_first say whose subsequences it takes), every file brings its own tables. ----
struct BatchFile {
    unsigned chunk0, n_chunks;          // 64-byte chunks of the concatenated scan buffer: bytes [chunk0 * 64, chunk0 * 64 + n_bytes)
    unsigned n_bytes;
    unsigned sub0, n_sub_max, n_sub;    // subsequence slots; n_sub = the ones that hold data once the stuffing is gone (device)
    unsigned long long u_off;           // byte offset of the file's unstuffed stream in U (a multiple of 4)
    unsigned u_words;                   // 32-bit words of U the file may read
    unsigned removed;                   // stuffing bytes removed (device)
    unsigned long long coeff_off;       // int16 offset of the file's coefficients
    unsigned total_blocks, nmcu, bpm, ncomp;
    unsigned cstart[3], ccount[3];      // component c owns blocks [cstart, cstart + ccount) of every MCU
    unsigned changed[2];                // per pass: lanes that moved / lanes left pending (device)
    unsigned error, pad;
    unsigned long long last_bit;
};

hipError_t launch_unstuff_count_batch(const uint8_t* S, const BatchFile* F, unsigned n_files, unsigned total_chunks, uint32_t* counts, hipStream_t s);
hipError_t launch_unstuff_copy_batch(const uint8_t* S, BatchFile* F, unsigned n_files, unsigned total_chunks, const unsigned long long* removed_before,
                                     uint8_t* U, hipStream_t s);       // also fills F[].removed, n_sub and resets the per-file flags
hipError_t launch_speculate_batch(const Setup* setups, const uint32_t* U, const BatchFile* F, const unsigned* wg_file, const unsigned* wg_first,
                                  unsigned n_wg, unsigned n_slots, unsigned long long* proposal, uint32_t* exit_state, hipStream_t s);
// active[f] (host-written between the phases): refinement passes -- the file still has lanes to settle; emit / DC pass -- the file converged
hipError_t launch_sync_batch(const Setup* setups, const uint32_t* U, BatchFile* F, const unsigned* wg_file, const unsigned* wg_first, unsigned n_wg,
                             const unsigned* active, uint32_t* exit_state, uint32_t* last_entry, unsigned* nblocks, int max_inner, hipStream_t s);
hipError_t launch_emit_batch(const Setup* setups, const uint32_t* U, BatchFile* F, const unsigned* wg_file, const unsigned* wg_first, unsigned n_wg,
                             const unsigned* active, const uint32_t* exit_state, const unsigned long long* blocks_before, int16_t* coeffs, hipStream_t s);
hipError_t launch_dc_prefix_batch(int16_t* coeffs, const BatchFile* F, const unsigned* active, unsigned n_files, hipStream_t s);

hipError_t launch_unstuff_count(const uint8_t* S, size_t n, uint32_t* counts, hipStream_t s);
hipError_t launch_unstuff_copy(const uint8_t* S, size_t n, const unsigned long long* removed_before, uint8_t* U, hipStream_t s);
// speculation pass: fills exit_state with the best available guess of every subsequence's true exit state
// (proposal: n_sub uint64 of scratch)
hipError_t launch_speculate(const Setup* S, const uint32_t* U, size_t u_words, unsigned n_sub, unsigned long long* proposal,
                            uint32_t* exit_state, hipStream_t s);
// u_words: 32-bit words of U that may be read (the rest of a workgroup's window reads as zero)
// changed[0] += number of lanes whose exit state moved, changed[1] += lanes left pending by the max_inner cut-off.  max_inner: propagation steps inside a workgroup (1: every lane
// decodes once from its predecessor's current exit state and nothing more)
hipError_t launch_sync(const Setup* S, const uint32_t* U, size_t u_words, unsigned n_sub, uint32_t* exit_state, uint32_t* last_entry,
                       unsigned* nblocks, unsigned* changed, int max_inner, hipStream_t s);
hipError_t launch_emit(const Setup* S, const uint32_t* U, size_t u_words, unsigned n_sub, const uint32_t* exit_state,
                       const unsigned long long* blocks_before, int16_t* out, unsigned* error, unsigned long long* last_bit, hipStream_t s);
hipError_t launch_dc_gather(const int16_t* coeffs, unsigned bpm, unsigned start, unsigned count, size_t n, unsigned long long* d, hipStream_t s);
hipError_t launch_dc_scatter(int16_t* coeffs, unsigned bpm, unsigned start, unsigned count, size_t n, const unsigned long long* before,
                             hipStream_t s);

}  // namespace huffdec
}  // namespace jpezy_dev
