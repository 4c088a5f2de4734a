// jpezy_huffdec.h -- GPU Huffman decoder of a baseline scan without restart markers (internal; see jpezy_huffdec.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "jpezy_huffdec_core.h"     // tables, entry format, the decode step (shared with the CPU harness tests/fuzz/huffdec_core_fuzz.cpp)

namespace jpezy_dev {
namespace huffdec {

unsigned subseq_bits();
size_t chunk_bytes();

// ---- batch form (round 3): many files per launch.  Every kernel above exists once more with the file as an index: a file owns a
// range of 64-byte chunks of the concatenated scans (unstuffing) and a range of subsequence slots (decoding); a workgroup never
// straddles two files (wg_file / wg_first say whose subsequences it takes), every file names its tables (setup).  A "file" is any
// independent stream that starts in the known state: a file's scan, or one restart interval of a scan. ----
struct BatchFile {
    unsigned chunk0, n_chunks;          // 64-byte chunks of the concatenated scan buffer: bytes [chunk0 * 64, chunk0 * 64 + n_bytes)
    unsigned n_bytes;
    unsigned sub0, n_sub_max, n_sub;    // subsequence slots; n_sub = the ones that hold data once the stuffing is gone (device)
    unsigned long long u_off;           // byte offset of the file's unstuffed stream in U (a multiple of 4)
    unsigned u_words;                   // 32-bit words of U the file may read
    unsigned removed;                   // stuffing bytes removed (device)
    unsigned first_marker;              // n_bytes may be an upper bound: offset of the first marker inside it (device, atomicMin; the host sets all ones)
    unsigned periodic;                  // lanes whose first walk left their subsequence in the state they entered it (device): a periodic stream
    unsigned long long coeff_off;       // int16 offset of the file's coefficients
    unsigned total_blocks, nmcu, bpm, ncomp;
    unsigned cstart[3], ccount[3];      // component c owns blocks [cstart, cstart + ccount) of every MCU
    unsigned changed[4];                // per launch: lanes that moved / lanes left pending / workgroup-last lanes among the moved /
                                        // lanes that moved at the launch's first step (device)
    unsigned error;
    unsigned setup;                     // which of the call's tables the file / stream decodes with
    unsigned long long last_bit;
};

hipError_t launch_unstuff_count_batch(const uint8_t* S, BatchFile* F, unsigned n_files, unsigned total_chunks, uint32_t* counts, hipStream_t s);
hipError_t launch_unstuff_copy_batch(const uint8_t* S, BatchFile* F, unsigned n_files, unsigned total_chunks, const unsigned long long* removed_before,
                                     uint8_t* U, hipStream_t s);       // also fills F[].removed, n_sub and resets the per-file flags
hipError_t launch_speculate_batch(const Setup* setups, const uint32_t* U, const BatchFile* F, const unsigned* wg_file, const unsigned* wg_first,
                                  unsigned n_wg, unsigned n_slots, unsigned long long* proposal, uint32_t* exit_state, hipStream_t s);
// active[f] (host-written between the phases): refinement passes -- the file still has lanes to settle; emit / DC pass -- the file converged
hipError_t launch_sync_batch(const Setup* setups, const uint32_t* U, BatchFile* F, const unsigned* wg_file, const unsigned* wg_first, unsigned n_wg,
                             const unsigned* active, uint32_t* exit_state, uint32_t* last_entry, unsigned* nblocks, int max_inner, hipStream_t s);
hipError_t launch_reset_changed_batch(BatchFile* F, const unsigned* active, unsigned n_files, hipStream_t s);   // changed[] = 0 for the active files
hipError_t launch_emit_batch(const Setup* setups, const uint32_t* U, BatchFile* F, const unsigned* wg_file, const unsigned* wg_first, unsigned n_wg,
                             const unsigned* active, const uint32_t* exit_state, const unsigned long long* blocks_before, int16_t* coeffs, hipStream_t s);
// short streams that share ONE set of tables (setups[0]) and start in the known state: a lane walks a whole stream (symbols, coefficients, DC
// predictors); needs the unstuffed streams (launch_unstuff_*_batch) and zeroed coefficients; sets F[].error / last_bit like the emit launch
hipError_t launch_stream_per_lane(const Setup* setups, const uint32_t* U, BatchFile* F, unsigned n_files, int16_t* coeffs, hipStream_t s);
hipError_t launch_dc_prefix_batch(int16_t* coeffs, const BatchFile* F, const unsigned* active, unsigned n_files, hipStream_t s);

// ---- single scan.  Round 4: the whole chain is enqueued without a host synchronisation in between -- the numbers the host used to fetch
// after the unstuffing (where the segment ends, how many subsequences hold data) stay on the device in a ScanState, which the kernels read.
struct ScanState {
    unsigned pad0;
    unsigned error;                     // emit launch: an invalid code
    unsigned long long last_bit;        // emit launch: the bit behind the last block (all ones: not reached)
    unsigned changed[4];                // first synchronisation launch (see launch_sync)
    unsigned long long first_marker;    // offset of the first marker in the uploaded bytes (all ones: none)
    unsigned long long removed;         // stuffing bytes removed in front of it
    unsigned changed2[4];               // second synchronisation launch
    unsigned n_sub;                     // subsequences that hold data
    unsigned periodic;                  // lanes that left their subsequence in the state they entered it (speculation): a periodic stream
    unsigned pad1[2];
};
static_assert(sizeof(ScanState) == 80, "ScanState is read back in one copy");
// A stream that does not synchronise (periodic data: flat areas) is the host decoder's: refinement would walk it lane after lane.  After the
// first synchronisation launch (counters `changed`, see launch_sync): more than half of the lanes moved at its first step AND more than a
// quarter are still pending after its propagation steps.  (Round 3 and most of round 4 judged by the first half alone, which also sent dense
// high-quality noise away -- blocks longer than a subsequence leave few lanes in step after the speculation, but the corrections then settle
// within a few steps: most of the wrong runs are short.)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline bool scan_hopeless(const unsigned* changed, unsigned n_sub) { return changed[3] > n_sub / 2u + 16u && changed[1] > n_sub / 4u; }
// Did the two synchronisation launches enqueued blindly settle the scan?  changed: the first launch's counters, changed2: the second's (which
// left at once unless the first one left something for it).  The host and the guarded coefficient / DC launches ask the same function.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline bool scan_settled(const unsigned* changed, const unsigned* changed2, unsigned n_sub)
{
    if (n_sub == 0u) return false;
    if (changed[1] == 0u && changed[2] == 0u) return true;                       // nothing pending after the first launch
    if (scan_hopeless(changed, n_sub)) return false;                             // the second launch left at once
    return changed2[1] == 0u && changed2[2] == 0u;
}
hipError_t launch_scan_state_init(ScanState* st, hipStream_t s);
// S: the file from the first byte of the scan on (n_max bytes).  The count launch also finds where the entropy-coded segment ends
// (st->first_marker: offset of the first 0xFF that is followed by anything but 0x00 or is the last byte; still all ones when there is none);
// the copy launch stops there and leaves the stuffing bytes it removed and the number of subsequences in st.
hipError_t launch_unstuff_count(const uint8_t* S, size_t n_max, uint32_t* counts, ScanState* st, hipStream_t s);
hipError_t launch_unstuff_copy(const uint8_t* S, size_t n_max, const unsigned long long* removed_before, uint8_t* U, ScanState* st, hipStream_t s);
// speculation pass: every lane walks from a guess `overflow` subsequences in front of its own through its own and leaves, for its own
// subsequence, the entry state it arrived in (last_entry), the exit state, the blocks completed and the marks -- the state of the
// synchronisation phase as if every lane had already confirmed its predecessor's exit state, true wherever the guesses have fallen into step.
// n_sub_max sizes the launch, st->n_sub says which lanes have data.
hipError_t launch_speculate(const Setup* S, const uint32_t* U, size_t u_words, unsigned n_sub_max, const ScanState* st, uint32_t* exit_state,
                            uint32_t* last_entry, unsigned* nblocks, uint32_t* marks, unsigned* mark_blocks, hipStream_t s);
// u_words: 32-bit words of U that may be read (the rest of a workgroup's window reads as zero)
// changed[0] += number of lanes whose exit state moved, changed[1] += lanes left pending by the max_inner cut-off, changed[2] += moved lanes that
// are a workgroup's last (the next workgroup may not have seen the new value), changed[3] += lanes whose first decode of the launch moved their state.  changed[1] == 0 && changed[2] == 0 after a launch: the states are
// the fixed point, i.e. the sequential decode.  max_inner: propagation steps inside a workgroup (1: every lane decodes once from its
// predecessor's current exit state and nothing more); a workgroup none of whose lanes has a new entry state leaves at once.
// prev (may be null): the counters of the launch before this one -- nothing pending there, or more than half of the proposals wrong at its
// first step (a stream that does not synchronise: the host decoder's), and this launch leaves at once: it can be enqueued blindly.
// marks / mark_blocks: (emit_parts() - 1) words per subsequence each -- the state at every emit_parts()-th of a subsequence and the blocks completed
// before it, left by a lane's last walk; the coefficient launch starts a lane at each (emit_parts() times shorter serial walks).
unsigned emit_parts();
hipError_t launch_sync(const Setup* S, const uint32_t* U, size_t u_words, unsigned n_sub_max, const ScanState* st, uint32_t* exit_state, uint32_t* last_entry,
                       unsigned* nblocks, uint32_t* marks, unsigned* mark_blocks, unsigned* changed, const unsigned* prev, int max_inner, hipStream_t s);
// guarded: the launch is enqueued before anybody has looked at the synchronisation launches' counters and leaves at once unless scan_settled()
hipError_t launch_emit(const Setup* S, const uint32_t* U, size_t u_words, unsigned n_sub_max, ScanState* st, const uint32_t* exit_state,
                       const uint32_t* marks, const unsigned* mark_blocks, const unsigned long long* blocks_before, int16_t* out, int16_t* dc_out,
                       bool guarded, hipStream_t s);       // dc_out (may be null): the DC differences go to dc_out[block], every block's
// DC differences -> values for all components of one scan in three launches (component q owns blocks [cstart[q], cstart[q] + ccount[q]) of
// every MCU); scratch: dc_prefix_scratch_ints(nmcu, largest ccount) ints
size_t dc_prefix_scratch_ints(size_t nmcu, unsigned max_count);
// guard (may be null): leave at once unless scan_settled(guard)
// dc (may be null): the differences lie in dc[block] instead of the coefficients' DC slots (launch_emit's dc_out); the values go to the coefficients
hipError_t launch_dc_prefix(int16_t* coeffs, int16_t* dc, unsigned bpm, unsigned ncomp, const unsigned cstart[3], const unsigned ccount[3], size_t nmcu,
                            int* scratch, const ScanState* guard, hipStream_t s);

}  // namespace huffdec
}  // namespace jpezy_dev
