// jpezy_entropy.h -- GPU Huffman coder + bit packer + byte stuffer (internal; see jpezy_entropy.hip)
#pragma once
#include "jpezy_experiment.h"
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace jpezy_dev {
namespace entropy {

// entry = (code << 8) | length in bits;  dc[t][category 0..11], ac[t][(run << 4) | size]   (t: 0 luma, 1 chroma)
// fast[t][(run << 6) | (v + 32)], run 0..15, v -32..31 (v != 0): the AC code of (run, size(v)) with the value bits already
// appended, (bits << 5) | total length (at most 16 + 6 bits) -- one lookup and one append per coefficient instead of
// category -> code lookup -> sign fix -> two appends (round 3; larger values and runs over 15 keep the general path)
struct alignas(16) CodeTables {
    uint32_t dc[2][16];
    uint32_t ac[2][256];
    uint32_t fast[2][1024];
};

struct Job {
    const int16_t* coeffs;        // device, [frame][mcu][bpm][64] zig-zag
    size_t coeffs_per_frame;      // int16 elements
    const CodeTables* tables;     // device
    unsigned blocks_per_frame;    // coded blocks: 6 per MCU (gray: the two chroma blocks are coded as zero blocks)
    int bpm;                      // stored blocks per MCU: 6 colour, 4 gray
    int n_frames;
};

inline size_t tiles256(size_t n) { return (n + 255) / 256; }

// device-resident form of launch_stuff: whole files (header, stuffed stream, EOI), sizes and per-frame verdicts on the device
struct FilePlan {
    const uint8_t* hdr = nullptr;       // nullptr: streams only (the host adds header and EOI)
    size_t hdr_len = 0;
    const unsigned* latched = nullptr;  // [frames] error flags as latched by launch_tile_bases
    long long* sizes = nullptr;         // [frames] file size, or JPEZY_E_FORMAT (-5) / JPEZY_E_NOSPACE (-6)
};

size_t scan_tmp_elems(size_t n);  // uint64 scratch elements launch_scan_u32 needs for n inputs
size_t chunk_bytes();             // granularity of the stuffing pass (64)
size_t tile_stream_bytes();       // stride of a tile's stream in the scratch S (worst case: 256 blocks x 208 bytes)
size_t assemble_piece_bytes();    // bytes of U one assembling / stuffing workgroup handles (16 KB): U strides are multiples of it

// out[0..n) exclusive prefix sums, out[n] the total
hipError_t launch_scan_u32(const uint32_t* in, unsigned long long* out, size_t n, unsigned long long* tmp, hipStream_t s);
hipError_t launch_scan_u64(const unsigned long long* in, unsigned long long* out, size_t n, unsigned long long* tmp, hipStream_t s);

// One coding pass (see jpezy_entropy.hip): every block is coded once, workgroup by workgroup ("tile" = 256 coded blocks of
// one frame), into tile streams S[frame][tile][tile_stream_bytes()], tile_total[frame][tile] bits each; status[frame] |= 1
// for a coefficient outside the Annex-K tables.
hipError_t launch_code_tiles(const Job& job, uint32_t* S, uint32_t* tile_total, unsigned* status, hipStream_t s);
// frame-relative bit offsets base[frame][tiles + 1], bytes[frame] = ceil(bits / 8), first_tile[frame][ft_stride] = the tile
// the first bit of every assemble_piece_bytes() of output lies in; latched != nullptr: latched[f] = status[f], status[f] = 0
hipError_t launch_tile_bases(const uint32_t* tile_total, unsigned tiles_per_frame, int n_frames, unsigned long long* base,
                             unsigned long long* bytes, uint32_t* first_tile, unsigned ft_stride, unsigned* status, unsigned* latched,
                             hipStream_t s);
// unstuffed streams U (stride a multiple of assemble_piece_bytes()) and the 0xFF bytes in front of every 64-byte chunk inside
// its piece (ff_loc[frame][chunk]) + per piece (ff_tile_total[frame][piece]): a prefix sum in two levels whose upper level
// every stuffing workgroup adds up for itself.  Frames with assemble_scans_tiles_itself(tiles): launch_tile_bases is NOT
// needed -- the kernel scans tile_total itself, publishes bytes[frame] and (latched != nullptr) latches + clears status;
// base / first_tile are then unused.  Larger frames: launch_tile_bases first (at most ft_stride pieces).
bool assemble_scans_tiles_itself(size_t tiles_per_frame);
hipError_t launch_assemble(const uint32_t* S, const uint32_t* tile_total, const unsigned long long* base, unsigned long long* bytes,
                           const uint32_t* first_tile, unsigned ft_stride, unsigned tiles_per_frame, int n_frames, uint32_t* U,
                           size_t u_stride_words, uint32_t* loc, uint32_t* ff_tile_total, unsigned* status, unsigned* latched, hipStream_t s);
hipError_t launch_stuff(const uint32_t* U, size_t u_stride_words, const unsigned long long* frame_bytes, int n_frames,
                        const uint32_t* ff_loc, const uint32_t* ff_tile_total, uint8_t* out, size_t out_stride, FilePlan plan, hipStream_t s);
// dst[f] = 0xFF bytes of frame f (the host-delivered form sizes its output buffer from it)
hipError_t launch_ff_frame_totals(const uint32_t* ff_tile_total, const unsigned long long* bytes, size_t u_stride_words, int n_frames,
                                  unsigned long long* dst, hipStream_t s);

}  // namespace entropy
}  // namespace jpezy_dev
