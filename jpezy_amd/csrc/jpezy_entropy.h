// jpezy_entropy.h -- GPU Huffman coder + bit packer + byte stuffer (internal; see jpezy_entropy.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace jpezy_dev {
namespace entropy {

// entry = (code << 8) | length in bits;  dc[t][category 0..11], ac[t][(run << 4) | size]   (t: 0 luma, 1 chroma)
struct CodeTables {
    uint32_t dc[2][16];
    uint32_t ac[2][256];
};

struct Job {
    const int16_t* coeffs;        // device, [frame][mcu][bpm][64] zig-zag
    size_t coeffs_per_frame;      // int16 elements
    const CodeTables* tables;     // device
    unsigned blocks_per_frame;    // coded blocks: 6 per MCU (gray: the two chroma blocks are coded as zero blocks)
    int bpm;                      // stored blocks per MCU: 6 colour, 4 gray
    int n_frames;
};

// A prefix sum kept in two levels: loc[i] = exclusive sum inside the 256-element tile of the kernel that produced the values
// (formed there, in registers and LDS), base[t] = exclusive sum of the tile totals (a scan over n/256 elements instead of n:
// two small launches).  at(i) for i in [0, n] -- at(n) is the grand total.
struct Offsets {
    const unsigned long long* base;   // [tiles256(n) + 1]
    const uint32_t* loc;              // [n]
    size_t n;
#ifdef __HIPCC__
    __device__ __forceinline__ unsigned long long at(size_t i) const
    {
        return i >= n ? base[(n + 255) >> 8] : base[i >> 8] + loc[i];
    }
#endif
};
inline size_t tiles256(size_t n) { return (n + 255) / 256; }

size_t scan_tmp_elems(size_t n);  // uint64 scratch elements launch_scan_u32 needs for n inputs
size_t chunk_bytes();             // granularity of the stuffing pass: U strides must be multiples of it

// code length of every block as an exclusive offset inside its 256-block tile (loc) + the tile totals; scan the totals with
// launch_scan_u32(tile_total, base, tiles256(n), tmp) to complete the Offsets
hipError_t launch_block_bits(const Job& job, uint32_t* loc, uint32_t* tile_total, unsigned* status, hipStream_t s);
// out[0..n) exclusive prefix sums, out[n] the total
hipError_t launch_scan_u32(const uint32_t* in, unsigned long long* out, size_t n, unsigned long long* tmp, hipStream_t s);
hipError_t launch_scan_u64(const unsigned long long* in, unsigned long long* out, size_t n, unsigned long long* tmp, hipStream_t s);
hipError_t launch_frame_totals(Offsets off, size_t per, int n_frames, unsigned long long* dst, hipStream_t s);
hipError_t launch_emit(const Job& job, Offsets bitoff, uint32_t* U, size_t u_stride_words, hipStream_t s);
// 0xFF bytes per 64-byte chunk of the unstuffed streams, again as tile-local offsets + tile totals
hipError_t launch_ff_count(const uint32_t* U, size_t u_stride_words, const unsigned long long* frame_bytes, int n_frames,
                           uint32_t* loc, uint32_t* tile_total, hipStream_t s);
hipError_t launch_stuff(const uint32_t* U, size_t u_stride_words, const unsigned long long* frame_bytes, int n_frames,
                        Offsets ff_before, uint8_t* out, size_t out_stride, hipStream_t s);

// device-resident variant (no host sync).  launch_zero_streams clears, per frame, the part of the unstuffed stream buffer
// the later kernels touch (the stream rounded up to a chunk, plus one) and publishes bytes[f] = ceil(bits of frame f / 8)
// from the block offsets (per = blocks per frame).  launch_plan_and_header: the fit/size/EOI decision and the
// header copy, one workgroup per frame; it consumes AND clears status[f] (the buffer must be zero before the first use).
hipError_t launch_zero_streams(uint32_t* U, size_t u_stride_words, Offsets off, size_t per, unsigned long long* bytes,
                               int n_frames, hipStream_t s);
hipError_t launch_plan_and_header(unsigned long long* bytes, Offsets ffoff, size_t chunks_per_frame, unsigned* status,
                                  int n_frames, const uint8_t* hdr, size_t hdr_len, uint8_t* out, size_t out_stride, long long* sizes,
                                  hipStream_t s);

// ---- one coding pass (see jpezy_entropy.hip): every block is coded once, workgroup by workgroup ("tile" = 256 coded
// blocks of one frame), into tile streams S[frame][tile][tile_stream_bytes()]; launch_tile_bases turns the tile totals into
// frame-relative bit offsets base[frame][tiles + 1], bytes[frame] and the first tile of every assemble_piece_bytes() of
// output; launch_assemble forms the unstuffed streams U (stride a multiple of assemble_piece_bytes()) and the 0xFF counts
// per 64-byte chunk in the two-level form launch_stuff / launch_plan_and_header read.
size_t tile_stream_bytes();
size_t assemble_piece_bytes();
hipError_t launch_code_tiles(const Job& job, uint32_t* S, uint32_t* tile_total, unsigned* status, hipStream_t s);
hipError_t launch_tile_bases(const uint32_t* tile_total, unsigned tiles_per_frame, int n_frames, unsigned long long* base,
                             unsigned long long* bytes, uint32_t* first_tile, unsigned ft_stride, hipStream_t s);
// u_stride_words * 4 / assemble_piece_bytes() pieces per frame, at most ft_stride (first_tile's frame stride)
hipError_t launch_assemble(const uint32_t* S, const unsigned long long* base, const unsigned long long* bytes, const uint32_t* first_tile,
                           unsigned ft_stride, unsigned tiles_per_frame, int n_frames, uint32_t* U, size_t u_stride_words, uint32_t* loc,
                           uint32_t* ff_tile_total, hipStream_t s);

}  // namespace entropy
}  // namespace jpezy_dev
