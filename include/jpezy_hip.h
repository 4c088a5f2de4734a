/*
 * jpezy_hip.h -- C-ABI of the MI355X (gfx950) baseline-JPEG hot path of falgon/jpezy.
 *
 * The reference has no FFI seam: its hot path is a set of private member functions of two class
 * templates.  This header is the seam a maintainer binds instead; every entry
 * point cites the reference code it replaces (paths relative to /root/reference/src/).
 *
 * Conventions: plain C, no exceptions cross the boundary.  Functions returning int give 0 on success
 * and a negative jpezy_status otherwise; jpezy_hip_last_error() (thread-local) says why.  The caller
 * owns every buffer it passes; the library owns streams and scratch inside the opaque context.  A
 * context is used by one thread at a time; distinct contexts (one per GPU) may run concurrently.
 * There is NO CPU fallback: without a HIP device every compute entry point fails with JPEZY_E_NODEVICE.
 *
 * Device pointers: every coefficient pointer handed to a *_dev / *_gpu entry point must be 16-byte aligned (the
 * kernels move coefficients as 16-byte accesses; one MCU is 768 or 512 bytes, so only the base matters) -- a
 * misaligned one is refused with JPEZY_E_BADARG.  Pixel planes may have any alignment (an unaligned kernel variant
 * takes them).
 * Streams: the *_dev entry points are asynchronous on the caller's stream and one context may be driven on several
 * streams by its one thread.  The context-wide device tables (dequantiser constants of jpezy_dequant_idct*_dev, the
 * cached JFIF header of jpezy_write_jpeg_gpu_dev) are rewritten only when the caller's tables / comment / size
 * change; the library then waits for the WHOLE device first (launches on other streams may still read them), and it
 * refuses (JPEZY_E_BADARG) to do so while `stream` is being captured into a hipGraph: make the first call with new
 * tables outside the capture.
 *
 * Coefficient buffer layout (both directions), per frame:
 *     int16_t coeffs[mcu_rows][mcu_cols][B][64]
 * MCUs row-major with mcu_cols = ceil(W/16), mcu_rows = ceil(H/16) (encoder/jpezy_encoder.hpp:55-56);
 * B = 6 blocks in the order Y0(top-left) Y1(top-right) Y2(bottom-left) Y3(bottom-right) Cb Cr
 * (jpezy_encoder.hpp:227-242), or B = 4 (Y0..Y3) when gray != 0 -- in GRAY_MODE the reference zeroes
 * the chroma blocks (jpezy_encoder.hpp:61-64) so they are not materialised; inside a block the 64
 * quantised coefficients are in ZIG-ZAG order: coeffs[n] = q[ZZ[n]] (jpezy.hpp:36-45).
 * Pixel planes are planar 8-bit r, g, b with row stride W, W*H bytes each (jpezy_encoder.hpp:105,266).
 */
#ifndef JPEZY_HIP_H
#define JPEZY_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jpezy_ctx jpezy_ctx;

enum jpezy_status {
    JPEZY_OK = 0,
    JPEZY_E_BADARG = -1,      /* null pointer, non-positive or > 65535 dimension, ...                */
    JPEZY_E_NODEVICE = -2,    /* no HIP device / device index out of range                           */
    JPEZY_E_HIP = -3,         /* a HIP runtime call failed (message in jpezy_hip_last_error)         */
    JPEZY_E_UNSUPPORTED = -4, /* decode layout other than jpezy's own 2x2,1x1,1x1 3-component files  */
    JPEZY_E_FORMAT = -5,      /* malformed JPEG / Huffman stream (the reference throws runtime_error) */
    JPEZY_E_NOSPACE = -6      /* output buffer too small                                             */
};

const char* jpezy_hip_last_error(void);
int jpezy_hip_device_count(void);
/* 1 when the library was built with a timing-probe switch that gives wrong results (development builds of tools/ab/ab_build.py,
 * jpezy_amd/csrc/jpezy_experiment.h); 0 for every build that may be shipped or tested. */
int jpezy_hip_is_experimental_build(void);

/* ---- context: one per GPU.  Owns a HIP stream, staging buffers and the device constant tables ---- */
jpezy_ctx* jpezy_ctx_create(int device);
void jpezy_ctx_destroy(jpezy_ctx* ctx);
int jpezy_ctx_sync(jpezy_ctx* ctx);
int jpezy_ctx_device(const jpezy_ctx* ctx);
void* jpezy_ctx_stream(const jpezy_ctx* ctx);   /* the context's own hipStream_t (non-blocking) */

/* ---- geometry helpers (jpezy_encoder.hpp:55-56) ---- */
int jpezy_mcu_cols(int W);
int jpezy_mcu_rows(int H);
size_t jpezy_coeff_count(int W, int H, int gray);   /* int16 elements per frame */

/*
 * ENCODE compute stage.  Replaces, for every MCU of every frame, the reference's
 *   encoder::make_YCC      (encoder/jpezy_encoder.hpp:90-144)   RGB->YCbCr, edge clamp, 2x2 decimation
 *   RGB::Y/Cb/Cr           (:244-256)                            truncating FP64 colour conversion
 *   encoder::DCT           (:146-166)                            8x8 FDCT, int(sum*cu*cv/4)
 *   encoder::quantization  (:168-172)                            Annex-K, C++ int division
 *   the ZZ read order of encode_huffman (:195,212)               zig-zag
 * i.e. everything in the MCU loop (:58-67) except encode_huffman.  Results are bit-identical to the
 * reference arithmetic evaluated in IEEE binary64 without contraction ("exactness").
 *
 * jpezy_fdct_quant: host buffers (what encoder::encode calls).  r,g,b: n_frames consecutive W*H planes.
 */
int jpezy_fdct_quant(jpezy_ctx* ctx, const uint8_t* r, const uint8_t* g, const uint8_t* b, int W, int H,
                     int gray, int n_frames, int16_t* coeffs);
/*
 * The host-buffer entry points (jpezy_fdct_quant, jpezy_dequant_idct, jpezy_encode_jpeg) stream: the call is cut into chunks of
 * about `n` bytes of input (default 4 MiB: MCU-row bands of a large frame, or several small frames) that flow through a ring of
 * pinned staging buffers -- upload of chunk c + 1, kernel of chunk c and download of chunk c - 1 overlap, PCIe runs in both
 * directions at once and the device footprint is the ring, not the batch (host path).  Tuning / test knob.
 */
void jpezy_ctx_set_host_chunk_bytes(jpezy_ctx* ctx, size_t n);
/*
 * jpezy_fdct_quant_dev: same, all pointers are DEVICE pointers; plane_stride = bytes between
 * consecutive frames of one plane (>= W*H); asynchronous on `stream`, a hipStream_t with HIP's own
 * meaning (NULL = the default stream; jpezy_ctx_stream() = the context's private stream).  Used by
 * batch drivers / the benchmark with inputs resident in HBM.
 */
int jpezy_fdct_quant_dev(jpezy_ctx* ctx, const uint8_t* d_r, const uint8_t* d_g, const uint8_t* d_b,
                         size_t plane_stride, int W, int H, int gray, int n_frames, int16_t* d_coeffs,
                         void* stream);

/*
 * DECODE compute stage.  Replaces, for every MCU, the reference's
 *   decoder::inverse_quantization (decoder/jpezy_decoder.hpp:645-650)
 *   decoder::inverse_dct          (:652-670)      int(sum/4 + 128), no clamp
 *   decode_mcu's replication      (:519-524)      nearest-neighbour chroma upsample
 *   decoder::make_rgb / to_r,g,b / revise_value (:531-578, 672-676)
 * for files with jpezy's own layout (3 components, sampling 2x2,1x1,1x1, 8-bit).  qt: four 64-entry
 * tables in NATURAL order (as analyze_dqt leaves them, :258-277); comp_tq[c] selects the table of
 * component c (Frame_component::Tq).  gray != 0 is GRAY_MODE: r = g = b = clamp(Y) (:561); coeffs keep
 * the 6-block layout (a decoded file always carries chroma blocks).
 */
int jpezy_dequant_idct(jpezy_ctx* ctx, const int16_t* coeffs, const uint16_t qt[4][64],
                       const uint8_t comp_tq[3], int W, int H, int gray, int n_frames, uint8_t* r,
                       uint8_t* g, uint8_t* b);
int jpezy_dequant_idct_dev(jpezy_ctx* ctx, const int16_t* d_coeffs, const uint16_t qt[4][64],
                           const uint8_t comp_tq[3], size_t plane_stride, int W, int H, int gray,
                           int n_frames, uint8_t* d_r, uint8_t* d_g, uint8_t* d_b, void* stream);

/*
 * DECODE compute stage for ANY baseline layout the reference's decode_mcu handles (decoder/jpezy_decoder.hpp:504-565):
 * 1 or 3 components, sampling factors 1..4 per direction (block placement as decode_mcu does it, :516-524), any table selectors.  coeffs as jpezy_read_jpeg delivers them
 * ([mcu][component blocks, ky outer, kx inner][64], zig-zag).  Blocks go through a fast FP64 inverse transform with a
 * guard band; a block with a sample inside the band is recomputed in the reference's own order (the 64-term sum, one
 * sample per lane), then the replication upsample and make_rgb run one thread per four pixels -- correct for every
 * layout, about a third of the speed of jpezy_dequant_idct, which handles jpezy_encode's own 2x2,1x1,1x1 files in one pass.
 * With ncomp == 1 the missing chroma planes read 0x80 as in the reference (:104-105).
 */
int jpezy_dequant_idct_generic(jpezy_ctx* ctx, const int16_t* coeffs, const uint16_t qt[4][64], int ncomp,
                               const uint8_t comp_h[3], const uint8_t comp_v[3], const uint8_t comp_tq[3], int W, int H,
                               int gray, uint8_t* r, uint8_t* g, uint8_t* b);
/*
 * The same on device memory, asynchronous on `stream` (a hipStream_t): d_coeffs as jpezy_read_jpeg_gpu leaves them,
 * d_r, d_g, d_b planes of W*H bytes.  precision: the SOF0 sample precision (8, or anything else for the reference's
 * 2048 level shift, decoder/jpezy_decoder.hpp:654).  The intermediate samples live in the context: one call in flight per
 * context.
 */
int jpezy_dequant_idct_generic_dev(jpezy_ctx* ctx, const int16_t* d_coeffs, const uint16_t qt[4][64], int ncomp,
                                   const uint8_t comp_h[3], const uint8_t comp_v[3], const uint8_t comp_tq[3], int precision,
                                   int W, int H, int gray, uint8_t* d_r, uint8_t* d_g, uint8_t* d_b, void* stream);

/*
 * The same for n_frames frames of ONE layout, size and set of quantiser tables in one pair of launches (the loop a caller with many
 * small files of a layout would otherwise run): frame f's coefficients at d_coeffs + f * (blocks of a frame) * 64, its planes at
 * d_r/d_g/d_b + f * plane_stride (>= W*H, a multiple of 4).  jpezy_decode_jpeg_batch uses it for the layouts that are not jpezy's own.
 */
int jpezy_dequant_idct_generic_batch_dev(jpezy_ctx* ctx, const int16_t* d_coeffs, const uint16_t qt[4][64], int ncomp,
                                         const uint8_t comp_h[3], const uint8_t comp_v[3], const uint8_t comp_tq[3], int precision,
                                         int W, int H, int gray, int n_frames, size_t plane_stride, uint8_t* d_r, uint8_t* d_g,
                                         uint8_t* d_b, void* stream);

/* Test hook: route EVERY coefficient / sample through the kernels' exact-order fallback (the path a
 * guard-band hit takes).  0 = normal, 1 = reference-order path, 2 = (encode variant 1 only) the FP64 second
 * level, which may still defer to the reference-order path, 3 = (encode variant 1 only) the per-lane evaluator a
 * quad falls back to when it has more guard-band hits than its queue holds.  Exists so the rare branches have
 * parity tests. */
void jpezy_ctx_set_force_exact(jpezy_ctx* ctx, int on);
/* Encode kernel variant: 0 = FP64 butterflies (round-1 kernel), 1 = packed-FP32 first level + FP64 second level +
 * reference-order third level, one quad of four MCUs per wave (the default).  Two independently written kernels with proven error
 * bounds that must agree bit for bit (tests/test_gpu_parity.py runs every case through both).
 * 2 and 3 are LABORATORY variants, present only in libraries built with -DJPEZY_WITH_LAB (python -m jpezy_amd._build --lab;
 * tools/ab/ab_build.py): variant 1's arithmetic in persistent workgroups -- 2: loader waves stream the pixels into an LDS ring by
 * LDS-DMA, 3: sixteen computing waves per CU prefetch their next quad into registers (frames whose rows divide into groups of 16
 * MCUs and 16-byte aligned planes; anything else is handed to variant 1's launch).  Both are parity-green and were measured NOT
 * faster than variant 1 (docs/ROUND5.md), so the shipped library does not contain them: it answers JPEZY_E_UNSUPPORTED and keeps
 * the context's kernel.  Returns JPEZY_OK, JPEZY_E_BADARG (no such variant) or JPEZY_E_UNSUPPORTED.  tests/test_gpu_persistent.py. */
int jpezy_ctx_set_variant(jpezy_ctx* ctx, int variant);
/*
 * Decode tolerance (opt-in; default 0).  BASELINE.json's north_star asks of the decoder "PPM output within +-1 LSB per
 * channel"; the default kernels deliver more (every byte equal to the reference's truncating FP64 arithmetic,
 * decoder/jpezy_decoder.hpp:652-676).  With on = 1, jpezy_dequant_idct[_dev] and jpezy_decode_jpeg[_batch] (jpezy's own
 * 2x2,1x1,1x1 layout) run the LUMA inverse transforms as FP32 butterflies without guard band or exact path: a luma
 * sample equals the reference's or differs from it by one, chroma samples, colour conversion and clamping stay
 * bit-exact, so every output byte is within one of the reference's (a unit on Cb would be 1.77 on B: chroma is never
 * relaxed).  Waves that hold out-of-range coefficients (|c * Q| > 2^15) still take the exact path.  Returns 0 or
 * JPEZY_E_BADARG.
 */
int jpezy_ctx_set_decode_tolerance(jpezy_ctx* ctx, int on);
/* Synchronises the device and returns how many coefficients/samples were resolved through the
 * exact-order fallback on this context since the previous call (the counter is then reset); -1 on error */
long jpezy_ctx_last_fallback_count(jpezy_ctx* ctx);

/*
 * HOST serial tail / head (stay on the CPU per BASELINE.json north_star).
 *
 * jpezy_write_jpeg replaces jpezy_writer::write_header/write_eoi (encoder/jpezy_writer.hpp:20-105) and
 * encoder::encode_huffman (encoder/jpezy_encoder.hpp:174-225) with the Annex-K tables of
 * encoder/huffman_table.hpp.  comment may be NULL/"" (no COM segment).  Returns bytes written (the
 * value encoder::encode returns, :76) or a negative status.
 */
long jpezy_write_jpeg(const int16_t* coeffs, int W, int H, int gray, const char* comment, uint8_t* out,
                      size_t cap);
size_t jpezy_jpeg_bound(int W, int H);   /* a cap that always suffices */
/*
 * The same serial tail for a batch of independent frames, spread over `threads` host threads (0 = all cores):
 * frame f reads coeffs + f*jpezy_coeff_count(W,H,gray) and writes at most `cap` bytes at out + f*cap; sizes[f]
 * receives the bytes written or a negative status.  Returns 0 if every frame succeeded.  Frames are independent
 * (pre_DC and the bit cursor are per file, encoder/jpezy_encoder.hpp:180-181), so this is plain host parallelism.
 */
int jpezy_write_jpeg_batch(const int16_t* coeffs, int W, int H, int gray, int n_frames, const char* comment,
                           uint8_t* out, size_t cap, long* sizes, int threads);

/*
 * The same tail on the GPU ("entropy coding off the critical path"): coefficients already in device
 * memory (the output of jpezy_fdct_quant_dev), bytes identical to jpezy_write_jpeg.  pre_DC (encoder/jpezy_encoder.hpp:
 * 180-181) becomes a read of the previous block's DC and the bit cursor a prefix sum of the blocks' code lengths (every
 * block is coded once; the sum is formed per 256 blocks and across them afterwards); the 0xFF00 stuffing of the reference's
 * bofstream is a second prefix sum.  out/sizes are host memory: frame f writes at
 * most cap bytes at out + f*cap and sizes[f] receives its length or a negative status.  Synchronous.
 */
long jpezy_write_jpeg_gpu(jpezy_ctx* ctx, const int16_t* d_coeffs, int W, int H, int gray, const char* comment,
                          uint8_t* out, size_t cap);
int jpezy_write_jpeg_gpu_batch(jpezy_ctx* ctx, const int16_t* d_coeffs, int W, int H, int gray, int n_frames,
                               const char* comment, uint8_t* out, size_t cap, long* sizes);
/*
 * Device-resident, asynchronous form of the same: everything is enqueued on `stream` (HIP semantics, NULL = default
 * stream; the coefficients must have been produced on it or be complete), nothing is copied to the host, no host
 * synchronisation.  Frame f's complete file (header, entropy-coded segment, EOI) is written at d_out + f*out_stride
 * and d_sizes[f] (device memory) receives its length, JPEZY_E_FORMAT or JPEZY_E_NOSPACE (out_stride too small; nothing
 * is written past it).  Scratch is sized for the worst case of 208 bytes per block and LIVES IN THE CONTEXT (the tile
 * streams, their bit totals, the unstuffed stream): calls of the entropy entry points (this one, jpezy_write_jpeg_gpu[_batch],
 * jpezy_encode_jpeg, jpezy_read_jpeg_gpu, jpezy_decode_jpeg) on one context must not overlap in time -- issue them on one
 * stream or order the streams with events; frames that are to be in flight together need a context each.
 */
int jpezy_write_jpeg_gpu_dev(jpezy_ctx* ctx, const int16_t* d_coeffs, int W, int H, int gray, int n_frames,
                             const char* comment, uint8_t* d_out, size_t out_stride, long long* d_sizes, void* stream);
/*
 * encoder::encode end to end (encoder/jpezy_encoder.hpp:38-77) with both stages on the GPU: host planar r,g,b in,
 * host .jpg bytes out; returns the byte count (the value encoder::encode returns) or a negative status.
 */
long jpezy_encode_jpeg(jpezy_ctx* ctx, const uint8_t* r, const uint8_t* g, const uint8_t* b, int W, int H, int gray,
                       const char* comment, uint8_t* out, size_t cap);

/*
 * MULTI-GPU, one host process (north_star: "host C++ ... partition the input batch and gather ... over xGMI").
 *
 * jpezy_shard_range: the partition rule of every batch entry point and of jpezy_amd/sharding.py -- shard k of n_shards owns the
 * contiguous units [*first, *first + *count) of n_units; counts differ by at most one, the earlier shards take the extra.
 *
 * jpezy_multi_create / jpezy_multi_encode / jpezy_multi_destroy: encoder::encode (encoder/jpezy_encoder.hpp:38-77) for n_frames
 * independent frames of one size, i.e. the loop a caller of the reference runs over encoder objects, spread over the n_dev GPUs
 * devices[0..n_dev) of this node; devices[0] is the ROOT (the calling thread drives it, one more host thread per further device).
 * The HANDLE owns, per entry of devices (a "lane"): a context, an upload stream, download streams and a ring of six slots, each a pinned
 * host buffer + a device buffer per direction, sized for chunk_frames frames (<= 0: about 28 MB of planes per chunk: 4 frames of 1080p,
 * one 4096^2 frame).  Nothing is allocated inside jpezy_multi_encode once the handle has served one call of the same shape, so a caller
 * with a stream of batches creates it once.  A call shards its frames over the lanes (jpezy_shard_range, any n_frames > 0, it may change
 * from call to call) and every lane streams its shard through its ring: feeder threads copy the caller's planes into pinned slots and
 * start the uploads, the lane's thread enqueues FDCT + Huffman stage per chunk, drainer threads bring the results back at their real
 * length -- upload of chunk c + 1, kernels of chunk c and download of chunk c - 1 overlap, PCIe runs in both directions.  The caller's
 * PAGEABLE memory never reaches the DMA engines (uploads straight from it run at an eighth of the link's rate); planes the caller has
 * pinned itself (hipHostMalloc / hipHostRegister; detected with hipPointerGetAttributes) are uploaded as they are, without the copy.
 * r, g, b: host memory, n_frames consecutive W*H planes each.  out says what is wanted and where:
 *   coeffs      NULL, or room for n_frames * jpezy_coeff_count(W, H, gray) int16 (the layout of jpezy_fdct_quant)
 *   jpg         NULL, or n_frames * jpg_stride bytes: frame f's complete file at jpg + f * jpg_stride (MCU loop AND Huffman tail on
 *               the frame's GPU, only the finished file travels, at its real length); jpg_sizes[f] (HOST memory, n_frames entries)
 *               receives its length, JPEZY_E_FORMAT or JPEZY_E_NOSPACE (jpg_stride too small for it)
 *   on_root_device  0: coeffs / jpg are HOST memory -- every device delivers its shard over its own PCIe link;
 *                   1: they are memory of devices[0] (coeffs 16-byte aligned) -- the consumer runs on that GPU: the other devices'
 *                      results are gathered into it chunk by chunk with hipMemcpyPeerAsync (xGMI between the GPUs of a node), the
 *                      root's own chunks are written in place.
 * THE GATHER IS hipMemcpyPeerAsync, NOT RCCL (north_star says "gather coefficient buffers with RCCL over xGMI"): one host process owns
 * every device here, so a peer copy is the point-to-point transfer over the same xGMI link, with no communicator to build, no
 * rendezvous and no second library in the link line of a C++ caller.  The one-process-per-GPU harness (jpezy_amd/sharding.py,
 * bench.py --gpus N) is where RCCL moves the same buffers (send/recv to the consumer rank).
 * An index may appear more than once in devices (several lanes on one GPU, each with its own context and streams): that is how the
 * multi-lane path is exercised on a one-GPU box (tests/test_gpu_multi.py).  jpezy_multi_encode returns JPEZY_OK, or the first failing
 * lane's code (message: which device and why); JPEZY_E_FORMAT when every lane ran but a frame was refused (see jpg_sizes).  Synchronous;
 * one call at a time per handle; the calling thread's current HIP device is what it was on return.
 * jpezy_multi_last_stats: per lane of the last call -- frames, wall time, the GPU time of its kernels (sum over its chunks, HIP
 * events), bytes uploaded and brought back, whether the planes were staged (1) or were the caller's pinned memory (0); returns the
 * number of lanes, fills at most cap entries.  jpezy_multi_chunk_frames: the chunk size the handle settled on.
 * jpezy_multi_feeder_threads / jpezy_multi_set_feeder_threads: the staging threads per lane (a core copies ~11 GB/s into pinned
 * memory, a PCIe link takes ~50: the default is what the host's cores allow when every lane runs its own, between 2 and 4 -- three keep a link busy; 1..6; tuning knob).
 *
 * jpezy_encode_batch_multi: the one-shot form (create, one call, destroy; chunk_frames is clamped to the largest shard so that a
 * single large frame does not reserve a ring for sixteen) -- what jpezy_encode --gpus N calls.
 * Replaces, for a batch: the caller's loop over encoder objects; inside each frame encoder/jpezy_encoder.hpp:55-67 and :174-225.
 */
typedef struct jpezy_multi_out {
    int16_t* coeffs;
    uint8_t* jpg;
    size_t jpg_stride;
    long long* jpg_sizes;
    int on_root_device;
} jpezy_multi_out;
typedef struct jpezy_multi_lane_stats {
    int device;
    int staged;
    long frames;
    double wall_ms, kernel_ms;
    unsigned long long bytes_up, bytes_down;
} jpezy_multi_lane_stats;
typedef struct jpezy_multi jpezy_multi;
void jpezy_shard_range(long n_units, int n_shards, int k, long* first, long* count);
jpezy_multi* jpezy_multi_create(const int* devices, int n_dev, int W, int H, int gray, int chunk_frames);
void jpezy_multi_destroy(jpezy_multi* m);
int jpezy_multi_encode(jpezy_multi* m, const uint8_t* r, const uint8_t* g, const uint8_t* b, int n_frames, const char* comment,
                       const jpezy_multi_out* out);
int jpezy_multi_last_stats(const jpezy_multi* m, jpezy_multi_lane_stats* stats, int cap);
int jpezy_multi_chunk_frames(const jpezy_multi* m);
int jpezy_multi_feeder_threads(const jpezy_multi* m);
int jpezy_multi_set_feeder_threads(jpezy_multi* m, int n);
int jpezy_encode_batch_multi(const int* devices, int n_dev, const uint8_t* r, const uint8_t* g, const uint8_t* b, int W, int H, int gray,
                             int n_frames, int chunk_frames, const char* comment, const jpezy_multi_out* out);

typedef struct jpezy_frame_info {
    int width, height, ncomp, precision;
    int H[3], V[3], Tq[3];
    int hmax, vmax, mcu_cols, mcu_rows, blocks_per_mcu;
    int restart_interval;
    int major_rev, minor_rev, units, hdensity, vdensity;
    int format;               /* 0 undefined, 1 JFIF, 2 JFXX (jpezy.hpp:155-160) */
    char comment[256];
    uint16_t qt[4][64];       /* natural order */
} jpezy_frame_info;

/*
 * jpezy_read_jpeg replaces decoder::analyze_header and the marker parsers (decoder/jpezy_decoder.hpp:
 * 171-502) and decoder::decode_huffman (:583-642).  coeffs (may be NULL: headers only) receives
 * [mcu][block][64] int16 in zig-zag order with DC prediction undone.
 */
int jpezy_read_jpeg(const uint8_t* data, size_t len, jpezy_frame_info* info, int16_t* coeffs,
                    size_t coeff_cap);
/*
 * The same head with the Huffman decoding on the GPU (decode side): the header is parsed on the
 * host, the entropy-coded segment is decoded by the self-synchronising parallel decoder of jpezy_huffdec.hip, the
 * coefficients ([mcu][block][64] zig-zag int16, coeff_cap elements) are left in DEVICE memory, ready for
 * jpezy_dequant_idct_dev / _generic.  A scan with restart intervals (DRI / RSTn, decoder/jpezy_decoder.hpp:152-163) is decoded as
 * that many independent streams when it is regular -- one RSTn behind every interval but the last, nothing else.  Anything irregular
 * (a missing or extra marker, an invalid code, an early end) is decoded by jpezy_read_jpeg's host decoder instead and uploaded, so
 * results and error codes are always those of jpezy_read_jpeg.  d_coeffs may be NULL (header only).  Synchronous.
 */
int jpezy_read_jpeg_gpu(jpezy_ctx* ctx, const uint8_t* data, size_t len, jpezy_frame_info* info, int16_t* d_coeffs,
                        size_t coeff_cap);
/*
 * decoder::decode end to end (decoder/jpezy_decoder.hpp:76-134): .jpg bytes in, planar r,g,b (plane_cap >= width*height
 * bytes each) out; info receives the header fields.  jpezy's own layout (3 components sampled 2x2/1x1/1x1, 8 bit) runs
 * Huffman decoding, dequantisation, IDCT and colour conversion on the device in one fused kernel; every other baseline
 * layout the reference accepts takes the same device Huffman decoder and the generic kernels.  r,g,b NULL: header only.
 */
int jpezy_decode_jpeg(jpezy_ctx* ctx, const uint8_t* data, size_t len, int gray, jpezy_frame_info* info, uint8_t* r,
                      uint8_t* g, uint8_t* b, size_t plane_cap);
/*
 * jpezy_decode_jpeg for n files at once (n decoder objects of the reference, decoder/jpezy_decoder.hpp:39-134, one per
 * file).  Files without restart intervals are grouped by size, layout (any the reference's decode_mcu handles: 1 or 3 components,
 * sampling factors 1..4) and quantiser tables and decoded TOGETHER: one sequence of Huffman-decoder launches per slice of 16 scans
 * (every file with its own code tables), one inverse-transform launch per slice (the fused kernel for jpezy's own 2x2/1x1/1x1 layout,
 * the generic kernels for the others), the planes of a slice copied out while the next slice is decoded -- a single file keeps 80
 * waves busy for a chain of launches that is pure latency, a slice fills the chip for the same chain; a batch is bounded by PCIe.
 * Everything else (irregular or non-converging streams, single files)
 * is decoded file by file, up to 8 in flight on child contexts, with the host decoder as the last word, as before.  data[i] / len[i]: file i;
 * r[i], g[i], b[i]: its planes (plane_cap[i] >= width*height bytes each; sizes come from a header-only jpezy_decode_jpeg or
 * jpezy_read_jpeg call); info[i] and status[i] (JPEZY_OK or that file's negative error code) are written per file.
 * Returns JPEZY_OK when every file decoded, else the first failing file's code (message: which file and why); the other
 * files' outputs are complete either way.
 */
int jpezy_decode_jpeg_batch(jpezy_ctx* ctx, int n, const uint8_t* const* data, const size_t* len, int gray,
                            jpezy_frame_info* info, uint8_t* const* r, uint8_t* const* g, uint8_t* const* b,
                            const size_t* plane_cap, int* status);
/* Synchronisation passes the last jpezy_read_jpeg_gpu call needed; 0 = the host decoder was used (test/diagnostic hook). */
int jpezy_ctx_last_huffdec_passes(jpezy_ctx* ctx);
/* Files of the last jpezy_decode_jpeg_batch call that were decoded by the batch form of the kernels (test/diagnostic hook). */
int jpezy_ctx_last_batch_fast_count(jpezy_ctx* ctx);
/* Scans shorter than n bytes are decoded on the host (default 32 KiB: the GPU decoder has ~0.32 ms of fixed cost, which the host decoder spends on ~30 KiB of scan); 0
 * sends every scan to the GPU decoder (tests). */
void jpezy_ctx_set_huffdec_min_bytes(jpezy_ctx* ctx, size_t n);


#ifdef __cplusplus
}
#endif
#endif
