/*
 * jpezy_oracle.h -- CPU restatement of falgon/jpezy's baseline-JPEG encode/decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under jpezy_amd/ (the product) includes, links or calls this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, as the checker.
 *
 * PARITY UNPINNED: the reference holds no golden vectors, cannot be built here (SrookCppLibraries and
 * Boost are absent; its cos table, 1/sqrt(2) and bit-stream writer live in the un-vendored, un-pinned
 * SrookCppLibraries), so this oracle is pinned only by (i) the reference's source text it restates line
 * by line, (ii) Annex-K tables re-derived from libjpeg, (iii) libjpeg decoding the files it writes.
 * The frozen choices (include/jpezy_constants.h): correctly rounded binary64 cosines, 1.0/sqrt(2.0) in
 * binary64, no FMA contraction, MSB-first bit packing with 0xFF00 stuffing and ZERO pad bits (JPEZY_PAD_BIT).
 * One more frozen reading: the DECODER writes `1.0 / srook::sqrt(2)` with an INT literal (ref
 * decoder/jpezy_decoder.hpp:655) where the encoder writes `srook::sqrt(2.0)` (ref encoder/jpezy_encoder.hpp:149).
 * Both are taken as the binary64 sqrt of 2.0 (srook::sqrt is a constexpr template; an overload returning an
 * integer would make the decoder's DC gain 1 instead of 1/sqrt 2 and every decoded image far too bright, which
 * the README's round trip rules out).  Whether the int overload rounds differently in the last place is
 * unknowable here; the alternative-constants build (tools/gen/gen_constants.py --variant alt1, make CONSTANTS=...)
 * exists so that a different value is a one-header change for oracle and product alike.
 *
 * All `ref` citations are into /root/reference/src/.
 */
#ifndef JPEZY_ORACLE_H
#define JPEZY_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- a1: colour conversion (ref encoder/jpezy_encoder.hpp:244-256) ---- */
int jo_rgb_y(uint8_t r, uint8_t g, uint8_t b);
int jo_rgb_cb(uint8_t r, uint8_t g, uint8_t b);
int jo_rgb_cr(uint8_t r, uint8_t g, uint8_t b);

/* ---- a4/a6: one block, natural order in, natural order out (ref jpezy_encoder.hpp:146-172) ---- */
void jo_fdct_block(const int pic[64], int out[64]);          /* DCT(): int(sum*cu*cv/4)          */
void jo_quantize_block(int blk[64], int cs);                 /* quantization(cs): blk[i] /= qt[i] */
/* ---- a12/a13 (ref decoder/jpezy_decoder.hpp:645-670) ---- */
void jo_idct_block(const int dct[64], int precision, int out[64]);

/* MCU grid (ref jpezy_encoder.hpp:55-56) */
int jo_mcu_cols(int W);
int jo_mcu_rows(int H);

/*
 * a2..a8: planar r,g,b (W*H each, row stride W) -> quantised coefficients.
 * Layout: [mcu_y][mcu_x][blk][64] int16, zig-zag order inside a block (out[n] = q[ZZ[n]]),
 * blk = Y0,Y1,Y2,Y3,Cb,Cr (colour, 6 blocks) or Y0..Y3 (gray: chroma blocks are identically zero in
 * the reference, jpezy_encoder.hpp:61-64, and are not materialised).
 * Only MCU rows [mcu_y0, mcu_y1) are computed (others untouched) so callers can band the work.
 */
void jo_encode_coeffs_rows(const uint8_t* r, const uint8_t* g, const uint8_t* b, int W, int H, int gray,
                           int mcu_y0, int mcu_y1, int16_t* coeffs);
void jo_encode_coeffs(const uint8_t* r, const uint8_t* g, const uint8_t* b, int W, int H, int gray,
                      int16_t* coeffs);

/*
 * a9 + a10: header + Huffman entropy coding + EOI (ref jpezy_encoder.hpp:38-77,174-242; jpezy_writer.hpp;
 * huffman_table.hpp).  `coeffs` as produced above.  Returns bytes written, or -1 if cap is too small /
 * a table index overflows (the reference throws std::runtime_error there).
 */
long jo_write_jpeg(const int16_t* coeffs, int W, int H, int gray, const char* comment, uint8_t* out,
                   size_t cap);
/* whole encoder: r,g,b -> .jpg bytes (comment as the CLI sets it: encode_io.hpp:149,181) */
long jo_encode_jpeg(const uint8_t* r, const uint8_t* g, const uint8_t* b, int W, int H, int gray,
                    uint8_t* out, size_t cap);

/* ---- decoder ---- */
typedef struct jo_frame_info {
    int width, height, ncomp, precision;
    int H[3], V[3], Tq[3];      /* per-component sampling factors / quant table selector            */
    int hmax, vmax;
    int mcu_cols, mcu_rows;     /* h_unit, v_unit (ref jpezy_decoder.hpp:94-98)                     */
    int blocks_per_mcu;         /* sum H*V                                                          */
    int restart_interval;
    int major_rev, minor_rev, units, hdensity, vdensity;
    int jfif;                   /* 1 JFIF, 2 JFXX, 0 undefined                                      */
    char comment[256];
    uint16_t qt[4][64];         /* natural order (de-zig-zagged at parse, ref :258-277)             */
} jo_frame_info;

/*
 * a11 + marker parser (ref jpezy_decoder.hpp:171-502, 583-642): parse headers, Huffman-decode the scan.
 * coeffs: [mcu][blk][64] int16 in ZIG-ZAG order (position k holds the k-th decoded coefficient; the
 * reference stores it at dct[ZZ[k]]), DC already un-differenced.  Pass coeffs=NULL to parse headers only.
 * Returns 0 ok, negative on error (the reference returns an empty optional).
 */
int jo_read_jpeg(const uint8_t* data, size_t len, jo_frame_info* info, int16_t* coeffs, size_t coeff_cap);

/*
 * a12..a15: dequantise + IDCT + nearest upsample + YCbCr->RGB + clamp (ref :504-578, 645-676).
 * r,g,b: W*H each, row stride W (the reference's vectors are larger; only the first W*H entries are
 * the image, decode_io.hpp:45-47).
 */
void jo_decode_planes_rows(const int16_t* coeffs, const jo_frame_info* info, int gray, int mcu_y0, int mcu_y1,
                           uint8_t* r, uint8_t* g, uint8_t* b);
void jo_decode_planes(const int16_t* coeffs, const jo_frame_info* info, int gray,
                      uint8_t* r, uint8_t* g, uint8_t* b);
int jo_decode_jpeg(const uint8_t* data, size_t len, int gray, jo_frame_info* info,
                   uint8_t* r, uint8_t* g, uint8_t* b, size_t plane_cap);

/* ---- PPM P3 (ref encoder/encode_io.hpp:45-101, decoder/decode_io.hpp:36-54) ---- */
/* returns 0 ok; *W,*H set; planes malloc'ed by the callee (free with jo_free). */
int jo_read_ppm_p3(const char* path, int* W, int* H, uint8_t** r, uint8_t** g, uint8_t** b);
long jo_format_ppm_p3(int W, int H, const uint8_t* r, const uint8_t* g, const uint8_t* b, char* out, size_t cap);
void jo_free(void* p);

/* constants, for tests */
const int* jo_zz(void);
const int* jo_qt(int cs);
const double* jo_cos_table(void);
double jo_inv_sqrt2(void);

#ifdef __cplusplus
}
#endif
#endif
