/*
 * jpezy_oracle.c -- CPU restatement of falgon/jpezy (see jpezy_oracle.h: TEST INFRASTRUCTURE ONLY,
 * PARITY UNPINNED).  Plain C99, double arithmetic in exactly the reference's operation order.
 * Build: gcc -O2 -ffp-contract=off (no -ffast-math, no -march=native) so results are IEEE-754 binary64
 * on every host.  `ref` = /root/reference/src/.
 */
#include "jpezy_oracle.h"
#include "../include/jpezy_constants.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ */
/* constants (ref jpezy.hpp:36-45, 131-152; jpezy_encoder.hpp:149, 271)                              */
/* ------------------------------------------------------------------------------------------------ */
static const int ZZ[64] = JPEZY_ZZ_INIT;
static const int QT_LUMA[64] = JPEZY_QT_LUMA_INIT;
static const int QT_CHROMA[64] = JPEZY_QT_CHROMA_INIT;
static const double COS_TABLE[64] = JPEZY_COS_INIT;
static const double INV_SQRT2 = JPEZY_INV_SQRT2;

const int* jo_zz(void) { return ZZ; }
const int* jo_qt(int cs) { return cs ? QT_CHROMA : QT_LUMA; }
const double* jo_cos_table(void) { return COS_TABLE; }
double jo_inv_sqrt2(void) { return INV_SQRT2; }

/* ------------------------------------------------------------------------------------------------ */
/* a1: RGB::Y / Cb / Cr  (ref encoder/jpezy_encoder.hpp:244-256) -- int() truncates toward zero      */
/* ------------------------------------------------------------------------------------------------ */
int jo_rgb_y(uint8_t r, uint8_t g, uint8_t b)
{
    return (int)((0.2990 * (int)r) + (0.5870 * (int)g) + (0.1140 * (int)b) - 128);
}
int jo_rgb_cb(uint8_t r, uint8_t g, uint8_t b)
{
    return (int)(-(0.1687 * (int)r) - (0.3313 * (int)g) + (0.5000 * (int)b));
}
int jo_rgb_cr(uint8_t r, uint8_t g, uint8_t b)
{
    return (int)((0.5000 * (int)r) - (0.4187 * (int)g) - (0.0813 * (int)b));
}

/* ------------------------------------------------------------------------------------------------ */
/* a4: encoder::DCT  (ref jpezy_encoder.hpp:146-166)                                                 */
/* ------------------------------------------------------------------------------------------------ */
void jo_fdct_block(const int pic[64], int out[64])
{
    const double dis_sqrt = INV_SQRT2;
    for (int i = 0; i < 8; ++i) {
        const double cv = i ? 1.0 : dis_sqrt;
        for (int j = 0; j < 8; ++j) {
            const double cu = j ? 1.0 : dis_sqrt;
            double sum = 0;
            for (int y = 0; y < 8; ++y) {
                for (int x = 0; x < 8; ++x) {
                    sum += pic[y * 8 + x] * COS_TABLE[j * 8 + x] * COS_TABLE[i * 8 + y];
                }
            }
            out[i * 8 + j] = (int)(sum * cu * cv / 4);
        }
    }
}

/* a6: encoder::quantization (ref jpezy_encoder.hpp:168-172): C++ int division, natural order */
void jo_quantize_block(int blk[64], int cs)
{
    const int* qt = cs ? QT_CHROMA : QT_LUMA;
    for (int i = 0; i < 64; ++i) blk[i] /= qt[i];
}

int jo_mcu_cols(int W) { return (W / 16) + ((W % 16) ? 1 : 0); }  /* ref :56 */
int jo_mcu_rows(int H) { return (H / 16) + ((H % 16) ? 1 : 0); }  /* ref :55 */

/* ------------------------------------------------------------------------------------------------ */
/* a2: encoder::make_YCC (ref jpezy_encoder.hpp:90-144)                                              */
/* ------------------------------------------------------------------------------------------------ */
static void make_ycc(const uint8_t* r, const uint8_t* g, const uint8_t* b, int W, int H, int ux, int uy,
                     int Y_block[4][64], int Cb_block[64], int Cr_block[64])
{
    int Crblock[4][64], Cbblock[4][64];
    for (int i = 0; i < 4; ++i) {
        int* yp = Y_block[i];
        int* cbp = Cbblock[i];
        int* crp = Crblock[i];
        const long sy0 = (long)uy * 16 + ((i > 1) ? 8 : 0);
        for (long sy = sy0; sy < sy0 + 8; ++sy) {
            const long ii = sy < H ? sy : H - 1;                     /* edge replication, ref :101 */
            const long sx0 = (long)ux * 16 + ((i & 1) ? 8 : 0);
            for (long sx = sx0; sx < sx0 + 8; ++sx) {
                const long jj = sx < W ? sx : W - 1;                 /* ref :104 */
                const long index = ii * W + jj;
                const uint8_t rv = r[index], gv = g[index], bv = b[index];
                *yp++ = jo_rgb_y(rv, gv, bv);
                *cbp++ = jo_rgb_cb(rv, gv, bv);
                *crp++ = jo_rgb_cr(rv, gv, bv);
            }
        }
    }
    /* 2x2 decimation: keep the top-left sample of every 2x2 (ref :116-143) */
    for (int i = 0; i < 4; ++i) {
        int n = (i == 0) ? 0 : (i == 1) ? 4 : (i == 2) ? 32 : 36;
        for (int y = 0; y < 8; y += 2) {
            for (int x = 0; x < 8; x += 2) {
                const int index = y * 8 + x;
                Cr_block[n] = Crblock[i][index];
                Cb_block[n] = Cbblock[i][index];
                ++n;
            }
            n += 4;
        }
    }
}

static void emit_block(const int pic[64], int cs, int16_t* out)
{
    int dct[64];
    jo_fdct_block(pic, dct);
    jo_quantize_block(dct, cs);
    for (int n = 0; n < 64; ++n) out[n] = (int16_t)dct[ZZ[n]];   /* a7: consumer reads dct[ZZ[n]] */
}

void jo_encode_coeffs_rows(const uint8_t* r, const uint8_t* g, const uint8_t* b, int W, int H, int gray,
                           int mcu_y0, int mcu_y1, int16_t* coeffs)
{
    const int HUnits = jo_mcu_cols(W);
    const int bpm = gray ? 4 : 6;
    int Y_block[4][64], Cb_block[64], Cr_block[64];
    for (int y = mcu_y0; y < mcu_y1; ++y) {
        for (int x = 0; x < HUnits; ++x) {                          /* ref :58-67 */
            int16_t* out = coeffs + ((size_t)y * HUnits + x) * bpm * 64;
            make_ycc(r, g, b, W, H, x, y, Y_block, Cb_block, Cr_block);
            for (int i = 0; i < 4; ++i) emit_block(Y_block[i], 0, out + i * 64);   /* make_MCU :229-233 */
            if (!gray) {
                emit_block(Cb_block, 1, out + 4 * 64);             /* :235-237 */
                emit_block(Cr_block, 2, out + 5 * 64);             /* :239-241 */
            }
        }
    }
}

void jo_encode_coeffs(const uint8_t* r, const uint8_t* g, const uint8_t* b, int W, int H, int gray,
                      int16_t* coeffs)
{
    jo_encode_coeffs_rows(r, g, b, W, H, gray, 0, jo_mcu_rows(H), coeffs);
}

/* ------------------------------------------------------------------------------------------------ */
/* Huffman tables.  The reference spells K.3-K.6 out as (size, code) arrays indexed run*10+s+(run==15)  */
/* (huffman_table.hpp:26-195); here the same arrays are rebuilt from the Annex-K BITS/HUFFVAL lists by  */
/* the canonical-code procedure (ISO/IEC 10918-1 Annex C), in that same index layout.                  */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    int size_tb[162];
    int code_tb[162];
    int n;
} enc_table;

static const uint8_t DC_L_BITS[16] = JPEZY_DC_LUMA_BITS_INIT, DC_L_VALS[] = JPEZY_DC_LUMA_VALS_INIT;
static const uint8_t DC_C_BITS[16] = JPEZY_DC_CHROMA_BITS_INIT, DC_C_VALS[] = JPEZY_DC_CHROMA_VALS_INIT;
static const uint8_t AC_L_BITS[16] = JPEZY_AC_LUMA_BITS_INIT, AC_L_VALS[] = JPEZY_AC_LUMA_VALS_INIT;
static const uint8_t AC_C_BITS[16] = JPEZY_AC_CHROMA_BITS_INIT, AC_C_VALS[] = JPEZY_AC_CHROMA_VALS_INIT;

static int ac_symbol_index(int sym)   /* run/size symbol -> index in the reference's table layout */
{
    const int run = sym >> 4, s = sym & 15;
    if (sym == 0x00) return 0;          /* EOB, YEOBidx / CEOBidx = 0 (huffman_table.hpp:122,194) */
    if (sym == 0xF0) return 151;        /* ZRL, YZRLidx / CZRLidx = 151 (:123,195)                */
    return run * 10 + s + (run == 15);
}

static void build_enc_table(const uint8_t bits[16], const uint8_t* vals, int nval, int is_ac, enc_table* t)
{
    memset(t, 0, sizeof *t);
    t->n = is_ac ? 162 : 12;
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
        for (int c = 0; c < bits[len - 1]; ++c, ++k) {
            const int sym = vals[k];
            const int idx = is_ac ? ac_symbol_index(sym) : sym;
            t->size_tb[idx] = len;
            t->code_tb[idx] = code++;
        }
        code <<= 1;
    }
    (void)nval;
}

static enc_table T_YDC, T_CDC, T_YAC, T_CAC;
static int tables_ready = 0;
static void ensure_tables(void)
{
    if (tables_ready) return;
    build_enc_table(DC_L_BITS, DC_L_VALS, JPEZY_DC_LUMA_NVAL, 0, &T_YDC);
    build_enc_table(DC_C_BITS, DC_C_VALS, JPEZY_DC_CHROMA_NVAL, 0, &T_CDC);
    build_enc_table(AC_L_BITS, AC_L_VALS, JPEZY_AC_LUMA_NVAL, 1, &T_YAC);
    build_enc_table(AC_C_BITS, AC_C_VALS, JPEZY_AC_CHROMA_NVAL, 1, &T_CAC);
    tables_ready = 1;
}

/* ------------------------------------------------------------------------------------------------ */
/* bit-stream writer: stands in for srook::io::jpeg::bofstream (NOT in the reference tree).  */
/* Frozen semantics: zero-initialised buffer; Bits(n) << v appends the low n bits of v MSB-first; a     */
/* completed entropy byte 0xFF is followed by a stuffed 0x00; Byte/Word/Bytes writes are raw and, when   */
/* issued mid-byte, first advance to the next byte (pad bits stay 0); writes past the buffer are dropped */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    uint8_t* buf;
    size_t cap, pos;
    int bitpos;        /* next bit to fill in buf[pos], 7 = byte empty */
    int overflow;
} bitw;

static void bw_inc(bitw* w)
{
    if (++w->pos >= w->cap) w->overflow = 1;
}
static void bw_byte(bitw* w, unsigned v)
{
    if (w->overflow) return;
    if (w->bitpos != 7) {
#if JPEZY_PAD_BIT   /* alternative frozen choice (include/jpezy_constants.h): pad with ones, stuff a padded 0xFF */
        w->buf[w->pos] |= (uint8_t)((1u << (w->bitpos + 1)) - 1u);
        const int full = w->buf[w->pos] == 0xFF;
        bw_inc(w);
        w->bitpos = 7;
        if (w->overflow) return;
        if (full) { w->buf[w->pos] = 0x00; bw_inc(w); if (w->overflow) return; }
#else
        bw_inc(w); w->bitpos = 7; if (w->overflow) return;
#endif
    }
    w->buf[w->pos] = (uint8_t)v;
    bw_inc(w);
}
static void bw_word(bitw* w, unsigned v) { bw_byte(w, (v >> 8) & 0xFF); bw_byte(w, v & 0xFF); }
static void bw_bytes(bitw* w, const void* p, size_t n)
{
    const uint8_t* s = (const uint8_t*)p;
    for (size_t i = 0; i < n; ++i) bw_byte(w, s[i]);
}
static void bw_bits(bitw* w, int nbits, int v)
{
    for (int i = nbits - 1; i >= 0; --i) {
        if (w->overflow) return;
        if ((v >> i) & 1) w->buf[w->pos] |= (uint8_t)(1u << w->bitpos);
        if (--w->bitpos < 0) {
            const int full = w->buf[w->pos] == 0xFF;
            w->bitpos = 7;
            bw_inc(w);
            if (full && !w->overflow) { w->buf[w->pos] = 0x00; bw_inc(w); }
        }
    }
}
static size_t bw_size(const bitw* w) { return w->pos + (w->bitpos != 7 ? 1 : 0); }

/* ------------------------------------------------------------------------------------------------ */
/* a10: jpezy_writer::write_header (ref jpezy_writer.hpp:20-94)                                       */
/* ------------------------------------------------------------------------------------------------ */
static void put_dht(bitw* w, int tc_th, const uint8_t bits[16], const uint8_t* vals, int nval)
{
    bw_byte(w, 0xFF); bw_byte(w, 0xC4);
    bw_word(w, (unsigned)(2 + 1 + 16 + nval));
    bw_byte(w, (unsigned)tc_th);
    bw_bytes(w, bits, 16);
    bw_bytes(w, vals, (size_t)nval);
}

static void write_header(bitw* w, int W, int H, const char* comment)
{
    const int dimension = 3, precision = 8;
    bw_byte(w, 0xFF); bw_byte(w, 0xD8);                               /* SOI */
    bw_byte(w, 0xFF); bw_byte(w, 0xE0);                               /* APP0 / JFIF */
    bw_word(w, 16);
    bw_bytes(w, "JFIF", 5);
    bw_word(w, 0x0102);
    bw_byte(w, 1);                                                    /* Units::dots_inch */
    bw_word(w, 96); bw_word(w, 96);
    bw_byte(w, 0); bw_byte(w, 0);
    if (comment && comment[0]) {                                      /* COM, ref :40-44 */
        const size_t n = strlen(comment);
        bw_byte(w, 0xFF); bw_byte(w, 0xFE);
        bw_word(w, (unsigned)(n + 3));
        bw_bytes(w, comment, n + 1);
    }
    bw_byte(w, 0xFF); bw_byte(w, 0xDB); bw_word(w, 67); bw_byte(w, 0);        /* DQT 0, zig-zag order */
    for (int i = 0; i < 64; ++i) bw_byte(w, (unsigned)QT_LUMA[ZZ[i]]);
    bw_byte(w, 0xFF); bw_byte(w, 0xDB); bw_word(w, 67); bw_byte(w, 1);        /* DQT 1 */
    for (int i = 0; i < 64; ++i) bw_byte(w, (unsigned)QT_CHROMA[ZZ[i]]);
    put_dht(w, 0x00, DC_L_BITS, DC_L_VALS, JPEZY_DC_LUMA_NVAL);               /* YDcDht */
    put_dht(w, 0x01, DC_C_BITS, DC_C_VALS, JPEZY_DC_CHROMA_NVAL);             /* CDcDht */
    put_dht(w, 0x10, AC_L_BITS, AC_L_VALS, JPEZY_AC_LUMA_NVAL);               /* YAcDht */
    put_dht(w, 0x11, AC_C_BITS, AC_C_VALS, JPEZY_AC_CHROMA_NVAL);             /* CAcDht */
    bw_byte(w, 0xFF); bw_byte(w, 0xC0);                               /* SOF0, ref :67-81 */
    bw_word(w, (unsigned)(3 * dimension + 8));
    bw_byte(w, (unsigned)precision);
    bw_word(w, (unsigned)H);
    bw_word(w, (unsigned)W);
    bw_byte(w, (unsigned)dimension);
    bw_byte(w, 0); bw_byte(w, 0x22); bw_byte(w, 0);
    for (unsigned i = 1; i < 3; ++i) { bw_byte(w, i); bw_byte(w, 0x11); bw_byte(w, 1); }
    bw_byte(w, 0xFF); bw_byte(w, 0xDA);                               /* SOS, ref :84-93 */
    bw_word(w, (unsigned)(2 * dimension + 6));
    bw_byte(w, (unsigned)dimension);
    for (int i = 0; i < dimension; ++i) { bw_byte(w, (unsigned)i); bw_byte(w, i == 0 ? 0 : 0x11); }
    bw_byte(w, 0); bw_byte(w, 63); bw_byte(w, 0);
}

/* ------------------------------------------------------------------------------------------------ */
/* a9: encoder::encode_huffman (ref jpezy_encoder.hpp:174-225).  blk is in zig-zag order, so          */
/* dct_data[ZZ[n]] of the reference is blk[n] here.                                                   */
/* ------------------------------------------------------------------------------------------------ */
static int encode_huffman(const int16_t* blk, int cs, int pre_DC[3], bitw* ofs, const enc_table* dcT,
                          const enc_table* acT, int eob_idx, int zrl_idx)
{
    const int diff = blk[0] - pre_DC[cs];
    pre_DC[cs] = blk[0];

    int di = 0;
    for (int abs_diff = abs(diff); abs_diff > 0; abs_diff >>= 1, ++di)
        ;
    if (di >= dcT->n) return -1;
    bw_bits(ofs, dcT->size_tb[di], dcT->code_tb[di]);
    if (di) bw_bits(ofs, di, diff < 0 ? diff - 1 : diff);

    int run = 0;
    for (int n = 1; n < 64; ++n) {
        int abs_coefficient = abs(blk[n]);
        if (abs_coefficient) {
            while (run > 15) {
                bw_bits(ofs, acT->size_tb[zrl_idx], acT->code_tb[zrl_idx]);
                run -= 16;
            }
            int s = 0;
            for (; abs_coefficient > 0; abs_coefficient >>= 1, ++s)
                ;
            const int a_di = run * 10 + s + (run == 15);
            if (a_di >= acT->n) return -1;
            bw_bits(ofs, acT->size_tb[a_di], acT->code_tb[a_di]);
            int v = blk[n];
            if (v < 0) --v;
            bw_bits(ofs, s, v);
            run = 0;
        } else {
            if (n == 63)
                bw_bits(ofs, acT->size_tb[eob_idx], acT->code_tb[eob_idx]);
            else
                ++run;
        }
    }
    return 0;
}

long jo_write_jpeg(const int16_t* coeffs, int W, int H, int gray, const char* comment, uint8_t* out,
                   size_t cap)
{
    ensure_tables();
    memset(out, 0, cap);
    bitw w = { out, cap, 0, 7, 0 };
    write_header(&w, W, H, comment);

    static const int16_t zero_blk[64] = { 0 };
    const int VUnits = jo_mcu_rows(H), HUnits = jo_mcu_cols(W);
    const int bpm = gray ? 4 : 6;
    int pre_DC[3] = { 0, 0, 0 };
    for (long m = 0; m < (long)VUnits * HUnits; ++m) {               /* make_MCU, ref :227-242 */
        const int16_t* mcu = coeffs + (size_t)m * bpm * 64;
        for (int i = 0; i < 4; ++i)
            if (encode_huffman(mcu + i * 64, 0, pre_DC, &w, &T_YDC, &T_YAC, 0, 151)) return -1;
        const int16_t* cb = gray ? zero_blk : mcu + 4 * 64;
        const int16_t* cr = gray ? zero_blk : mcu + 5 * 64;
        if (encode_huffman(cb, 1, pre_DC, &w, &T_CDC, &T_CAC, 0, 151)) return -1;
        if (encode_huffman(cr, 2, pre_DC, &w, &T_CDC, &T_CAC, 0, 151)) return -1;
    }
    bw_byte(&w, 0xFF); bw_byte(&w, 0xD9);                             /* write_eoi, ref jpezy_writer.hpp:101-105 */
    if (w.overflow) return -1;
    return (long)bw_size(&w);
}

long jo_encode_jpeg(const uint8_t* r, const uint8_t* g, const uint8_t* b, int W, int H, int gray,
                    uint8_t* out, size_t cap)
{
    const size_t nmcu = (size_t)jo_mcu_cols(W) * jo_mcu_rows(H);
    int16_t* coeffs = (int16_t*)malloc(nmcu * (gray ? 4 : 6) * 64 * sizeof(int16_t));
    if (!coeffs) return -1;
    jo_encode_coeffs(r, g, b, W, H, gray, coeffs);
    /* the CLI's comment strings: encode_io.hpp:149 (colour), :181 (gray) */
    const long n = jo_write_jpeg(coeffs, W, H, gray, gray ? "Encoded by JPEZY" : "Encoded by jpezy", out, cap);
    free(coeffs);
    return n;
}

/* ------------------------------------------------------------------------------------------------ */
/* decoder side: bit reader standing in for srook::io::jpeg::bifstream, marker parser, Huffman head    */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    const uint8_t* p;
    size_t len, pos;
    int bitpos;        /* bits left in cur, 0 = need a new byte */
    unsigned cur;
} bitr;

static int br_byte(bitr* s)                      /* raw byte, discards pending bits */
{
    s->bitpos = 0;
    if (s->pos >= s->len) return -1;
    return s->p[s->pos++];
}
static int br_word(bitr* s)
{
    const int a = br_byte(s), b = br_byte(s);
    if (a < 0 || b < 0) return -1;
    return (a << 8) | b;
}
static void br_skip(bitr* s, long n)
{
    s->bitpos = 0;
    if (n > 0) s->pos += (size_t)n;
    if (s->pos > s->len) s->pos = s->len;
}
static int br_bit(bitr* s)                       /* entropy bit: 0xFF00 -> 0xFF */
{
    if (s->bitpos == 0) {
        if (s->pos >= s->len) return -1;
        s->cur = s->p[s->pos++];
        if (s->cur == 0xFF && s->pos < s->len && s->p[s->pos] == 0x00) s->pos++;
        s->bitpos = 8;
    }
    --s->bitpos;
    return (int)((s->cur >> s->bitpos) & 1u);
}
static int br_bits(bitr* s, int n, int* out)
{
    int v = 0;
    for (int i = 0; i < n; ++i) {
        const int b = br_bit(s);
        if (b < 0) return -1;
        v = (v << 1) | b;
    }
    *out = v;
    return 0;
}

typedef struct {
    int n;
    int sizeTP[256], codeTP[256], valueTP[256];
} dec_table;

typedef struct {
    bitr in;
    jo_frame_info* info;
    dec_table ht[2][4];
    int Td[3], Ta[3];
    int decodable;     /* property::AnalyzedResult bits: 1 htable, 2 qtable, 4 jfif, 8 comment, 16 start */
    int enable;
} dstate;

enum { M_ERROR = 0xFF };

/* ref jpezy_decoder.hpp:486-502 */
static int get_marker(dstate* d)
{
    for (;;) {
        int c = br_byte(&d->in);
        if (c < 0) return -1;
        if (c == 0xFF) {
            c = br_byte(&d->in);
            if (c < 0) return -1;
            if (c) {
                if (c > 0x02 && c < 0xC0) return M_ERROR;
                return c;
            }
        }
    }
}

/* ref :190-256 */
static int analyze_dht(dstate* d, long size)
{
    const size_t end_add = d->in.pos + (size_t)size;
    do {
        const int uc = br_byte(&d->in);
        if (uc < 0) return -1;
        const int tc = uc >> 4, th = uc & 0x0f;
        if (tc > 1 || th > 3) return -1;
        dec_table* t = &d->ht[tc][th];
        int cc[16], n = 0;
        for (int i = 0; i < 16; ++i) {
            cc[i] = br_byte(&d->in);
            if (cc[i] < 0) return -1;
            n += cc[i];
        }
        if (n > 256) return -1;
        t->n = n;
        for (int i = 1, k = 0; i <= 16; ++i)
            for (int j = 1; j <= cc[i - 1]; ++j, ++k) t->sizeTP[k] = i;
        if (n > 0) {
            int k = 0, code = 0, si = t->sizeTP[0];
            for (;;) {
                for (; k < n && t->sizeTP[k] == si; ++k, ++code) t->codeTP[k] = code;
                if (k >= n) break;
                do {
                    code <<= 1;
                    ++si;
                } while (t->sizeTP[k] != si);
            }
        }
        for (int k = 0; k < n; ++k) {
            const int v = br_byte(&d->in);
            if (v < 0) return -1;
            t->valueTP[k] = v;
        }
    } while (d->in.pos < end_add);
    return 0;
}

/* ref :258-277 */
static int analyze_dqt(dstate* d, long size)
{
    const size_t end_add = d->in.pos + (size_t)size;
    do {
        const int c = br_byte(&d->in);
        if (c < 0) return -1;
        uint16_t* q = d->info->qt[c & 0x3];
        if (!(c >> 4)) {
            for (int i = 0; i < 64; ++i) {
                const int t = br_byte(&d->in);
                if (t < 0) return -1;
                q[ZZ[i]] = (uint16_t)t;
            }
        } else {
            for (int i = 0; i < 64; ++i) {
                const int t = br_word(&d->in);
                if (t < 0) return -1;
                q[ZZ[i]] = (uint16_t)t;
            }
        }
    } while (d->in.pos < end_add);
    return 0;
}

/* ref :279-305 */
static int analyze_frame(dstate* d)
{
    jo_frame_info* f = d->info;
    f->precision = br_byte(&d->in);
    f->height = br_word(&d->in);
    f->width = br_word(&d->in);
    f->ncomp = br_byte(&d->in);
    if (f->ncomp != 3 && f->ncomp != 1) return -1;
    for (int i = 0; i < f->ncomp; ++i) {
        (void)br_byte(&d->in);                        /* C */
        const int c = br_byte(&d->in);
        if (c < 0) return -1;
        f->H[i] = c >> 4;
        if (f->H[i] > f->hmax) f->hmax = f->H[i];
        f->V[i] = c & 0xf;
        if (f->V[i] > f->vmax) f->vmax = f->V[i];
        f->Tq[i] = br_byte(&d->in);
    }
    return 0;
}

/* ref :307-334 */
static int analyze_scan(dstate* d)
{
    const int ns = br_byte(&d->in);
    if (ns < 0 || ns > 3) return -1;
    for (int i = 0; i < ns; ++i) {
        (void)br_byte(&d->in);                        /* Cs */
        const int c = br_byte(&d->in);
        if (c < 0) return -1;
        d->Td[i] = c >> 4;
        if (d->Td[i] > 2) return -1;
        d->Ta[i] = c & 0xf;
        if (d->Ta[i] > 2) return -1;
    }
    (void)br_byte(&d->in); (void)br_byte(&d->in); (void)br_byte(&d->in);   /* Ss, Se, Ah/Al: unused */
    return 0;
}

/* ref :360-484; returns AnalyzedResult bits, or -1 for "throw" */
static int analyze_marker(dstate* d)
{
    jo_frame_info* f = d->info;
    long length;
    const int mark = get_marker(d);
    if (mark < 0) return -1;
    switch (mark) {
    case 0xC0:                                        /* SOF0 */
        (void)br_word(&d->in);
        if (analyze_frame(d)) return -1;
        break;
    case 0xC4:                                        /* DHT */
        length = br_word(&d->in) - 2;
        if (analyze_dht(d, length)) return -1;
        return 0x01;
    case 0xDC:                                        /* DNL */
        (void)br_word(&d->in);
        f->height = br_word(&d->in);
        break;
    case 0xDB:                                        /* DQT */
        length = br_word(&d->in) - 2;
        if (analyze_dqt(d, length)) return -1;
        return 0x02;
    case 0xD9:                                        /* EOI */
        d->enable = 0;
        break;
    case 0xDA:                                        /* SOS */
        (void)br_word(&d->in);
        if (analyze_scan(d)) return -1;
        return 0x10;
    case 0xDD:                                        /* DRI */
        (void)br_word(&d->in);
        f->restart_interval = br_word(&d->in);
        break;
    case 0xFE: {                                      /* COM */
        length = br_word(&d->in) - 2;
        size_t k = 0;
        for (long i = 0; i < length; ++i) {
            const int c = br_byte(&d->in);
            if (c < 0) return -1;
            if (c && k + 1 < sizeof f->comment) f->comment[k++] = (char)c;
        }
        f->comment[k] = 0;
        return 0x08;
    }
    /* unsupported frames: the reference builds a runtime_error but never throws it (:420) */
    case 0xC1: case 0xC2: case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB:
    case 0xCD: case 0xCE: case 0xCF: case 0xDF: case 0xCC: case 0xDE:
        break;
    case 0xE0: {                                      /* APP0, ref :422-448 */
        length = br_word(&d->in) - 2;
        if (length >= 4) {
            char id[5];
            for (int i = 0; i < 5; ++i) id[i] = (char)br_byte(&d->in);
            if (!memcmp(id, "JFIF", 4)) {
                f->jfif = 1;                          /* analyze_jfif, :336-350 */
                f->major_rev = br_byte(&d->in);
                f->minor_rev = br_byte(&d->in);
                f->units = br_byte(&d->in);
                f->hdensity = br_word(&d->in);
                f->vdensity = br_word(&d->in);
                (void)br_byte(&d->in); (void)br_byte(&d->in);
                d->decodable |= 0x04;
                br_skip(&d->in, length - 14);
            } else if (!memcmp(id, "JFXX", 4)) {
                f->jfif = 2;
                (void)br_byte(&d->in);
                br_skip(&d->in, length - 1);
            } else {
                br_skip(&d->in, length - 4);
            }
        } else {
            br_skip(&d->in, length);
        }
        break;
    }
    case 0xE1: case 0xE2: case 0xE3: case 0xE4: case 0xE5: case 0xE6: case 0xE7: case 0xE8:
    case 0xE9: case 0xEA: case 0xEB: case 0xEC: case 0xED: case 0xEE: case 0xEF:
        length = br_word(&d->in) - 2;
        br_skip(&d->in, length);
        break;
    default:
        return -1;                                    /* "Marker error" */
    }
    return 0;
}

/* ref :171-188 */
static int analyze_header(dstate* d)
{
    do {
        const int m = get_marker(d);
        if (m < 0) return -1;
        if (m == 0xD8) d->enable = 1;
    } while (!d->enable);
    while (d->enable) {
        const int rbits = analyze_marker(d);
        if (rbits < 0) return -1;
        d->decodable |= rbits;
        if (d->decodable & 0x10) return 0;
    }
    return -1;
}

/* ref :626-642 -- note the table is picked with Td for DC *and* AC (:630) */
static int decode_huffman_impl(dstate* d, int is_ac, int sc)
{
    const dec_table* t = &d->ht[is_ac][d->Td[sc]];
    int code = 0, length = 0, k = 0;
    while (k < t->n && length < 16) {
        ++length;
        code <<= 1;
        const int next = br_bit(&d->in);
        if (next < 0) return next;
        code |= next;
        for (; k < t->n && t->sizeTP[k] == length; ++k)
            if (t->codeTP[k] == code) return t->valueTP[k];
    }
    return -2;
}

/* ref :583-624; blk in zig-zag order (blk[k] is the reference's dct[ZZ[k]]) */
static int decode_huffman(dstate* d, int sc, int pred_dct[3], int16_t* blk)
{
    int dc_diff = 0;
    int category = decode_huffman_impl(d, 0, sc);
    if (category > 0) {
        if (br_bits(&d->in, category, &dc_diff)) return -1;
        if ((dc_diff & (1 << (category - 1))) == 0) dc_diff -= (1 << category) - 1;
    } else if (category < 0) {
        return -1;
    }
    pred_dct[sc] += dc_diff;
    blk[0] = (int16_t)pred_dct[sc];

    for (int k = 1; k < 64;) {
        category = decode_huffman_impl(d, 1, sc);
        if (!category) {
            for (; k < 64; ++k) blk[k] = 0;
            break;
        } else if (category < 0) {
            return -1;
        }
        int run = category >> 4, acv = 0;
        category &= 0x0f;
        if (category) {
            if (br_bits(&d->in, category, &acv)) return -1;
            if (!(acv & (1 << (category - 1)))) acv -= (1 << category) - 1;
        }
        if ((run + k) > 63) return -1;
        for (; run-- > 0; ++k) blk[k] = 0;
        blk[k++] = (int16_t)acv;
    }
    return 0;
}

int jo_read_jpeg(const uint8_t* data, size_t len, jo_frame_info* info, int16_t* coeffs, size_t coeff_cap)
{
    dstate* d = (dstate*)calloc(1, sizeof *d);
    if (!d) return -1;
    memset(info, 0, sizeof *info);
    info->hdensity = info->vdensity = 1;             /* decoder ctor defaults, ref :54-55 */
    d->in.p = data; d->in.len = len;
    d->info = info;
    int rc = analyze_header(d);
    if (!rc && !(d->decodable & (0x01 | 0x02 | 0x10))) rc = -1;     /* ref :89 */
    if (!rc && (info->hmax <= 0 || info->vmax <= 0 || info->ncomp <= 0)) rc = -1;
    if (!rc) {
        const int Vblock = (info->height >> 3) + ((info->height & 7) > 0);   /* get_blocks, :166-169 */
        const int Hblock = (info->width >> 3) + ((info->width & 7) > 0);
        info->mcu_cols = (Hblock / info->hmax) + ((Hblock % info->hmax) ? 1 : 0);
        info->mcu_rows = (Vblock / info->vmax) + ((Vblock % info->vmax) ? 1 : 0);
        info->blocks_per_mcu = 0;
        for (int i = 0; i < info->ncomp; ++i) {
            if (info->H[i] <= 0 || info->V[i] <= 0) { rc = -1; break; }
            info->blocks_per_mcu += info->H[i] * info->V[i];
        }
    }
    if (!rc && coeffs) {
        const size_t nmcu = (size_t)info->mcu_cols * info->mcu_rows;
        if (coeff_cap < nmcu * info->blocks_per_mcu * 64) rc = -3;
        int pred_dct[3] = { 0, 0, 0 };
        size_t restart_counter = 0;
        int16_t* out = coeffs;
        for (size_t m = 0; !rc && m < nmcu; ++m) {
            for (int sc = 0; !rc && sc < info->ncomp; ++sc)                   /* decode_mcu, :504-528 */
                for (int kb = 0; kb < info->H[sc] * info->V[sc]; ++kb, out += 64)
                    if (decode_huffman(d, sc, pred_dct, out)) { rc = -2; break; }
            if (!rc && info->restart_interval) {                              /* :152-163 */
                if (++restart_counter >= (size_t)info->restart_interval) {
                    restart_counter = 0;
                    const int mark = get_marker(d);
                    if (mark >= 0xD0 && mark <= 0xD7) pred_dct[0] = pred_dct[1] = pred_dct[2] = 0;
                }
            }
        }
    }
    free(d);
    return rc;
}

/* ------------------------------------------------------------------------------------------------ */
/* a13: decoder::inverse_dct (ref jpezy_decoder.hpp:652-670)                                          */
/* ------------------------------------------------------------------------------------------------ */
void jo_idct_block(const int dct[64], int precision, int out[64])
{
    const int sl = precision == 8 ? 128 : 2048;
    const double disqrt2 = INV_SQRT2;
    for (int y = 0; y < 8; ++y) {
        for (int x = 0; x < 8; ++x) {
            double sum = 0;
            for (int v = 0; v < 8; ++v) {
                const double cv = (!v) ? disqrt2 : 1.0;
                for (int u = 0; u < 8; ++u) {
                    const double cu = (!u) ? disqrt2 : 1.0;
                    sum += cu * cv * dct[v * 8 + u] * COS_TABLE[u * 8 + x] * COS_TABLE[v * 8 + y];
                }
            }
            out[y * 8 + x] = (int)(sum / 4 + sl);
        }
    }
}

/* ref :567-578, 672-676 */
static double to_r(double yp, double vp) { return yp + (vp - 0x80) * 1.4020; }
static double to_g(double yp, double up, double vp) { return yp - (up - 0x80) * 0.3441 - (vp - 0x80) * 0.7139; }
static double to_b(double yp, double up) { return yp + (up - 0x80) * 1.7718; }
static uint8_t revise_value(double v) { return (v < 0.0) ? 0 : (v > 255.0) ? 255 : (uint8_t)v; }

void jo_decode_planes_rows(const int16_t* coeffs, const jo_frame_info* info, int gray, int mcu_y0, int mcu_y1,
                           uint8_t* r, uint8_t* g, uint8_t* b)
{
    const int hmax = info->hmax, vmax = info->vmax;
    const int W = info->width, Himg = info->height;
    const int end_x = hmax * 8, end_y = vmax * 8;
    const int unit_size = hmax * vmax * 64;
    int* comp[3];
    for (int i = 0; i < 3; ++i) {
        comp[i] = (int*)malloc(sizeof(int) * (size_t)unit_size);
        for (int k = 0; k < unit_size; ++k) comp[i][k] = i ? 0x80 : 0;        /* ref :104-105 */
    }
    for (int uy = mcu_y0; uy < mcu_y1; ++uy) {
        for (int ux = 0; ux < info->mcu_cols; ++ux) {
            const int16_t* in = coeffs + ((size_t)uy * info->mcu_cols + ux) * info->blocks_per_mcu * 64;
            for (int sc = 0; sc < info->ncomp; ++sc) {                        /* decode_mcu, :504-528 */
                const int num_v = info->V[sc], num_h = info->H[sc];
                const int dupc_y = vmax / num_v, dupc_x = hmax / num_h;
                const int v_step = hmax * 8;
                for (int ky = 0; ky < num_v; ++ky) {
                    for (int kx = 0; kx < num_h; ++kx, in += 64) {
                        int dct[64], block[64];
                        const uint16_t* q = info->qt[info->Tq[sc] & 3];
                        for (int k = 0; k < 64; ++k) dct[ZZ[k]] = in[k];      /* decode_huffman stores dct[ZZ[k]] */
                        for (int i = 0; i < 64; ++i) dct[i] *= q[i];          /* a12, :645-650 */
                        jo_idct_block(dct, info->precision, block);
                        int* tp = comp[sc] + ky * v_step * 8 + kx * 8;
                        for (int y_u = 0; y_u < 8 * dupc_y; ++y_u)
                            for (int x_u = 0; x_u < 8 * dupc_x; ++x_u)
                                tp[y_u * v_step + x_u] = block[(y_u / dupc_y) * 8 + (x_u / dupc_x)];
                    }
                }
            }
            /* make_rgb, :531-565 (rows >= height land past W*H in the reference's vectors: dropped) */
            for (int pic_y = 0; pic_y < end_y; ++pic_y) {
                const long row = (long)uy * end_y + pic_y;
                if (row >= Himg) break;
                for (int pic_x = 0; pic_x < end_x; ++pic_x) {
                    const long col = (long)ux * end_x + pic_x;
                    if (col >= W) break;
                    const int yv = comp[0][pic_y * end_x + pic_x];
                    const int uv = comp[1][pic_y * end_x + pic_x];
                    const int vv = comp[2][pic_y * end_x + pic_x];
                    const long index = row * W + col;
                    if (!gray) {
                        r[index] = revise_value(to_r(yv, vv));
                        g[index] = revise_value(to_g(yv, uv, vv));
                        b[index] = revise_value(to_b(yv, uv));
                    } else {
                        g[index] = b[index] = r[index] = revise_value(yv);
                    }
                }
            }
        }
    }
    for (int i = 0; i < 3; ++i) free(comp[i]);
}

void jo_decode_planes(const int16_t* coeffs, const jo_frame_info* info, int gray,
                      uint8_t* r, uint8_t* g, uint8_t* b)
{
    jo_decode_planes_rows(coeffs, info, gray, 0, info->mcu_rows, r, g, b);
}

int jo_decode_jpeg(const uint8_t* data, size_t len, int gray, jo_frame_info* info,
                   uint8_t* r, uint8_t* g, uint8_t* b, size_t plane_cap)
{
    int rc = jo_read_jpeg(data, len, info, NULL, 0);
    if (rc) return rc;
    if ((size_t)info->width * info->height > plane_cap) return -3;
    const size_t ncoef = (size_t)info->mcu_cols * info->mcu_rows * info->blocks_per_mcu * 64;
    int16_t* coeffs = (int16_t*)malloc(ncoef * sizeof(int16_t));
    if (!coeffs) return -1;
    rc = jo_read_jpeg(data, len, info, coeffs, ncoef);
    if (!rc) jo_decode_planes(coeffs, info, gray, r, g, b);
    free(coeffs);
    return rc;
}

/* ------------------------------------------------------------------------------------------------ */
/* PPM P3 (ref encoder/encode_io.hpp:45-101; decoder/decode_io.hpp:36-54)                             */
/* ------------------------------------------------------------------------------------------------ */
static int next_line(FILE* f, char** line, size_t* cap, int* hit_eof)   /* jump_comment, :49-55 */
{
    for (;;) {
        size_t n = 0;
        int c;
        *hit_eof = 0;
        while ((c = fgetc(f)) != EOF && c != '\n') {
            if (n + 2 > *cap) { *cap = *cap ? *cap * 2 : 256; *line = (char*)realloc(*line, *cap); }
            (*line)[n++] = (char)c;
        }
        if (n + 1 > *cap) { *cap = *cap ? *cap * 2 : 256; *line = (char*)realloc(*line, *cap); }
        (*line)[n] = 0;
        if (c == EOF) { *hit_eof = 1; return n > 0; }       /* std::getline sets eofbit */
        if (!strchr(*line, '#')) return 1;                  /* lines containing '#' are skipped */
    }
}

int jo_read_ppm_p3(const char* path, int* W, int* H, uint8_t** r, uint8_t** g, uint8_t** b)
{
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    char* line = NULL;
    size_t cap = 0;
    int eof = 0, rc = -1;
    uint8_t* img = NULL;
    size_t nimg = 0, cimg = 0;
    do {
        next_line(f, &line, &cap, &eof);
        if (strcmp(line, "P3")) break;                      /* :63 */
        next_line(f, &line, &cap, &eof);
        int w, h, consumed = 0;
        if (sscanf(line, "%d %d%n", &w, &h, &consumed) != 2) break;   /* exactly two tokens, :68-72 */
        while (line[consumed] == ' ' || line[consumed] == '\t' || line[consumed] == '\r') ++consumed;
        if (line[consumed]) break;
        next_line(f, &line, &cap, &eof);                    /* max_color: parsed, unused (:77) */
        for (;;) {
            next_line(f, &line, &cap, &eof);
            if (eof) break;                                 /* a last unterminated line is dropped (:80) */
            char* p = line;
            for (;;) {
                while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
                if (!*p) break;
                const long v = strtol(p, &p, 10);
                if (nimg == cimg) { cimg = cimg ? cimg * 2 : 4096; img = (uint8_t*)realloc(img, cimg); }
                img[nimg++] = (uint8_t)v;
            }
        }
        const size_t npx = nimg / 3;
        *W = w; *H = h;
        *r = (uint8_t*)malloc(npx ? npx : 1); *g = (uint8_t*)malloc(npx ? npx : 1); *b = (uint8_t*)malloc(npx ? npx : 1);
        for (size_t i = 0; i < npx; ++i) { (*r)[i] = img[3 * i]; (*g)[i] = img[3 * i + 1]; (*b)[i] = img[3 * i + 2]; }
        rc = (npx >= (size_t)w * h) ? 0 : -2;
    } while (0);
    free(img);
    free(line);
    fclose(f);
    return rc;
}

long jo_format_ppm_p3(int W, int H, const uint8_t* r, const uint8_t* g, const uint8_t* b, char* out, size_t cap)
{
    size_t n = 0;
    int k = snprintf(out, cap, "P3\n# Decoded by jpezy\n%d %d\n255\n", W, H);
    if (k < 0 || (size_t)k >= cap) return -1;
    n = (size_t)k;
    for (long i = 0; i < (long)W * H; ++i) {
        if (n + 16 > cap) return -1;
        n += (size_t)sprintf(out + n, "%u %u %u\n", r[i], g[i], b[i]);
    }
    return (long)n;
}

void jo_free(void* p) { free(p); }
