// jpezy_host_codec.cpp -- the serial tail (encode) and head (decode) of the codec, on the host.
//
// BASELINE.json north_star keeps Huffman coding with the fixed Annex-K tables on the CPU; the GPU hands
// over / receives zig-zagged int16 coefficients.  This file produces exactly the bytes the reference's
// jpezy_writer + encoder::encode_huffman would (ref encoder/jpezy_writer.hpp:20-105,
// encoder/jpezy_encoder.hpp:174-242) and parses what decoder::analyze_header / decode_huffman accept
// (ref decoder/jpezy_decoder.hpp:171-502, 583-642), but is organised for throughput: symbol-indexed code
// LUTs, a 64-bit bit accumulator, canonical max-code decoding with an 8-bit first-level lookup.
#include "jpezy_host_codec.h"

#include <cstring>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "../../include/jpezy_constants.h"

namespace jpezy_host {
namespace {

const uint8_t kZZ[64] = JPEZY_ZZ_INIT;
const uint8_t kQtLuma[64] = JPEZY_QT_LUMA_INIT;
const uint8_t kQtChroma[64] = JPEZY_QT_CHROMA_INIT;

struct HuffSpec {
    uint8_t bits[16];
    const uint8_t* vals;
    int nval;
};
const uint8_t kDcLVals[] = JPEZY_DC_LUMA_VALS_INIT, kDcCVals[] = JPEZY_DC_CHROMA_VALS_INIT;
const uint8_t kAcLVals[] = JPEZY_AC_LUMA_VALS_INIT, kAcCVals[] = JPEZY_AC_CHROMA_VALS_INIT;
const HuffSpec kSpec[4] = {   // order of the DHT segments in the file: YDc, CDc, YAc, CAc (jpezy_writer.hpp:61-64)
    { JPEZY_DC_LUMA_BITS_INIT, kDcLVals, JPEZY_DC_LUMA_NVAL },
    { JPEZY_DC_CHROMA_BITS_INIT, kDcCVals, JPEZY_DC_CHROMA_NVAL },
    { JPEZY_AC_LUMA_BITS_INIT, kAcLVals, JPEZY_AC_LUMA_NVAL },
    { JPEZY_AC_CHROMA_BITS_INIT, kAcCVals, JPEZY_AC_CHROMA_NVAL },
};
const uint8_t kSpecId[4] = { 0x00, 0x01, 0x10, 0x11 };

// ---- encoder LUT: symbol -> (code, length), canonical codes of Annex C ----
struct EncLut {
    uint16_t code[256];
    uint8_t len[256];
};
struct EncTables {
    EncLut t[4];
    EncTables()
    {
        for (int k = 0; k < 4; ++k) {
            std::memset(&t[k], 0, sizeof t[k]);
            unsigned code = 0;
            int p = 0;
            for (int l = 1; l <= 16; ++l) {
                for (int c = 0; c < kSpec[k].bits[l - 1]; ++c, ++p) {
                    t[k].code[kSpec[k].vals[p]] = (uint16_t)code++;
                    t[k].len[kSpec[k].vals[p]] = (uint8_t)l;
                }
                code <<= 1;
            }
        }
    }
};
const EncTables& enc_tables()
{
    static const EncTables e;
    return e;
}

// ---- MSB-first bit packer with 0xFF00 stuffing; raw byte writes realign on a byte (zero pad bits) ----
class BitSink {
public:
    BitSink(uint8_t* p, size_t cap) : p_(p), end_(p + cap), cur_(p) {}
    bool ok() const { return !overflow_; }
    size_t size() const { return (size_t)(cur_ - p_); }

    void raw(unsigned v)
    {
        flush_partial();
        if (cur_ < end_) *cur_++ = (uint8_t)v; else overflow_ = true;
    }
    void raw16(unsigned v) { raw(v >> 8); raw(v & 0xFF); }
    void raw_n(const void* s, size_t n)
    {
        const uint8_t* b = static_cast<const uint8_t*>(s);
        for (size_t i = 0; i < n; ++i) raw(b[i]);
    }
    inline void bits(unsigned v, int n)   // append the low n bits of v
    {
        acc_ = (acc_ << n) | (v & ((1u << n) - 1u));
        nacc_ += n;
        while (nacc_ >= 8) {
            const unsigned byte = (unsigned)(acc_ >> (nacc_ - 8)) & 0xFFu;
            nacc_ -= 8;
            put(byte);
            if (byte == 0xFF) put(0x00);
        }
    }
    void flush_partial()
    {
        if (nacc_ > 0) {   // pad bits: JPEZY_PAD_BIT (frozen bofstream semantics): 0 -- the byte cannot be 0xFF
            unsigned byte = (unsigned)(acc_ << (8 - nacc_)) & 0xFFu;
#if JPEZY_PAD_BIT
            byte |= (1u << (8 - nacc_)) - 1u;
            put(byte);
            if (byte == 0xFF) put(0x00);
#else
            put(byte);
#endif
            nacc_ = 0;
        }
        acc_ = 0;
    }

private:
    inline void put(unsigned b)
    {
        if (cur_ < end_) *cur_++ = (uint8_t)b; else overflow_ = true;
    }
    uint8_t* p_;
    uint8_t* end_;
    uint8_t* cur_;
    uint64_t acc_ = 0;
    int nacc_ = 0;
    bool overflow_ = false;
};

inline int bit_length(unsigned v) { return v ? 32 - __builtin_clz(v) : 0; }

// one block: DC difference category + AC run/size symbols (ref jpezy_encoder.hpp:174-225)
inline bool put_block(BitSink& o, const int16_t* z, int& pred, const EncLut& dc, const EncLut& ac)
{
    const int diff = z[0] - pred;
    pred = z[0];
    const int di = bit_length((unsigned)(diff < 0 ? -diff : diff));
    if (di > 11) return false;                                    // reference: throw runtime_error (:186)
    o.bits(dc.code[di], dc.len[di]);
    if (di) o.bits((unsigned)(diff < 0 ? diff - 1 : diff), di);

    int run = 0;
    for (int n = 1; n < 64; ++n) {
        const int v = z[n];
        if (v == 0) {
            ++run;
            continue;
        }
        while (run > 15) {
            o.bits(ac.code[0xF0], ac.len[0xF0]);                  // ZRL
            run -= 16;
        }
        const int s = bit_length((unsigned)(v < 0 ? -v : v));
        if (s > 10) return false;                                 // outside K.5/K.6 (reference: :207 / aliasing)
        const int sym = (run << 4) | s;
        o.bits(ac.code[sym], ac.len[sym]);
        o.bits((unsigned)(v < 0 ? v - 1 : v), s);
        run = 0;
    }
    if (run) o.bits(ac.code[0x00], ac.len[0x00]);                 // EOB when the block ends in zeros (:219-220)
    return true;
}

void put_header(BitSink& o, int W, int H, const char* comment)
{
    static const uint8_t soi_app0[] = { 0xFF, 0xD8, 0xFF, 0xE0, 0x00, 0x10, 'J', 'F', 'I', 'F', 0x00,
                                        0x01, 0x02, 0x01, 0x00, 0x60, 0x00, 0x60, 0x00, 0x00 };
    o.raw_n(soi_app0, sizeof soi_app0);
    if (comment && *comment) {
        const size_t n = std::strlen(comment);
        o.raw(0xFF); o.raw(0xFE);
        o.raw16((unsigned)(n + 3));
        o.raw_n(comment, n + 1);
    }
    for (int t = 0; t < 2; ++t) {
        const uint8_t* q = t ? kQtChroma : kQtLuma;
        o.raw(0xFF); o.raw(0xDB); o.raw16(67); o.raw((unsigned)t);
        for (int i = 0; i < 64; ++i) o.raw(q[kZZ[i]]);
    }
    for (int k = 0; k < 4; ++k) {
        o.raw(0xFF); o.raw(0xC4);
        o.raw16((unsigned)(19 + kSpec[k].nval));
        o.raw(kSpecId[k]);
        o.raw_n(kSpec[k].bits, 16);
        o.raw_n(kSpec[k].vals, (size_t)kSpec[k].nval);
    }
    const uint8_t sof_sos[] = { 0xFF, 0xC0, 0x00, 0x11, 0x08, (uint8_t)(H >> 8), (uint8_t)H, (uint8_t)(W >> 8), (uint8_t)W,
                                0x03, 0x00, 0x22, 0x00, 0x01, 0x11, 0x01, 0x02, 0x11, 0x01,
                                0xFF, 0xDA, 0x00, 0x0C, 0x03, 0x00, 0x00, 0x01, 0x11, 0x02, 0x11, 0x00, 0x3F, 0x00 };
    o.raw_n(sof_sos, sizeof sof_sos);
}

}  // namespace

size_t write_header(int W, int H, const char* comment, uint8_t* out, size_t cap)
{
    BitSink o(out, cap);
    put_header(o, W, H, comment);
    return o.ok() ? o.size() : 0;
}

void enc_code_tables(uint16_t code[4][256], uint8_t len[4][256])
{
    const EncTables& T = enc_tables();
    for (int k = 0; k < 4; ++k) {
        std::memcpy(code[k], T.t[k].code, sizeof T.t[k].code);
        std::memcpy(len[k], T.t[k].len, sizeof T.t[k].len);
    }
}

size_t jpeg_bound(int W, int H)
{
    const size_t nmcu = (size_t)((W + 15) / 16) * (size_t)((H + 15) / 16);
    // worst case per coefficient: 16-bit code + 10 value bits, doubled by byte stuffing
    return 1024 + nmcu * 6 * 64 * 7;
}

long write_jpeg(const int16_t* coeffs, int W, int H, bool gray, const char* comment, uint8_t* out, size_t cap,
                std::string* err)
{
    if (!coeffs || !out || W <= 0 || H <= 0 || W > 65535 || H > 65535) {
        if (err) *err = "write_jpeg: bad argument";
        return JPEZY_E_BADARG;
    }
    const EncTables& T = enc_tables();
    BitSink o(out, cap);
    put_header(o, W, H, comment);

    static const int16_t kZeroBlock[64] = { 0 };
    const size_t nmcu = (size_t)((W + 15) / 16) * (size_t)((H + 15) / 16);
    const int bpm = gray ? 4 : 6;
    int pred[3] = { 0, 0, 0 };                                     // pre_DC, never reset (no RSTn)
    for (size_t mcu = 0; mcu < nmcu; ++mcu) {
        const int16_t* z = coeffs + mcu * (size_t)bpm * 64;
        bool good = true;
        for (int i = 0; i < 4; ++i) good &= put_block(o, z + i * 64, pred[0], T.t[0], T.t[2]);
        good &= put_block(o, gray ? kZeroBlock : z + 256, pred[1], T.t[1], T.t[3]);
        good &= put_block(o, gray ? kZeroBlock : z + 320, pred[2], T.t[1], T.t[3]);
        if (!good) {
            if (err) *err = "write_jpeg: coefficient outside the Annex-K code tables";
            return JPEZY_E_FORMAT;
        }
        if (!o.ok()) break;
    }
    o.raw(0xFF); o.raw(0xD9);
    if (!o.ok()) {
        if (err) *err = "write_jpeg: output buffer too small";
        return JPEZY_E_NOSPACE;
    }
    return (long)o.size();
}

// ======================================================================================================
// reader
// ======================================================================================================
namespace {

struct DecTable {
    int n = 0;
    uint8_t val[256];
    int32_t maxcode[18];   // maxcode[l]: largest code of length l (or -1)
    int32_t valptr[17];
    int32_t mincode[17];
    uint8_t look_len[256]; // 8-bit first-level lookup: code length (0 = longer than 8)
    bool prefix_code = true; // false: the counts over-subscribe the code space; no lookup, the bit-by-bit search decides
    uint8_t look_val[256];
    // a table no DHT segment defined holds no codes: every symbol read from it fails, like the reference's empty
    // huffman_table (decode_huffman_impl finds nothing and throws, :629-641)
    DecTable()
    {
        std::memset(look_len, 0, sizeof look_len);
        std::memset(look_val, 0, sizeof look_val);
        std::memset(val, 0, sizeof val);
        for (int l = 0; l < 18; ++l) maxcode[l] = -1;
        for (int l = 0; l < 17; ++l) valptr[l] = mincode[l] = 0;
    }
};

void build_dec(DecTable& t, const uint8_t bits[16], const uint8_t* vals, int n)
{
    t.n = n;
    std::memcpy(t.val, vals, (size_t)n);
    std::memset(t.look_len, 0, sizeof t.look_len);
    int code = 0, p = 0;
    t.prefix_code = true;
    for (int l = 1; l <= 16; ++l) {
        if (code + (int)bits[l - 1] > (1 << l)) t.prefix_code = false;
        if (bits[l - 1]) {
            t.valptr[l] = p;
            t.mincode[l] = code;
            for (int c = 0; c < bits[l - 1]; ++c, ++p, ++code) {
                if (l <= 8 && code < (1 << l)) {       // an over-subscribed DHT yields codes no l-bit read can equal: not in the LUT
                    const int lo = code << (8 - l);
                    for (int f = 0; f < (1 << (8 - l)); ++f) {
                        t.look_len[lo + f] = (uint8_t)l;
                        t.look_val[lo + f] = vals[p];
                    }
                }
            }
            t.maxcode[l] = code - 1;
        } else {
            t.maxcode[l] = -1;
            t.valptr[l] = p;
            t.mincode[l] = code;
        }
        code <<= 1;
    }
    t.maxcode[17] = 0x7FFFFFFF;
    if (!t.prefix_code) std::memset(t.look_len, 0, sizeof t.look_len);
}

struct Reader {
    const uint8_t* p;
    size_t len, pos = 0;
    uint64_t acc = 0;
    int nacc = 0;
    int pad = 0;       // zero bytes fed past the end of the data

    int byte() { nacc = 0; acc = 0; return pos < len ? p[pos++] : -1; }
    int word() { const int a = byte(), b = byte(); return (a < 0 || b < 0) ? -1 : (a << 8) | b; }
    void skip(long n) { nacc = 0; acc = 0; if (n > 0) pos += (size_t)n; if (pos > len) pos = len; }

    inline void fill()
    {
        while (nacc <= 56) {
            unsigned b = 0;
            if (pos < len) {
                b = p[pos++];
                if (b == 0xFF && pos < len && p[pos] == 0x00) ++pos;   // stuffed zero
            } else {
                ++pad;
            }
            acc = (acc << 8) | b;
            nacc += 8;
        }
    }
    inline int peek8() { if (nacc < 8) fill(); return (int)((acc >> (nacc - 8)) & 0xFF); }
    inline int get(int n)
    {
        if (n == 0) return 0;
        if (nacc < n) fill();
        nacc -= n;
        return (int)((acc >> nacc) & ((1u << n) - 1u));
    }
    // Before a byte-level marker scan (restart handling): give back the whole bytes that were buffered
    // but not used.  A stuffed 0xFF00 pair was consumed as one buffered byte.
    void realign()
    {
        int whole = nacc / 8 - pad;
        size_t q = pos;
        for (; whole > 0 && q > 0; --whole) {
            --q;
            if (q > 0 && p[q] == 0x00 && p[q - 1] == 0xFF) --q;
        }
        pos = q;
        nacc = 0; acc = 0; pad = 0;
    }
};

inline int decode_symbol(Reader& r, const DecTable& t)
{
    const int look = r.peek8();
    int l = t.look_len[look];
    if (l) {
        r.nacc -= l;
        return t.look_val[look];
    }
    int code = t.prefix_code ? r.get(8) : 0;
    for (l = t.prefix_code ? 9 : 1; l <= 16; ++l) {
        code = (code << 1) | r.get(1);
        // the codes of length l are mincode..maxcode; both tests, so that a DHT whose counts do not describe a prefix code
        // (over-subscribed, or with gaps) behaves like the reference's search for an equal (length, code) pair (:629-640)
        if (code >= t.mincode[l] && code <= t.maxcode[l]) return t.val[t.valptr[l] + code - t.mincode[l]];
    }
    return -1;
}

inline int extend(int v, int cat) { return (v & (1 << (cat - 1))) ? v : v - ((1 << cat) - 1); }

int next_marker(Reader& r)   // ref decoder/jpezy_decoder.hpp:486-502
{
    for (;;) {
        int c = r.byte();
        if (c < 0) return -1;
        if (c != 0xFF) continue;
        c = r.byte();
        if (c < 0) return -1;
        if (c == 0) continue;
        if (c > 0x02 && c < 0xC0) return 0xFF;
        return c;
    }
}

}  // namespace

namespace {
int read_jpeg_impl(const uint8_t* data, size_t len, jpezy_frame_info* info, int16_t* coeffs, size_t coeff_cap, std::string* err,
                   ScanSetup* setup);
}

int read_jpeg(const uint8_t* data, size_t len, jpezy_frame_info* info, int16_t* coeffs, size_t coeff_cap,
              std::string* err)
{
    return read_jpeg_impl(data, len, info, coeffs, coeff_cap, err, nullptr);
}

int parse_header(const uint8_t* data, size_t len, jpezy_frame_info* info, ScanSetup* setup, std::string* err)
{
    if (!setup) { if (err) *err = "parse_header: bad argument"; return JPEZY_E_BADARG; }
    std::memset(setup, 0, sizeof *setup);
    return read_jpeg_impl(data, len, info, nullptr, 0, err, setup);
}

namespace {
int read_jpeg_impl(const uint8_t* data, size_t len, jpezy_frame_info* info, int16_t* coeffs, size_t coeff_cap, std::string* err,
                   ScanSetup* setup)
{
    auto fail = [&](int code, const char* msg) { if (err) *err = msg; return code; };
    if (!data || !info) return fail(JPEZY_E_BADARG, "read_jpeg: bad argument");
    std::memset(info, 0, sizeof *info);
    info->hdensity = info->vdensity = 1;

    Reader r{ data, len };
    DecTable* ht = new DecTable[8];   // [tc*4 + th]
    struct Free { DecTable* p; ~Free() { delete[] p; } } guard{ ht };
    int Td[3] = { 0, 0, 0 };
    bool have_ht = false, have_qt = false, have_sos = false, have_sof = false;

    // SOI
    for (;;) {
        const int mk = next_marker(r);
        if (mk < 0) return fail(JPEZY_E_FORMAT, "read_jpeg: no SOI");
        if (mk == 0xD8) break;
    }
    while (!have_sos) {
        const int mk = next_marker(r);
        if (mk < 0) return fail(JPEZY_E_FORMAT, "read_jpeg: truncated header");
        switch (mk) {
        case 0xC0: {   // SOF0 (:279-305)
            r.word();
            info->precision = r.byte();
            info->height = r.word();
            info->width = r.word();
            info->ncomp = r.byte();
            if (info->ncomp != 3 && info->ncomp != 1) return fail(JPEZY_E_FORMAT, "read_jpeg: dimension not supported");
            for (int i = 0; i < info->ncomp; ++i) {
                r.byte();
                const int c = r.byte();
                if (c < 0) return fail(JPEZY_E_FORMAT, "read_jpeg: truncated SOF0");
                info->H[i] = c >> 4;
                info->V[i] = c & 15;
                if (info->H[i] > info->hmax) info->hmax = info->H[i];
                if (info->V[i] > info->vmax) info->vmax = info->V[i];
                info->Tq[i] = r.byte();
            }
            have_sof = true;
            break;
        }
        case 0xC4: {   // DHT (:190-256)
            const long seg = r.word() - 2;
            const size_t end = r.pos + (size_t)(seg > 0 ? seg : 0);
            do {
                const int id = r.byte();
                if (id < 0 || (id >> 4) > 1 || (id & 15) > 3) return fail(JPEZY_E_FORMAT, "read_jpeg: bad DHT id");
                uint8_t bits[16], vals[256];
                int n = 0;
                for (int i = 0; i < 16; ++i) { const int b = r.byte(); if (b < 0) return fail(JPEZY_E_FORMAT, "read_jpeg: truncated DHT"); bits[i] = (uint8_t)b; n += b; }
                if (n > 256) return fail(JPEZY_E_FORMAT, "read_jpeg: invalid size table");
                for (int i = 0; i < n; ++i) { const int b = r.byte(); if (b < 0) return fail(JPEZY_E_FORMAT, "read_jpeg: truncated DHT"); vals[i] = (uint8_t)b; }
                build_dec(ht[(id >> 4) * 4 + (id & 15)], bits, vals, n);
                if (setup) {
                    const int slot = (id >> 4) * 4 + (id & 15);
                    setup->present[slot] = 1;
                    setup->nvals[slot] = n;
                    std::memcpy(setup->bits[slot], bits, 16);
                    std::memcpy(setup->vals[slot], vals, (size_t)n);
                }
            } while (r.pos < end);
            have_ht = true;
            break;
        }
        case 0xDB: {   // DQT (:258-277)
            const long seg = r.word() - 2;
            const size_t end = r.pos + (size_t)(seg > 0 ? seg : 0);
            do {
                const int c = r.byte();
                if (c < 0) return fail(JPEZY_E_FORMAT, "read_jpeg: truncated DQT");
                uint16_t* q = info->qt[c & 3];
                for (int i = 0; i < 64; ++i) {
                    const int v = (c >> 4) ? r.word() : r.byte();
                    if (v < 0) return fail(JPEZY_E_FORMAT, "read_jpeg: truncated DQT");
                    q[kZZ[i]] = (uint16_t)v;
                }
            } while (r.pos < end);
            have_qt = true;
            break;
        }
        case 0xDC: r.word(); info->height = r.word(); break;                 // DNL
        case 0xDD: r.word(); info->restart_interval = r.word(); break;       // DRI
        case 0xD9: return fail(JPEZY_E_FORMAT, "read_jpeg: EOI before SOS");   // analyze_header throws (:187)
        case 0xDA: {   // SOS (:307-334)
            r.word();
            const int ns = r.byte();
            if (ns < 0 || ns > 3) return fail(JPEZY_E_FORMAT, "read_jpeg: bad SOS");
            for (int i = 0; i < ns; ++i) {
                r.byte();
                const int c = r.byte();
                if (c < 0 || (c >> 4) > 2 || (c & 15) > 2) return fail(JPEZY_E_FORMAT, "read_jpeg: bad SOS table id");
                Td[i] = c >> 4;
            }
            r.byte(); r.byte(); r.byte();
            have_sos = true;
            break;
        }
        case 0xFE: {   // COM (:405-410)
            const long seg = r.word() - 2;
            size_t k = 0;
            for (long i = 0; i < seg; ++i) {
                const int c = r.byte();
                if (c < 0) return fail(JPEZY_E_FORMAT, "read_jpeg: truncated COM");
                if (c && k + 1 < sizeof info->comment) info->comment[k++] = (char)c;
            }
            info->comment[k] = 0;
            break;
        }
        case 0xE0: {   // APP0 (:422-448)
            const long seg = r.word() - 2;
            if (seg >= 4) {
                char id[5];
                for (char& ch : id) ch = (char)r.byte();
                if (!std::memcmp(id, "JFIF", 4)) {
                    info->format = 1;
                    info->major_rev = r.byte();
                    info->minor_rev = r.byte();
                    info->units = r.byte();
                    info->hdensity = r.word();
                    info->vdensity = r.word();
                    r.byte(); r.byte();
                    r.skip(seg - 14);
                } else if (!std::memcmp(id, "JFXX", 4)) {
                    info->format = 2;
                    r.byte();
                    r.skip(seg - 1);
                } else {
                    r.skip(seg - 4);
                }
            } else {
                r.skip(seg);
            }
            break;
        }
        // frames the reference does not support: it builds an exception object and never throws it (:412-421)
        case 0xC1: case 0xC2: case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB:
        case 0xCD: case 0xCE: case 0xCF: case 0xDF: case 0xCC: case 0xDE:
            break;
        default:
            if (mk >= 0xE1 && mk <= 0xEF) { r.skip(r.word() - 2); break; }
            return fail(JPEZY_E_FORMAT, "read_jpeg: Marker error");          // :480-481
        }
    }
    if (!(have_ht || have_qt || have_sos)) return fail(JPEZY_E_FORMAT, "read_jpeg: not decodable");
    if (!have_sof || info->hmax <= 0 || info->vmax <= 0) return fail(JPEZY_E_FORMAT, "read_jpeg: no frame header");

    const int Vblock = (info->height >> 3) + ((info->height & 7) > 0);   // get_blocks (:166-169)
    const int Hblock = (info->width >> 3) + ((info->width & 7) > 0);
    info->mcu_cols = Hblock / info->hmax + ((Hblock % info->hmax) ? 1 : 0);
    info->mcu_rows = Vblock / info->vmax + ((Vblock % info->vmax) ? 1 : 0);
    info->blocks_per_mcu = 0;
    for (int i = 0; i < info->ncomp; ++i) {
        if (info->H[i] <= 0 || info->V[i] <= 0) return fail(JPEZY_E_FORMAT, "read_jpeg: zero sampling factor");
        info->blocks_per_mcu += info->H[i] * info->V[i];
    }
    if (setup) {
        setup->scan_pos = r.pos;
        for (int i = 0; i < 3; ++i) setup->Td[i] = Td[i];
    }
    if (!coeffs) return JPEZY_OK;

    const size_t nmcu = (size_t)info->mcu_cols * info->mcu_rows;
    if (coeff_cap < nmcu * (size_t)info->blocks_per_mcu * 64) return fail(JPEZY_E_NOSPACE, "read_jpeg: coefficient buffer too small");

    int pred[3] = { 0, 0, 0 };
    size_t restart_counter = 0;
    int16_t* z = coeffs;
    for (size_t mcu = 0; mcu < nmcu; ++mcu) {
        for (int sc = 0; sc < info->ncomp; ++sc) {
            // the reference selects BOTH tables with scomp[sc].Td (:630)
            const DecTable& dc = ht[0 * 4 + Td[sc]];
            const DecTable& ac = ht[1 * 4 + Td[sc]];
            for (int kb = info->H[sc] * info->V[sc]; kb > 0; --kb, z += 64) {
                int cat = decode_symbol(r, dc);
                if (cat < 0 || cat > 16) return fail(JPEZY_E_FORMAT, "read_jpeg: decode_huffman");
                if (cat) pred[sc] += extend(r.get(cat), cat);
                z[0] = (int16_t)pred[sc];
                int k = 1;
                while (k < 64) {
                    const int rs = decode_symbol(r, ac);
                    if (rs < 0) return fail(JPEZY_E_FORMAT, "read_jpeg: decode_huffman");
                    if (rs == 0) break;
                    const int run = rs >> 4, s = rs & 15;
                    if (run + k > 63) return fail(JPEZY_E_FORMAT, "read_jpeg: decode_huffman");
                    for (int i = 0; i < run; ++i) z[k++] = 0;
                    z[k++] = (int16_t)(s ? extend(r.get(s), s) : 0);
                }
                while (k < 64) z[k++] = 0;
                if (r.pad > 16) return fail(JPEZY_E_FORMAT, "read_jpeg: truncated scan");
            }
        }
        if (info->restart_interval && ++restart_counter >= (size_t)info->restart_interval) {   // :152-163
            restart_counter = 0;
            r.realign();
            const int mk = next_marker(r);
            if (mk >= 0xD0 && mk <= 0xD7) pred[0] = pred[1] = pred[2] = 0;
        }
    }
    return JPEZY_OK;
}
}  // namespace

size_t entropy_segment_length(const uint8_t* scan, size_t n)
{
    size_t i = 0;
#if defined(__SSE2__)
    const __m128i ff = _mm_set1_epi8((char)0xFF), zero = _mm_setzero_si128();
    for (; i + 17 <= n; i += 16) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(scan + i));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(scan + i + 1));
        const int hit = _mm_movemask_epi8(_mm_andnot_si128(_mm_cmpeq_epi8(b, zero), _mm_cmpeq_epi8(a, ff)));   // 0xFF, next != 0x00
        if (hit) return i + (size_t)__builtin_ctz((unsigned)hit);
    }
#endif
    for (; i < n; ++i)
        if (scan[i] == 0xFF && (i + 1 >= n || scan[i + 1] != 0x00)) return i;
    return n;
}

}  // namespace jpezy_host
