// jpezy_host_codec.h -- host-side serial tail/head of the codec (internal C++ API behind the C-ABI):
// JFIF writer + Annex-K Huffman encoder, marker parser + Huffman decoder.  Product code: independent of
// oracle/ (which restates the same reference functions for checking).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/jpezy_hip.h"

namespace jpezy_host {

// ref encoder/jpezy_writer.hpp:20-105 + encoder/jpezy_encoder.hpp:174-242
long write_jpeg(const int16_t* coeffs, int W, int H, bool gray, const char* comment, uint8_t* out, size_t cap,
                std::string* err);
size_t jpeg_bound(int W, int H);
// the bytes before the entropy-coded segment (SOI .. SOS, 644 with the default comment); 0 if cap is too small
size_t write_header(int W, int H, const char* comment, uint8_t* out, size_t cap);
// canonical (code, length) per symbol of the four Annex-K tables in DHT order YDc, CDc, YAc, CAc (for the GPU coder)
void enc_code_tables(uint16_t code[4][256], uint8_t len[4][256]);

// What the entropy decoder needs besides jpezy_frame_info: where the scan data start, the raw DHT specifications
// (slot = tc*4 + th: 0..3 DC, 4..7 AC) and the table selector of each scan component (the reference uses Td for both
// the DC and the AC table, decoder/jpezy_decoder.hpp:630).
struct ScanSetup {
    size_t scan_pos;
    int Td[3];
    uint8_t present[8];
    int nvals[8];
    uint8_t bits[8][16];
    uint8_t vals[8][256];
};
// header only (the marker parser of read_jpeg): fills info and setup
int parse_header(const uint8_t* data, size_t len, jpezy_frame_info* info, ScanSetup* setup, std::string* err);

// ref decoder/jpezy_decoder.hpp:171-502, 583-642
int read_jpeg(const uint8_t* data, size_t len, jpezy_frame_info* info, int16_t* coeffs, size_t coeff_cap,
              std::string* err);

// Length of the entropy-coded segment that starts at scan[0]: it ends before the first marker -- a 0xFF followed by anything but 0x00 --
// or before a 0xFF that is the last byte (decoder::decode_huffman reads on until its bit reader meets one, ref decoder/jpezy_decoder.hpp:
// 583-642); n when there is none.  Compressed data holds a 0xFF every ~256 bytes, so this is a 16-bytes-at-a-time compare, not a memchr
// per 0xFF.
size_t entropy_segment_length(const uint8_t* scan, size_t n);

}  // namespace jpezy_host
