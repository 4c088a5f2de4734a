// encode_io.hpp -- jpezy::encode_io (ASCII PPM P3 reader), to_jpeg, gray_scale and the pipe operators, mirroring
// src/encoder/encode_io.hpp:35-208.  The reference's parser goes line by line through std::list and boost::split and
// is 12x slower than its encoder (README.md:49,52); this one is a single pass over the file with the SAME acceptance
// rules: lines containing '#' are dropped (:53), "P3" must be a line of its own (:63), width/height must be one line
// of exactly two space-separated tokens (:68-72), max_color is parsed and unused (:77), a last line without a
// trailing newline is dropped (:80), an empty token inside a pixel line is an error (std::stoi("") throws).
#ifndef JPEZY_AMD_HOST_ENCODE_IO_HPP
#define JPEZY_AMD_HOST_ENCODE_IO_HPP
#include <array>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <thread>
#include <fstream>
#include <iostream>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "jpezy_encoder.hpp"
#include "pnm_stream.hpp"

namespace jpezy {

struct to_jpeg {
    explicit constexpr to_jpeg(const char* file_) : file(file_) {}
    const char* file;
};

struct gray_scale_t {};
inline constexpr gray_scale_t gray_scale{};

struct encode_io : pnm_stream {
    encode_io(const char* file_name) : pnm_stream(true, 0, 0, 0)
    {
        std::string text;
        if (std::FILE* fp = std::fopen(file_name, "rb")) {
            std::fseek(fp, 0, SEEK_END);
            const long n = std::ftell(fp);
            std::fseek(fp, 0, SEEK_SET);
            text.resize(n > 0 ? static_cast<std::size_t>(n) : 0);
            if (n > 0 && std::fread(text.data(), 1, text.size(), fp) != text.size()) text.clear();
            std::fclose(fp);
        } else {
            initializing_succeed = false;
            return;
        }
        std::size_t pos = 0;
        bool eof = false;
        // jump_comment (:49-55): next line that holds no '#'; sets eof when the line ended at end of file
        auto next_line = [&]() -> std::string_view {
            for (;;) {
                if (pos >= text.size()) { eof = true; return {}; }
                const std::size_t nl = text.find('\n', pos);
                std::string_view line;
                if (nl == std::string::npos) { line = std::string_view(text).substr(pos); pos = text.size(); eof = true; }
                else { line = std::string_view(text).substr(pos, nl - pos); pos = nl + 1; }
                if (eof || line.find('#') == std::string_view::npos) return line;
            }
        };
        auto is_sp = [](char c) { return std::isspace(static_cast<unsigned char>(c)) != 0; };

        if (next_line() != "P3") { initializing_succeed = false; return; }
        {
            const std::string_view wh = next_line();
            std::vector<std::string> tok(1);
            for (char c : wh) { if (is_sp(c)) tok.emplace_back(); else tok.back().push_back(c); }
            if (tok.size() != 2) { initializing_succeed = false; return; }
            width = static_cast<std::size_t>(std::stoi(tok[0]));
            height = static_cast<std::size_t>(std::stoi(tok[1]));
        }
        max_color = static_cast<std::size_t>(std::stoi(std::string(next_line())));

        // Pixel lines.  Every rule above is local to a line, so the body is cut into pieces at line ends and the pieces
        // are tokenised by all host cores at once (a 4096x4096 P3 file is 180 MB of text: 1 s on one core, the GPU
        // stage behind it takes milliseconds); results are joined in file order.
        struct Part { std::vector<value_type> vals; bool bad = false; };
        static const auto sp = [] {
            std::array<bool, 256> t{};
            for (unsigned char c : { ' ', '\t', '\n', '\v', '\f', '\r' }) t[c] = true;
            return t;
        }();
        const char* const base = text.data();
        auto parse_piece = [&](std::size_t b, std::size_t e, Part& out) {      // [b, e): whole lines, each ending in '\n'
            std::vector<value_type> vals;                                      // local: neighbouring Parts share cache lines
            vals.reserve((e - b) / 3);
            const char* p = base + b;
            const char* const end = base + e;
            while (p < end) {
                const char* nl = static_cast<const char*>(std::memchr(p, '\n', static_cast<std::size_t>(end - p)));
                if (!nl) break;                                                // unterminated last line: dropped (:80)
                if (std::memchr(p, '#', static_cast<std::size_t>(nl - p))) { p = nl + 1; continue; }   // comment line (:53)
                const char* i = p;
                for (;;) {
                    const char* j = i;
                    while (j < nl && !sp[static_cast<unsigned char>(*j)]) ++j;
                    if (j == i) {                          // empty token
                        if (i == nl) break;                // ... the trailing one is popped (:84-85)
                        out.bad = true;
                        return;
                    }
                    int v = 0;
                    bool digits = false;
                    const char* k = i;
                    const bool neg = *k == '-';
                    if (neg || *k == '+') ++k;
                    for (; k < j && *k >= '0' && *k <= '9'; ++k) { v = v * 10 + (*k - '0'); digits = true; }
                    if (!digits) { out.bad = true; return; }
                    vals.push_back(static_cast<value_type>(neg ? -v : v));
                    if (j == nl) break;
                    i = j + 1;
                }
                p = nl + 1;
            }
            out.vals = std::move(vals);
        };
        const std::size_t body = pos, total = text.size();
        unsigned nt = std::thread::hardware_concurrency();
        if (nt == 0) nt = 4;
        if (nt > 16) nt = 16;
        if (const char* e = std::getenv("JPEZY_IO_THREADS")) nt = static_cast<unsigned>(std::max(1, std::atoi(e)));
        if (total - body < (std::size_t(4) << 20)) nt = 1;
        std::vector<std::size_t> cut(nt + 1, total);
        cut[0] = body;
        for (unsigned t = 1; t < nt; ++t) {
            const std::size_t guess = body + (total - body) / nt * t;
            const std::size_t nl = text.find('\n', guess);
            cut[t] = nl == std::string::npos ? total : nl + 1;
            if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
        }
        std::vector<Part> parts(nt);
        {
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < nt; ++t) pool.emplace_back(parse_piece, cut[t], cut[t + 1], std::ref(parts[t]));
            parse_piece(cut[0], cut[1], parts[0]);
            for (auto& th : pool) th.join();
        }
        std::size_t nvals = 0;
        for (const Part& pt : parts) {
            if (pt.bad) throw std::invalid_argument("stoi");                   // what std::stoi("") throws in the reference
            nvals += pt.vals.size();
        }
        std::vector<value_type> img;
        img.reserve(nvals);
        for (const Part& pt : parts) img.insert(img.end(), pt.vals.begin(), pt.vals.end());
        rgb_img.resize(img.size() / 3);
        for (std::size_t px = 0; px < rgb_img.size(); ++px) rgb_img[px] = { img[3 * px], img[3 * px + 1], img[3 * px + 2] };
        std::cout << "width: " << width << " height: " << height << std::endl;
        if (rgb_img.size() < width * height) initializing_succeed = false;   // the reference would read out of bounds
    }

    // (extension, jpezy_encode --gpus: the batch form needs the planes of several files at once)
    std::size_t image_width() const noexcept { return width; }
    std::size_t image_height() const noexcept { return height; }
    void append_planes(std::vector<std::uint8_t>& r, std::vector<std::uint8_t>& g, std::vector<std::uint8_t>& b) const
    {
        const std::size_t n = width * height;
        for (std::size_t i = 0; i < n; ++i) {
            r.push_back(std::to_integer<std::uint8_t>(rgb_img[i][0]));
            g.push_back(std::to_integer<std::uint8_t>(rgb_img[i][1]));
            b.push_back(std::to_integer<std::uint8_t>(rgb_img[i][2]));
        }
    }

private:
    friend std::ostream& operator<<(std::ostream& os, const encode_io& pnm)   // Mode::PPM / --debug (:104-119)
    {
        pnm.report_error(__func__);
        os << "P3\n" << pnm.width << " " << pnm.height << "\n" << pnm.max_color << "\n";
        write_p3_pixels(os, pnm.rgb_img.size(), [&pnm](std::size_t i) {
            const auto& px = pnm.rgb_img[i];
            return std::array<unsigned, 3>{ std::to_integer<unsigned>(px[0]), std::to_integer<unsigned>(px[1]), std::to_integer<unsigned>(px[2]) };
        });
        return os;
    }

    std::tuple<std::vector<rgb_type>, std::vector<rgb_type>, std::vector<rgb_type>> split_rgb() const   // :121-133
    {
        std::vector<rgb_type> r(rgb_img.size()), g(rgb_img.size()), b(rgb_img.size());
        for (std::size_t i = 0; i < rgb_img.size(); ++i) { r[i] = rgb_img[i][0]; g[i] = rgb_img[i][1]; b[i] = rgb_img[i][2]; }
        return { std::move(r), std::move(g), std::move(b) };
    }

    template <class MODE_TAG>
    std::size_t run_encoder(const char* file, const char* comment) const
    {
        const property pr = make_property({ width, height, 3, 8, comment, property::Format::JFIF, byte(1), byte(2),
                                            property::Units::dots_inch, 96, 96, 0, 0, property::ExtensionCodes::undefined,
                                            property::AnalyzedResult::Yet });
        auto [r, g, b] = split_rgb();
        encoder enc(pr, r, g, b);
        return enc.template encode<MODE_TAG>(file);
    }

    friend std::ofstream& operator<<(std::ofstream& ofs, const std::pair<const to_jpeg, const encode_io&>& pnm)   // :135-169
    {
        ofs.close();
        pnm.second.report_error(__func__);
        const std::size_t size = pnm.second.run_encoder<COLOR_MODE>(pnm.first.file, "Encoded by jpezy");
        std::cout << "Output size: " << size << " byte" << std::endl;
        return ofs;
    }

    friend std::ofstream& operator<<(std::ofstream& ofs, const std::pair<gray_scale_t, std::pair<const to_jpeg, const encode_io&>>& pnm)   // :171-196
    {
        ofs.close();
        pnm.second.second.report_error(__func__);
        const std::size_t size = pnm.second.second.run_encoder<GRAY_MODE>(pnm.second.first.file, "Encoded by JPEZY");
        std::cout << "Output size: " << size << " srook::byte" << std::endl;      // sic (:193)
        return ofs;
    }

    friend std::pair<gray_scale_t, std::pair<const to_jpeg, const encode_io&>>
    operator|(const std::pair<const to_jpeg, const encode_io&>& pnm, const gray_scale_t& gr)
    {
        return { gr, pnm };
    }

    friend std::pair<const to_jpeg, const encode_io&> operator|(const encode_io& pnm, const to_jpeg& jpeg_tag) noexcept
    {
        return { jpeg_tag, pnm };
    }
};

}  // namespace jpezy
#endif
