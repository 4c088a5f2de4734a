// decode_io.hpp -- jpezy::decode_io<Range>: planes -> ASCII PPM (P3), mirrors src/decoder/decode_io.hpp:27-57.
#ifndef JPEZY_AMD_HOST_DECODE_IO_HPP
#define JPEZY_AMD_HOST_DECODE_IO_HPP
#include <algorithm>
#include <cstdlib>
#include <functional>
#include <ostream>
#include <string>
#include <thread>
#include <vector>

#include "pnm_stream.hpp"

namespace jpezy {

template <class Range>
struct decode_io : pnm_stream {
    decode_io(std::size_t w, std::size_t h, const Range& r, const Range& g, const Range& b)
        : pnm_stream(true, w, h, 255), r_(r), g_(g), b_(b)
    {
        if (!(r.size() == g.size() && g.size() == b.size())) initializing_succeed = false;
    }

    friend std::ostream& operator<<(std::ostream& ofs, const decode_io& io)
    {
        if (!io.initializing_succeed) io.report_error(__func__);
        ofs << "P3\n# Decoded by jpezy\n" << io.width << " " << io.height << "\n" << io.max_color << "\n";
        // "r g b\n" per pixel, first width*height entries of the planes (:45-52).  A 4096x4096 frame is 187 MB of text:
        // the pixel range is cut into pieces, every host core formats one into its own buffer, the buffers are written
        // in order (JPEZY_IO_THREADS overrides the thread count).
        const std::size_t n = std::min<std::size_t>(io.width * io.height, io.r_.size());
        unsigned nt = std::thread::hardware_concurrency();
        if (nt == 0) nt = 4;
        if (nt > 16) nt = 16;
        if (const char* e = std::getenv("JPEZY_IO_THREADS")) nt = static_cast<unsigned>(std::max(1, std::atoi(e)));
        if (n < (std::size_t(1) << 20)) nt = 1;
        std::vector<std::string> text(nt);
        auto format = [&io](std::size_t b, std::size_t e, std::string& out) {
            std::string slab;                                   // local: the strings of neighbouring pieces share cache lines
            slab.resize((e - b) * 12);
            char* p = slab.data();
            auto put = [&p](unsigned v) {
                if (v >= 100) *p++ = char('0' + v / 100);
                if (v >= 10) *p++ = char('0' + (v / 10) % 10);
                *p++ = char('0' + v % 10);
            };
            for (std::size_t i = b; i < e; ++i) {
                put(std::to_integer<unsigned>(io.r_[i])); *p++ = ' ';
                put(std::to_integer<unsigned>(io.g_[i])); *p++ = ' ';
                put(std::to_integer<unsigned>(io.b_[i])); *p++ = '\n';
            }
            slab.resize(static_cast<std::size_t>(p - slab.data()));
            out = std::move(slab);
        };
        {
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < nt; ++t) pool.emplace_back(format, n * t / nt, n * (t + 1) / nt, std::ref(text[t]));
            format(0, n / nt, text[0]);
            for (auto& th : pool) th.join();
        }
        for (unsigned t = 0; t < nt && ofs; ++t) ofs.write(text[t].data(), static_cast<std::streamsize>(text[t].size()));
        return ofs;
    }

private:
    const Range &r_, &g_, &b_;
};

template <class Range>
decode_io(std::size_t, std::size_t, const Range&, const Range&, const Range&) -> decode_io<Range>;

}  // namespace jpezy
#endif
