// pnm_stream.hpp -- common base of the PPM adaptors (mirrors the reference's src/pnm_stream.hpp).
#ifndef JPEZY_AMD_HOST_PNM_STREAM_HPP
#define JPEZY_AMD_HOST_PNM_STREAM_HPP
#include <algorithm>
#include <array>
#include <cstdlib>
#include <functional>
#include <ostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "jpezy.hpp"

namespace jpezy {

struct pnm_stream {
    pnm_stream() = default;
    pnm_stream(bool ok, std::size_t w, std::size_t h, std::size_t max) : initializing_succeed(ok), width(w), height(h), max_color(max) {}
    explicit operator bool() const noexcept { return initializing_succeed; }

protected:
    using value_type = byte;
    using rgb_type = byte;
    bool initializing_succeed = true;
    std::size_t width = 0, height = 0, max_color = 0;
    std::vector<std::array<rgb_type, 3>> rgb_img;

    // n lines "r g b\n" (values 0..255) to os.  A 4096x4096 frame is 187 MB of text: the pixel range is cut into pieces,
    // every host core formats one into its own buffer, the buffers are written in order (JPEZY_IO_THREADS overrides the
    // thread count).  px(i) -> {r, g, b}.
    template <class Px>
    static void write_p3_pixels(std::ostream& os, std::size_t n, Px px)
    {
        unsigned nt = std::thread::hardware_concurrency();
        if (nt == 0) nt = 4;
        if (nt > 16) nt = 16;
        if (const char* e = std::getenv("JPEZY_IO_THREADS")) nt = static_cast<unsigned>(std::max(1, std::atoi(e)));
        if (n < (std::size_t(1) << 20)) nt = 1;
        std::vector<std::string> text(nt);
        auto format = [&px](std::size_t b, std::size_t e, std::string& out) {
            std::string slab;                                   // local: the strings of neighbouring pieces share cache lines
            slab.resize((e - b) * 12);
            char* p = slab.data();
            auto put = [&p](unsigned v) {
                if (v >= 100) *p++ = char('0' + v / 100);
                if (v >= 10) *p++ = char('0' + (v / 10) % 10);
                *p++ = char('0' + v % 10);
            };
            for (std::size_t i = b; i < e; ++i) {
                const std::array<unsigned, 3> v = px(i);
                put(v[0]); *p++ = ' ';
                put(v[1]); *p++ = ' ';
                put(v[2]); *p++ = '\n';
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
        for (unsigned t = 0; t < nt && os; ++t) os.write(text[t].data(), static_cast<std::streamsize>(text[t].size()));
    }

    void report_error(const char* func) const
    {
        if (initializing_succeed) return;
        throw std::runtime_error(std::string("Initializing was failed: ") + func);
    }
};

}  // namespace jpezy
#endif
