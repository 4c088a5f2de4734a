// pnm_stream.hpp -- common base of the PPM adaptors (mirrors the reference's src/pnm_stream.hpp).
#ifndef JPEZY_AMD_HOST_PNM_STREAM_HPP
#define JPEZY_AMD_HOST_PNM_STREAM_HPP
#include <array>
#include <stdexcept>
#include <string>
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

    void report_error(const char* func) const
    {
        if (initializing_succeed) return;
        throw std::runtime_error(std::string("Initializing was failed: ") + func);
    }
};

}  // namespace jpezy
#endif
