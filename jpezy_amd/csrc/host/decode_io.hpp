// decode_io.hpp -- jpezy::decode_io<Range>: planes -> ASCII PPM (P3), mirrors src/decoder/decode_io.hpp:27-57.
#ifndef JPEZY_AMD_HOST_DECODE_IO_HPP
#define JPEZY_AMD_HOST_DECODE_IO_HPP
#include <algorithm>
#include <array>
#include <ostream>
#include <string>

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
        // "r g b\n" per pixel, first width*height entries of the planes (:45-52)
        const std::size_t n = std::min<std::size_t>(io.width * io.height, io.r_.size());
        write_p3_pixels(ofs, n, [&io](std::size_t i) {
            return std::array<unsigned, 3>{ std::to_integer<unsigned>(io.r_[i]), std::to_integer<unsigned>(io.g_[i]),
                                            std::to_integer<unsigned>(io.b_[i]) };
        });
        return ofs;
    }

private:
    const Range &r_, &g_, &b_;
};

template <class Range>
decode_io(std::size_t, std::size_t, const Range&, const Range&, const Range&) -> decode_io<Range>;

}  // namespace jpezy
#endif
