// jpezy_encoder.hpp -- jpezy::encoder<T>, same surface as the reference's src/encoder/jpezy_encoder.hpp:22-77.
// encode<MODE_TAG>() splits the reference's per-MCU loop (:58-67) in two stages, both on the MI355X through one
// C-ABI call (jpezy_encode_jpeg): the batched compute stage (make_YCC + DCT + quantization + zig-zag for ALL MCUs),
// then the entropy stage (encode_huffman :174-225 + the bit packer) -- legal because encode_huffman's only
// cross-block state is pre_DC[3], which is the previous block's DC, and the bit cursor, which is a prefix sum of
// code lengths.  The JFIF header and EOI are written by the host.  Output bytes are identical
// to the reference arithmetic.  Define JPEZY_HOST_ENTROPY to keep the serial tail on the host
// (jpezy_fdct_quant + jpezy_write_jpeg): same bytes.
#ifndef JPEZY_AMD_HOST_ENCODER_HPP
#define JPEZY_AMD_HOST_ENCODER_HPP
#include <cstdio>
#include <type_traits>
#include <vector>

#include "jpezy.hpp"

namespace jpezy {

template <class T>
struct encoder {
    static_assert(sizeof(T) == 1, "jpezy::encoder works on 8-bit samples");
    encoder(const property& pr_, const std::vector<T>& r_, const std::vector<T>& g_, const std::vector<T>& b_)
        : pr(pr_), r(r_), g(g_), b(b_)      // the object copies the planes (ref :266) and refers to the property (:265)
    {}

    static constexpr int block_size = 8;

    template <class MODE_TAG = COLOR_MODE>
    std::size_t encode(const char* output_file)
    {
        constexpr bool gray = std::is_same_v<MODE_TAG, GRAY_MODE>;
        const int W = static_cast<int>(pr.template get<property::At::HSize>());
        const int H = static_cast<int>(pr.template get<property::At::VSize>());
        if (W <= 0 || H <= 0 || r.size() < std::size_t(W) * H || g.size() < std::size_t(W) * H || b.size() < std::size_t(W) * H)
            throw std::runtime_error("encode");
        std::FILE* fp = std::fopen(output_file, "wb");
        if (!fp) throw std::runtime_error("write_header");          // jpezy_writer::write_header (:22-23)
        std::vector<std::uint8_t> out(jpezy_jpeg_bound(W, H));
        long n = 0;
        {
            raii_messenger mes("Write JPEG Header ...");          // the header is produced with the entropy data below
        }
        try {
            raii_messenger mes("Encoding ...");
            jpezy_ctx* ctx = detail::device_context();
#ifdef JPEZY_HOST_ENTROPY
            std::vector<std::int16_t> coeffs(jpezy_coeff_count(W, H, gray));
            if (jpezy_fdct_quant(ctx, reinterpret_cast<const std::uint8_t*>(r.data()), reinterpret_cast<const std::uint8_t*>(g.data()),
                                 reinterpret_cast<const std::uint8_t*>(b.data()), W, H, gray, 1, coeffs.data()) != JPEZY_OK)
                throw std::runtime_error(std::string("jpezy_fdct_quant: ") + jpezy_hip_last_error());
            n = jpezy_write_jpeg(coeffs.data(), W, H, gray, pr.template get<property::At::Comment>().c_str(), out.data(), out.size());
#else
            n = jpezy_encode_jpeg(ctx, reinterpret_cast<const std::uint8_t*>(r.data()), reinterpret_cast<const std::uint8_t*>(g.data()),
                                  reinterpret_cast<const std::uint8_t*>(b.data()), W, H, gray,
                                  pr.template get<property::At::Comment>().c_str(), out.data(), out.size());
            if (n < 0 && n != JPEZY_E_FORMAT) throw std::runtime_error(std::string("jpezy_encode_jpeg: ") + jpezy_hip_last_error());
#endif
            if (n < 0) throw std::runtime_error("encode_huffman");   // ref :186-187, 207-208
        } catch (...) {
            std::fclose(fp);
            throw;
        }
        {
            raii_messenger mes("Write EOI ...");
        }
        const std::size_t wrote = std::fwrite(out.data(), 1, static_cast<std::size_t>(n), fp);
        std::fclose(fp);
        if (wrote != static_cast<std::size_t>(n)) throw std::runtime_error("output_file");
        return static_cast<std::size_t>(n);                         // bofstream::wrote_size(), ref :76
    }

private:
    const property& pr;
    const std::vector<T> r, g, b;
};

template <class T>
encoder(const property&, const std::vector<T>&, const std::vector<T>&, const std::vector<T>&) -> encoder<T>;

}  // namespace jpezy
#endif
