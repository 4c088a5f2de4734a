// jpezy_decoder.hpp -- jpezy::decoder<BuildMode>, same surface as the reference's src/decoder/jpezy_decoder.hpp:39-136.
// decode<MODE_TAG>() parses the markers on the host (header pass of jpezy_read_jpeg) and hands the file to
// jpezy_decode_jpeg: decode_huffman (:583-642), inverse_quantization + inverse_dct + upsampling (:504-528, 645-670) and
// make_rgb (:531-578) for ALL MCUs -- for jpezy_encode's own layout all on the MI355X (self-synchronising parallel
// Huffman decoder, fused IDCT kernel), for other baseline layouts the host Huffman decoder and the generic kernels.
#ifndef JPEZY_AMD_HOST_DECODER_HPP
#define JPEZY_AMD_HOST_DECODER_HPP
#include <array>
#include <cstdio>
#include <cstring>
#include <memory>
#include <optional>
#include <type_traits>
#include <vector>

#include "jpezy.hpp"

namespace jpezy {

template <class BuildMode = Release>
struct decoder {
    explicit decoder(const char* filename)
    {
        if (std::FILE* fp = std::fopen(filename, "rb")) {           // the reference opens its bifstream here (:68)
            std::fseek(fp, 0, SEEK_END);
            const long n = std::ftell(fp);
            std::fseek(fp, 0, SEEK_SET);
            if (n > 0) {
                file.resize(static_cast<std::size_t>(n));
                if (std::fread(file.data(), 1, file.size(), fp) != file.size()) file.clear();
            }
            std::fclose(fp);
        }
    }

    static constexpr bool is_release_mode = std::is_same_v<Release, BuildMode>;
    static constexpr std::size_t rgb_size = 3, block_size = 8, blocks_size = 64, mcu_size = 4;

    template <class MODE_TAG = COLOR_MODE>
    std::optional<std::array<std::vector<byte>, 3>> decode()
    {
        constexpr bool gray = std::is_same_v<MODE_TAG, GRAY_MODE>;
        raii_messenger mes("process started...");
        std::cout << '\n';

        jpezy_frame_info info;
        {
            std::unique_ptr<raii_messenger> hm;
            if constexpr (!is_release_mode) hm = std::make_unique<raii_messenger>("analyzing header...", "\t");
            if constexpr (!is_release_mode) trace_markers();
            if (jpezy_read_jpeg(file.data(), file.size(), &info, nullptr, 0) != JPEZY_OK) return {};   // :82-86
        }
        pr = property{ static_cast<std::size_t>(info.width), static_cast<std::size_t>(info.height), info.ncomp, info.precision,
                       info.comment, info.format == 1 ? property::Format::JFIF : info.format == 2 ? property::Format::JFXX : property::Format::undefined,
                       byte(info.major_rev), byte(info.minor_rev),
                       info.units == 1 ? property::Units::dots_inch : info.units == 2 ? property::Units::dots_cm : property::Units::undefined,
                       info.hdensity, info.vdensity, 0, 0, property::ExtensionCodes::undefined,
                       property::is_htable | property::is_qtable | property::is_start_data | (info.format ? property::is_jfif : 0) |
                           (info.comment[0] ? property::is_comment : 0) };
        disp_info("\t");

        std::unique_ptr<raii_messenger> mes_dec;
        if constexpr (!is_release_mode) mes_dec = std::make_unique<raii_messenger>("decoding started...", "\t");

        // the reference sizes its planes to the padded MCU grid (:94-101); the first W*H entries are the image
        const std::size_t rgb_s = static_cast<std::size_t>(info.mcu_rows) * info.vmax * 8 * static_cast<std::size_t>(info.mcu_cols) * info.hmax * 8;
        std::array<std::vector<byte>, 3> rgb;
        for (auto& v : rgb) v.resize(rgb_s);
        jpezy_ctx* ctx = detail::device_context();
        // decode_huffman + inverse_quantization + inverse_dct + upsampling + make_rgb (:504-578, 583-670) for all MCUs: one
        // C-ABI call; for jpezy_encode's own layout every stage runs on the device
        const int rc = jpezy_decode_jpeg(ctx, reinterpret_cast<const std::uint8_t*>(file.data()), file.size(), gray, &info,
                                         reinterpret_cast<std::uint8_t*>(rgb[0].data()), reinterpret_cast<std::uint8_t*>(rgb[1].data()),
                                         reinterpret_cast<std::uint8_t*>(rgb[2].data()), rgb_s);
        if (rc == JPEZY_E_FORMAT || rc == JPEZY_E_NOSPACE) {
            std::cerr << "decode_mcu(): throw exception from " << jpezy_hip_last_error() << std::endl;   // :109-114
            return {};
        }
        if (rc != JPEZY_OK) {
            std::cerr << "make_rgb(): throw exception from " << jpezy_hip_last_error() << std::endl;
            return {};
        }
        return { std::move(rgb) };
    }

    property pr;

private:
    void disp_info(const char* indent = "")   // ref :139-150 (spelling as in the reference)
    {
        using At = property::At;
        std::cout << indent << "Loaded JPEG: " << pr.get<At::HSize>() << "x" << pr.get<At::VSize>() << ", "
                  << "presicion " << pr.get<At::SamplePrecision>() << ", "
                  << "\"" << pr.get<At::Comment>() << "\", "
                  << (pr.get<At::Format>() == property::Format::JFIF ? "JFIF" : pr.get<At::Format>() == property::Format::JFXX ? "JFXX" : "undefined")
                  << " standart " << std::to_integer<unsigned>(pr.get<At::MajorRevisions>()) << ".0"
                  << std::to_integer<unsigned>(pr.get<At::MinorRevisions>()) << ", "
                  << (pr.get<At::Units>() == property::Units::dots_inch ? "dots inch" : pr.get<At::Units>() == property::Units::dots_cm ? "dots cm" : "undefined")
                  << ", frames " << pr.get<At::Dimension>() << ", density " << pr.get<At::HDensity>() << "x" << pr.get<At::VDensity>()
                  << "\n" << std::endl;
    }

    // -v: the per-marker trace of decoder<Debug> (ref :369-423), from a walk over the segment headers
    void trace_markers() const
    {
        std::size_t p = 0;
        const std::size_t n = file.size();
        auto seg_len = [&](std::size_t at) { return at + 3 < n ? (std::size_t(file[at + 2]) << 8) | file[at + 3] : 0; };
        while (p + 1 < n) {
            if (file[p] != 0xFF || file[p + 1] == 0x00 || file[p + 1] == 0xFF) { ++p; continue; }
            const unsigned m = file[p + 1];
            const char* name = m == 0xE0 ? "APP0" : m == 0xFE ? "COM" : m == 0xDB ? "DQT" : m == 0xC4 ? "DHT" : m == 0xC0 ? "SOF0" :
                               m == 0xDA ? "SOS" : m == 0xDD ? "DRI" : m == 0xDC ? "DNL" : nullptr;
            if (m == 0xD8) { p += 2; continue; }
            if (name) std::cout << (m == 0xE0 ? "\n" : "") << "\t\tfound marker: [" << name << "]" << std::endl;
            if (m == 0xDA) break;
            p += 2 + seg_len(p);
        }
    }

    std::vector<std::uint8_t> file;
};

}  // namespace jpezy
#endif
