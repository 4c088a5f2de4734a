// jpezy.hpp -- shared core of the host-side class surface (namespace jpezy), mirroring the reference's
// src/jpezy.hpp: mode tags, markers, the Annex-K tables, `property` and its accessors, the timing messenger
// and the logo.  srook:: vocabulary types become std:: ones (SrookCppLibraries is not available) and
// Boost.Parameter's named arguments become a plain aggregate (`property_args`).
#ifndef JPEZY_AMD_HOST_JPEZY_HPP
#define JPEZY_AMD_HOST_JPEZY_HPP

#include <array>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <iostream>
#include <optional>
#include <stdexcept>
#include <string>

#include "jpezy_constants.h"
#include "jpezy_hip.h"

namespace jpezy {

using byte = std::byte;

inline void disp_logo()   // ref jpezy.hpp:20-29
{
    static const char* const rows[] = {
        "   _", "  (_)_ __   ___ _____   _", "  | | '_ \\ / _ \\_  / | | | ", "  | | |_) |  __// /| |_| |",
        " _/ | .__/ \\___/___|\\__, |", "|__/|_|             |___/\tby roki",
    };
    for (const char* r : rows) std::cout << r << "\n";
    std::cout << std::endl;
}

// compile-time tags (ref jpezy.hpp:31-34)
struct Release;
struct Debug;
struct COLOR_MODE;
struct GRAY_MODE;

inline constexpr std::array<int, 64> ZZ = JPEZY_ZZ_INIT;                  // ref :36-45
inline constexpr std::array<int, 64> YQuantumTb = JPEZY_QT_LUMA_INIT;     // Annex K.1, ref :131-140
inline constexpr std::array<int, 64> CQuantumTb = JPEZY_QT_CHROMA_INIT;   // Annex K.2, ref :143-152

enum class MARKER : unsigned char {   // the markers the codec names (ref :47-127)
    SOF0 = 0xc0, DHT = 0xc4, RST0 = 0xd0, RST7 = 0xd7, SOI = 0xd8, EOI = 0xd9, SOS = 0xda, DQT = 0xdb,
    DNL = 0xdc, DRI = 0xdd, APP0 = 0xe0, COM = 0xfe, Marker = 0xff
};

struct property {   // ref :154-342
    enum class Format { undefined, JFIF, JFXX };
    enum class Units { undefined, dots_inch, dots_cm };
    enum class ExtensionCodes { undefined = 0, JPEG = 0x10, oneByte_pixel = 0x11, threeByte_pixel = 0x13 };
    enum AnalyzedResult { Yet = 0, is_htable = 0x01, is_qtable = 0x02, is_jfif = 0x04, is_comment = 0x08, is_start_data = 0x10 };
    enum class At {
        HSize, VSize, Dimension, SamplePrecision, Comment, Format, MajorRevisions, MinorRevisions, Units,
        HDensity, VDensity, HThumbnail, VThumbnail, ExtensionCode, Decodable, ELEMENT_SIZE
    };

    std::size_t width = 0, height = 0;
    int dimension = 0, sample_precision = 0;
    std::string comment;
    Format format = Format::undefined;
    byte major_rev{ 0 }, minor_rev{ 0 };
    Units uni = Units::undefined;
    int width_density = 1, height_density = 1, width_thumbnail = 0, height_thumbnail = 0;
    ExtensionCodes ext = ExtensionCodes::undefined;
    int decodable = Yet;

    property() = default;
    property(std::size_t w, std::size_t h, int dim, int sample_pre, std::string com, Format form, byte marev, byte mirev,
             Units u, int wd, int hd, int wt, int ht, ExtensionCodes e, int decflag = Yet)
        : width(w), height(h), dimension(dim), sample_precision(sample_pre), comment(std::move(com)), format(form),
          major_rev(marev), minor_rev(mirev), uni(u), width_density(wd), height_density(hd), width_thumbnail(wt),
          height_thumbnail(ht), ext(e), decodable(decflag)
    {}

    template <At at>
    const auto& get() const noexcept { return get_impl<at>(*this); }
    template <At at>
    auto& get() noexcept { return get_impl<at>(*this); }
    template <std::size_t at>
    const auto& get() const noexcept { return get_impl<static_cast<At>(at)>(*this); }

private:
    template <At at, class Self>
    static auto& get_impl(Self& s) noexcept
    {
        if constexpr (at == At::HSize) return s.width;
        else if constexpr (at == At::VSize) return s.height;
        else if constexpr (at == At::Dimension) return s.dimension;
        else if constexpr (at == At::SamplePrecision) return s.sample_precision;
        else if constexpr (at == At::Comment) return s.comment;
        else if constexpr (at == At::Format) return s.format;
        else if constexpr (at == At::MajorRevisions) return s.major_rev;
        else if constexpr (at == At::MinorRevisions) return s.minor_rev;
        else if constexpr (at == At::Units) return s.uni;
        else if constexpr (at == At::HDensity) return s.width_density;
        else if constexpr (at == At::VDensity) return s.height_density;
        else if constexpr (at == At::HThumbnail) return s.width_thumbnail;
        else if constexpr (at == At::VThumbnail) return s.height_thumbnail;
        else if constexpr (at == At::ExtensionCode) return s.ext;
        else return s.decodable;
    }
};

// stands in for the reference's Boost.Parameter pack (ref :346-386): designated initialisers give the same
// "named argument" call sites, e.g. make_property({.width = w, .height = h, .dimension = 3, ...})
struct property_args {
    std::size_t width = 0, height = 0;
    int dimension = 0, sample_precision = 0;
    std::string comment;
    property::Format format = property::Format::undefined;
    byte major_rev{ 0 }, minor_rev{ 0 };
    property::Units units = property::Units::undefined;
    int width_density = 1, height_density = 1, width_thumbnail = 0, height_thumbnail = 0;
    property::ExtensionCodes extension_code = property::ExtensionCodes::undefined;
    int decodable = property::AnalyzedResult::Yet;
};
inline property make_property(const property_args& a)
{
    return property{ a.width, a.height, a.dimension, a.sample_precision, a.comment, a.format, a.major_rev, a.minor_rev,
                     a.units, a.width_density, a.height_density, a.width_thumbnail, a.height_thumbnail, a.extension_code,
                     a.decodable };
}

// "message ... Done! Processing time: X(sec)" with millisecond resolution (ref :388-432)
struct raii_messenger {
    explicit raii_messenger(const char* message, const char* ind = "") : mes(message), indent(ind)
    {
        std::cout << indent << mes << " ";
        start = std::chrono::system_clock::now();
    }
    void restart(const char* str = nullptr)
    {
        if (!stoped) return;
        if (str) std::cout << str << std::endl; else std::cout << mes << " ";
        start = std::chrono::system_clock::now();
        stoped = false;
    }
    std::optional<float> stop()
    {
        if (stoped) return std::nullopt;
        const auto end = std::chrono::system_clock::now();
        const float time = static_cast<float>(std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count()) / 1000;
        std::cout << indent << "Done! Processing time: " << time << "(sec)" << std::endl;
        stoped = true;
        return time;
    }
    ~raii_messenger() { stop(); }

private:
    std::chrono::system_clock::time_point start;
    const char *mes, *indent;
    bool stoped = false;
};

// One GPU context per process for the class surface (the C-ABI allows one per GPU; JPEZY_DEVICE picks it).
namespace detail {
struct ctx_holder {
    jpezy_ctx* ctx = nullptr;
    ~ctx_holder() { if (ctx) jpezy_ctx_destroy(ctx); }
};
inline jpezy_ctx* device_context()
{
    static ctx_holder h;
    if (!h.ctx) {
        int dev = 0;
        if (const char* e = std::getenv("JPEZY_DEVICE")) dev = std::atoi(e);
        h.ctx = jpezy_ctx_create(dev);
        if (!h.ctx) throw std::runtime_error(std::string("jpezy: ") + jpezy_hip_last_error());
    }
    return h.ctx;
}
}  // namespace detail

}  // namespace jpezy
#endif
