// jpezy_decode <input.(jpg | jpeg)> ( <output.ppm | [OPT: --gray]> | -v )
// Same argv rules, transcript and exit codes as the reference's src/decoder/main.cpp.
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string_view>

#include "decode_io.hpp"
#include "jpezy_decoder.hpp"

namespace {

int disp_error()
{
    std::cerr << "Usage: jpezy_decode <input.(jpg | jpeg)> ( <output.ppm | [OPT: --gray]> | -v )" << std::endl;
    return EXIT_FAILURE;
}

bool has_ext(std::string_view s, std::string_view ext)
{
    return s.find(ext, s.find_first_of('.')) != std::string_view::npos;
}

template <class CL, class T>
int output(jpezy::decoder<T>& dec, const char* out)
{
    auto raw_op = dec.template decode<CL>();
    if (!raw_op) {
        std::cerr << "decode failed" << std::endl;
        return EXIT_FAILURE;
    }
    const auto raw = std::move(raw_op.value());
    const auto& [r, g, b] = raw;
    using At = jpezy::property::At;
    jpezy::decode_io dec_io(dec.pr.template get<At::HSize>(), dec.pr.template get<At::VSize>(), r, g, b);
    std::ofstream ofs(out, std::ios_base::out | std::ios_base::trunc);
    ofs << dec_io;
    std::cout << "Decoded image: Netpbm image data, size = " << dec.pr.template get<At::HSize>() << " x "
              << dec.pr.template get<At::VSize>() << ", pixmap, ASCII text" << std::endl;
    return EXIT_SUCCESS;
}

template <class T>
int run(const char* in, const char* out, bool gray)
{
    jpezy::disp_logo();
    jpezy::decoder<T> dec(in);
    return gray ? output<jpezy::GRAY_MODE>(dec, out) : output<jpezy::COLOR_MODE>(dec, out);
}

}  // namespace

int main(const int argc, const char* argv[])
{
    if (argc > 5 || argc < 3) return disp_error();

    const std::string_view sv0 = argv[1], sv1 = argv[2];
    const std::string_view sv2 = argc > 3 ? std::string_view(argv[3]) : std::string_view();
    const std::string_view sv3 = argc > 4 ? std::string_view(argv[4]) : std::string_view();

    if (!((has_ext(sv0, "jpeg") || has_ext(sv0, "jpg")) && has_ext(sv1, "ppm"))) return disp_error();

    const bool gray = sv2.find("--gray") != std::string_view::npos || sv3.find("--gray") != std::string_view::npos;
    const bool verbose = sv2.find("-v") != std::string_view::npos || sv3.find("-v") != std::string_view::npos;
    try {
        return verbose ? run<jpezy::Debug>(argv[1], argv[2], gray) : run<jpezy::Release>(argv[1], argv[2], gray);
    } catch (const std::runtime_error& e) {
        std::cerr << e.what() << std::endl;
        return EXIT_FAILURE;
    }
}
