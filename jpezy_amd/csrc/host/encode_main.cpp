// jpezy_encode <input.ppm> ( <output.(jpeg | jpg) [OPT: --gray]> | <output.ppm> | --debug )
// Same argv rules, transcript and exit codes as the reference's src/encoder/main.cpp; the codec underneath is
// the MI355X path (jpezy_encoder.hpp).
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string_view>

#include "encode_io.hpp"

namespace {

int disp_error()
{
    std::cerr << "Usage: jpezy_encode <input.ppm> ( <ouput.(jpeg | jpg) [OPT: --gray]> | <output.ppm> | --debug )" << std::endl;
    return EXIT_FAILURE;
}

enum class Mode { JPEG, GRAY, PPM, DEBUG, UD };

// "jpeg"/"jpg"/"ppm" anywhere after the first '.', as the reference's find(pattern, find_first_of('.')) (:71-84)
bool has_ext(std::string_view s, std::string_view ext)
{
    return s.find(ext, s.find_first_of('.')) != std::string_view::npos;
}

}  // namespace

int main(const int argc, const char* argv[])
{
    if (argc < 3) return disp_error();

    Mode m1 = Mode::UD, m2 = Mode::UD;
    const std::string_view sv1 = argv[2];
    const std::string_view sv2 = argc > 3 ? std::string_view(argv[3]) : std::string_view();   // the reference reads argv[3] unguarded

    if (has_ext(sv1, "jpeg") || has_ext(sv1, "jpg")) {
        m1 = Mode::JPEG;
        if (sv2.find("--gray") != std::string_view::npos) m2 = Mode::GRAY;
    } else if (has_ext(sv1, "ppm")) {
        m1 = Mode::PPM;
    } else if (sv1 == "--debug") {
        m1 = Mode::DEBUG;
    } else {
        return disp_error();
    }

    jpezy::disp_logo();

    jpezy::raii_messenger section("Reading the input file...");
    jpezy::encode_io pnm(argv[1]);
    if (!pnm) {
        std::cerr << "The file is not found or the formatting error" << std::endl;
        return disp_error();
    }
    const auto t1 = section.stop();
    section.restart("Start encoding and writing ...");

    try {
        if (m1 == Mode::JPEG) {
            std::ofstream ofs(argv[2], std::ios::binary);
            if (m2 == Mode::GRAY) ofs << (pnm | jpezy::to_jpeg(argv[2]) | jpezy::gray_scale);
            else ofs << (pnm | jpezy::to_jpeg(argv[2]));
        } else if (m1 == Mode::PPM) {
            std::ofstream ofs(argv[2]);
            static_cast<std::ostream&>(ofs) << pnm;
        } else {
            std::cout << pnm << std::endl;
        }
    } catch (const std::runtime_error& e) {
        std::cerr << e.what() << std::endl;
        return EXIT_FAILURE;
    }

    const auto t2 = section.stop();
    if (t1 && t2) std::cout << "Total processing time: " << *t1 + *t2 << std::endl;
    return EXIT_SUCCESS;
}
