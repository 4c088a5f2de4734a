// jpezy_encode <input.ppm> ( <output.(jpeg | jpg) [OPT: --gray]> | <output.ppm> | --debug )
// Same argv rules, transcript and exit codes as the reference's src/encoder/main.cpp; the codec underneath is
// the MI355X path (jpezy_encoder.hpp).
// Extension (not in the reference):  jpezy_encode --gpus N [--gray] <in1.ppm> <out1.jpg> [<in2.ppm> <out2.jpg> ...]
// encodes a list of files on up to N GPUs of this node through jpezy_encode_batch_multi: runs of consecutive inputs of one size
// form a batch, a batch is sharded over the GPUs frame by frame.
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string_view>

#include "encode_io.hpp"

#include <cstdint>
#include <vector>

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

// jpezy_encode --gpus N [--gray] in out [in out ...]
int batch_main(const int argc, const char* argv[])
{
    const int want = std::atoi(argv[2]);
    int a = 3;
    bool gray = false;
    if (a < argc && std::string_view(argv[a]) == "--gray") { gray = true; ++a; }
    if (want <= 0 || a >= argc || (argc - a) % 2 != 0) {
        std::cerr << "Usage: jpezy_encode --gpus N [--gray] <in1.ppm> <out1.jpg> [<in2.ppm> <out2.jpg> ...]" << std::endl;
        return EXIT_FAILURE;
    }
    const int have = jpezy_hip_device_count();
    if (have <= 0) { std::cerr << "jpezy_encode: no HIP device (the jpezy hot path has no CPU fallback)" << std::endl; return EXIT_FAILURE; }
    std::vector<int> devices;
    for (int d = 0; d < want && d < have; ++d) devices.push_back(d);
    jpezy::disp_logo();
    const int n_files = (argc - a) / 2;
    int f = 0;
    jpezy_ctx* single = nullptr;
    while (f < n_files) {
        std::vector<std::uint8_t> r, g, b;
        std::size_t W = 0, H = 0;
        int n = 0;
        for (; f + n < n_files; ++n) {                      // a run of consecutive inputs of one size
            jpezy::encode_io pnm(argv[a + 2 * (f + n)]);
            if (!pnm) { std::cerr << "The file is not found or the formatting error" << std::endl; return EXIT_FAILURE; }
            if (n == 0) { W = pnm.image_width(); H = pnm.image_height(); }
            else if (pnm.image_width() != W || pnm.image_height() != H) break;
            pnm.append_planes(r, g, b);
        }
        const std::size_t stride = jpezy_jpeg_bound(static_cast<int>(W), static_cast<int>(H));
        std::vector<std::uint8_t> jpg(stride * static_cast<std::size_t>(n));
        std::vector<long long> sizes(static_cast<std::size_t>(n));
        const char* comment = gray ? "Encoded by JPEZY" : "Encoded by jpezy";
        if (n == 1) {
            // a run of one frame has nothing to shard: the ordinary single-file path (streams the frame band by band, no ring of
            // whole-frame slots: a 16K x 16K file would otherwise reserve several GB of pinned memory for nothing)
            if (!single) single = jpezy_ctx_create(devices[0]);
            if (!single) { std::cerr << "jpezy_ctx_create: " << jpezy_hip_last_error() << std::endl; return EXIT_FAILURE; }
            sizes[0] = jpezy_encode_jpeg(single, r.data(), g.data(), b.data(), static_cast<int>(W), static_cast<int>(H), gray ? 1 : 0, comment, jpg.data(), stride);
            if (sizes[0] < 0) { std::cerr << "jpezy_encode_jpeg: " << jpezy_hip_last_error() << std::endl; jpezy_ctx_destroy(single); return EXIT_FAILURE; }
        } else {
            jpezy_multi_out out{ nullptr, jpg.data(), stride, sizes.data(), 0 };
            const int rc = jpezy_encode_batch_multi(devices.data(), static_cast<int>(devices.size()), r.data(), g.data(), b.data(), static_cast<int>(W),
                                                    static_cast<int>(H), gray ? 1 : 0, n, 0, comment, &out);
            if (rc != JPEZY_OK) { std::cerr << "jpezy_encode_batch_multi: " << jpezy_hip_last_error() << std::endl; if (single) jpezy_ctx_destroy(single); return EXIT_FAILURE; }
        }
        for (int i = 0; i < n; ++i) {
            const char* name = argv[a + 2 * (f + i) + 1];
            std::ofstream ofs(name, std::ios::binary);
            ofs.write(reinterpret_cast<const char*>(jpg.data() + stride * static_cast<std::size_t>(i)), static_cast<std::streamsize>(sizes[static_cast<std::size_t>(i)]));
            if (!ofs) { std::cerr << "output_file" << std::endl; return EXIT_FAILURE; }
            std::cout << name << ": Output size: " << sizes[static_cast<std::size_t>(i)] << " byte" << std::endl;
        }
        f += n;
    }
    if (single) jpezy_ctx_destroy(single);
    std::cout << "Encoded " << n_files << " file(s) on " << devices.size() << " GPU(s)" << std::endl;
    return EXIT_SUCCESS;
}

}  // namespace

int main(const int argc, const char* argv[])
{
    if (argc >= 3 && std::string_view(argv[1]) == "--gpus") return batch_main(argc, argv);
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
