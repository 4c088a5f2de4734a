// ASan/UBSan harness for the P3 reader of the CLI (jpezy::encode_io): every file named on the command line is parsed;
// exceptions are what the reference throws too (std::stoi), out-of-bounds accesses are what this looks for.
#include <cstdio>
#include <exception>
#include <fstream>
#include <string>
#include <vector>

#include "../../jpezy_amd/csrc/host/encode_io.hpp"

int main(int argc, char** argv)
{
    std::vector<std::string> files;
    for (int i = 1; i < argc; ++i) {
        if (argv[i][0] == '@') {
            std::ifstream l(argv[i] + 1);
            for (std::string s; std::getline(l, s);) if (!s.empty()) files.push_back(s);
        } else files.push_back(argv[i]);
    }
    size_t ok = 0, bad = 0, threw = 0;
    std::fclose(stdout);                                   // the reader prints "width: .. height: .."
    for (const std::string& f : files) {
        try {
            jpezy::encode_io io(f.c_str());
            if (io) ++ok; else ++bad;
        } catch (const std::exception&) {
            ++threw;
        }
    }
    std::fprintf(stderr, "%zu files: %zu parsed, %zu refused, %zu threw\n", files.size(), ok, bad, threw);
    return 0;
}
