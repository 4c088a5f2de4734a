// CPU harness for the core of the GPU Huffman decoder (jpezy_amd/csrc/jpezy_huffdec_core.h: the two lookup tables, their builder and
// the one-symbol decode step the kernels run) -- built by g++ with AddressSanitizer + UBSan and driven by tests/fuzz/run_host_fuzz.py;
// no HIP involved.  Every file named on the command line (or listed in @file) that the host parser accepts and that has no restart
// intervals is decoded TWICE: by the host decoder (jpezy_host::read_jpeg, the authoritative one) and by walking the unstuffed scan
// with decode_step from the known start state, block after block, followed by the DC prefix sums -- what the device does, minus the
// parallelism.  The walk may decline (an invalid code, a block that runs past 63, a last block that ends behind the data: the GPU path
// then hands the file to the host decoder); when it does not, the host decoder must accept the file too and every coefficient must be
// equal.  Files written by an encoder (names starting with "seed") must not be declined.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../jpezy_amd/csrc/jpezy_host_codec.h"
#include "../../jpezy_amd/csrc/jpezy_huffdec_core.h"

namespace HD = jpezy_dev::huffdec;

static std::vector<unsigned char> slurp(const char* path)
{
    std::ifstream f(path, std::ios::binary);
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// the cursor of the walk: bits of a byte vector, most significant first, zeros behind its end (the device reads its window the same way)
struct HostCursor {
    const std::vector<unsigned char>* u;
    unsigned pos;
    uint32_t peek32() const
    {
        uint64_t v = 0;
        const size_t b0 = pos >> 3;
        for (size_t i = 0; i < 5; ++i) v = (v << 8) | (b0 + i < u->size() ? (*u)[b0 + i] : 0u);
        return (uint32_t)(v >> (8 - (pos & 7u)));
    }
    uint32_t prefetch() const { return 0; }
    void advance(unsigned np, uint32_t) { pos = np; }
};

// Every 16-bit window of the stream against the canonical code itself (the loop of ITU T.81 F.2.2.3 over bits / vals): the entry the
// step would pick -- l1 by the first L1_BITS bits, hi by the last HI_BITS when the window starts with 16 - HI_BITS ones -- must carry the
// code's length, value-bit count and run, and E_NOT_A_CODE where no code matches.  A table the builder declines must have a code longer
// than L1_BITS bits below the hi region (or counts that are no prefix code).  Returns false on any mismatch.
static bool check_table(const HD::Table& t, bool built, const uint8_t bits[16], const uint8_t* vals, bool dc, const char* what)
{
    unsigned first[17], code = 0, at[17];
    int p = 0;
    bool prefix_code = true, long_low = false;
    for (int l = 1; l <= 16; ++l) {
        first[l] = code; at[l] = (unsigned)p;
        if (code + bits[l - 1] > (1u << l)) prefix_code = false;
        if (l > HD::L1_BITS && bits[l - 1] && (code << (16 - l)) < HD::HI_FIRST) long_low = true;
        code = (code + bits[l - 1]) << 1;
        p += bits[l - 1];
    }
    if (!built) {
        if (prefix_code && !long_low) { std::fprintf(stderr, "%s: the builder declined a table it can express\n", what); return false; }
        return true;
    }
    if (!prefix_code || long_low) { std::fprintf(stderr, "%s: the builder accepted a table it cannot express\n", what); return false; }
    for (unsigned w = 0; w < 0x10000u; ++w) {
        unsigned want = HD::E_NOT_A_CODE;
        for (int l = 1; l <= 16; ++l) {
            const unsigned c = w >> (16 - l);
            if (c >= first[l] && c < first[l] + bits[l - 1]) {
                const unsigned sym = vals[at[l] + (c - first[l])];
                if (dc) { if (sym <= 16) want = HD::make_entry((unsigned)l, sym, 0u); }
                else want = HD::make_entry((unsigned)l, sym & 15u, sym == 0 ? HD::E_RUN_END : sym >> 4);
                break;
            }
        }
        const unsigned got = w >= HD::HI_FIRST ? t.hi[w - HD::HI_FIRST] : t.l1[w >> (16 - HD::L1_BITS)];
        if (got != want) { std::fprintf(stderr, "%s: window %04x decodes to %04x, the canonical code says %04x\n", what, w, got, want); return false; }
    }
    return true;
}

int main(int argc, char** argv)
{
    std::vector<std::string> files;
    for (int i = 1; i < argc; ++i) {
        if (argv[i][0] == '@') {
            std::ifstream l(argv[i] + 1);
            for (std::string s; std::getline(l, s);) if (!s.empty()) files.push_back(s);
        } else files.push_back(argv[i]);
    }
    size_t walked = 0, declined = 0, skipped = 0, host_rejected_after_decline = 0, tables_checked = 0;
    for (const std::string& path : files) {
        const std::vector<unsigned char> d = slurp(path.c_str());
        const bool is_seed = path.find("/seed") != std::string::npos;
        jpezy_frame_info info;
        std::memset(&info, 0, sizeof info);
        jpezy_host::ScanSetup setup;
        std::string err;
        if (jpezy_host::parse_header(d.data(), d.size(), &info, &setup, &err) < 0) { ++skipped; continue; }
        const size_t nmcu = (size_t)info.mcu_cols * info.mcu_rows;
        const int bpm = info.blocks_per_mcu;
        const size_t total_blocks = nmcu * (size_t)bpm, ncoef = total_blocks * 64;
        // what jpezy_read_jpeg_gpu sends to the subsequence kernels (jpezy_capi.hip): the same admission rules
        bool takes = info.restart_interval == 0 && bpm >= 1 && bpm <= 48 && ncoef > 0 && ncoef <= (size_t(1) << 26) && setup.scan_pos < d.size() &&
                     total_blocks / 4 <= d.size();
        for (int i = 0; i < info.ncomp && takes; ++i)
            takes = setup.Td[i] >= 0 && setup.Td[i] <= 2 && setup.present[setup.Td[i]] && setup.present[4 + setup.Td[i]];
        if (!takes) { ++skipped; continue; }
        std::vector<HD::Setup> hs(1);
        HD::Setup& S = hs[0];
        std::memset(&S, 0, sizeof S);
        bool tables = true;
        for (int td = 0; td < 3; ++td) {
            if (setup.present[td]) {
                const bool b = HD::build_dev_table(S.dc[td], setup.bits[td], setup.vals[td], setup.nvals[td], true);
                if (!check_table(S.dc[td], b, setup.bits[td], setup.vals[td], true, path.c_str())) return 1;
                tables = b && tables;
            }
            if (setup.present[4 + td]) {
                const bool b = HD::build_dev_table(S.ac[td], setup.bits[4 + td], setup.vals[4 + td], setup.nvals[4 + td], false);
                if (!check_table(S.ac[td], b, setup.bits[4 + td], setup.vals[4 + td], false, path.c_str())) return 1;
                tables = b && tables;
            }
            tables_checked += (setup.present[td] ? 1 : 0) + (setup.present[4 + td] ? 1 : 0);
        }
        int seq[48], nb = 0;
        for (int q = 0; q < info.ncomp; ++q)
            for (int t = info.H[q] * info.V[q]; t > 0 && nb < 48; --t) seq[nb++] = setup.Td[q];
        int period = nb;
        for (int pd = 1; pd < nb; ++pd) {
            if (nb % pd) continue;
            bool same = true;
            for (int i = pd; i < nb && same; ++i) same = seq[i] == seq[i - pd];
            if (same) { period = pd; break; }
        }
        unsigned tdmask = 0;
        if (!tables || !HD::pack_td_sequence(seq, period, &tdmask)) { ++skipped; continue; }

        // the entropy-coded segment and its unstuffed form
        const unsigned char* scan = d.data() + setup.scan_pos;
        const size_t n = jpezy_host::entropy_segment_length(scan, d.size() - setup.scan_pos);
        std::vector<unsigned char> U;
        U.reserve(n);
        for (size_t i = 0; i < n; ++i) {
            U.push_back(scan[i]);
            if (scan[i] == 0xFF && i + 1 < n && scan[i + 1] == 0x00) ++i;
        }
        // the walk
        std::vector<int16_t> co(ncoef, 0);
        const uint16_t* tabs = reinterpret_cast<const uint16_t*>(&S);
        HostCursor c{ &U, 0 };
        HD::Walk wk;
        wk.init(0, 0, tdmask);
        bool ok = true;
        const unsigned long long limit = (unsigned long long)U.size() * 8 + 64;
        while (ok && wk.nblocks < total_blocks) {
            ok = HD::decode_step<true>(tabs, (unsigned)period, tdmask, c, wk, 0ull, (unsigned)total_blocks, co.data());
            if (c.pos > limit) ok = false;
        }
        ok = ok && c.pos <= U.size() * 8;                       // the last block is complete inside the data
        std::vector<int16_t> want(ncoef);
        jpezy_frame_info info2;
        std::memset(&info2, 0, sizeof info2);
        const int rc = jpezy_host::read_jpeg(d.data(), d.size(), &info2, want.data(), want.size(), &err);
        if (!ok) {
            ++declined;
            if (rc < 0) ++host_rejected_after_decline;
            if (is_seed) { std::fprintf(stderr, "%s: an encoder's file was declined by the walk\n", path.c_str()); return 1; }
            continue;
        }
        if (rc < 0) { std::fprintf(stderr, "%s: the walk decoded a file the host decoder rejects (%s)\n", path.c_str(), err.c_str()); return 1; }
        // DC differences -> values per component (pre_DC)
        int pred[3] = { 0, 0, 0 };
        for (size_t blk = 0; blk < total_blocks; ++blk) {
            const int b = (int)(blk % (size_t)bpm);
            int comp = 0, at = 0;
            for (int q = 0; q < info.ncomp; ++q) { if (b >= at) comp = q; at += info.H[q] * info.V[q]; }
            pred[comp] += co[blk * 64];
            co[blk * 64] = (int16_t)pred[comp];
        }
        if (std::memcmp(co.data(), want.data(), ncoef * sizeof(int16_t)) != 0) {
            std::fprintf(stderr, "%s: coefficients differ from the host decoder's\n", path.c_str());
            return 1;
        }
        ++walked;
    }
    // random canonical codes (not only the ones files carry): counts per length drawn so that they stay a prefix code, random symbols
    size_t random_tables = 0, random_declined = 0;
    {
        uint64_t rs = 0x6A70657A79ull;
        auto rnd = [&rs](unsigned n) { rs = rs * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)((rs >> 33) % n); };
        for (int it = 0; it < 3000; ++it) {
            uint8_t bits[16], vals[256];
            unsigned space = 0x10000u - 1u, total = 0;                      // (the all-ones code stays free, as JPEG asks)
            const int lmin = 1 + (int)rnd(4);
            for (int l = 1; l <= 16; ++l) {
                const unsigned unit = 1u << (16 - l), room = space / unit;
                unsigned n = 0;
                if (l >= lmin && room) n = rnd(room < 24 ? room + 1 : 24);
                if (l == 16) n = room < 255u - total ? room : 255u - total;  // fill what is left at the bottom
                if (total + n > 255) n = 255 - total;
                bits[l - 1] = (uint8_t)n; total += n; space -= n * unit;
            }
            const bool dc = (it & 3) == 0;
            for (unsigned k = 0; k < total; ++k) vals[k] = (uint8_t)(dc ? rnd(20) : rnd(256));
            HD::Table t;
            const bool b = HD::build_dev_table(t, bits, vals, (int)total, dc);
            if (!check_table(t, b, bits, vals, dc, "random table")) return 1;
            ++random_tables;
            random_declined += b ? 0 : 1;
        }
    }
    std::printf("decode step on the CPU: %zu files walked and equal to the host decoder, %zu declined (%zu of them rejected by the host decoder too), %zu not for this path; "
                "%zu tables of files and %zu random ones (%zu declined by the builder) checked window by window against their canonical codes\n",
                walked, declined, host_rejected_after_decline, skipped, tables_checked, random_tables, random_declined);
    return 0;
}
