// pack_fuzz.cpp -- host-side ingest logic (libgwaspp_amd/csrc/gw_pack.cpp) under
// AddressSanitizer + UBSan: random genotype text, random sizes, both packings.
// Built and run by tests/test_host_sanitizers.py (CPU only; GPU sanitizers are
// not available on this pool).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "gw_pack.h"

int main(int argc, char **argv) {
    const unsigned iters = argc > 1 ? std::atoi(argv[1]) : 2000;
    std::mt19937 rng(12345);
    const char alphabet[] = "ACGT0NB? 12";
    unsigned long accepted = 0, rejected = 0, checksum = 0;
    for (unsigned it = 0; it < iters; ++it) {
        const uint32_t n = 1 + rng() % 300;
        const uint32_t samples = (rng() % 8 == 0) ? n + 1 + rng() % 3 : (rng() % 4 == 0 ? rng() % (n + 1) : n);
        // mostly well-formed rows (two alleles of a bi-allelic SNP), sometimes noise
        const bool noisy = rng() % 5 == 0;
        const char a1 = "ACGT"[rng() % 4], a2 = "ACGT"[rng() % 4];
        std::string text;
        for (uint32_t s = 0; s < samples; ++s) {
            if (noisy) {
                text.push_back(alphabet[rng() % (sizeof alphabet - 1)]);
                text.push_back(alphabet[rng() % (sizeof alphabet - 1)]);
            } else {
                switch (rng() % 5) {
                case 0: text += "00"; break;
                case 1: text.push_back(a1); text.push_back(a1); break;
                case 2: text.push_back(a1); text.push_back(a2); break;
                default: text.push_back(a2); text.push_back(a2); break;
                }
            }
            text.push_back('\t');
        }
        if (rng() % 16 == 0 && !text.empty()) text.pop_back();  // missing final delimiter
        for (int block = 0; block < 2; ++block) {
            const uint32_t blocks = block ? gw::block_row_blocks(n) : gw::stream_row_blocks(n);
            std::vector<uint16_t> row(blocks);  // exact size: any overrun trips ASan
            const int rc = gw::pack_genotype_row(text.data(), text.data() + text.size(), n, block != 0, row.data());
            (rc == 0 ? accepted : rejected)++;
            for (uint16_t v : row) checksum = checksum * 31 + v;
        }
        std::vector<int32_t> idx(rng() % 50);
        for (auto &v : idx) v = (int32_t)(rng() % (n + 2)) - 1;  // sometimes -1 or n: must be refused
        std::vector<uint16_t> mask(gw::stream_blocks(n));
        const long members = gw::build_stream_mask(idx.data(), idx.size(), n, mask.data());
        checksum = checksum * 31 + (unsigned long)(members + 7);
        uint16_t header = 0;
        for (int k = 0; k < 6; ++k) checksum = checksum * 31 + (unsigned long)gw::label_from_header(header, rng() % 16);
    }
    std::printf("accepted %lu rejected %lu checksum %lu\n", accepted, rejected, checksum);
    return accepted > 0 && rejected > 0 ? 0 : 1;
}
