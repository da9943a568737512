// gw_pack.cpp -- see gw_pack.h.
#include "gw_pack.h"

#include <cstring>
#include <vector>

namespace gw {

namespace {

// Row-label automaton.  A row may show at most two homozygotes and one
// heterozygote spelling; the order in which they first appear decides their
// 2-bit codes and is recorded in bits 15..12 of the row header
// (common_genotype.h:257-304):
//   stage 0 (order 0x0000) nothing seen
//   stage 1 (order 0x1000) one homozygote
//   stage 2 (order 0x2000) the heterozygote only
//   stage 3 (order 0x3000) two homozygotes
//   stage 4 (order 0x4000) one homozygote + the heterozygote
//   stage 5 (order 0x7000) all three
struct Move {
    int8_t next;     // next stage, -1 = the reference asserts
    uint8_t code;    // 2-bit code given to the new genotype
    uint8_t shift;   // header nibble the encoding is stored in
    uint16_t keep;   // header bits that survive (clear_code)
};
// kMoves[stage][is_homozygote]
const Move kMoves[6][2] = {
    /*0*/ {{2, 2, 4, 0x0000}, {1, 1, 8, 0x0000}},
    /*1*/ {{4, 2, 4, 0x0F00}, {3, 3, 0, 0x0F00}},
    /*2*/ {{-1, 0, 0, 0}, {4, 1, 8, 0x00F0}},
    /*3*/ {{5, 2, 4, 0x0F0F}, {-1, 0, 0, 0}},
    /*4*/ {{-1, 0, 0, 0}, {5, 3, 0, 0x0FF0}},
    /*5*/ {{-1, 0, 0, 0}, {-1, 0, 0, 0}},
};
const uint16_t kOrderBits[6] = {0x0000, 0x1000, 0x2000, 0x3000, 0x4000, 0x7000};

struct SymbolTable {
    uint8_t v[256];
    SymbolTable() {
        std::memset(v, 4, sizeof v);  // everything outside "ACGT" is unknown (quirk Q2)
        v[(unsigned char)'A'] = 0;
        v[(unsigned char)'C'] = 1;
        v[(unsigned char)'G'] = 2;
        v[(unsigned char)'T'] = 3;
    }
};
const SymbolTable kSymbols;

}  // namespace

RowLabeller::RowLabeller() : stage_(0), header_(0) { std::memset(assigned_, 0, sizeof assigned_); }

int RowLabeller::code_for(unsigned enc) {
    if (assigned_[enc]) return assigned_[enc];
    const bool hom = (enc >> 2) == (enc & 3u);  // AA CC GG TT = 0,5,10,15 (table5.cpp:1220-1223)
    const Move &m = kMoves[stage_][hom ? 1 : 0];
    if (m.next < 0) return 0;
    stage_ = (uint8_t)m.next;
    header_ = (uint16_t)((header_ & m.keep) | kOrderBits[stage_] | (uint16_t)(enc << m.shift));
    assigned_[enc] = m.code;
    return m.code;
}

int label_from_header(uint16_t &header, unsigned enc) {
    const uint16_t order = header & 0xF000;
    int stage = 0;
    for (int i = 0; i < 6; ++i)
        if (kOrderBits[i] == order) stage = i;
    const unsigned aa = (header >> 8) & 0xFu, ab = (header >> 4) & 0xFu, bb = header & 0xFu;
    const bool has_aa = stage == 1 || stage >= 3, has_ab = stage == 2 || stage >= 4, has_bb = stage == 3 || stage == 5;
    if (has_aa && enc == aa) return 1;
    if (has_ab && enc == ab) return 2;
    if (has_bb && enc == bb) return 3;
    const bool hom = (enc >> 2) == (enc & 3u);
    const Move &m = kMoves[stage][hom ? 1 : 0];
    if (m.next < 0) return 0;
    header = (uint16_t)((header & m.keep) | kOrderBits[m.next] | (uint16_t)(enc << m.shift));
    return m.code;
}

// both allele letters at once: (first << 8 | second) -> 4 * a + b, or 16 where either is not in "ACGT"
struct PairTable {
    uint8_t v[65536];
    PairTable() {
        for (unsigned x = 0; x < 65536u; ++x) {
            const unsigned a = kSymbols.v[x >> 8], b = kSymbols.v[x & 255u];
            v[x] = (uint8_t)((a > 3u || b > 3u) ? 16u : 4u * a + b);
        }
    }
};
const PairTable kPairs;

int pack_genotype_row(const char *begin, const char *end, uint32_t n_samples, bool block_layout,
                      uint16_t *row) {
    const uint32_t row_blocks = block_layout ? block_row_blocks(n_samples) : stream_row_blocks(n_samples);
    std::memset(row, 0, (size_t)row_blocks * sizeof(uint16_t));  // "unknown" everywhere
    if (begin >= end) return 0;
    RowLabeller lab;
    uint16_t *lo = row + 1;
    uint16_t *hi = row + 1 + stream_blocks(n_samples);
    const size_t len = (size_t)(end - begin);
    const uint32_t n_text = (uint32_t)((len + 2u) / 3u);  // genotypes in the text (3 bytes each, the last may lack its delimiter)
    if (n_text > n_samples) return -1;
    // code of every genotype spelling seen so far in this row (0 = not yet seen / missing); slot 16 = missing
    uint8_t code_of[17];
    std::memset(code_of, 0, sizeof code_of);
    bool seen[17];
    std::memset(seen, 0, sizeof seen);
    seen[16] = true;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(begin);
    // whole genotypes whose two letters are both inside the text
    const uint32_t n_full = (uint32_t)((len + 1u) / 3u) <= n_text ? (uint32_t)((len + 1u) / 3u) : n_text;
    auto classify = [&](unsigned enc) -> int {  // first appearance of a spelling: the labelling automaton
        const int code = lab.code_for(enc);
        if (code == 0) return -1;
        code_of[enc] = (uint8_t)code;
        seen[enc] = true;
        return 0;
    };
    uint32_t sample = 0;
    const uint32_t per_block = block_layout ? 8u : 16u;
    while (sample < n_full) {
        const uint32_t k_end = (n_full - sample < per_block) ? n_full - sample : per_block;
        uint32_t w_lo = 0, w_hi = 0, w_blk = 0;
        for (uint32_t k = 0; k < k_end; ++k, p += 3) {
            const unsigned enc = kPairs.v[((unsigned)p[0] << 8) | p[1]];
            if (!seen[enc] && classify(enc) != 0) return -1;
            const uint32_t code = code_of[enc];
            w_lo |= (code & 1u) << k;
            w_hi |= (code >> 1) << k;
            w_blk |= code << (2u * k);
        }
        if (block_layout) {
            lo[sample >> 3] = (uint16_t)w_blk;
        } else {
            lo[sample >> 4] = (uint16_t)w_lo;
            hi[sample >> 4] = (uint16_t)w_hi;
        }
        sample += k_end;
    }
    // a trailing one-letter genotype (text ends inside it) is a missing call: its bits stay 0
    row[0] = lab.header();
    return 0;
}

long build_stream_mask(const int32_t *idx, size_t n_idx, uint32_t n_samples, uint16_t *mask) {
    const uint32_t blocks = stream_blocks(n_samples);
    std::memset(mask, 0, (size_t)blocks * sizeof(uint16_t));
    long members = 0;
    for (size_t k = 0; k < n_idx; ++k) {
        const int32_t i = idx[k];
        if (i < 0 || (uint32_t)i >= n_samples) return -1;
        const uint16_t bit = (uint16_t)(1u << (i & 15));
        if (!(mask[i >> 4] & bit)) {  // a std::set<int> in the reference: duplicates once
            mask[i >> 4] |= bit;
            ++members;
        }
    }
    return members;
}

}  // namespace gw
