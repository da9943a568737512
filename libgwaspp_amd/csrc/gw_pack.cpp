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

int pack_genotype_row(const char *begin, const char *end, uint32_t n_samples, bool block_layout,
                      uint16_t *row) {
    const uint32_t row_blocks = block_layout ? block_row_blocks(n_samples) : stream_row_blocks(n_samples);
    std::memset(row, 0, (size_t)row_blocks * sizeof(uint16_t));  // "unknown" everywhere
    if (begin >= end) return 0;
    RowLabeller lab;
    uint16_t *lo = row + 1;
    uint16_t *hi = row + 1 + stream_blocks(n_samples);
    uint32_t sample = 0;
    for (const char *p = begin; p < end; p += 3, ++sample) {
        if (sample >= n_samples) return -1;
        const unsigned a = kSymbols.v[(unsigned char)p[0]];
        const unsigned b = (p + 1 < end) ? kSymbols.v[(unsigned char)p[1]] : 4u;
        if (a > 3u || b > 3u) continue;  // missing call: both bits stay 0
        const int code = lab.code_for(4u * a + b);
        if (code == 0) return -1;
        if (block_layout) {
            lo[sample >> 3] |= (uint16_t)(code << (2u * (sample & 7u)));
        } else {
            const uint16_t bit = (uint16_t)(1u << (sample & 15u));
            if (code & 1) lo[sample >> 4] |= bit;
            if (code & 2) hi[sample >> 4] |= bit;
        }
    }
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
