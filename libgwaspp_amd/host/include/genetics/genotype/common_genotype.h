// genetics/genotype/common_genotype.h -- result records of the scan path, binary
// compatible with the reference's (genotype/common_genotype.h:44-50 header_table,
// :67-75 frequency_table, :101-106 marginal_information, :163-171
// contingency_table, :182-192 extended_contingency_table = CONTIN_TABLE_T).
#pragma once
#include <cstring>

#include "common.h"

namespace libgwaspp {
namespace genetics {

enum { GENOTYPE_COUNT = 4 };

// genotype encodings (4*a1+a2 over "ACGT") behind the row's 2-bit codes
union header_table {
    ulong l;
    ushort header[4];
    struct { ushort xx, aa, ab, bb; };
};
inline void ResetHeaderTable(header_table &h) { h.l = 0xFFFF000000000000ul; }

// per-SNP genotype counts | AA | Aa | aa | xx |
union frequency_table {
    uint freq[GENOTYPE_COUNT];
    struct { ulong l0, l1; };
    struct { uint aa, ab, bb, xx; };
};
inline void CopyFrequencyTable(frequency_table &d, const frequency_table &s) { d.l0 = s.l0; d.l1 = s.l1; }
inline void ResetFrequencyTable(frequency_table &f) { f.l0 = 0; f.l1 = 0; }

struct marginal_information {
    frequency_table margins, cases, controls;
    double dMarginalEntropy, dMarginalEntropy_Y;
    double dPbc[2 * GENOTYPE_COUNT];  // P(genotype | status): cases[4], controls[4]
    double dPca[2 * GENOTYPE_COUNT];  // P(status | genotype): cases[4], controls[4]
};

// 3x3 joint genotype counts, marker A down, marker B across, + one xx total
union contingency_table {
    uint contin[10];
    struct { uint AA_BB, AA_Bb, AA_bb, Aa_BB, Aa_Bb, Aa_bb, aa_BB, aa_Bb, aa_bb, xx_xx; };
};

// 4x4 with the missing-call margins
union extended_contingency_table {
    uint contin[16];
    struct {
        uint AA_BB, AA_Bb, AA_bb, AA_xx, Aa_BB, Aa_Bb, Aa_bb, Aa_xx, aa_BB, aa_Bb, aa_bb, aa_xx,
             xx_BB, xx_Bb, xx_bb, xx_xx;
    };
};
typedef extended_contingency_table CONTIN_TABLE_T;
enum { CONTIN_ROW_COUNT = GENOTYPE_COUNT, CONTIN_COLUMN_COUNT = GENOTYPE_COUNT };

inline void ResetContingencyTable(CONTIN_TABLE_T &c) { std::memset(c.contin, 0, sizeof c.contin); }
inline void CopyContingencyTable(CONTIN_TABLE_T &d, const CONTIN_TABLE_T &s) { std::memcpy(d.contin, s.contin, sizeof d.contin); }

}  // namespace genetics
}  // namespace libgwaspp
