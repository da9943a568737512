// common_genotype_func.cpp -- see genetics/genotype/common_genotype_func.h.
#include "genetics/genotype/common_genotype_func.h"

#include <cmath>
#include <cstdlib>

namespace libgwaspp {
namespace genetics {

void printFrequencyDistribution(const frequency_table &ft, std::ostream &out, bool include_header) {
    if (include_header) out << "AA\tAa\taa\txx" << std::endl;
    out << ft.aa << "\t" << ft.ab << "\t" << ft.bb << "\t" << ft.xx << std::endl;
}

void printContingencyTable(const CONTIN_TABLE_T &ct, std::ostream &out, bool include_header) {
    static const char *row_name[4] = {"AA", "Aa", "aa", "xx"};
    if (include_header) out << "\tBB\tBb\tbb\txx" << std::endl;
    for (int r = 0; r < CONTIN_ROW_COUNT; ++r) {
        out << row_name[r];
        for (int c = 0; c < CONTIN_COLUMN_COUNT; ++c) out << "\t" << ct.contin[r * CONTIN_COLUMN_COUNT + c];
        out << std::endl;
    }
}

std::ostream &operator<<(std::ostream &out, CONTIN_TABLE_T &ct) {
    printContingencyTable(ct, out, true);
    return out;
}

std::ostream &operator<<(std::ostream &out, frequency_table &ft) {
    printFrequencyDistribution(ft, out, true);
    return out;
}

void computeMarginalInformation(const frequency_table &cases, const frequency_table &ctrls, uint nIndivids,
                                marginal_information &m) {
    CopyFrequencyTable(m.cases, cases);
    CopyFrequencyTable(m.controls, ctrls);
    const uint nCases = cases.aa + cases.ab + cases.bb + cases.xx;
    const uint nControls = ctrls.aa + ctrls.ab + ctrls.bb + ctrls.xx;
    if (nCases + nControls != nIndivids) std::abort();  // assert, common_genotype_func.cpp:180
    m.dMarginalEntropy = 0.0;
    m.dMarginalEntropy_Y = 0.0;
    const double n = (double)nIndivids;
    for (int g = 0; g < GENOTYPE_COUNT; ++g) {
        const uint ca = cases.freq[g], co = ctrls.freq[g], mar = ca + co;
        m.margins.freq[g] = mar;
        if (mar > 0) {
            const double p = (double)mar / n;
            m.dMarginalEntropy += -(p)*std::log(p);
        }
        if (ca > 0) {
            const double p = (double)ca / n;
            m.dMarginalEntropy_Y += -(p)*std::log(p);
            m.dPbc[g] = (double)ca / (double)nCases;
            m.dPca[g] = (double)ca / (double)mar;
        }
        if (co > 0) {
            const double p = (double)co / n;
            m.dMarginalEntropy_Y += -(p)*std::log(p);
            m.dPbc[GENOTYPE_COUNT + g] = (double)co / (double)nControls;
            m.dPca[GENOTYPE_COUNT + g] = (double)co / (double)mar;
        }
    }
}

}  // namespace genetics
}  // namespace libgwaspp
