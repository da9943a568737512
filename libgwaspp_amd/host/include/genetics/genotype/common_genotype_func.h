// genetics/genotype/common_genotype_func.h -- arithmetic on the count records
// (reference: genotype/common_genotype_func.h:92-104, .cpp:37-152,173-219).
#pragma once
#include <iostream>

#include "genetics/genotype/common_genotype.h"

namespace libgwaspp {
namespace genetics {

void printFrequencyDistribution(const frequency_table &ft, std::ostream &out, bool include_header = false);
void printContingencyTable(const CONTIN_TABLE_T &ct, std::ostream &out, bool include_header = false);
std::ostream &operator<<(std::ostream &out, CONTIN_TABLE_T &ct);
std::ostream &operator<<(std::ostream &out, frequency_table &ft);

// entropies and conditional genotype/status probabilities of one SNP
// (common_genotype_func.cpp:173-219); entries of zero-count classes are left
// untouched, as the reference leaves them (quirk Q6).
void computeMarginalInformation(const frequency_table &cases, const frequency_table &ctrls, uint nIndivids,
                                marginal_information &m);

}  // namespace genetics
}  // namespace libgwaspp
