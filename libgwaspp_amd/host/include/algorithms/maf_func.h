// algorithms/maf_func.h -- the per-SNP scan harnesses above the table API
// (reference: algorithms/maf_func.h:46-54, maf_func.cpp:224-335).
#pragma once
#include <iostream>

#include "genetics/genetic_data.h"
#include "genetics/genotype/geno_table.h"

namespace libgwaspp {
namespace algorithms {

using namespace libgwaspp::genetics;

// maf_func.h:46-54.  NB the reference divides (2aa+ab) by tot, not 2*tot, and
// flips values below 0.5; kept bit for bit.
inline void MinorAlleleFrequency(const frequency_table &ft, double &tot, double &maf) {
    tot = ft.aa;
    maf = 2.0 * tot;
    tot += ft.ab;
    maf += ft.ab;
    tot += ft.bb;
    maf /= tot;
    if (maf < 0.5) maf = 1.0 - maf;
}

void compute_maf_perform(GeneticData *gd, std::ostream *out);         // maf_func.cpp:224-236
void select_cc_maf(GeneticData *gd, std::ostream *out);               // :238-269
void genotype_dist_performance(GeneticData *gd, std::ostream *out);   // :271-290
void inline_cc_maf(GeneticData *gd, std::ostream *out);               // :292-316
void inline_maf_print(GeneticData *gd, std::ostream *out);            // :318-335: "xx aa ab bb" per SNP
// case/control analogue of inline_maf_print: one "xx aa ab bb" line per SNP
// into each stream -- the format of genotype_set_builder.pl's
// *.expected.{case,control}.dist files
void inline_cc_maf_print(GeneticData *gd, std::ostream *case_out, std::ostream *ctrl_out);

}  // namespace algorithms
}  // namespace libgwaspp
