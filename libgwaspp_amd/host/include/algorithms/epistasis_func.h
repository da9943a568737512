// algorithms/epistasis_func.h -- the SNP-pair harnesses above the table API
// (reference: algorithms/epistasis_func.h:58-92, epistasis_func.cpp:105-135
// ContingencyPerformance, :349-506 computeBoost, :508-704 computeGTest,
// :706-721 computeMargins).
#pragma once
#include <iostream>
#include <utility>
#include <vector>

#include "genetics/genetic_data.h"
#include "genetics/genotype/geno_table.h"

namespace libgwaspp {
namespace algorithms {

using namespace libgwaspp::genetics;

const int GT_COUNT = 3;
const int GT_BUFFER_COUNT = 2;
const int GT_BUFFER_SIZE = GT_COUNT * GT_BUFFER_COUNT;

typedef std::pair<uint, uint> SNPPair;
typedef std::pair<SNPPair, double> SNPInteractionPair;

// marginal_information of every SNP (after selectCaseControl), :706-721
void computeMargins(GenoTable &gt, int nIndivids, marginal_information *&pMargins, int &nMarkerCount);

// BOOST: KSA pre-screen of all pairs (threshold 30.0), then the exact G test on
// the pairs that passed; prints the reference's report
void computeBoost(GeneticData *gd, std::ostream *out);

// the second half of computeBoost (:488-505): "Located ...", computeGTest on the passing pairs, the report lines
void boostStage2AndReport(GenoTable &gt, marginal_information *pMargins, int nIndivids,
                          std::vector<SNPInteractionPair> &passingThreshold, std::ostream *out);

// stage 2: iterative proportional fit + exact measure + odds-ratio z per listed
// pair; passingThreshold[k].second is replaced by the exact measure
void computeGTest(GenoTable &gt, marginal_information *pMargins, uint nIndivids,
                  std::vector<SNPInteractionPair> &passingThreshold, std::vector<double> &zval);

// one pair, from its margin-assisted tables (the body of computeGTest's loop)
void gtestOnePair(const CONTIN_TABLE_T &contin_ca, const CONTIN_TABLE_T &contin_co, const marginal_information &m1,
                  const marginal_information &m2, uint nIndivids, double &interMeasure, double &z);
// the pre-screen's interaction measure of one pair (:424-470)
double boostMeasureOnePair(const CONTIN_TABLE_T &contin_ca, const CONTIN_TABLE_T &contin_co,
                           const marginal_information &m1, const marginal_information &m2, int nIndivids);

// 4x4 tables of every pair i < j, timing harness (:105-135)
void ContingencyPerformance(GeneticData *gd, std::ostream *out);
// margin-assisted case/control tables of every pair after selectCaseControl + computeMargins, timing
// harness (:137-203)
void ContingencyCCPerformance(GeneticData *gd, std::ostream *out);
// case/control tables of every pair through the inline-mask getter + the two-locus log-likelihood-ratio
// test of every pair, timing harness (:307-347).  The harness makes its own CaseControlSet: columns 0, 2, ... 398
// are cases, 1, 3, ... 399 controls (as far as the table has them).
void EpistasisPerformance(GeneticData *gd, std::ostream *out);
// 2 x log-likelihood ratio of the saturated two-locus model against the Kirkwood superposition approximation,
// from the 3 x 3 case and control tables of a pair.  The library's own pairwise_epi_test
// (epistasis_func.cpp:723-843) is broken as shipped; this follows the correct statement of
// the same test, src/test/pairwise.c:50-133.
double pairwise_epi_test(const CONTIN_TABLE_T &contin_ca, const CONTIN_TABLE_T &contin_co);
// upper tail of chi-square with 4 degrees of freedom = pchisq(x, 4.0, 0, 0) (pairwise.c:44): exp(-x/2)(1 + x/2)
double chisq4_upper_tail(double x);

}  // namespace algorithms
}  // namespace libgwaspp
