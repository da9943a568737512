// genetics/genotype/single_marker_analyzable.h -- the per-SNP half of the table's operator API
// (reference: genotype/single_marker_analyzable.h:129-164).  The result holders it fills live in
// scan_results.h.
#pragma once
#include "genetics/analyzable/case_control_set.h"
#include "genetics/genotype/case_control_selectable.h"
#include "genetics/genotype/scan_results.h"

namespace libgwaspp {
namespace genetics {

class SingleMarkerAnalyzable : public virtual CaseControlSelectable {
public:
    virtual ~SingleMarkerAnalyzable() {}

    // counts of one SNP over every sample
    virtual void getGenotypeDistribution(uint rIdx, GenotypeDistribution &dist) = 0;

    // counts of one SNP split by phenotype:
    //   with the masks of `ccs` applied on the fly ...
    virtual void getCaseControlGenotypeDistribution(uint rIdx, CaseControlSet &ccs,
                                                    CaseControlGenotypeDistribution &ccgd) = 0;
    //   ... or after selectCaseControl(), optionally with the SNP's marginal information
    virtual void getCaseControlGenotypeDistribution(uint rIdx, CaseControlGenotypeDistribution &ccgd) = 0;
    virtual void getCaseControlGenotypeDistribution(uint rIdx, CaseControlGenotypeDistribution &ccgd,
                                                    marginal_information &m) = 0;

    // legacy look-up-table path of the older tables (assert(false) in the 2-bit stream table)
    virtual void selectMarker(uint rIdx) = 0;
};

}  // namespace genetics
}  // namespace libgwaspp
