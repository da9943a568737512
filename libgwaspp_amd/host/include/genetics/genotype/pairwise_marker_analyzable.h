// genetics/genotype/pairwise_marker_analyzable.h -- the SNP-pair half of the table's operator API
// (reference: genotype/pairwise_marker_analyzable.h:171-203).  The result holders it fills live in
// scan_results.h.
#pragma once
#include "genetics/analyzable/case_control_set.h"
#include "genetics/genotype/case_control_selectable.h"
#include "genetics/genotype/scan_results.h"

namespace libgwaspp {
namespace genetics {

class PairwiseMarkerAnalyzable : public virtual CaseControlSelectable {
public:
    virtual ~PairwiseMarkerAnalyzable() {}

    // joint genotype counts of two SNPs over every sample
    virtual void getContingencyTable(uint rIdx1, uint rIdx2, ContingencyTable &ct) = 0;
    virtual void getContingencyTable(uint rIdx1, uint rIdx2, ushort *column_set, ContingencyTable &ct) = 0;

    // joint counts split by phenotype:
    //   with the masks of `ccs` applied on the fly ...
    virtual void getCaseControlContingencyTable(uint rIdx1, uint rIdx2, CaseControlSet &ccs,
                                                CaseControlContingencyTable &ccct) = 0;
    //   ... or after selectCaseControl(), optionally helped by the two SNPs' margins
    virtual void getCaseControlContingencyTable(uint rIdx1, uint rIdx2, CaseControlContingencyTable &ccct) = 0;
    virtual void getCaseControlContingencyTable(uint rIdx1, uint rIdx2, const marginal_information &m1,
                                                const marginal_information &m2,
                                                CaseControlContingencyTable &ccct) = 0;

    // legacy look-up-table path of the older tables
    virtual void selectMarkerPair(uint rIdx1, uint rIdx2) = 0;
};

}  // namespace genetics
}  // namespace libgwaspp
