// genetics/genotype/single_marker_analyzable.h -- per-SNP operator API and its
// caller-owned result holders (reference: genotype/single_marker_analyzable.h:
// GenotypeDistribution :41-79, CaseControlGenotypeDistribution :81-127,
// SingleMarkerAnalyzable :129-164).
#pragma once
#include "genetics/analyzable/case_control_set.h"
#include "genetics/genotype/case_control_selectable.h"
#include "genetics/genotype/common_genotype.h"

namespace libgwaspp {
namespace genetics {

class GenotypeDistribution {
public:
    GenotypeDistribution() { reset(); }
    uint getCurrentIndex() { return current_rIdx; }
    const frequency_table *getDistribution() { return &distribution; }
    const header_table *getGenotypes() { return &genotypes; }
    void setDistribution(const frequency_table &ft) { CopyFrequencyTable(distribution, ft); }
    void reset() {
        ResetFrequencyTable(distribution);
        ResetHeaderTable(genotypes);
        current_rIdx = (uint)-1;
    }
    virtual ~GenotypeDistribution() {}

protected:
    uint current_rIdx;
    frequency_table distribution;
    header_table genotypes;
};

class CaseControlGenotypeDistribution {
public:
    CaseControlGenotypeDistribution() { reset(); }
    uint getCurrentIndex() { return current_rIdx; }
    const frequency_table *getCaseDistribution() { return &case_dist; }
    const frequency_table *getControlDistribution() { return &control_dist; }
    const header_table *getGenotypes() { return &genotypes; }
    void setCaseDistribution(const frequency_table &ft) { CopyFrequencyTable(case_dist, ft); }
    void setControlDistribution(const frequency_table &ft) { CopyFrequencyTable(control_dist, ft); }
    void reset() {
        ResetFrequencyTable(case_dist);
        ResetFrequencyTable(control_dist);
        ResetHeaderTable(genotypes);
        current_rIdx = (uint)-1;
    }
    virtual ~CaseControlGenotypeDistribution() {}

protected:
    uint current_rIdx;
    frequency_table case_dist, control_dist;
    header_table genotypes;
};

class SingleMarkerAnalyzable : public virtual CaseControlSelectable {
public:
    virtual void selectMarker(uint rIdx) = 0;
    virtual void getGenotypeDistribution(uint rIdx, GenotypeDistribution &dist) = 0;
    // masks applied inline
    virtual void getCaseControlGenotypeDistribution(uint rIdx, CaseControlSet &ccs,
                                                    CaseControlGenotypeDistribution &ccgd) = 0;
    // after selectCaseControl()
    virtual void getCaseControlGenotypeDistribution(uint rIdx, CaseControlGenotypeDistribution &ccgd) = 0;
    virtual void getCaseControlGenotypeDistribution(uint rIdx, CaseControlGenotypeDistribution &ccgd,
                                                    marginal_information &m) = 0;
    virtual ~SingleMarkerAnalyzable() {}
};

}  // namespace genetics
}  // namespace libgwaspp
