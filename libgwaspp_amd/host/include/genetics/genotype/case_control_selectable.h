// genetics/genotype/case_control_selectable.h -- "install a phenotype split"
// capability (reference: genotype/case_control_selectable.h:38-56).
#pragma once
#include "genetics/analyzable/case_control_set.h"
#include "genetics/genotype/common_genotype.h"

namespace libgwaspp {
namespace genetics {

class CaseControlSelectable {
public:
    CaseControlSelectable() : nCaseCount(0), nControlCount(0), nIndivids(0) {}
    virtual void selectCaseControl(CaseControlSet &ccs) = 0;
    virtual ~CaseControlSelectable() {}

protected:
    uint nCaseCount, nControlCount, nIndivids;
};

}  // namespace genetics
}  // namespace libgwaspp
