// genetics/genotype/pairwise_marker_analyzable.h -- SNP-pair operator API and
// result holders (reference: genotype/pairwise_marker_analyzable.h:
// ContingencyTable :44-101, CaseControlContingencyTable :103-169,
// PairwiseMarkerAnalyzable :171-203).
#pragma once
#include <iostream>

#include "genetics/analyzable/case_control_set.h"
#include "genetics/genotype/case_control_selectable.h"
#include "genetics/genotype/common_genotype.h"
#include "genetics/genotype/common_genotype_func.h"

namespace libgwaspp {
namespace genetics {

class ContingencyTable {
public:
    ContingencyTable() { reset(); }
    uint getMarkerAIndex() { return ma_rIdx; }
    uint getMarkerBIndex() { return mb_rIdx; }
    void setMarkerAIndex(uint r) { ma_rIdx = r; }
    void setMarkerBIndex(uint r) { mb_rIdx = r; }
    const CONTIN_TABLE_T *getContingencyTable() { return &cont; }
    const header_table *getMarkerAHeader() { return &ma_header; }
    const header_table *getMarkerBHeader() { return &mb_header; }
    void setContingency(const CONTIN_TABLE_T &ct) { CopyContingencyTable(cont, ct); }
    void setHeaderA(const header_table &h) { ma_header.l = h.l; }
    void setHeaderB(const header_table &h) { mb_header.l = h.l; }
    void reset() {
        ResetHeaderTable(ma_header);
        ResetHeaderTable(mb_header);
        ResetContingencyTable(cont);
        ma_rIdx = mb_rIdx = (uint)-1;
    }
    virtual ~ContingencyTable() {}

protected:
    uint ma_rIdx, mb_rIdx;
    CONTIN_TABLE_T cont;
    header_table ma_header, mb_header;
};

class CaseControlContingencyTable {
public:
    CaseControlContingencyTable() { reset(); }
    uint getMarkerAIndex() { return ma_rIdx; }
    uint getMarkerBIndex() { return mb_rIdx; }
    void setMarkerAIndex(uint r) { ma_rIdx = r; }
    void setMarkerBIndex(uint r) { mb_rIdx = r; }
    const header_table *getMarkerAHeader() { return &ma_header; }
    const header_table *getMarkerBHeader() { return &mb_header; }
    const CONTIN_TABLE_T *getCaseContingencyTable() { return &case_contin; }
    const CONTIN_TABLE_T *getControlContingencyTable() { return &control_contin; }
    void setCaseContingency(const CONTIN_TABLE_T &ct) { CopyContingencyTable(case_contin, ct); }
    void setControlContingency(const CONTIN_TABLE_T &ct) { CopyContingencyTable(control_contin, ct); }
    void updateContingencyTables(const CONTIN_TABLE_T &cs, const CONTIN_TABLE_T &ct) {
        setCaseContingency(cs);
        setControlContingency(ct);
    }
    void setHeaderA(const header_table &h) { ma_header.l = h.l; }
    void setHeaderB(const header_table &h) { mb_header.l = h.l; }
    void reset() {
        ResetHeaderTable(ma_header);
        ResetHeaderTable(mb_header);
        ResetContingencyTable(case_contin);
        ResetContingencyTable(control_contin);
        ma_rIdx = mb_rIdx = (uint)-1;
    }
    virtual ~CaseControlContingencyTable() {}
    friend std::ostream &operator<<(std::ostream &out, CaseControlContingencyTable &t) {
        out << "Cases" << std::endl << t.case_contin << "Controls" << std::endl << t.control_contin;
        return out;
    }

protected:
    uint ma_rIdx, mb_rIdx;
    CONTIN_TABLE_T case_contin, control_contin;
    header_table ma_header, mb_header;
};

class PairwiseMarkerAnalyzable : public virtual CaseControlSelectable {
public:
    virtual void selectMarkerPair(uint rIdx1, uint rIdx2) = 0;
    virtual void getContingencyTable(uint rIdx1, uint rIdx2, ContingencyTable &ct) = 0;
    virtual void getContingencyTable(uint rIdx1, uint rIdx2, ushort *column_set, ContingencyTable &ct) = 0;
    // masks applied inline
    virtual void getCaseControlContingencyTable(uint rIdx1, uint rIdx2, CaseControlSet &ccs,
                                                CaseControlContingencyTable &ccct) = 0;
    // after selectCaseControl()
    virtual void getCaseControlContingencyTable(uint rIdx1, uint rIdx2, CaseControlContingencyTable &ccct) = 0;
    virtual void getCaseControlContingencyTable(uint rIdx1, uint rIdx2, const marginal_information &m1,
                                                const marginal_information &m2,
                                                CaseControlContingencyTable &ccct) = 0;
    virtual ~PairwiseMarkerAnalyzable() {}
};

}  // namespace genetics
}  // namespace libgwaspp
