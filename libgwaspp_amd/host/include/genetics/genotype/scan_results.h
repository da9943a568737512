// genetics/genotype/scan_results.h -- the caller-owned value objects that the table's getters fill
// (SURVEY.md 8b "Ownership").  They carry the reference's names and accessors --
//   GenotypeDistribution, CaseControlGenotypeDistribution   (single_marker_analyzable.h:41-127)
//   ContingencyTable, CaseControlContingencyTable           (pairwise_marker_analyzable.h:44-169)
// -- so that code written against libgwaspp compiles unchanged; here they are thin views over
// small building blocks (a marker tag and count payloads) instead of four hand-rolled classes.
#pragma once
#include <iostream>

#include "genetics/genotype/common_genotype.h"
#include "genetics/genotype/common_genotype_func.h"

namespace libgwaspp {
namespace genetics {

namespace detail {

const uint kNoMarker = (uint)-1;

// which SNP row a result belongs to, and which genotype encodings its codes stand for
struct MarkerTag {
    uint row;
    header_table codes;
    MarkerTag() { clear(); }
    void clear() {
        row = kNoMarker;
        ResetHeaderTable(codes);
    }
};

// one frequency_table with value semantics
struct Counts {
    frequency_table ft;
    Counts() { ResetFrequencyTable(ft); }
    void zero() { ResetFrequencyTable(ft); }
    void assign(const frequency_table &src) { CopyFrequencyTable(ft, src); }
};

// one 4x4 joint-genotype table with value semantics
struct Joint {
    CONTIN_TABLE_T tab;
    Joint() { ResetContingencyTable(tab); }
    void zero() { ResetContingencyTable(tab); }
    void assign(const CONTIN_TABLE_T &src) { CopyContingencyTable(tab, src); }
};

}  // namespace detail

// ---- single marker ---------------------------------------------------------
class GenotypeDistribution {
public:
    virtual ~GenotypeDistribution() {}
    uint getCurrentIndex() { return tag.row; }
    const frequency_table *getDistribution() { return &all.ft; }
    const header_table *getGenotypes() { return &tag.codes; }
    void setDistribution(const frequency_table &ft) { all.assign(ft); }
    void reset() { all.zero(); tag.clear(); }

protected:
    detail::MarkerTag tag;
    detail::Counts all;
};

class CaseControlGenotypeDistribution {
public:
    virtual ~CaseControlGenotypeDistribution() {}
    uint getCurrentIndex() { return tag.row; }
    const frequency_table *getCaseDistribution() { return &cases.ft; }
    const frequency_table *getControlDistribution() { return &controls.ft; }
    const header_table *getGenotypes() { return &tag.codes; }
    void setCaseDistribution(const frequency_table &ft) { cases.assign(ft); }
    void setControlDistribution(const frequency_table &ft) { controls.assign(ft); }
    void reset() { cases.zero(); controls.zero(); tag.clear(); }

protected:
    detail::MarkerTag tag;
    detail::Counts cases, controls;
};

// ---- marker pair -----------------------------------------------------------
namespace detail {
// the two markers of a pair result
struct PairTags {
    MarkerTag a, b;
    void clear() { a.clear(); b.clear(); }
};
}  // namespace detail

class ContingencyTable {
public:
    virtual ~ContingencyTable() {}
    uint getMarkerAIndex() { return who.a.row; }
    uint getMarkerBIndex() { return who.b.row; }
    void setMarkerAIndex(uint r) { who.a.row = r; }
    void setMarkerBIndex(uint r) { who.b.row = r; }
    const header_table *getMarkerAHeader() { return &who.a.codes; }
    const header_table *getMarkerBHeader() { return &who.b.codes; }
    void setHeaderA(const header_table &h) { who.a.codes.l = h.l; }
    void setHeaderB(const header_table &h) { who.b.codes.l = h.l; }
    const CONTIN_TABLE_T *getContingencyTable() { return &joint.tab; }
    void setContingency(const CONTIN_TABLE_T &ct) { joint.assign(ct); }
    void reset() { who.clear(); joint.zero(); }

protected:
    detail::PairTags who;
    detail::Joint joint;
};

class CaseControlContingencyTable {
public:
    virtual ~CaseControlContingencyTable() {}
    uint getMarkerAIndex() { return who.a.row; }
    uint getMarkerBIndex() { return who.b.row; }
    void setMarkerAIndex(uint r) { who.a.row = r; }
    void setMarkerBIndex(uint r) { who.b.row = r; }
    const header_table *getMarkerAHeader() { return &who.a.codes; }
    const header_table *getMarkerBHeader() { return &who.b.codes; }
    void setHeaderA(const header_table &h) { who.a.codes.l = h.l; }
    void setHeaderB(const header_table &h) { who.b.codes.l = h.l; }
    const CONTIN_TABLE_T *getCaseContingencyTable() { return &cases.tab; }
    const CONTIN_TABLE_T *getControlContingencyTable() { return &controls.tab; }
    void setCaseContingency(const CONTIN_TABLE_T &ct) { cases.assign(ct); }
    void setControlContingency(const CONTIN_TABLE_T &ct) { controls.assign(ct); }
    void updateContingencyTables(const CONTIN_TABLE_T &cs, const CONTIN_TABLE_T &ct) {
        cases.assign(cs);
        controls.assign(ct);
    }
    void reset() { who.clear(); cases.zero(); controls.zero(); }
    friend std::ostream &operator<<(std::ostream &out, CaseControlContingencyTable &t) {
        return out << "Cases" << std::endl << t.cases.tab << "Controls" << std::endl << t.controls.tab;
    }

protected:
    detail::PairTags who;
    detail::Joint cases, controls;
};

}  // namespace genetics
}  // namespace libgwaspp
