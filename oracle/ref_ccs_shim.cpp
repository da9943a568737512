// ref_ccs_shim.cpp -- ORACLE-SIDE driver for the REAL reference CaseControlSet.
//
// Test infrastructure only.  Compiled by oracle/Makefile together with the
// reference's own, unmodified sources where they lie under /root/reference:
//   src/libgwaspp/genetics/analyzable/case_control_set.cpp
//   src/util/index_set/index_set.cpp
// into oracle/_ref/libref_ccs.so.  No reference source is copied and no
// stand-in header is written: these two translation units build from the
// checked-in tree as they are (libgwasppConfig.h is checked in under
// src/libgwaspp/).  The genotype-table translation units are NOT buildable
// here -- genotype/common_genotype_func.h:36-37 includes Rmath.h, which this
// image does not have -- so the real reference pins the phenotype masks only.
//
// The C entry points below hand the reference object's mask bytes to
// tests/test_oracle_ref.py, which compares them with gwo_ccs_set().
#include <cstdint>
#include <cstring>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "genetics/analyzable/case_control_set.h"
#include "util/index_set/index_set.h"

namespace {
// the reference's `indexable` is a 3-method naming interface; samples are
// simply named by their column number here
struct ColumnNames : util::indexable {
    int n;
    explicit ColumnNames(int n_) : n(n_) {}
    int operator()(const std::string &id) { return std::stoi(id); }
    std::string operator()(int idx) const { std::ostringstream s; s << idx; return s.str(); }
    int size() const { return n; }
};
}  // namespace

extern "C" {

// Builds the reference CaseControlSet for n_samples columns with the given
// case / control column sets and copies out its four masks.
//   block_case/block_ctrl   : (n/8+1 -> x4) uint16, 0b11 per member
//   stream_case/stream_ctrl : (n/16+1 -> x4) uint16, one bit per member
// counts[0..2] = getCaseCount, getControlCount, getTotalCount.
int ref_ccs_build(int n_samples, const int32_t *cases, int n_cases, const int32_t *ctrls,
                  int n_ctrls, uint16_t *block_case, uint16_t *block_ctrl, uint16_t *stream_case,
                  uint16_t *stream_ctrl, uint32_t *counts) {
    ColumnNames names(n_samples);
    std::vector<int> all(n_samples);
    for (int i = 0; i < n_samples; ++i) all[i] = i;
    util::IndexedSet columns(&names, all);

    libgwaspp::genetics::CaseControlSet ccs(&columns);
    std::set<int> cs(cases, cases + n_cases), ct(ctrls, ctrls + n_ctrls);
    ccs.setCases(cs);
    ccs.setControls(ct);

    const unsigned n = (unsigned)n_samples;
    unsigned bc = n / 8 + 1, sc = n / 16 + 1;
    bc = (bc + 3) & ~3u;
    sc = (sc + 3) & ~3u;
    std::memcpy(block_case, ccs.case_begin(), bc * 2);
    std::memcpy(block_ctrl, ccs.control_begin(), bc * 2);
    std::memcpy(stream_case, ccs.stream_case_begin(), sc * 2);
    std::memcpy(stream_ctrl, ccs.stream_control_begin(), sc * 2);
    counts[0] = ccs.getCaseCount();
    counts[1] = ccs.getControlCount();
    counts[2] = ccs.getTotalCount();
    return 0;
}

}  // extern "C"
