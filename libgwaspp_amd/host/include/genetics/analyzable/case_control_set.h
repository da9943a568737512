// genetics/analyzable/case_control_set.h -- phenotype masks in both packings
// of the reference's CaseControlSet (analyzable/case_control_set.h:44-96,
// case_control_set.cpp:34-191): a 2-bit-replicated mask (0b11 per member) for
// the block table and a 1-bit "stream" mask for the bit-plane table, both
// padded to 64-bit words.
#pragma once
#include <set>
#include <vector>

#include "libgwaspp.h"
#include "util/index_set/indexer.h"

namespace libgwaspp {
namespace genetics {

class CaseControlSet {
public:
    explicit CaseControlSet(const util::indexer *idx);

    const util::indexer *getPossibleIndices() { return possible_indices; }
    uint getMaximumIndex() const { return max_index; }
    uint getCaseCount() const { return case_count; }
    uint getControlCount() const { return ctrl_count; }
    uint getTotalCount() const { return case_count + ctrl_count; }

    void setCases(const std::set<int> &case_idx);
    void setControls(const std::set<int> &ctrl_idx);
    void setAllAsCases();
    void setAllAsControls();
    void reset();

    const ushort *case_begin() const { return block_case.data(); }
    const ushort *case_end() const { return block_case.data() + block_case.size(); }
    const ushort *control_begin() const { return block_ctrl.data(); }
    const ushort *control_end() const { return block_ctrl.data() + block_ctrl.size(); }
    const ushort *stream_case_begin() const { return stream_case.data(); }
    const ushort *stream_case_end() const { return stream_case.data() + stream_case.size(); }
    const ushort *stream_control_begin() const { return stream_ctrl.data(); }
    const ushort *stream_control_end() const { return stream_ctrl.data() + stream_ctrl.size(); }

    bool isCase(uint idx) const;
    bool isControl(uint idx) const;

    virtual ~CaseControlSet() {}

protected:
    void fill(const std::set<int> &idx, std::vector<ushort> &block, std::vector<ushort> &stream, uint &count);
    const util::indexer *possible_indices;
    uint max_index;
    uint case_count, ctrl_count;
    std::vector<ushort> block_case, block_ctrl, stream_case, stream_ctrl;
};

}  // namespace genetics
}  // namespace libgwaspp
