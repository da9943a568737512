// case_control_set.cpp -- see genetics/analyzable/case_control_set.h.
#include "genetics/analyzable/case_control_set.h"

#include <algorithm>
#include <cstdlib>

namespace libgwaspp {
namespace genetics {

namespace {
// one pad block, then up to a multiple of the 64-bit word (case_control_set.cpp:37-52)
uint padded_blocks(uint n, uint per_block_shift) { return (((n >> per_block_shift) + 1u) + 3u) & ~3u; }
}  // namespace

CaseControlSet::CaseControlSet(const util::indexer *idx)
    : possible_indices(idx), max_index((uint)idx->included_size() - 1u), case_count(0), ctrl_count(0) {
    const uint n = (uint)idx->included_size();
    block_case.assign(padded_blocks(n, 3), 0);
    block_ctrl.assign(padded_blocks(n, 3), 0);
    stream_case.assign(padded_blocks(n, 4), 0);
    stream_ctrl.assign(padded_blocks(n, 4), 0);
}

void CaseControlSet::fill(const std::set<int> &idx, std::vector<ushort> &block, std::vector<ushort> &stream,
                          uint &count) {
    std::fill(block.begin(), block.end(), 0);
    std::fill(stream.begin(), stream.end(), 0);
    for (int i : idx) {
        if (i < 0 || (uint)i > max_index) std::abort();  // assert(*it <= max_index), :87
        block[i >> 3] |= (ushort)(0x3u << ((i & 7) << 1));
        stream[i >> 4] |= (ushort)(1u << (i & 15));
    }
    count = (uint)idx.size();
}

void CaseControlSet::setCases(const std::set<int> &idx) { fill(idx, block_case, stream_case, case_count); }
void CaseControlSet::setControls(const std::set<int> &idx) { fill(idx, block_ctrl, stream_ctrl, ctrl_count); }

void CaseControlSet::setAllAsCases() {
    // every bit incl. the pad bits of the stream mask, only the last block
    // trimmed in the block mask (case_control_set.cpp:152-169)
    std::fill(block_case.begin(), block_case.end(), 0xFFFF);
    std::fill(block_ctrl.begin(), block_ctrl.end(), 0);
    std::fill(stream_case.begin(), stream_case.end(), 0xFFFF);
    std::fill(stream_ctrl.begin(), stream_ctrl.end(), 0);
    block_case.back() &= (ushort)(0xFFFF >> ((7 - (max_index & 7)) << 1));
    case_count = max_index + 1;
    ctrl_count = 0;
}

void CaseControlSet::setAllAsControls() {
    std::fill(block_case.begin(), block_case.end(), 0);
    std::fill(block_ctrl.begin(), block_ctrl.end(), 0xFFFF);
    std::fill(stream_case.begin(), stream_case.end(), 0);
    std::fill(stream_ctrl.begin(), stream_ctrl.end(), 0xFFFF);
    block_ctrl.back() &= (ushort)(0xFFFF >> ((7 - (max_index & 7)) << 1));
    ctrl_count = max_index + 1;
    case_count = 0;
}

void CaseControlSet::reset() {
    std::fill(block_case.begin(), block_case.end(), 0);
    std::fill(block_ctrl.begin(), block_ctrl.end(), 0);
    std::fill(stream_case.begin(), stream_case.end(), 0);
    std::fill(stream_ctrl.begin(), stream_ctrl.end(), 0);
}

bool CaseControlSet::isCase(uint i) const { return i <= max_index && (stream_case[i >> 4] >> (i & 15) & 1u); }
bool CaseControlSet::isControl(uint i) const { return i <= max_index && (stream_ctrl[i >> 4] >> (i & 15) & 1u); }

}  // namespace genetics
}  // namespace libgwaspp
