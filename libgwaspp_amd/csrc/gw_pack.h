// gw_pack.h -- host-side ingest logic of the drop-in table: genotype text ->
// 2-bit codes with the reference's first-seen labelling, and phenotype index
// sets -> bit masks.  Pure host C++ (text parsing is not a GPU problem);
// product code, independent of oracle/.
#pragma once

#include <cstddef>
#include <cstdint>

namespace gw {

// geometry of the reference's two 2-bit row packings (uint16 blocks)
//   stream: compressed_genotype_table5.cpp:52-72   block: table3.cpp:52-67
inline uint32_t stream_blocks(uint32_t n_samples) { return ((n_samples / 16u + 1u) + 3u) & ~3u; }
inline uint32_t block_blocks(uint32_t n_samples) { return ((n_samples / 8u + 1u) + 3u) & ~3u; }
inline uint32_t stream_row_blocks(uint32_t n) { return 2u * stream_blocks(n) + 1u; }
inline uint32_t block_row_blocks(uint32_t n) { return block_blocks(n) + 1u; }

// Tracks which genotypes a row has shown so far and which 2-bit code each got
// (HeaderStateMachine, genotype/common_genotype.h:257-304).
class RowLabeller {
public:
    RowLabeller();
    // encoding = 4*a1 + a2 over "ACGT" (table5.cpp:110-118).  Returns the
    // 2-bit code (1 first-seen homozygote, 2 heterozygote, 3 the other
    // homozygote) or 0 if the reference would assert on this row.
    int code_for(unsigned encoding);
    uint16_t header() const { return header_; }

private:
    uint8_t assigned_[16];
    uint8_t stage_;  // index into the transition table
    uint16_t header_;
};

// Single-call form of the labeller for GenoTable::addGenotype(r, c, gt)
// (table5.cpp:156-194): continues from the state recorded in `header`, returns
// the 2-bit code of `encoding` (0 = the reference asserts) and updates header.
int label_from_header(uint16_t &header, unsigned encoding);

// text "AC<d>AA<d>..." (two allele symbols + one delimiter per sample; what
// TpedGenotypeFile hands over, individual/tped_genotype_file.cpp:127-190) ->
// one row in the chosen reference packing, header block first.
// Returns 0, or -1 where the reference asserts / more samples than columns.
int pack_genotype_row(const char *begin, const char *end, uint32_t n_samples, bool block_layout,
                      uint16_t *row);

// CaseControlSet::setCases/setControls in the stream packing
// (analyzable/case_control_set.cpp:77-150).  Returns member count, -1 on an
// out-of-range index.
long build_stream_mask(const int32_t *idx, size_t n_idx, uint32_t n_samples, uint16_t *mask);

}  // namespace gw
