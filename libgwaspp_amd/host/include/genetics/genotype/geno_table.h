// genetics/genotype/geno_table.h -- THE DROP-IN BOUNDARY: the abstract genotype
// table of the reference (genotype/geno_table.h:47-87) = Table<DataBlock> +
// SingleMarkerAnalyzable + PairwiseMarkerAnalyzable + CaseControlSelectable.
// HipGenotypeTable (hip_genotype_table.h) is the MI355X implementation.
#pragma once
#include <string>

#include "genetics/genotype/case_control_selectable.h"
#include "genetics/genotype/pairwise_marker_analyzable.h"
#include "genetics/genotype/single_marker_analyzable.h"
#include "libgwaspp.h"
#include "util/index_set/indexer.h"
#include "util/table/table.h"

#ifndef GENOTYPE_ALPHABET
#define GENOTYPE_ALPHABET "ACGT"  // genotype/genotype.h:80-82
#endif
#ifndef MAX_ALLELE_COUNT
#define MAX_ALLELE_COUNT 2  // genotype/genotype.h:39-41
#endif

namespace libgwaspp {
namespace genetics {

class GenoTable : public util::Table<DataBlock>,
                  public PairwiseMarkerAnalyzable,
                  public SingleMarkerAnalyzable,
                  public virtual CaseControlSelectable {
public:
    GenoTable(util::indexer *markers, util::indexer *individuals)
        : util::Table<DataBlock>(markers, individuals), beg(nullptr), end(nullptr), possible_genotypes_size(0) {}

    virtual void addGenotype(int rIdx, int cIdx, const std::string &gt) = 0;
    virtual void addGenotypeRow(int rIdx, std::string::const_iterator &it, std::string::const_iterator &it_end,
                                char delim) = 0;
    virtual void addGenotypeRow(int rIdx, const char *p_begin, const char *p_end, char delim) = 0;

    virtual ushort encodeGenotype(const std::string &gt) = 0;
    virtual const char *decodeGenotype(ushort encoded_gt) = 0;
    virtual bool isGenotypeHomozygous(ushort enc) = 0;

    const char *getCallAt(uint marker_idx, uint individ_idx) { return decodeGenotype(getGenotypeAt(marker_idx, individ_idx)); }
    DataBlock getGenotypeAt(uint marker_idx, uint individ_idx) { return (*this)((int)marker_idx, (int)individ_idx); }

    uint getPossibleGenotypeCount() const { return possible_genotypes_size; }
    virtual ushort *possible_genotypes_begin() const { return beg; }
    virtual ushort *possible_genotypes_end() const { return end; }

    virtual ~GenoTable() { delete[] beg; }

protected:
    ushort *beg, *end;
    uint possible_genotypes_size;
};

}  // namespace genetics
}  // namespace libgwaspp
