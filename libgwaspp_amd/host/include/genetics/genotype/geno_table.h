// genetics/genotype/geno_table.h -- THE DROP-IN BOUNDARY.
//
// The reference's abstract genotype table (genotype/geno_table.h:47-87): storage contract
// (util::Table<DataBlock>) + per-SNP operators (SingleMarkerAnalyzable) + SNP-pair operators
// (PairwiseMarkerAnalyzable) + phenotype selection (CaseControlSelectable).  A table implementation
// supplies ingest, the genotype vocabulary and the operators; GeneticData picks the implementation
// (genetic_data.cpp:46-80).  HipGenotypeTable (hip_genotype_table.h) is the MI355X one.
#pragma once
#include <string>

#include "genetics/genotype/case_control_selectable.h"
#include "genetics/genotype/pairwise_marker_analyzable.h"
#include "genetics/genotype/single_marker_analyzable.h"
#include "libgwaspp.h"
#include "util/index_set/indexer.h"
#include "util/table/table.h"

// allele symbols and ploidy of the reference build (genotype/genotype.h:80-82, :39-41)
#ifndef GENOTYPE_ALPHABET
#define GENOTYPE_ALPHABET "ACGT"
#endif
#ifndef MAX_ALLELE_COUNT
#define MAX_ALLELE_COUNT 2
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
    virtual ~GenoTable() { delete[] beg; }

    // ---- ingest: one cell, or one SNP row of text "AC<delim>AA<delim>..."
    virtual void addGenotype(int rIdx, int cIdx, const std::string &gt) = 0;
    virtual void addGenotypeRow(int rIdx, const char *p_begin, const char *p_end, char delim) = 0;
    virtual void addGenotypeRow(int rIdx, std::string::const_iterator &it, std::string::const_iterator &it_end,
                                char delim) = 0;

    // ---- genotype vocabulary: text <-> encoding (4*a1 + a2 over the alphabet, 0xFFFF = unknown)
    virtual ushort encodeGenotype(const std::string &gt) = 0;
    virtual const char *decodeGenotype(ushort encoded_gt) = 0;
    virtual bool isGenotypeHomozygous(ushort enc) = 0;
    uint getPossibleGenotypeCount() const { return possible_genotypes_size; }
    virtual ushort *possible_genotypes_begin() const { return beg; }
    virtual ushort *possible_genotypes_end() const { return end; }

    // ---- one call, decoded
    DataBlock getGenotypeAt(uint marker_idx, uint individ_idx) {
        return (*this)(static_cast<int>(marker_idx), static_cast<int>(individ_idx));
    }
    const char *getCallAt(uint marker_idx, uint individ_idx) {
        return decodeGenotype(getGenotypeAt(marker_idx, individ_idx));
    }

protected:
    ushort *beg, *end;  // the encodings a cell can hold
    uint possible_genotypes_size;
};

}  // namespace genetics
}  // namespace libgwaspp
