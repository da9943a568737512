// genetics/genetic_data.h -- the factory seam and the accessors of the
// reference's GeneticData facade that the scan harnesses use (reference:
// genetics/genetic_data.h:77-160, genetic_data.cpp:46-80,116-158).  The domain
// model behind the real facade (individual / marker / phenotype collections)
// is metadata bookkeeping outside the scan path and is not mirrored: rows and
// columns are plain id lists here.
#pragma once
#include <set>
#include <string>
#include <vector>

#include "genetics/analyzable/case_control_set.h"
#include "genetics/genotype/geno_table.h"

namespace libgwaspp {
namespace genetics {

// the reference's levels (genetic_data.h:77) + the new HBM-resident table
enum eCompressionLevel {
    eBasicCompression = 0, eByteCompression, eHalfByteCompression, e2BitBlockCompression, e3BitStream,
    e2BitStream, eHip2BitStream
};

class GeneticData {
public:
    explicit GeneticData(eCompressionLevel comp_level = eHip2BitStream, int device = 0);
    virtual ~GeneticData();

    void createGenotypedIndividuals(const std::vector<std::string> &ids);
    void createGenotypedMarkers(const std::vector<std::string> &ids);
    // n entries named by their ordinal (tables generated in place: gwas_hip --simulate)
    void createGenotypedIndividuals(int n);
    void createGenotypedMarkers(int n);
    int getGenotypedIndividualsCount() const { return genotyped_individs ? genotyped_individs->included_size() : 0; }
    int getGenotypedMarkersCount() const { return genotyped_markers ? genotyped_markers->included_size() : 0; }
    int getGenotypedIndividualIndex(const std::string &id) const { return genotyped_individs->orderOf(id); }
    std::string getGenotypedMarkerID(int order) { return genotyped_markers->getIDAtOrderedIndex(order); }
    std::string getGenotypedIndividualID(int order) { return genotyped_individs->getIDAtOrderedIndex(order); }

    // genetic_data.cpp:46-80: picks the table implementation
    void updateGenotypeTable();
    GenoTable *getGenotypeTable() { return geno_tbl; }

    void addGenotypeRow(int r, const char *b, const char *e, char delim) { geno_tbl->addGenotypeRow(r, b, e, delim); }
    void addGenotype(int r, int c, const std::string &s) { geno_tbl->addGenotype(r, c, s); }
    const char *getGenotype(int r, int c) { return geno_tbl->decodeGenotype((*geno_tbl)(r, c)); }

    // genetic_data.cpp:116-158: ids -> column indices; an id in both sets stays a case
    void setCaseControlSet(std::set<std::string> *case_set, std::set<std::string> *ctrl_set);
    // the same from column indices
    void setCaseControlColumns(const std::set<int> &cases, const std::set<int> &controls);
    CaseControlSet *getCaseControlSet() { return ccs; }

private:
    eCompressionLevel compression_level;
    int device_id;
    util::indexer *genotyped_individs, *genotyped_markers;
    GenoTable *geno_tbl;
    CaseControlSet *ccs;
};

}  // namespace genetics
}  // namespace libgwaspp
