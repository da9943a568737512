// genetics/individual/tplink_files.h -- transposed-PLINK ingest of the scan
// path, restating the three readers the reference's CLI
// chains (src/test/gwas_basic.cpp:126-193):
//   TfamPhenotypeFile   individual/tfam_phenotype_file.cpp:34-119  -> column ids
//   TpedGenotypeFile    individual/tped_genotype_file.cpp:127-190  -> addGenotypeRow per line
//   TFamAnnotationFile  individual/tfam_annotation_file.cpp:38-89  -> case/control sets
#pragma once
#include <string>

#include "genetics/genetic_data.h"

namespace libgwaspp {
namespace genetics {

// ids "<family>-<individual>" of every .tfam line, in file order
bool readTfamIndividuals(const std::string &tfam, GeneticData *gd);
// marker ids (column 2) of every .tped line; creates the table, then packs
// every genotype row through GenoTable::addGenotypeRow
// (first_marker, n_markers: only that range of the file's lines becomes the table -- a rank's shard of a job;
// n_markers < 0 = to the end; ids_only: the table is created with the markers' rows left empty -- a rank that
// receives the rows from another one, gw_table_replicate)
bool readTpedGenotypes(const std::string &tped, GeneticData *gd, char delim = '\t', long first_marker = 0,
                       long n_markers = -1, bool ids_only = false);
// number of marker lines in a .tped (the job's n_total when it is sharded)
long countTpedMarkers(const std::string &tped);
// sixth .tfam column: '1' case, '0' control, anything else neither
bool readTfamCaseControl(const std::string &tfam, GeneticData *gd);

}  // namespace genetics
}  // namespace libgwaspp
