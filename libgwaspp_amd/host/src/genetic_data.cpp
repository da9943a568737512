// genetic_data.cpp -- see genetics/genetic_data.h.
#include "genetics/genetic_data.h"

#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "genetics/genotype/hip_genotype_table.h"

namespace libgwaspp {
namespace genetics {

GeneticData::GeneticData(eCompressionLevel comp_level, int device)
    : compression_level(comp_level), device_id(device), genotyped_individs(nullptr), genotyped_markers(nullptr),
      geno_tbl(nullptr), ccs(nullptr) {}

GeneticData::~GeneticData() {
    delete geno_tbl;
    delete ccs;
    delete genotyped_individs;
    delete genotyped_markers;
}

void GeneticData::createGenotypedIndividuals(const std::vector<std::string> &ids) {
    delete genotyped_individs;
    genotyped_individs = new util::SequentialIndexer(ids);
}

void GeneticData::createGenotypedMarkers(const std::vector<std::string> &ids) {
    delete genotyped_markers;
    genotyped_markers = new util::SequentialIndexer(ids);
}

void GeneticData::createGenotypedIndividuals(int n) {
    delete genotyped_individs;
    genotyped_individs = new util::SequentialIndexer(n);
}

void GeneticData::createGenotypedMarkers(int n) {
    delete genotyped_markers;
    genotyped_markers = new util::SequentialIndexer(n);
}

void GeneticData::setCaseControlColumns(const std::set<int> &cases, const std::set<int> &controls) {
    if (!ccs) ccs = new CaseControlSet(genotyped_individs);
    ccs->reset();
    std::set<int> ctrl_only(controls);
    for (int i : cases) ctrl_only.erase(i);  // a column in both sets stays a case (genetic_data.cpp:148-153)
    ccs->setCases(cases);
    ccs->setControls(ctrl_only);
}

void GeneticData::updateGenotypeTable() {
    delete geno_tbl;
    geno_tbl = nullptr;
    switch (compression_level) {
    case eHip2BitStream:
        geno_tbl = new HipGenotypeTable(genotyped_markers, genotyped_individs, device_id);
        break;
    default:
        // the CPU tables of the reference (genetic_data.cpp:60-79) are not part of this build
        std::fprintf(stderr, "GeneticData: compression level %d is a CPU table of the reference; this build only "
                             "provides eHip2BitStream\n", (int)compression_level);
        std::abort();
    }
}

void GeneticData::setCaseControlSet(std::set<std::string> *case_set, std::set<std::string> *ctrl_set) {
    if (!ccs) ccs = new CaseControlSet(genotyped_individs);
    ccs->reset();
    std::set<int> cases, controls;
    if (case_set)
        for (const std::string &id : *case_set) {
            const int i = getGenotypedIndividualIndex(id);
            if (i >= 0) cases.insert(i);
            else std::cout << "Case Individual not genotyped: " << id << std::endl;
        }
    if (ctrl_set)
        for (const std::string &id : *ctrl_set) {
            const int i = getGenotypedIndividualIndex(id);
            if (i >= 0) controls.insert(i);
            else std::cout << "Control Individual not genotyped: " << id << std::endl;
        }
    for (int i : cases)  // a column in both sets stays a case (genetic_data.cpp:148-153)
        controls.erase(i);
    ccs->setCases(cases);
    ccs->setControls(controls);
}

}  // namespace genetics
}  // namespace libgwaspp
