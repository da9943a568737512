// tplink_files.cpp -- see genetics/individual/tplink_files.h.
#include "genetics/individual/tplink_files.h"

#include <fstream>
#include <iostream>
#include <sstream>

namespace libgwaspp {
namespace genetics {

namespace {
bool blank(const std::string &s) { return s.find_first_not_of(" \t\r\n") == std::string::npos; }
}  // namespace

bool readTfamIndividuals(const std::string &tfam, GeneticData *gd) {
    std::ifstream in(tfam.c_str());
    if (!in.is_open()) return false;
    std::vector<std::string> ids;
    std::string ln, fam, ind;
    while (std::getline(in, ln)) {
        if (blank(ln)) continue;
        std::istringstream p(ln);
        p >> fam >> ind;
        ids.push_back(fam + "-" + ind);  // id = family + "-" + individual (tfam_annotation_file.cpp:56-58)
    }
    gd->createGenotypedIndividuals(ids);
    return true;
}

bool readTpedGenotypes(const std::string &tped, GeneticData *gd, char delim) {
    std::ifstream in(tped.c_str());
    if (!in.is_open()) return false;
    // pass 1: marker ids (tped_genotype_file.cpp:104-125 parseNextMarkerRecord)
    std::vector<std::string> markers;
    std::string ln, chr, rs;
    while (std::getline(in, ln)) {
        if (blank(ln)) continue;
        std::istringstream p(ln);
        p >> chr >> rs;
        markers.push_back(rs);
    }
    gd->createGenotypedMarkers(markers);
    gd->updateGenotypeTable();
    // pass 2: genotype rows (parseNextGenotypeRecord, :127-190)
    in.clear();
    in.seekg(0, std::ios::beg);
    const size_t n = (size_t)gd->getGenotypedIndividualsCount();
    std::string buf;
    int row = 0;
    while (std::getline(in, ln)) {
        if (blank(ln)) continue;
        size_t b = ln.find_first_not_of(" \t\r\n"), e = ln.find_last_not_of(" \t\r\n") + 1;
        size_t pos = b;
        for (int seen = 0; seen < 4 && pos < e; ++pos)  // skip chrom, rs, distance, position
            if (ln[pos] == ' ' || ln[pos] == '\t') ++seen;
        // alleles sit two characters apart ("A C A A ..."): drop the blank inside
        // each pair, put `delim` after it, map 1..4 to ACGT (:161-185)
        buf.clear();
        buf.reserve(3 * n);
        bool second = false;
        for (; pos < e; pos += 2) {
            char c = ln[pos];
            switch (c) {
            case '1': c = 'A'; break;
            case '2': c = 'C'; break;
            case '3': c = 'G'; break;
            case '4': c = 'T'; break;
            default: break;
            }
            buf.push_back(c);
            if (second) buf.push_back(delim);
            second = !second;
        }
        gd->addGenotypeRow(row++, buf.data(), buf.data() + buf.size(), delim);
    }
    return true;
}

bool readTfamCaseControl(const std::string &tfam, GeneticData *gd) {
    std::ifstream in(tfam.c_str());
    if (!in.is_open()) return false;
    std::set<std::string> cases, controls;
    std::string ln, fam, ind, pat, mat, sex, status;
    while (std::getline(in, ln)) {
        if (blank(ln)) continue;
        std::istringstream p(ln);
        p >> fam >> ind >> pat >> mat >> sex >> status;
        if (status.empty()) continue;
        if (status[0] == '1') cases.insert(fam + "-" + ind);
        else if (status[0] == '0') controls.insert(fam + "-" + ind);
    }
    std::cout << "Setting " << cases.size() << " cases." << std::endl;
    std::cout << "Setting " << controls.size() << " controls." << std::endl;
    gd->setCaseControlSet(&cases, &controls);
    return true;
}

}  // namespace genetics
}  // namespace libgwaspp
