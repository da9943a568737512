// tplink_files.cpp -- see genetics/individual/tplink_files.h.
#include "genetics/individual/tplink_files.h"

#include <algorithm>
#include <atomic>
#include <fstream>
#include <iostream>
#include <sstream>
#include <thread>

#include "genetics/genotype/hip_genotype_table.h"

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

long countTpedMarkers(const std::string &tped) {
    std::ifstream in(tped.c_str());
    if (!in.is_open()) return -1;
    long n = 0;
    std::string ln;
    while (std::getline(in, ln))
        if (!blank(ln)) ++n;
    return n;
}

bool readTpedGenotypes(const std::string &tped, GeneticData *gd, char delim, long first_marker, long n_markers,
                       bool ids_only) {
    std::ifstream in(tped.c_str());
    if (!in.is_open()) return false;
    const long last_marker = n_markers < 0 ? -1 : first_marker + n_markers;  // exclusive; -1 = to the end
    // pass 1: marker ids (tped_genotype_file.cpp:104-125 parseNextMarkerRecord)
    std::vector<std::string> markers;
    std::string ln, chr, rs;
    long line_no = 0;
    while (std::getline(in, ln)) {
        if (blank(ln)) continue;
        const long k = line_no++;
        if (k < first_marker || (last_marker >= 0 && k >= last_marker)) continue;
        std::istringstream p(ln);
        p >> chr >> rs;
        markers.push_back(rs);
    }
    gd->createGenotypedMarkers(markers);
    gd->updateGenotypeTable();
    if (ids_only) return true;  // the rows arrive some other way (gw_table_replicate from the rank that read them)
    // pass 2: genotype rows (parseNextGenotypeRecord, :127-190), a batch of lines at a time: the
    // lines are normalised ("A C A A ..." -> "AC<d>AA<d>...", 1..4 -> ACGT) by all host cores and
    // handed to the table together
    in.clear();
    in.seekg(0, std::ios::beg);
    const size_t n = (size_t)gd->getGenotypedIndividualsCount();
    HipGenotypeTable *hip = dynamic_cast<HipGenotypeTable *>(gd->getGenotypeTable());
    const size_t batch_lines = std::max<size_t>(1, std::min<size_t>(4096, (256u << 20) / (4 * n + 64)));
    auto normalise = [n, delim](const std::string &line, std::string &buf) {
        size_t b = line.find_first_not_of(" \t\r\n"), e = line.find_last_not_of(" \t\r\n") + 1;
        size_t pos = b;
        for (int seen = 0; seen < 4 && pos < e; ++pos)  // skip chrom, rs, distance, position
            if (line[pos] == ' ' || line[pos] == '\t') ++seen;
        buf.clear();
        buf.reserve(3 * n);
        bool second = false;
        for (; pos < e; pos += 2) {  // alleles sit two characters apart (:161-185)
            char c = line[pos];
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
    };
    std::vector<std::string> lines, texts;
    int row = 0;
    bool eof = false;
    line_no = 0;
    while (!eof) {
        lines.clear();
        while (lines.size() < batch_lines) {
            if (!std::getline(in, ln)) { eof = true; break; }
            if (blank(ln)) continue;
            const long k = line_no++;
            if (k < first_marker) continue;
            if (last_marker >= 0 && k >= last_marker) { eof = true; break; }
            lines.push_back(ln);
        }
        if (lines.empty()) break;
        texts.resize(lines.size());
        unsigned workers = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), (unsigned)lines.size()));
        std::atomic<size_t> next(0);
        auto work = [&]() {
            for (size_t k; (k = next.fetch_add(1)) < lines.size();) normalise(lines[k], texts[k]);
        };
        std::vector<std::thread> pool;
        for (unsigned w = 1; w < workers; ++w) pool.emplace_back(work);
        work();
        for (std::thread &th : pool) th.join();
        if (hip) {
            std::vector<std::pair<const char *, const char *> > rows(texts.size());
            for (size_t k = 0; k < texts.size(); ++k) rows[k] = std::make_pair(texts[k].data(), texts[k].data() + texts[k].size());
            hip->addGenotypeRows(row, rows, delim);
            row += (int)texts.size();
        } else {
            for (const std::string &t : texts) gd->addGenotypeRow(row++, t.data(), t.data() + t.size(), delim);
        }
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
