// gwas_hip.cpp -- command-line harness over the HipGenotypeTable, after the
// reference's src/test/gwas_basic.cpp (flags :246-334, dispatch :195-237):
// reads a transposed-PLINK pair, builds the table through the same
// GeneticData / GenoTable calls, runs one of the reference's scan harnesses.
//
//   gwas_hip -g x.tped -p x.tfam [-o out] MODE
//   --test-inline-maf      "xx aa ab bb" per SNP (inline_maf_print; the format of
//                          genotype_set_builder.pl's *.expected.dist)
//   --test-inline-cc-maf   same for cases and controls -> <out>.case / <out>.control
//   --inline-cc-maf / --select-cc-maf / --dist-perform    timing harnesses
//   --valid-geno           every call decoded back (validate_func.cpp:31-55)
//   --contin-debug         4x4 table of every SNP pair (epistasis_func.cpp:84-103)
//   --epi-debug            case/control tables of every pair, all three getters
//   --test-boost-epi       computeBoost: KSA pre-screen of all pairs + exact G test on the hits
//   --contin-perform       ContingencyPerformance timing harness
//   --contin-cc-perform    ContingencyCCPerformance: margin-assisted case/control tables of every pair
//   --epi-perform          EpistasisPerformance: case/control tables + two-locus LLR test of every pair
//   --assoc                counts + MAF/HWE/chi-square of every SNP
//   --world W --rank R --comm-id FILE   (with --assoc) one process per GPU: rank R ingests and scans its SNP
//                          range of the .tped, the records are gathered to rank 0 over RCCL, rank 0 prints all of them
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>

#include "algorithms/epistasis_func.h"
#include "algorithms/maf_func.h"
#include "algorithms/sharded_scan.h"
#include "genetics/genotype/hip_genotype_table.h"
#include "genetics/individual/tplink_files.h"

using namespace libgwaspp::genetics;
using namespace libgwaspp::algorithms;

static void usage() {
    std::cerr << "usage: gwas_hip -g <tped> -p <tfam> [-o <out>] [--device N] [--scan-window ROWS] [--no-prefetch] "
                 "[--world W --rank R --comm-id FILE (with --assoc)] "
                 "--test-inline-maf|--test-inline-cc-maf|--inline-cc-maf|--select-cc-maf|--dist-perform|"
                 "--valid-geno|--contin-debug|--epi-debug|--margins|--test-boost-epi|--contin-perform|--contin-cc-perform|--epi-perform|--assoc\n";
    std::exit(2);
}

static void print_table(std::ostream &o, const CONTIN_TABLE_T &t) {
    for (int i = 0; i < 16; ++i) o << t.contin[i] << (i == 15 ? "\n" : "\t");
}

int main(int argc, char **argv) {
    std::string tped, tfam, out_path, mode;
    int device = 0, scan_window = 0;
    bool prefetch = true;
    ShardInfo shard;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if ((a == "-g" || a == "--genotype") && i + 1 < argc) tped = argv[++i];
        else if ((a == "-p" || a == "--phenotype") && i + 1 < argc) tfam = argv[++i];
        else if ((a == "-o" || a == "--output") && i + 1 < argc) out_path = argv[++i];
        else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--scan-window" && i + 1 < argc) scan_window = std::atoi(argv[++i]);  // rows per device scan
        else if (a == "--no-prefetch") prefetch = false;
        else if (a == "--world" && i + 1 < argc) shard.world = std::atoi(argv[++i]);
        else if (a == "--rank" && i + 1 < argc) shard.rank = std::atoi(argv[++i]);
        else if (a == "--comm-id" && i + 1 < argc) shard.id_file = argv[++i];
        else if (a == "-t") {}  // transposed input is the only form here
        else if (a.rfind("--", 0) == 0) mode = a;
        else usage();
    }
    if (tped.empty() || tfam.empty() || mode.empty()) usage();

    shard.device = device;
    const bool sharded = shard.world > 1 || !shard.id_file.empty();
    if (sharded && (mode != "--assoc" || shard.rank < 0 || shard.rank >= shard.world || (shard.world > 1 && shard.id_file.empty())))
        usage();
    long n_total = 0;
    unsigned long first = 0, last = 0;
    if (sharded) {
        n_total = countTpedMarkers(tped);
        if (n_total < 0) { std::cerr << "cannot read " << tped << std::endl; return 1; }
        shardRange((unsigned long)n_total, shard.rank, shard.world, first, last);
    }
    GeneticData gd(eHip2BitStream, device);
    if (!readTfamIndividuals(tfam, &gd) ||
        !(sharded ? readTpedGenotypes(tped, &gd, '\t', (long)first, (long)(last - first)) : readTpedGenotypes(tped, &gd)) ||
        !readTfamCaseControl(tfam, &gd)) {
        std::cerr << "cannot read " << tped << " / " << tfam << std::endl;
        return 1;
    }
    std::ofstream file;
    std::ostream *out = &std::cout;
    if (!out_path.empty() && mode != "--test-inline-cc-maf") {
        file.open(out_path.c_str());
        out = &file;
    }
    GenoTable &gt = *gd.getGenotypeTable();
    if (HipGenotypeTable *hip = dynamic_cast<HipGenotypeTable *>(&gt)) {
        if (scan_window > 0) hip->setScanWindow((uint)scan_window);
        hip->setPrefetch(prefetch);
    }
    const int M = gd.getGenotypedMarkersCount(), N = gd.getGenotypedIndividualsCount();

    if (mode == "--test-inline-maf") {
        inline_maf_print(&gd, out);
    } else if (mode == "--test-inline-cc-maf") {
        if (out_path.empty()) usage();
        std::ofstream fc((out_path + ".case").c_str()), ft((out_path + ".control").c_str());
        inline_cc_maf_print(&gd, &fc, &ft);
    } else if (mode == "--inline-cc-maf") {
        inline_cc_maf(&gd, out);
    } else if (mode == "--select-cc-maf") {
        select_cc_maf(&gd, out);
    } else if (mode == "--dist-perform") {
        genotype_dist_performance(&gd, out);
    } else if (mode == "--valid-geno") {
        for (int r = 0; r < M; ++r) {
            *out << gd.getGenotypedMarkerID(r);
            for (int c = 0; c < N; ++c) *out << "\t" << gt.getCallAt(r, c);
            *out << "\n";
        }
    } else if (mode == "--contin-debug") {
        ContingencyTable ct;
        for (int i = 0; i < M; ++i)
            for (int j = i + 1; j < M; ++j) {
                gt.getContingencyTable(i, j, ct);
                *out << i << "\t" << j << "\t";
                print_table(*out, *ct.getContingencyTable());
            }
    } else if (mode == "--epi-debug") {
        CaseControlSet &ccs = *gd.getCaseControlSet();
        CaseControlContingencyTable a, b, c;
        CaseControlGenotypeDistribution ccgd;
        gt.selectCaseControl(ccs);
        std::vector<marginal_information> margins(M);
        for (int i = 0; i < M; ++i) gt.getCaseControlGenotypeDistribution(i, ccgd, margins[i]);
        for (int i = 0; i < M; ++i)
            for (int j = i + 1; j < M; ++j) {
                gt.getCaseControlContingencyTable(i, j, ccs, a);
                gt.getCaseControlContingencyTable(i, j, b);
                gt.getCaseControlContingencyTable(i, j, margins[i], margins[j], c);
                for (CaseControlContingencyTable *t : {&a, &b, &c}) {
                    *out << i << "\t" << j << "\tcase\t";
                    print_table(*out, *t->getCaseContingencyTable());
                    *out << i << "\t" << j << "\tctrl\t";
                    print_table(*out, *t->getControlContingencyTable());
                }
            }
    } else if (mode == "--margins") {
        // computeMargins' loop (epistasis_func.cpp:706-721): the three-argument getter after selectCaseControl
        CaseControlSet &ccs = *gd.getCaseControlSet();
        CaseControlGenotypeDistribution ccgd;
        gt.selectCaseControl(ccs);
        char buf[64];
        for (int i = 0; i < M; ++i) {
            marginal_information m;
            std::memset(&m, 0xFF, sizeof m);  // the getter must fill every field
            gt.getCaseControlGenotypeDistribution(i, ccgd, m);
            *out << i;
            for (int g = 0; g < 4; ++g) *out << "\t" << m.margins.freq[g];
            for (int g = 0; g < 4; ++g) *out << "\t" << ccgd.getCaseDistribution()->freq[g];
            for (int g = 0; g < 4; ++g) *out << "\t" << ccgd.getControlDistribution()->freq[g];
            std::snprintf(buf, sizeof buf, "\t%.17g\t%.17g", m.dMarginalEntropy, m.dMarginalEntropy_Y);
            *out << buf;
            for (int k = 0; k < 8; ++k) { std::snprintf(buf, sizeof buf, "\t%.17g", m.dPbc[k]); *out << buf; }
            for (int k = 0; k < 8; ++k) { std::snprintf(buf, sizeof buf, "\t%.17g", m.dPca[k]); *out << buf; }
            *out << "\n";
        }
    } else if (mode == "--test-boost-epi") {
        computeBoost(&gd, out);
    } else if (mode == "--contin-perform") {
        ContingencyPerformance(&gd, out);
    } else if (mode == "--contin-cc-perform") {
        ContingencyCCPerformance(&gd, out);
    } else if (mode == "--epi-perform") {
        EpistasisPerformance(&gd, out);
    } else if (mode == "--assoc" && sharded) {
        std::vector<uint> counts;
        std::vector<double> stats;
        shardedCaseControlScan(&gd, shard, (unsigned long)n_total, counts, stats);
        if (shard.rank == 0) {
            *out << std::setprecision(17);
            for (long i = 0; i < n_total; ++i) {
                for (int k = 0; k < 8; ++k) *out << (k ? "\t" : "") << counts[(size_t)i * 8 + k];
                for (int k = 0; k < 8; ++k) *out << "\t" << stats[(size_t)i * 8 + k];
                *out << "\n";
            }
        }
    } else if (mode == "--assoc") {
        HipGenotypeTable &ht = static_cast<HipGenotypeTable &>(gt);
        CaseControlSet &ccs = *gd.getCaseControlSet();
        CaseControlGenotypeDistribution ccgd;
        *out << std::setprecision(17);
        for (int i = 0; i < M; ++i) {
            gt.getCaseControlGenotypeDistribution(i, ccs, ccgd);
            const frequency_table &ca = *ccgd.getCaseDistribution(), &co = *ccgd.getControlDistribution();
            *out << ca.aa << "\t" << ca.ab << "\t" << ca.bb << "\t" << ca.xx << "\t" << co.aa << "\t" << co.ab
                 << "\t" << co.bb << "\t" << co.xx;
            const double *s = ht.getAssociationStatistics(i);
            for (int k = 0; k < 8; ++k) *out << "\t" << s[k];
            *out << "\n";
        }
    } else {
        usage();
    }
    return 0;
}
