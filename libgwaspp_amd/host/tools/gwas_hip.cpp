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
//   --time-boost           computeBoost phase by phase (select + margins, screen, stage 2 on the device, stage 2 on one
//                          host thread): one JSON line
//   --contin-perform       ContingencyPerformance timing harness
//   --contin-cc-perform    ContingencyCCPerformance: margin-assisted case/control tables of every pair
//   --epi-perform          EpistasisPerformance: case/control tables + two-locus LLR test of every pair
//   --assoc                counts + MAF/HWE/chi-square of every SNP
//   --world W --rank R --comm-id FILE --comm-nonce X   one process per GPU (X: a number the ranks of ONE job share,
//                          e.g. the launcher's start time -- a leftover id file of an earlier job is then ignored):
//       with --assoc           rank R ingests and scans its SNP range of the .tped, the records are gathered to rank 0
//                              over RCCL, rank 0 prints all of them
//       with --test-boost-epi  rank 0 ingests the .tped, its rows are replicated to the other ranks' tables (RCCL
//                              broadcast), the pair triangle is screened in row blocks dealt over the ranks, the hit
//                              lists are all-gathered, rank 0 runs the exact test on the hits and prints the report
//   --simulate N M [--seed S] [--missing-q16 Q]   instead of -g / -p: the table (N individuals x M markers, restated
//                          data/simulate_data.cpp) and the phenotype are generated in HBM through the C-ABI
//       with --time-inline-cc-maf   times inline_cc_maf (maf_func.cpp:292-316) walking the drop-in table's per-row
//                              getter over all M rows, window prefetch on and off, beside the kernel time of the same
//                              scan through the C-ABI's device form; checks the walk's counts against the batch scan;
//                              prints one JSON line
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <set>
#include <string>
#include <vector>

#include "algorithms/epistasis_func.h"
#include "algorithms/maf_func.h"
#include "algorithms/sharded_scan.h"
#include "genetics/genotype/hip_genotype_table.h"
#include "genetics/individual/tplink_files.h"
#include "gwas_scan.h"

using namespace libgwaspp::genetics;
using namespace libgwaspp::algorithms;

static void usage() {
    std::cerr << "usage: gwas_hip (-g <tped> -p <tfam> | --simulate N M [--seed S] [--missing-q16 Q]) [-o <out>] [--device N] "
                 "[--scan-window ROWS] [--no-prefetch] [--mask-check-every-call] "
                 "[--world W --rank R --comm-id FILE --comm-nonce X (with --assoc or --test-boost-epi)] --time-inline-cc-maf|"
                 "--test-inline-maf|--test-inline-cc-maf|--inline-cc-maf|--select-cc-maf|--dist-perform|"
                 "--valid-geno|--contin-debug|--epi-debug|--margins|--test-boost-epi|--time-boost|--contin-perform|--contin-cc-perform|--epi-perform|--assoc\n";
    std::exit(2);
}

static void print_table(std::ostream &o, const CONTIN_TABLE_T &t) {
    for (int i = 0; i < 16; ++i) o << t.contin[i] << (i == 15 ? "\n" : "\t");
}

int main(int argc, char **argv) {
    std::string tped, tfam, out_path, mode;
    int device = 0, scan_window = 0;
    bool prefetch = true, mask_check_every_call = false;
    long sim_n = 0, sim_m = 0;
    unsigned long long sim_seed = 0x5EED, sim_missing = 0;
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
        else if (a == "--comm-nonce" && i + 1 < argc) shard.nonce = std::strtoull(argv[++i], nullptr, 0);
        else if (a == "--simulate" && i + 2 < argc) { sim_n = std::atol(argv[++i]); sim_m = std::atol(argv[++i]); }
        else if (a == "--seed" && i + 1 < argc) sim_seed = std::strtoull(argv[++i], nullptr, 0);
        else if (a == "--missing-q16" && i + 1 < argc) sim_missing = std::strtoull(argv[++i], nullptr, 0);
        else if (a == "--mask-check-every-call") mask_check_every_call = true;
        else if (a == "-t") {}  // transposed input is the only form here
        else if (a.rfind("--", 0) == 0) mode = a;
        else usage();
    }
    const bool simulated = sim_n > 0 && sim_m > 0;
    if ((!simulated && (tped.empty() || tfam.empty())) || mode.empty()) usage();

    shard.device = device;
    const bool sharded = shard.world > 1 || !shard.id_file.empty();
    const bool sharded_pairs = sharded && mode == "--test-boost-epi";
    if (sharded && ((mode != "--assoc" && !sharded_pairs) || simulated || shard.rank < 0 || shard.rank >= shard.world ||
                    (shard.world > 1 && (shard.id_file.empty() || shard.nonce == 0))))
        usage();
    long n_total = 0;
    unsigned long first = 0, last = 0;
    if (sharded && !sharded_pairs) {
        n_total = countTpedMarkers(tped);
        if (n_total < 0) { std::cerr << "cannot read " << tped << std::endl; return 1; }
        shardRange((unsigned long)n_total, shard.rank, shard.world, first, last);
    }
    GeneticData gd(eHip2BitStream, device);
    if (simulated) {
        gd.createGenotypedIndividuals((int)sim_n);
        gd.createGenotypedMarkers((int)sim_m);
        gd.updateGenotypeTable();
        HipGenotypeTable &ht = static_cast<HipGenotypeTable &>(*gd.getGenotypeTable());
        gw_table *h = ht.handle();
        if (gw_table_simulate_exact(h, sim_seed, 0, (uint32_t)sim_missing) != GW_OK || gw_table_simulate_phenotype(h, sim_seed) != GW_OK) {
            std::cerr << "simulate: " << gw_last_error() << std::endl;
            return 1;
        }
        ht.tableChanged();
        // the phenotype as a CaseControlSet of the host side (what a caller of the reference holds)
        const uint32_t nb = gw_stream_blocks((uint32_t)sim_n);
        std::vector<uint16_t> cs(nb), ct(nb);
        uint32_t nc = 0, nt = 0;
        if (gw_table_get_case_control(h, cs.data(), ct.data(), nb, &nc, &nt) != GW_OK) { std::cerr << gw_last_error() << std::endl; return 1; }
        std::set<int> cases, ctrls;
        for (long c = 0; c < sim_n; ++c) {
            if (cs[c >> 4] >> (c & 15) & 1) cases.insert(cases.end(), (int)c);
            if (ct[c >> 4] >> (c & 15) & 1) ctrls.insert(ctrls.end(), (int)c);
        }
        gd.setCaseControlColumns(cases, ctrls);
    } else if (!readTfamIndividuals(tfam, &gd) ||
               !(sharded_pairs ? readTpedGenotypes(tped, &gd, '\t', 0, -1, /*ids_only=*/shard.rank != 0)
                 : sharded     ? readTpedGenotypes(tped, &gd, '\t', (long)first, (long)(last - first))
                               : readTpedGenotypes(tped, &gd)) ||
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
        hip->setMaskCheckEveryCall(mask_check_every_call);
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
    } else if (mode == "--test-boost-epi" && sharded) {
        computeBoostSharded(&gd, shard, out);
    } else if (mode == "--test-boost-epi") {
        computeBoost(&gd, out);
    } else if (mode == "--time-boost") {
        // computeBoost end to end (epistasis_func.cpp:349-506), phase by phase, on the drop-in table: selectCaseControl +
        // computeMargins, the pre-screen of all pairs, stage 2 (computeGTest) on the device -- and stage 2 once more on one
        // host thread over device-fetched tables (round 3's form), the baseline the device form is stated against.
        HipGenotypeTable &ht = static_cast<HipGenotypeTable &>(gt);
        CaseControlSet &ccs = *gd.getCaseControlSet();
        auto seconds = [](std::chrono::steady_clock::time_point t0) {
            return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        };
        const int nIndivids = (int)(ccs.getCaseCount() + ccs.getControlCount());
        double t_margins = 0, t_screen = 0, t_stage2_dev = 0, t_stage2_host = 0;
        size_t n_hits = 0, n_over = 0;
        double max_dev_host = 0.0;
        for (int pass = 0; pass < 2; ++pass) {   // pass 0 warms (allocations, first launches), pass 1 is reported
            // the masks handed over again: compaction, margins and the tiled copy are rebuilt inside the timed region
            gw_table *h = ht.handle();
            const uint32_t nb = gw_stream_blocks((uint32_t)gt.column_size());
            if (gw_table_set_case_control(h, ccs.stream_case_begin(), ccs.stream_control_begin(), nb, ccs.getCaseCount(),
                                          ccs.getControlCount()) != GW_OK) { std::cerr << gw_last_error() << std::endl; return 1; }
            ht.tableChanged();
            auto t0 = std::chrono::steady_clock::now();
            gt.selectCaseControl(ccs);
            marginal_information *pMargins = nullptr;
            int nMarkerCount = 0;
            computeMargins(gt, nIndivids, pMargins, nMarkerCount);
            t_margins = seconds(t0);
            t0 = std::chrono::steady_clock::now();
            std::vector<SNPInteractionPair> passing;
            ht.boostScreen(30.0, passing);
            t_screen = seconds(t0);
            std::vector<SNPInteractionPair> on_dev(passing), on_host(passing);
            std::vector<double> z_dev, z_host;
            t0 = std::chrono::steady_clock::now();
            ht.setHostGTest(false);
            computeGTest(gt, pMargins, (uint)nIndivids, on_dev, z_dev);
            t_stage2_dev = seconds(t0);
            t0 = std::chrono::steady_clock::now();
            ht.setHostGTest(true);
            computeGTest(gt, pMargins, (uint)nIndivids, on_host, z_host);
            t_stage2_host = seconds(t0);
            ht.setHostGTest(false);
            n_hits = passing.size();
            n_over = 0;
            max_dev_host = 0.0;
            for (size_t k = 0; k < n_hits; ++k) {
                if (on_dev[k].second > 30.0) ++n_over;
                const double dm = std::fabs(on_dev[k].second - on_host[k].second), dz = std::fabs(z_dev[k] - z_host[k]);
                if (dm == dm && dm > max_dev_host) max_dev_host = dm;
                if (dz == dz && dz > max_dev_host) max_dev_host = dz;
            }
            delete[] pMargins;
        }
        char line[768];
        std::snprintf(line, sizeof line,
                      "{\"mode\": \"time-boost\", \"n_samples\": %d, \"n_snps\": %d, \"pairs\": %.0f, \"hits_over_30\": %zu, "
                      "\"hits_over_30_after_stage2\": %zu, \"select_and_margins_s\": %.6f, \"screen_s\": %.6f, \"stage2_device_s\": %.6f, "
                      "\"stage2_one_host_thread_s\": %.6f, \"compute_boost_end_to_end_s\": %.6f, \"max_abs_diff_device_vs_host_stage2\": %.3g}",
                      (int)gt.column_size(), M, 0.5 * (double)M * (double)(M - 1), n_hits, n_over, t_margins, t_screen, t_stage2_dev,
                      t_stage2_host, t_margins + t_screen + t_stage2_dev, max_dev_host);
        *out << line << std::endl;
    } else if (mode == "--time-inline-cc-maf") {
        // What a libgwaspp maintainer gets from the drop-in table: the reference's own harness loop walking the
        // per-row getter, served from windowed device scans -- beside the kernel rate of the same table.
        HipGenotypeTable &ht = static_cast<HipGenotypeTable &>(gt);
        CaseControlSet &ccs = *gd.getCaseControlSet();
        auto seconds = [](std::chrono::steady_clock::time_point t0) {
            return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        };
        double walk[2] = {0, 0};
        for (int pf = 1; pf >= 0; --pf) {
            ht.setPrefetch(pf != 0);
            inline_cc_maf(&gd, nullptr);   // warm: pinned / pageable buffers, first-touch of the caches
            ht.tableChanged();
            const auto t0 = std::chrono::steady_clock::now();
            inline_cc_maf(&gd, nullptr);
            walk[pf] = seconds(t0);
            ht.tableChanged();
        }
        // the walk's counts against one batch scan of the whole table
        std::vector<uint32_t> counts((size_t)M * GW_N_COUNTS);
        gw_table *h = ht.handle();
        if (gw_scan_case_control(h, 0, (uint64_t)M, counts.data(), nullptr) != GW_OK) { std::cerr << gw_last_error() << std::endl; return 1; }
        ht.setPrefetch(true);
        CaseControlGenotypeDistribution ccgd;
        bool same = true;
        for (int i = 0; i < M && same; ++i) {
            gt.getCaseControlGenotypeDistribution(i, ccs, ccgd);
            const frequency_table &ca = *ccgd.getCaseDistribution(), &co = *ccgd.getControlDistribution();
            const uint32_t *c = &counts[(size_t)i * GW_N_COUNTS];
            same = ca.aa == c[0] && ca.ab == c[1] && ca.bb == c[2] && ca.xx == c[3] && co.aa == c[4] && co.ab == c[5] &&
                   co.bb == c[6] && co.xx == c[7];
        }
        // kernel time of the same scan, records left in HBM (C-ABI device form, HIP events)
        void *d_rec = nullptr;
        float ms = 0.0f;
        const int reps = 10;
        h = ht.handle();
        if (gw_device_malloc(device, (size_t)M * 96u, &d_rec) != GW_OK) { std::cerr << gw_last_error() << std::endl; return 1; }
        char *d_stats = static_cast<char *>(d_rec) + (size_t)M * 32u;
        gw_scan_case_control_device(h, 0, (uint64_t)M, d_rec, d_stats, nullptr);
        gw_timer_start(h, nullptr);
        for (int k = 0; k < reps; ++k) gw_scan_case_control_device(h, 0, (uint64_t)M, d_rec, d_stats, nullptr);
        if (gw_timer_stop(h, nullptr, &ms) != GW_OK) { std::cerr << gw_last_error() << std::endl; return 1; }
        gw_device_free(device, d_rec);
        char buf[1024];
        std::snprintf(buf, sizeof buf,
                      "{\"harness\": \"inline_cc_maf over HipGenotypeTable (maf_func.cpp:292-316)\", \"n_samples\": %d, "
                      "\"n_snps\": %d, \"scan_window_rows\": %d, \"walk_seconds_prefetch\": %.6f, "
                      "\"walk_seconds_no_prefetch\": %.6f, \"snps_per_s_prefetch\": %.1f, \"snps_per_s_no_prefetch\": %.1f, "
                      "\"kernel_ms\": %.6f, \"kernel_snps_per_s\": %.1f, \"kernel\": \"%s\", \"walk_equals_batch_scan\": %s}",
                      N, M, scan_window > 0 ? scan_window : (1 << 16), walk[1], walk[0], M / walk[1], M / walk[0], ms / reps,
                      M / (ms / reps * 1e-3), gw_last_scan_kernel(h), same ? "true" : "false");
        *out << buf << std::endl;
        if (!same) return 3;
    } else if (mode == "--mask-edit-selftest") {
        // The reference's CaseControlSet has no change counter: a set edited IN PLACE (same object, same counts) must
        // still reach the device.  Expected counts come from one batch scan per mask state through the C-ABI.
        HipGenotypeTable &ht = static_cast<HipGenotypeTable &>(gt);
        CaseControlSet &ccs = *gd.getCaseControlSet();
        ht.setScanWindow(scan_window > 0 ? (uint)scan_window : 64u);
        std::set<int> cases, ctrls;
        for (int c = 0; c < N; ++c) {
            if (ccs.isCase((uint)c)) cases.insert(c);
            else if (ccs.isControl((uint)c)) ctrls.insert(c);
        }
        // expected counts of a mask state: installed straight through the C-ABI (not through the table object), one
        // batch scan; the caller puts the first state back afterwards so that device and table object agree again
        auto batch = [&](const std::set<int> &cs_, const std::set<int> &ct_, std::vector<uint32_t> &v) {
            std::vector<int32_t> ic(cs_.begin(), cs_.end()), it(ct_.begin(), ct_.end());
            const uint32_t nb = gw_stream_blocks((uint32_t)N);
            std::vector<uint16_t> mc(nb), mt(nb);
            uint32_t nc = 0, nt = 0;
            v.resize((size_t)M * GW_N_COUNTS);
            gw_table *h = ht.handle();
            return gw_build_stream_mask(ic.data(), ic.size(), (uint32_t)N, mc.data(), &nc) == GW_OK &&
                   gw_build_stream_mask(it.data(), it.size(), (uint32_t)N, mt.data(), &nt) == GW_OK &&
                   gw_table_set_case_control(h, mc.data(), mt.data(), nb, nc, nt) == GW_OK &&
                   gw_scan_case_control(h, 0, (uint64_t)M, v.data(), nullptr) == GW_OK;
        };
        auto row_equals = [&](int i, const std::vector<uint32_t> &v, CaseControlGenotypeDistribution &d) {
            const frequency_table &ca = *d.getCaseDistribution(), &co = *d.getControlDistribution();
            const uint32_t *c = &v[(size_t)i * GW_N_COUNTS];
            return ca.aa == c[0] && ca.ab == c[1] && ca.bb == c[2] && ca.xx == c[3] && co.aa == c[4] && co.ab == c[5] &&
                   co.bb == c[6] && co.xx == c[7];
        };
        CaseControlGenotypeDistribution d;
        std::vector<uint32_t> before, after;
        if (cases.empty() || ctrls.empty()) { *out << "FAIL needs both groups" << std::endl; return 3; }
        // second state: one member moves from the cases to the controls and one back -- both counts stay what they were
        std::set<int> cases2(cases), ctrls2(ctrls);
        const int a = *cases.begin(), b = *ctrls.rbegin();
        cases2.erase(a); ctrls2.erase(b); cases2.insert(b); ctrls2.insert(a);
        if (!batch(cases2, ctrls2, after) || !batch(cases, ctrls, before)) { *out << "FAIL batch scans" << std::endl; return 3; }
        gt.getCaseControlGenotypeDistribution(0, ccs, d);          // installs the set (first state)
        if (!row_equals(0, before, d)) { *out << "FAIL first walk" << std::endl; return 3; }
        ccs.setCases(cases2);                                      // edited in place
        ccs.setControls(ctrls2);
        bool ok = true;
        if (mask_check_every_call) {  // every byte compared on every call: the very next getter sees the edit
            gt.getCaseControlGenotypeDistribution(1, ccs, d);
            ok = row_equals(1, after, d);
            *out << (ok ? "PASS" : "FAIL") << " edit seen by the next getter call (every-call check)" << std::endl;
        } else {
            // (1) a new window of rows compares every byte
            const int w = scan_window > 0 ? scan_window : 64;
            if (M > w) {
                gt.getCaseControlGenotypeDistribution((uint)w, ccs, d);
                const bool seen = row_equals(w, after, d);
                *out << (seen ? "PASS" : "FAIL") << " edit seen at the next window" << std::endl;
                ok = ok && seen;
            }
            // (2) selectCaseControl compares every byte, whatever was sampled
            ccs.setCases(cases);
            ccs.setControls(ctrls);
            gt.selectCaseControl(ccs);
            gt.getCaseControlGenotypeDistribution(0, d);
            const bool back = row_equals(0, before, d);
            *out << (back ? "PASS" : "FAIL") << " edit seen by selectCaseControl" << std::endl;
            ok = ok && back;
            // (3) ADVICE r03: a relabelling with EQUAL group sizes, then a second walk that starts over at row 0 of a
            // table that fits one window (no new window, no selectCaseControl in between): the walk's first call is not
            // row + 1 of the previous one, so every mask byte is compared and the whole walk sees the new labels --
            // wherever in the mask the changed individuals sit
            ht.setScanWindow(1u << 16);
            std::set<int> cases3(cases), ctrls3(ctrls);
            {   // swap a case and a control from the MIDDLE of the cohort (not among the sampled words' neighbours)
                std::set<int>::const_iterator ic = cases.begin(), it = ctrls.begin();
                std::advance(ic, (long)cases.size() / 2 + 3);
                std::advance(it, (long)ctrls.size() / 2 + 5);
                cases3.erase(*ic); ctrls3.erase(*it); cases3.insert(*it); ctrls3.insert(*ic);
            }
            std::vector<uint32_t> third;
            if (!batch(cases3, ctrls3, third) || !batch(cases, ctrls, before)) { *out << "FAIL batch scans" << std::endl; return 3; }
            bool walk1 = true, walk2 = true;
            for (int i = 0; i < M; ++i) {                           // first walk, first labelling
                gt.getCaseControlGenotypeDistribution((uint)i, ccs, d);
                walk1 = walk1 && row_equals(i, before, d);
            }
            ccs.setCases(cases3);                                  // same object, same counts, other members
            ccs.setControls(ctrls3);
            for (int i = 0; i < M; ++i) {                           // second walk from row 0, no call in between
                gt.getCaseControlGenotypeDistribution((uint)i, ccs, d);
                walk2 = walk2 && row_equals(i, third, d);
            }
            *out << (walk1 && walk2 && third != before ? "PASS" : "FAIL")
                 << " relabelling with equal group sizes seen by a second walk from row 0 inside one window" << std::endl;
            ok = ok && walk1 && walk2 && third != before;
        }
        const bool differ = after != before;
        *out << (differ ? "PASS" : "FAIL") << " the two mask states give different counts" << std::endl;
        if (!ok || !differ) return 3;
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
