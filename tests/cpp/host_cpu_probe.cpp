// host_cpu_probe.cpp -- exercises the GPU-free parts of the C++ mirror
// (libgwaspp_amd/host) and prints what they produce, one record per line, for
// tests/test_host_mirror.py to compare with the oracle and oracle/_ref.
//   host_cpu_probe ccs <n> <case idx,...> <ctrl idx,...>
//   host_cpu_probe marginal <8 counts>
//   host_cpu_probe maf <4 counts>
//   host_cpu_probe gtest <9 case cells> <9 control cells>   (3x3 tables, no missing calls)
//   host_cpu_probe llr <9 case cells> <9 control cells>     pairwise_epi_test + chi-square(4) upper tail
#include <cstdio>
#include <cstdlib>
#include <set>
#include <sstream>
#include <string>

#include "algorithms/epistasis_func.h"
#include "algorithms/maf_func.h"
#include "genetics/analyzable/case_control_set.h"
#include "genetics/genotype/common_genotype_func.h"

using namespace libgwaspp::genetics;

static std::set<int> parse_set(const char *s) {
    std::set<int> out;
    std::stringstream ss(s);
    std::string tok;
    while (std::getline(ss, tok, ','))
        if (!tok.empty() && tok != "-") out.insert(std::atoi(tok.c_str()));
    return out;
}

static void dump(const char *name, const ushort *b, const ushort *e) {
    std::printf("%s", name);
    for (; b != e; ++b) std::printf(" %04x", (unsigned)*b);
    std::printf("\n");
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    std::string mode = argv[1];
    if (mode == "ccs" && argc == 5) {
        util::SequentialIndexer cols(std::atoi(argv[2]));
        CaseControlSet ccs(&cols);
        ccs.setCases(parse_set(argv[3]));
        ccs.setControls(parse_set(argv[4]));
        dump("block_case", ccs.case_begin(), ccs.case_end());
        dump("block_ctrl", ccs.control_begin(), ccs.control_end());
        dump("stream_case", ccs.stream_case_begin(), ccs.stream_case_end());
        dump("stream_ctrl", ccs.stream_control_begin(), ccs.stream_control_end());
        std::printf("counts %u %u %u\n", ccs.getCaseCount(), ccs.getControlCount(), ccs.getTotalCount());
        ccs.setAllAsCases();
        dump("all_block_case", ccs.case_begin(), ccs.case_end());
        dump("all_stream_case", ccs.stream_case_begin(), ccs.stream_case_end());
        return 0;
    }
    if (mode == "marginal" && argc == 10) {
        frequency_table ca, co;
        for (int i = 0; i < 4; ++i) { ca.freq[i] = std::atoi(argv[2 + i]); co.freq[i] = std::atoi(argv[6 + i]); }
        uint n = 0;
        for (int i = 0; i < 4; ++i) n += ca.freq[i] + co.freq[i];
        marginal_information m;
        for (int i = 0; i < 8; ++i) m.dPbc[i] = m.dPca[i] = 0.0;
        computeMarginalInformation(ca, co, n, m);
        std::printf("%u %u %u %u %.17g %.17g", m.margins.aa, m.margins.ab, m.margins.bb, m.margins.xx,
                    m.dMarginalEntropy, m.dMarginalEntropy_Y);
        for (int i = 0; i < 8; ++i) std::printf(" %.17g", m.dPbc[i]);
        for (int i = 0; i < 8; ++i) std::printf(" %.17g", m.dPca[i]);
        std::printf("\n");
        return 0;
    }
    if (mode == "gtest" && argc == 20) {
        CONTIN_TABLE_T ca, co;
        ResetContingencyTable(ca);
        ResetContingencyTable(co);
        frequency_table f1c, f1t, f2c, f2t;
        ResetFrequencyTable(f1c); ResetFrequencyTable(f1t); ResetFrequencyTable(f2c); ResetFrequencyTable(f2t);
        uint n = 0;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                const uint x = std::atoi(argv[2 + 3 * a + b]), y = std::atoi(argv[11 + 3 * a + b]);
                ca.contin[4 * a + b] = x;
                co.contin[4 * a + b] = y;
                f1c.freq[a] += x; f2c.freq[b] += x;
                f1t.freq[a] += y; f2t.freq[b] += y;
                n += x + y;
            }
        marginal_information m1 = marginal_information(), m2 = marginal_information();
        computeMarginalInformation(f1c, f1t, n, m1);
        computeMarginalInformation(f2c, f2t, n, m2);
        double measure, z;
        libgwaspp::algorithms::gtestOnePair(ca, co, m1, m2, n, measure, z);
        std::printf("%.17g %.17g %.17g\n", libgwaspp::algorithms::boostMeasureOnePair(ca, co, m1, m2, (int)n), measure, z);
        return 0;
    }
    if (mode == "llr" && argc == 20) {
        CONTIN_TABLE_T ca, co;
        ResetContingencyTable(ca);
        ResetContingencyTable(co);
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                ca.contin[4 * a + b] = std::atoi(argv[2 + 3 * a + b]);
                co.contin[4 * a + b] = std::atoi(argv[11 + 3 * a + b]);
            }
        ca.contin[15] = 7;  // missing-call cells are outside the test
        co.contin[3] = 5;
        const double ll = libgwaspp::algorithms::pairwise_epi_test(ca, co);
        std::printf("%.17g %.17g\n", ll, libgwaspp::algorithms::chisq4_upper_tail(ll));
        return 0;
    }
    if (mode == "maf" && argc == 6) {
        frequency_table ft;
        for (int i = 0; i < 4; ++i) ft.freq[i] = std::atoi(argv[2 + i]);
        double tot, maf;
        libgwaspp::algorithms::MinorAlleleFrequency(ft, tot, maf);
        std::printf("%.17g %.17g\n", tot, maf);
        return 0;
    }
    return 2;
}
