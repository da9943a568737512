// maf_func.cpp -- see algorithms/maf_func.h.  Each harness is the reference's
// loop over the SNP rows; the per-row getters it calls are served from batched
// GPU scans inside HipGenotypeTable, so the "i <lapse>" line of the first row
// of a scan window carries the whole window's kernel time.
#include "algorithms/maf_func.h"

#include <chrono>

namespace libgwaspp {
namespace algorithms {

namespace {
struct Lapse {  // INIT_LAPSE_TIME / RECORD_START / RECORD_STOP / PRINT_LAPSE (util/time/timing.h:68-109)
    std::chrono::steady_clock::time_point t0;
    void start() { t0 = std::chrono::steady_clock::now(); }
    double stop() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
void line(std::ostream *out, int i, double seconds) {
    if (out) *out << i << "\t" << seconds << std::endl;
}
}  // namespace

void compute_maf_perform(GeneticData *gd, std::ostream *) {
    GenoTable &gt = *gd->getGenotypeTable();
    GenotypeDistribution dist;
    const frequency_table &ft = *dist.getDistribution();
    double tot, maf;
    for (int i = 0, m = gd->getGenotypedMarkersCount(); i < m; ++i) {
        gt.getGenotypeDistribution(i, dist);
        MinorAlleleFrequency(ft, tot, maf);
    }
}

void select_cc_maf(GeneticData *gd, std::ostream *out) {
    GenoTable &gt = *gd->getGenotypeTable();
    CaseControlGenotypeDistribution ccgd;
    const frequency_table &cases = *ccgd.getCaseDistribution(), &ctrls = *ccgd.getControlDistribution();
    double tot, maf;
    Lapse t;
    t.start();
    gt.selectCaseControl(*gd->getCaseControlSet());
    line(out, -1, t.stop());
    for (int i = 0, m = gd->getGenotypedMarkersCount(); i < m; ++i) {
        t.start();
        gt.getCaseControlGenotypeDistribution(i, ccgd);
        line(out, i, t.stop());
        MinorAlleleFrequency(cases, tot, maf);
        MinorAlleleFrequency(ctrls, tot, maf);
    }
}

void genotype_dist_performance(GeneticData *gd, std::ostream *out) {
    GenoTable &gt = *gd->getGenotypeTable();
    GenotypeDistribution dist;
    Lapse t;
    for (int i = 0, m = gd->getGenotypedMarkersCount(); i < m; ++i) {
        t.start();
        gt.getGenotypeDistribution(i, dist);
        line(out, i, t.stop());
    }
}

void inline_cc_maf(GeneticData *gd, std::ostream *out) {
    GenoTable &gt = *gd->getGenotypeTable();
    CaseControlGenotypeDistribution ccgd;
    const frequency_table &cases = *ccgd.getCaseDistribution(), &ctrls = *ccgd.getControlDistribution();
    double tot, maf;
    CaseControlSet &ccs = *gd->getCaseControlSet();
    Lapse t;
    for (int i = 0, m = gd->getGenotypedMarkersCount(); i < m; ++i) {
        t.start();
        gt.getCaseControlGenotypeDistribution(i, ccs, ccgd);
        line(out, i, t.stop());
        MinorAlleleFrequency(cases, tot, maf);
        MinorAlleleFrequency(ctrls, tot, maf);
    }
}

void inline_maf_print(GeneticData *gd, std::ostream *out) {
    GenoTable &gt = *gd->getGenotypeTable();
    const int individ_count = gd->getGenotypedIndividualsCount();
    GenotypeDistribution dist;
    const frequency_table &ft = *dist.getDistribution();
    double tot, maf;
    for (int i = 0, m = gd->getGenotypedMarkersCount(); i < m; ++i) {
        gt.getGenotypeDistribution(i, dist);
        MinorAlleleFrequency(ft, tot, maf);
        // the missing count is printed as N - tot, not ft.xx (maf_func.cpp:332)
        *out << (int)(individ_count - tot) << "\t" << ft.aa << "\t" << ft.ab << "\t" << ft.bb << std::endl;
    }
}

void inline_cc_maf_print(GeneticData *gd, std::ostream *case_out, std::ostream *ctrl_out) {
    GenoTable &gt = *gd->getGenotypeTable();
    CaseControlGenotypeDistribution ccgd;
    const frequency_table &ca = *ccgd.getCaseDistribution(), &co = *ccgd.getControlDistribution();
    CaseControlSet &ccs = *gd->getCaseControlSet();
    for (int i = 0, m = gd->getGenotypedMarkersCount(); i < m; ++i) {
        gt.getCaseControlGenotypeDistribution(i, ccs, ccgd);
        *case_out << ca.xx << "\t" << ca.aa << "\t" << ca.ab << "\t" << ca.bb << std::endl;
        *ctrl_out << co.xx << "\t" << co.aa << "\t" << co.ab << "\t" << co.bb << std::endl;
    }
}

}  // namespace algorithms
}  // namespace libgwaspp
