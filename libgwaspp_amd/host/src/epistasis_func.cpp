// epistasis_func.cpp -- see algorithms/epistasis_func.h.
#include "algorithms/epistasis_func.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>

#include "genetics/genotype/hip_genotype_table.h"

namespace libgwaspp {
namespace algorithms {

void computeMargins(GenoTable &gt, int, marginal_information *&pMargins, int &nMarkerCount) {
    nMarkerCount = gt.row_size();
    delete[] pMargins;
    pMargins = new marginal_information[nMarkerCount]();  // zeroed: entries of empty classes stay 0 (quirk Q6)
    CaseControlGenotypeDistribution ccgd;
    for (int i = 0; i < nMarkerCount; ++i) gt.getCaseControlGenotypeDistribution(i, ccgd, pMargins[i]);
}

double boostMeasureOnePair(const CONTIN_TABLE_T &ca, const CONTIN_TABLE_T &co, const marginal_information &m1,
                           const marginal_information &m2, int nIndivids) {
    double tao = 0.0, interMeasure = 0.0;
    for (int a = 0; a < GT_COUNT; ++a)
        for (int b = 0; b < GT_COUNT; ++b) {
            const uint x = ca.contin[CONTIN_COLUMN_COUNT * a + b], y = co.contin[CONTIN_COLUMN_COUNT * a + b];
            const double dPab = (double)(x + y) / (double)m2.margins.freq[b];
            const double tmp2 = (dPab) * (m2.dPbc[b]) * (m1.dPca[a]);
            const double tmp3 = (dPab) * (m2.dPbc[GENOTYPE_COUNT + b]) * (m1.dPca[GENOTYPE_COUNT + a]);
            tao += tmp2 + tmp3;
            if (x > 0) {
                const double tmp1 = (double)x / nIndivids;
                interMeasure += tmp1 * std::log(tmp1);
                if (tmp2 > 0) interMeasure += -tmp1 * std::log(tmp2);
            }
            if (y > 0) {
                const double tmp1 = (double)y / nIndivids;
                interMeasure += tmp1 * std::log(tmp1);
                if (tmp3 > 0) interMeasure += -tmp1 * std::log(tmp3);
            }
        }
    return (interMeasure + std::log(tao)) * nIndivids * 2.0;
}

void gtestOnePair(const CONTIN_TABLE_T &contin_ca, const CONTIN_TABLE_T &contin_co, const marginal_information &m1,
                  const marginal_information &m2, uint nIndivids, double &interMeasure, double &z) {
    const int J = GT_COUNT * GT_COUNT;
    double mu[2][J], mu0[2][J];  // [0] cases, [1] controls
    for (int i = 0; i < J; ++i) { mu[0][i] = mu[1][i] = 1.0; mu0[0][i] = mu0[1][i] = 0.0; }
    // the reference sums |mu - mu0| eighteen times over the first element (:552-556)
    double muError = 2.0 * J * std::fabs(mu[0][0] - mu0[0][0]);
    while (muError > 0.001) {
        std::memcpy(mu0, mu, sizeof mu);
        double mu_ik[GT_BUFFER_SIZE] = {0}, mu_jk[GT_BUFFER_SIZE] = {0};
        for (int a = 0; a < GT_COUNT; ++a)
            for (int b = 0; b < GT_COUNT; ++b) {
                const int i = GT_COUNT * a + b;
                const uint tot = contin_ca.contin[CONTIN_COLUMN_COUNT * a + b] + contin_co.contin[CONTIN_COLUMN_COUNT * a + b];
                const double mu_ij = mu[0][i] + mu[1][i];
                if (mu_ij > 0) {
                    mu[0][i] = mu[0][i] * (tot) / mu_ij;
                    mu[1][i] = mu[1][i] * (tot) / mu_ij;
                } else {
                    mu[0][i] = 0;
                    mu[1][i] = 0;
                }
                mu_ik[a] += mu[0][i];
                mu_ik[GT_COUNT + a] += mu[1][i];
                mu_jk[b] += mu[0][i];
                mu_jk[GT_COUNT + b] += mu[1][i];
            }
        muError = 0.0;
        for (int a = 0; a < GT_COUNT; ++a) {
            const double tmp1 = mu_ik[a] > 0 ? m1.cases.freq[a] / mu_ik[a] : 0.0;
            const double tmp2 = mu_ik[GT_COUNT + a] > 0 ? m1.controls.freq[a] / mu_ik[GT_COUNT + a] : 0.0;
            for (int b = 0; b < GT_COUNT; ++b) {
                const int i = GT_COUNT * a + b;
                const double tmp3 = mu_jk[b] > 0 ? m2.cases.freq[b] / mu_jk[b] : 0.0;
                const double tmp4 = mu_jk[GT_COUNT + b] > 0 ? m2.controls.freq[b] / mu_jk[GT_COUNT + b] : 0.0;
                mu[0][i] = mu[0][i] * (tmp1)*tmp3;
                mu[1][i] = mu[1][i] * (tmp2)*tmp4;
                muError += std::fabs(mu[0][i] - mu0[0][i]);
                muError += std::fabs(mu[1][i] - mu0[1][i]);
            }
        }
    }
    double tao = 0.0;
    interMeasure = 0.0;
    for (int a = 0; a < GT_COUNT; ++a)
        for (int b = 0; b < GT_COUNT; ++b) {
            const int i = GT_COUNT * a + b;
            const uint x = contin_ca.contin[CONTIN_COLUMN_COUNT * a + b], y = contin_co.contin[CONTIN_COLUMN_COUNT * a + b];
            double tmp1, tmp2;
            if (x > 0) { tmp1 = (double)x / nIndivids; interMeasure += tmp1 * std::log(tmp1); } else tmp1 = 0.0;
            if (mu[0][i] > 0) { tmp2 = mu[0][i] / nIndivids; interMeasure += -tmp1 * std::log(tmp2); tao += tmp2; }
            if (y > 0) { tmp1 = (double)y / nIndivids; interMeasure += tmp1 * std::log(tmp1); } else tmp1 = 0.0;
            if (mu[1][i] > 0) { tmp2 = mu[1][i] / nIndivids; interMeasure += -tmp1 * std::log(tmp2); tao += tmp2; }
        }
    interMeasure = (interMeasure + std::log(tao)) * nIndivids * 2.0;

    // allele-level 2x2 per group and the difference of the log odds ratios (:686-701); uint arithmetic
    uint aj[8];
    const uint *t = contin_ca.contin;
    for (int g = 0; g < 2; ++g, t = contin_co.contin) {
        aj[4 * g + 0] = (t[0] << 2) + (t[1] << 1) + (t[4] << 1) + t[5];
        aj[4 * g + 1] = (t[2] << 2) + (t[1] << 1) + (t[6] << 1) + t[5];
        aj[4 * g + 2] = (t[8] << 2) + (t[9] << 1) + (t[4] << 1) + t[5];
        aj[4 * g + 3] = (t[10] << 2) + (t[9] << 1) + (t[6] << 1) + t[5];
    }
    const double or_aff = std::log((double)(aj[0] * aj[3]) / (double)(aj[1] * aj[2]));
    const double v_aff = 1 / (double)aj[0] + 1 / (double)aj[1] + 1 / (double)aj[2] + 1 / (double)aj[3];
    const double or_unf = std::log((double)(aj[4] * aj[7]) / (double)(aj[5] * aj[6]));
    const double v_unf = 1 / (double)aj[4] + 1 / (double)aj[5] + 1 / (double)aj[6] + 1 / (double)aj[7];
    z = (or_aff - or_unf) / std::sqrt(v_aff + v_unf);
}

void computeGTest(GenoTable &gt, marginal_information *pMargins, uint nIndivids,
                  std::vector<SNPInteractionPair> &passingThreshold, std::vector<double> &zval) {
    HipGenotypeTable *hip = dynamic_cast<HipGenotypeTable *>(&gt);
    if (hip && !hip->hostGTest()) {
        // stage 2 on the device (round 4): tables, proportional fitting, measure and z for every pair in two launches
        hip->boostGTest(passingThreshold, zval);
        return;
    }
    if (hip) {
        // (setHostGTest: round 3's form, kept as the timed baseline) the tables of all passing pairs in one device
        // launch instead of one per pair, the fitting on this thread
        std::vector<uint> pairs;
        pairs.reserve(2 * passingThreshold.size());
        for (const SNPInteractionPair &p : passingThreshold) {
            pairs.push_back(p.first.first);
            pairs.push_back(p.first.second);
        }
        std::vector<CONTIN_TABLE_T> cases, ctrls;
        hip->getSelectedCaseControlContingencyTables(pairs.data(), passingThreshold.size(), cases, ctrls);
        for (size_t k = 0; k < passingThreshold.size(); ++k) {
            SNPInteractionPair &p = passingThreshold[k];
            double z;
            gtestOnePair(cases[k], ctrls[k], pMargins[p.first.first], pMargins[p.first.second], nIndivids, p.second, z);
            zval.push_back(z);
        }
        return;
    }
    CaseControlContingencyTable ccct;
    for (SNPInteractionPair &p : passingThreshold) {
        const uint i = p.first.first, j = p.first.second;
        gt.getCaseControlContingencyTable(i, j, pMargins[i], pMargins[j], ccct);
        double z;
        gtestOnePair(*ccct.getCaseContingencyTable(), *ccct.getControlContingencyTable(), pMargins[i], pMargins[j],
                     nIndivids, p.second, z);
        zval.push_back(z);
    }
}

void computeBoost(GeneticData *gd, std::ostream *out) {
    CaseControlSet &ccs = *gd->getCaseControlSet();
    const int nIndivids = (int)(ccs.getCaseCount() + ccs.getControlCount());
    GenoTable &gt = *gd->getGenotypeTable();
    gt.selectCaseControl(ccs);

    marginal_information *pMargins = nullptr;
    int nMarkerCount = 0;
    computeMargins(gt, nIndivids, pMargins, nMarkerCount);
    const double thresholdRecord = 30.0;  // :390
    *out << "Pre-screening " << nMarkerCount << " SNP interactions" << std::endl;

    std::vector<SNPInteractionPair> passingThreshold;
    const auto t0 = std::chrono::steady_clock::now();
    HipGenotypeTable *hip = dynamic_cast<HipGenotypeTable *>(&gt);
    if (hip) {
        // the whole O(M^2) loop (:393-486) is one device pass
        hip->boostScreen(thresholdRecord, passingThreshold);
    } else {
        CaseControlContingencyTable ccct;
        for (int i = 0; i < nMarkerCount; ++i)
            for (int j = i + 1; j < nMarkerCount; ++j) {
                gt.getCaseControlContingencyTable(i, j, pMargins[i], pMargins[j], ccct);
                const double m = boostMeasureOnePair(*ccct.getCaseContingencyTable(), *ccct.getControlContingencyTable(),
                                                     pMargins[i], pMargins[j], nIndivids);
                if (m > thresholdRecord) passingThreshold.push_back(SNPInteractionPair(SNPPair(i, j), m));
            }
    }
    *out << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << std::endl;
    boostStage2AndReport(gt, pMargins, nIndivids, passingThreshold, out);
    delete[] pMargins;
}

void boostStage2AndReport(GenoTable &gt, marginal_information *pMargins, int nIndivids,
                          std::vector<SNPInteractionPair> &passingThreshold, std::ostream *out) {
    const double thresholdRecord = 30.0;  // :390
    *out << "Located " << passingThreshold.size() << " potential interactions" << std::endl;
    *out << "Performing deeper analysis of SNPs" << std::endl;

    std::vector<double> zval;
    computeGTest(gt, pMargins, (uint)nIndivids, passingThreshold, zval);
    uint idx = 0;
    char line[256];
    for (size_t k = 0; k < passingThreshold.size(); ++k) {
        if (passingThreshold[k].second > thresholdRecord) {
            std::snprintf(line, sizeof line, "%7d\t%7d\t%7d\t%f\t%f\t%f\t%f", (int)idx++,
                          (int)passingThreshold[k].first.first, (int)passingThreshold[k].first.second, 0.0, 0.0,
                          passingThreshold[k].second, zval[k]);
            *out << line << std::endl;
        }
    }
}

void ContingencyPerformance(GeneticData *gd, std::ostream *out) {
    GenoTable &gt = *gd->getGenotypeTable();
    const int m = gd->getGenotypedMarkersCount();
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long checksum = 0;
    if (HipGenotypeTable *hip = dynamic_cast<HipGenotypeTable *>(&gt)) {
        // the same i < j walk, handed to the device a million pairs at a time
        std::vector<uint> pairs, tables;
        const size_t batch = 1u << 20;
        pairs.reserve(2 * batch);
        auto flush = [&]() {
            if (pairs.empty()) return;
            hip->getContingencyTables(pairs.data(), pairs.size() / 2, tables);
            for (uint v : tables) checksum += v;
            pairs.clear();
        };
        for (int i = 0; i < m; ++i)
            for (int j = i + 1; j < m; ++j) {
                pairs.push_back((uint)i);
                pairs.push_back((uint)j);
                if (pairs.size() == 2 * batch) flush();
            }
        flush();
    } else {
        ContingencyTable ct;
        for (int i = 0; i < m; ++i)
            for (int j = i + 1; j < m; ++j) {
                gt.getContingencyTable(i, j, ct);
                for (int c = 0; c < 16; ++c) checksum += ct.getContingencyTable()->contin[c];
            }
    }
    *out << "Contingencies " << m << ": " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()
         << " (sum of all cells " << checksum << ")" << std::endl;
}

void ContingencyCCPerformance(GeneticData *gd, std::ostream *out) {
    CaseControlSet &ccs = *gd->getCaseControlSet();
    GenoTable &gt = *gd->getGenotypeTable();
    const int nIndivids = (int)(ccs.getCaseCount() + ccs.getControlCount());
    gt.selectCaseControl(ccs);                                    // :151
    marginal_information *pMargins = nullptr;
    int m = 0;
    computeMargins(gt, nIndivids, pMargins, m);                   // :156
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long checksum = 0, n_pairs = 0;
    if (HipGenotypeTable *hip = dynamic_cast<HipGenotypeTable *>(&gt)) {
        // the i < j walk of :187-201, a million pairs per launch
        std::vector<uint> pairs;
        std::vector<CONTIN_TABLE_T> cs, ct;
        const size_t batch = 1u << 20;
        pairs.reserve(2 * batch);
        auto flush = [&]() {
            if (pairs.empty()) return;
            hip->getSelectedCaseControlContingencyTables(pairs.data(), pairs.size() / 2, cs, ct);
            for (size_t k = 0; k < cs.size(); ++k)
                for (int c = 0; c < 16; ++c) checksum += cs[k].contin[c] + ct[k].contin[c];
            n_pairs += cs.size();
            pairs.clear();
        };
        for (int i = 0; i < m; ++i)
            for (int j = i + 1; j < m; ++j) {
                pairs.push_back((uint)i);
                pairs.push_back((uint)j);
                if (pairs.size() == 2 * batch) flush();
            }
        flush();
    } else {
        CaseControlContingencyTable ccct;
        for (int i = 0; i < m; ++i)
            for (int j = i + 1; j < m; ++j, ++n_pairs) {
                gt.getCaseControlContingencyTable(i, j, pMargins[i], pMargins[j], ccct);
                for (int c = 0; c < 16; ++c)
                    checksum += ccct.getCaseContingencyTable()->contin[c] + ccct.getControlContingencyTable()->contin[c];
            }
    }
    delete[] pMargins;
    // every exact table sums to its group: the checksum is pairs x (cases + controls)
    *out << "Case/Control Contingencies " << m << ": "
         << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " (" << n_pairs
         << " pairs, sum of all cells " << checksum << ")" << std::endl;
}

double pairwise_epi_test(const CONTIN_TABLE_T &contin_ca, const CONTIN_TABLE_T &contin_co) {
    // src/test/pairwise.c:50-133; the 3 x 3 cells of the 4 x 4 tables (missing calls are outside the test)
    long s[3][3], t[3][3], both[3][3];
    long s_a[3] = {0, 0, 0}, s_b[3] = {0, 0, 0}, t_a[3] = {0, 0, 0}, t_b[3] = {0, 0, 0}, all_a[3] = {0, 0, 0},
         all_b[3] = {0, 0, 0};
    long ns = 0, nt = 0, n = 0;
    for (int a = 0; a < 3; ++a) {          // step 1: margins (:70-89)
        for (int b = 0; b < 3; ++b) {
            s[a][b] = contin_ca.contin[4 * a + b];
            t[a][b] = contin_co.contin[4 * a + b];
            both[a][b] = s[a][b] + t[a][b];
            s_a[a] += s[a][b];
            s_b[b] += s[a][b];
            t_a[a] += t[a][b];
            t_b[b] += t[a][b];
            all_a[a] += both[a][b];
            all_b[b] += both[a][b];
        }
        ns += s_a[a];
        nt += t_a[a];
        n += all_a[a];
    }
    double ll = 0.0, tao = 0.0;
    for (int a = 0; a < 3; ++a) {          // steps 2 and 3 (:92-131), same operation order
        const double psa = (double)s_a[a] / all_a[a], pta = 1.0 - psa;
        for (int b = 0; b < 3; ++b) {
            const double pab = (double)both[a][b] / all_b[b];
            const double pbs = (double)s_b[b] / ns, pbt = (double)t_b[b] / nt;
            if (s[a][b] > 0) ll += s[a][b] * std::log((double)s[a][b] / n);
            if (t[a][b] > 0) ll += t[a][b] * std::log((double)t[a][b] / n);
            const double ps = pab * pbs * psa, pt = pab * pbt * pta;
            tao += ps + pt;
            if (ps > 0) ll -= s[a][b] * std::log(ps);
            if (pt > 0) ll -= t[a][b] * std::log(pt);
        }
    }
    ll += n * std::log(tao);
    return 2.0 * ll;
}

double chisq4_upper_tail(double x) { return x <= 0.0 ? 1.0 : std::exp(-0.5 * x) * (1.0 + 0.5 * x); }

void EpistasisPerformance(GeneticData *gd, std::ostream *out) {
    GenoTable &gt = *gd->getGenotypeTable();
    const int m = gd->getGenotypedMarkersCount(), n = gd->getGenotypedIndividualsCount();
    CaseControlSet ccs(gt.getColumnSet());                       // :313-324: the harness's own groups
    std::set<int> cases, ctrls;
    for (int i = 0; i < 200; ++i) {
        if ((i << 1) < n) cases.insert(i << 1);
        if ((i << 1) + 1 < n) ctrls.insert((i << 1) + 1);
    }
    ccs.setCases(cases);
    ccs.setControls(ctrls);
    const auto t0 = std::chrono::steady_clock::now();
    double ll_sum = 0.0, p_min = 1.0;
    unsigned long n_pairs = 0, n_nan = 0;
    auto test = [&](const CONTIN_TABLE_T &ca, const CONTIN_TABLE_T &co) {
        const double ll = pairwise_epi_test(ca, co);             // :336-339
        const double pval = chisq4_upper_tail(ll);
        if (ll == ll) { ll_sum += ll; p_min = std::min(p_min, pval); } else ++n_nan;
        ++n_pairs;
    };
    if (HipGenotypeTable *hip = dynamic_cast<HipGenotypeTable *>(&gt)) {
        std::vector<uint> pairs, cs, ct;
        const size_t batch = 1u << 20;
        pairs.reserve(2 * batch);
        auto flush = [&]() {
            if (pairs.empty()) return;
            hip->getCaseControlContingencyTables(pairs.data(), pairs.size() / 2, ccs, cs, ct);
            CONTIN_TABLE_T a, b;
            for (size_t k = 0; k < pairs.size() / 2; ++k) {
                std::memcpy(a.contin, &cs[16 * k], sizeof a.contin);
                std::memcpy(b.contin, &ct[16 * k], sizeof b.contin);
                test(a, b);
            }
            pairs.clear();
        };
        for (int i = 0; i < m; ++i)
            for (int j = i + 1; j < m; ++j) {
                pairs.push_back((uint)i);
                pairs.push_back((uint)j);
                if (pairs.size() == 2 * batch) flush();
            }
        flush();
    } else {
        CaseControlContingencyTable ccct;
        for (int i = 0; i < m; ++i)
            for (int j = i + 1; j < m; ++j) {
                gt.getCaseControlContingencyTable(i, j, ccs, ccct);
                test(*ccct.getCaseContingencyTable(), *ccct.getControlContingencyTable());
            }
    }
    char buf[160];
    std::snprintf(buf, sizeof buf, " (%lu pairs, %lu undefined, sum of LLR %.10g, smallest p %.6g)", n_pairs, n_nan, ll_sum, p_min);
    *out << "Case/Control Contingencies " << m << ": "
         << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << buf << std::endl;
}

}  // namespace algorithms
}  // namespace libgwaspp
