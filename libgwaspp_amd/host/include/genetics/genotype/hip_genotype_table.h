// genetics/genotype/hip_genotype_table.h -- the MI355X-backed GenoTable.
//
// A new subclass of the reference's abstract table (genotype/geno_table.h:47-87)
// next to CompressedGenotypeTable3/4/5: the packed genotypes live in HBM (one
// gw_table handle of include/gwas_scan.h), the per-row getters of
// SingleMarkerAnalyzable are served from BATCHED device scans.  The callers in
// algorithms/maf_func.cpp only ever walk r = 0..M-1 in order
// (maf_func.cpp:304-314), so the first getter call of a window scans the whole
// window on the GPU and the following calls read the host-side result cache.
//
// Semantics kept from CompressedGenotypeTable5 (the 2-bit stream table), incl.
// its per-method xx conventions:
//   getGenotypeDistribution            xx = missing + pad bits   (table5.cpp:600)
//   getCaseControlGenotypeDistribution xx = group - aa - ab - bb (table5.cpp:649,653)
//   getContingencyTable / cc (inline)  xx_xx absorbs non-members and pad bits (table5.cpp:856)
//   cc (selected)                      pad bits of the compacted rows (table5.cpp:896-987)
//   cc (margin-assisted)               exact tables (table5.cpp:989-1150)
// Errors abort, like the reference's asserts.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <future>
#include <vector>

#include "genetics/genotype/geno_table.h"

struct gw_table;  // include/gwas_scan.h
struct gw_marginal_information;
struct gw_comm;

namespace libgwaspp {
namespace genetics {

class HipGenotypeTable : public GenoTable {
public:
    HipGenotypeTable(util::indexer *markers, util::indexer *individs, int device = 0);
    virtual ~HipGenotypeTable();

    // ---- ingest (geno_table.h:51-53)
    void addGenotype(int rIdx, int cIdx, const std::string &gt) override;
    void addGenotypeRow(int rIdx, std::string::const_iterator &it, std::string::const_iterator &it_end, char delim) override;
    void addGenotypeRow(int rIdx, const char *p_begin, const char *p_end, char delim) override;

    ushort encodeGenotype(const std::string &gt) override;
    const char *decodeGenotype(ushort encoded_gt) override;
    bool isGenotypeHomozygous(ushort encoded_gt) override;
    DataBlock operator()(int r, int c) override;

    // ---- single marker (single_marker_analyzable.h:132-145)
    void selectMarker(uint rIdx) override;
    void getGenotypeDistribution(uint rIdx, GenotypeDistribution &dist) override;
    void getCaseControlGenotypeDistribution(uint rIdx, CaseControlSet &ccs, CaseControlGenotypeDistribution &ccgd) override;
    void getCaseControlGenotypeDistribution(uint rIdx, CaseControlGenotypeDistribution &ccgd) override;
    void getCaseControlGenotypeDistribution(uint rIdx, CaseControlGenotypeDistribution &ccgd, marginal_information &m) override;

    // ---- pairs (pairwise_marker_analyzable.h:175-187)
    void selectMarkerPair(uint rIdx1, uint rIdx2) override;
    void getContingencyTable(uint rIdx1, uint rIdx2, ContingencyTable &ct) override;
    void getContingencyTable(uint rIdx1, uint rIdx2, ushort *column_set, ContingencyTable &ct) override;
    void getCaseControlContingencyTable(uint rIdx1, uint rIdx2, CaseControlSet &ccs, CaseControlContingencyTable &ccct) override;
    void getCaseControlContingencyTable(uint rIdx1, uint rIdx2, CaseControlContingencyTable &ccct) override;
    void getCaseControlContingencyTable(uint rIdx1, uint rIdx2, const marginal_information &m1,
                                        const marginal_information &m2, CaseControlContingencyTable &ccct) override;

    // ---- mask (case_control_selectable.h:42)
    void selectCaseControl(CaseControlSet &ccs) override;

    // ---- batch extensions (what the per-row getters are served from)
    // addGenotypeRow for rows rBegin, rBegin+1, ...: packed by all host cores, one upload
    void addGenotypeRows(int rBegin, const std::vector<std::pair<const char *, const char *> > &rows, char delim);
    // association statistics of row r computed in the same kernel pass
    // (include/gwas_scan.h gw_scan_case_control: MAF x3, HWE x2, chi-square x2, call rate)
    const double *getAssociationStatistics(uint rIdx);
    // many pairs in one launch; out = n x 16 (and n x 16 for controls)
    void getCaseControlContingencyTables(const uint *pairs, size_t n_pairs, CaseControlSet &ccs,
                                         std::vector<uint> &cases, std::vector<uint> &ctrls);
    // getContingencyTable for many pairs in one launch; out = n x 16
    void getContingencyTables(const uint *pairs, size_t n_pairs, std::vector<uint> &tables);
    // the post-select, margin-assisted getter (pairwise_marker_analyzable.h:185, table5.cpp:989-1150)
    // for many pairs in one launch: computeGTest's tables (epistasis_func.cpp:508-704)
    void getSelectedCaseControlContingencyTables(const uint *pairs, size_t n_pairs,
                                                 std::vector<CONTIN_TABLE_T> &cases, std::vector<CONTIN_TABLE_T> &ctrls);
    // computeBoost's whole pre-screening loop (algorithms/epistasis_func.cpp:393-486) in one device
    // pass: pairs (i, j, measure) with measure > threshold, in the reference's (i, j) visiting order
    void boostScreen(double threshold, std::vector<std::pair<std::pair<uint, uint>, double> > &passing);
    // computeGTest (algorithms/epistasis_func.cpp:508-704) for every pair of `pairs` on the device (gw_boost_gtest): the
    // IPF-fitted measure replaces .second, zval receives the odds-ratio z values.  setHostGTest(true) keeps stage 2 on one
    // host thread over device-fetched tables (round 3's form: the baseline `gwas_hip --time-boost` times beside it)
    void boostGTest(std::vector<std::pair<std::pair<uint, uint>, double> > &pairs, std::vector<double> &zval);
    void setHostGTest(bool on) { host_gtest = on; }
    bool hostGTest() const { return host_gtest; }
    void setScanWindow(uint rows) { window = rows ? rows : 1; invalidate(); }
    // scan window k+1 on a worker thread while the caller walks window k (on by default)
    void setPrefetch(bool on) { drop_prefetch(); prefetch = on; }
    // the reference's CaseControlSet has no change counter, so a set edited in place is recognised from its mask
    // bytes: all of them at selectCaseControl(), at every new window of rows and whenever a getter call is not the
    // strict successor (row + 1) of the previous one -- so a second walk from row 0 after a relabelling sees it; inside
    // a strictly sequential walk object + counts, and 32 sampled words per mask on every 16th call (the reference's
    // loops never edit a set mid-walk).  true = compare all bytes on every getter call (N/4 bytes of host reads per call)
    void setMaskCheckEveryCall(bool on) { check_every_call = on; }
    gw_table *handle() { settle(); return tbl; }
    // the rows were changed through handle() (gw_table_simulate, gw_table_upload_rows): drop every host cache
    void tableChanged() { invalidate(); }
    // the rows of rank `root`'s table into this one (gw_table_replicate: one RCCL broadcast), caches dropped
    void replicateFrom(gw_comm *comm, int root);

protected:
    void initialize() override;
    void invalidate();
    // makes ccs the device's masks unless it already is.  whole = compare every mask byte with the installed
    // copy; otherwise a per-call check (object, counts, 32 sampled words per mask)
    void install(CaseControlSet &ccs, bool whole = false);
    bool masks_unchanged(CaseControlSet &ccs, bool whole);
    void fill_dist_cache(uint rIdx);
    void fill_cc_cache(uint rIdx);
    // the window after the one just served, scanned ahead by a worker thread.  The C-ABI is one-thread-per-
    // table: every method that enters it calls settle() first (the worker is then the only caller)
    void start_prefetch(int kind, ulong begin);
    bool take_prefetch(int kind, ulong begin);   // true: the caches now hold that window
    void settle();                               // wait for the worker; its result stays usable
    void drop_prefetch();                        // wait and discard (data or masks changed)
    void pair_cc(uint r1, uint r2, CONTIN_TABLE_T &cs, CONTIN_TABLE_T &ct);

    gw_table *tbl;
    int device_id;
    uint n_individs, ref_words;
    uint window;

    // result caches (host)
    std::vector<uint> dist_cache, cc_cache;
    std::vector<double> stat_cache;
    ulong dist_begin, dist_count, cc_begin, cc_count, mar_begin;
    std::vector<gw_marginal_information> mar_cache;  // a window of device-computed marginal_information
    bool prefetch;
    struct Ahead {
        std::future<int> done;
        bool valid;
        int kind;  // 0 distribution, 1 case/control
        ulong begin, count;
        std::vector<uint> counts;
        std::vector<double> stats;
        Ahead() : valid(false), kind(0), begin(0), count(0) {}
    } ahead;

    // installed masks
    const CaseControlSet *installed;
    uint installed_cases, installed_ctrls;
    std::vector<ushort> installed_case_mask, installed_ctrl_mask;   // what the device holds (stream packing)
    uint64_t installed_sample[2];
    bool check_every_call;
    uint calls_since_sample;
    uint last_cc_row;   // the row of the previous masked getter call (0xFFFFFFFF: none)
    bool host_gtest;

    // pair getters: the reference's loops walk (i, j), (i, j+1), ... -- a run of pairs of the same first marker
    // is fetched in one launch (the run grows x4 while the walk stays sequential, 1 .. 4096 pairs)
    struct PairRun {
        bool valid, masked;
        uint first, second_begin, count, next_len;
        std::vector<uint> plain, cases, ctrls;  // count x 16
        PairRun() : valid(false), masked(false), first(0), second_begin(0), count(0), next_len(1) {}
    } run;
    const uint *pair_run(uint r1, uint r2, bool masked, const uint **ctrl_out);

    // operator()(r,c): one decoded row
    std::vector<ushort> row_cache;
    long row_cached;

    char gt_text[17][3];
};

}  // namespace genetics
}  // namespace libgwaspp
