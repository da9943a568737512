// hip_genotype_table.cpp -- see genetics/genotype/hip_genotype_table.h.
#include "genetics/genotype/hip_genotype_table.h"

#include <algorithm>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gwas_scan.h"

namespace libgwaspp {
namespace genetics {

namespace {

// the reference aborts through assert() (kept on at -O3, CMakeLists.txt:34);
// a failed C-ABI call is the same class of error
void die_unless_ok(int rc, const char *what) {
    if (rc != GW_OK) {
        std::fprintf(stderr, "HipGenotypeTable: %s failed: %s\n", what, gw_last_error());
        std::abort();
    }
}
void unsupported(const char *what) {
    // e.g. table5.cpp:434-441, 802-804: assert(false) in the reference too
    std::fprintf(stderr, "HipGenotypeTable: %s is not implemented (assert(false) in CompressedGenotypeTable5)\n", what);
    std::abort();
}

const char kAlphabet[] = GENOTYPE_ALPHABET;

inline void to_freq(const uint *c, frequency_table &ft) {
    ft.aa = c[0]; ft.ab = c[1]; ft.bb = c[2]; ft.xx = c[3];
}

}  // namespace

HipGenotypeTable::HipGenotypeTable(util::indexer *markers, util::indexer *individs, int device)
    : GenoTable(markers, individs), tbl(nullptr), device_id(device), n_individs(0), ref_words(0),
      window(1u << 16), dist_begin(0), dist_count(0), cc_begin(0), cc_count(0), mar_begin(0), prefetch(true), installed(nullptr),
      installed_cases(0), installed_ctrls(0), check_every_call(false), calls_since_sample(0), last_cc_row(0xFFFFFFFFu), host_gtest(false), row_cached(-1) {
    installed_sample[0] = installed_sample[1] = 0;
    initialize();
}

void HipGenotypeTable::initialize() {
    n_individs = (uint)max_column;
    die_unless_ok(gw_table_create(device_id, n_individs, (uint64_t)max_row, &tbl), "gw_table_create");
    ref_words = gw_stream_blocks(n_individs) / BLOCKS_PER_PWORD;
    // genotype vocabulary: 16 allele pairs + "unknown" (table5.cpp:37, 104-118)
    possible_genotypes_size = 17;
    beg = new ushort[possible_genotypes_size];
    end = beg + possible_genotypes_size;
    for (uint e = 0; e < 16; ++e) {
        beg[e] = (ushort)e;
        gt_text[e][0] = kAlphabet[e >> 2];
        gt_text[e][1] = kAlphabet[e & 3];
        gt_text[e][2] = '\0';
    }
    beg[16] = 0xFFFF;
    std::strcpy(gt_text[16], "00");  // err_lookup, table5.cpp:147-149
    std::cout << "Initializing HipGenotypeTable ... " << max_row << " markers x " << max_column
              << " individuals resident in HBM" << std::endl;
    // (the reference's constructors log to cout too; this line is the one behaviour a drop-in user has to know about)
    std::cout << "HipGenotypeTable: a CaseControlSet edited in place is recognised from its mask bytes -- all of them at "
                 "selectCaseControl, at each new window of rows and at any getter call that is not row + 1 of the previous "
                 "one; sampled words inside a strictly sequential walk (setMaskCheckEveryCall(true): all bytes, every call)"
              << std::endl;
}

HipGenotypeTable::~HipGenotypeTable() {
    drop_prefetch();
    gw_table_destroy(tbl);
}

void HipGenotypeTable::replicateFrom(gw_comm *comm, int root) {
    drop_prefetch();
    die_unless_ok(gw_table_replicate(tbl, comm, root, nullptr), "gw_table_replicate");
    invalidate();
}

void HipGenotypeTable::invalidate() {
    drop_prefetch();
    run.valid = false;
    dist_count = cc_count = 0;
    mar_cache.clear();
    row_cached = -1;
}

void HipGenotypeTable::settle() {
    if (ahead.valid && ahead.done.valid()) ahead.done.wait();
}

void HipGenotypeTable::drop_prefetch() {
    if (ahead.valid && ahead.done.valid()) ahead.done.wait();
    ahead.valid = false;
}

void HipGenotypeTable::start_prefetch(int kind, ulong begin) {
    if (!prefetch || begin >= (ulong)max_row) return;
    drop_prefetch();
    ahead.kind = kind;
    ahead.begin = begin;
    ahead.count = std::min<ulong>(window, (ulong)max_row - begin);
    ahead.counts.resize(ahead.count * (kind ? GW_N_COUNTS : 4));
    if (kind) ahead.stats.resize(ahead.count * GW_N_STATS);
    gw_table *h = tbl;
    Ahead *a = &ahead;
    ahead.done = std::async(std::launch::async, [h, a]() -> int {
        return a->kind ? gw_scan_case_control(h, a->begin, a->count, a->counts.data(), a->stats.data())
                       : gw_scan_distribution(h, a->begin, a->count, a->counts.data());
    });
    ahead.valid = true;
}

bool HipGenotypeTable::take_prefetch(int kind, ulong begin) {
    if (!ahead.valid || ahead.kind != kind || ahead.begin != begin) return false;
    const int rc = ahead.done.get();
    ahead.valid = false;
    if (rc != GW_OK) return false;  // the caller scans again on its own thread and reports the error
    if (kind) {
        cc_cache.swap(ahead.counts);
        stat_cache.swap(ahead.stats);
        cc_begin = begin;
        cc_count = ahead.count;
    } else {
        dist_cache.swap(ahead.counts);
        dist_begin = begin;
        dist_count = ahead.count;
    }
    return true;
}

// --------------------------------------------------------------------- ingest

void HipGenotypeTable::addGenotypeRow(int rIdx, const char *p_begin, const char *p_end, char delim) {
    drop_prefetch();  // the C-ABI is entered by one thread at a time; the data change
    die_unless_ok(gw_table_add_genotype_row(tbl, (uint64_t)rIdx, p_begin, p_end, delim), "addGenotypeRow");
    invalidate();
}

void HipGenotypeTable::addGenotypeRow(int rIdx, std::string::const_iterator &it, std::string::const_iterator &it_end,
                                      char delim) {
    if (it >= it_end) return;
    const char *b = &*it;
    addGenotypeRow(rIdx, b, b + (it_end - it), delim);
    it = it_end;
}

void HipGenotypeTable::addGenotypeRows(int rBegin, const std::vector<std::pair<const char *, const char *> > &rows,
                                       char delim) {
    drop_prefetch();
    std::vector<const char *> b(rows.size()), e(rows.size());
    for (size_t k = 0; k < rows.size(); ++k) { b[k] = rows[k].first; e[k] = rows[k].second; }
    die_unless_ok(gw_table_add_genotype_rows(tbl, (uint64_t)rBegin, rows.size(), b.data(), e.data(), delim, 0),
                  "addGenotypeRows");
    invalidate();
}

void HipGenotypeTable::addGenotype(int rIdx, int cIdx, const std::string &gt) {
    drop_prefetch();
    // read-modify-write of one row in the reference's stream packing
    gw_geometry g;
    die_unless_ok(gw_table_geometry(tbl, &g), "gw_table_geometry");
    std::vector<ushort> row(g.stream_row_bytes / 2);
    die_unless_ok(gw_table_download_rows(tbl, rIdx, 1, row.data(), g.stream_row_bytes, GW_LAYOUT_STREAM), "download");
    const ushort enc = encodeGenotype(gt);
    int code = 0;
    if (enc != 0xFFFF) die_unless_ok(gw_header_assign(&row[0], enc, &code), "addGenotype (row labelling)");
    const ushort bit = (ushort)(1u << (cIdx & 15));
    ushort &lo = row[1 + (cIdx >> 4)], &hi = row[1 + g.stream_blocks + (cIdx >> 4)];
    lo = (ushort)((lo & ~bit) | ((code & 1) ? bit : 0));
    hi = (ushort)((hi & ~bit) | ((code & 2) ? bit : 0));
    die_unless_ok(gw_table_upload_rows(tbl, rIdx, 1, row.data(), g.stream_row_bytes, GW_LAYOUT_STREAM), "upload");
    invalidate();
}

ushort HipGenotypeTable::encodeGenotype(const std::string &gt) {
    if (gt.length() != MAX_ALLELE_COUNT) std::abort();  // assert in table5.cpp:369
    int v[2];
    for (int i = 0; i < 2; ++i) {
        const char *p = std::strchr(kAlphabet, gt[i]);
        if (!p || !gt[i]) return 0xFFFF;
        v[i] = (int)(p - kAlphabet);
    }
    return (ushort)(4 * v[0] + v[1]);
}

const char *HipGenotypeTable::decodeGenotype(ushort enc) { return gt_text[enc < 16 ? enc : 16]; }

bool HipGenotypeTable::isGenotypeHomozygous(ushort enc) { return enc < 16 && (enc >> 2) == (enc & 3); }

DataBlock HipGenotypeTable::operator()(int r, int c) {
    settle();
    gw_geometry g;
    die_unless_ok(gw_table_geometry(tbl, &g), "gw_table_geometry");
    if (row_cached != r) {
        row_cache.resize(g.stream_row_bytes / 2);
        die_unless_ok(gw_table_download_rows(tbl, r, 1, row_cache.data(), g.stream_row_bytes, GW_LAYOUT_STREAM),
                      "download");
        row_cached = r;
    }
    const ushort bit = (ushort)(1u << (c & 15));
    const bool lo = row_cache[1 + (c >> 4)] & bit, hi = row_cache[1 + g.stream_blocks + (c >> 4)] & bit;
    const ushort head = row_cache[0];
    if (lo) return hi ? (head & 0x000F) : ((head >> 8) & 0x000F);  // BB : AA (table5.cpp:399-406)
    return hi ? ((head >> 4) & 0x000F) : 0xFFFF;                    // AB : unknown
}

// ------------------------------------------------------------- single marker

void HipGenotypeTable::selectMarker(uint) { unsupported("selectMarker"); }
void HipGenotypeTable::selectMarkerPair(uint, uint) { unsupported("selectMarkerPair"); }
void HipGenotypeTable::getContingencyTable(uint, uint, ushort *, ContingencyTable &) {
    unsupported("getContingencyTable(column_set)");
}

void HipGenotypeTable::fill_dist_cache(uint rIdx) {
    const ulong begin = (rIdx / window) * (ulong)window;
    const ulong n = std::min<ulong>(window, (ulong)max_row - begin);
    if (!take_prefetch(0, begin)) {
        drop_prefetch();
        dist_cache.resize(n * 4);
        die_unless_ok(gw_scan_distribution(tbl, begin, n, dist_cache.data()), "gw_scan_distribution");
        dist_begin = begin;
        dist_count = n;
    }
    start_prefetch(0, begin + n);  // the callers walk r = 0 .. M-1 in order (maf_func.cpp:271-290, 318-335)
}

void HipGenotypeTable::getGenotypeDistribution(uint rIdx, GenotypeDistribution &dist) {
    if (rIdx < dist_begin || rIdx >= dist_begin + dist_count) fill_dist_cache(rIdx);
    frequency_table ft;
    to_freq(&dist_cache[(rIdx - dist_begin) * 4], ft);
    // the stream table counts its pad bits as xx here (table5.cpp:600, quirk Q3)
    ft.xx += ref_words * 64u - n_individs;
    dist.setDistribution(ft);
}

// Masks.  The reference's CaseControlSet (analyzable/case_control_set.h:48-91) exposes the mask pointers and the
// two counts and nothing else: no block count accessor, no change counter.  Everything install() needs follows from
// those: the block count is a function of the cohort size (case_control_set.cpp:35-52 == gw_stream_blocks), and
// "did the set change since it was installed" is answered from the mask bytes themselves.  This file compiles
// unchanged against the reference's own headers (tests/test_build_guards.py).
namespace {
// a few words spread over the mask: the per-call check between two whole comparisons
inline uint64_t mask_sample(const ushort *m, uint blocks) {
    const uint words = blocks / BLOCKS_PER_PWORD;
    const uint64_t *w = reinterpret_cast<const uint64_t *>(m);
    uint64_t h = 0x9E3779B97F4A7C15ull;
    const uint step = words > 32u ? words / 32u : 1u;
    for (uint k = 0; k < words; k += step) {
        uint64_t v;
        std::memcpy(&v, w + k, sizeof v);
        h = (h ^ v) * 0xBF58476D1CE4E5B9ull;
        h ^= h >> 29;
    }
    return h;
}
}  // namespace

bool HipGenotypeTable::masks_unchanged(CaseControlSet &ccs, bool whole) {
    if (installed != &ccs || installed_cases != ccs.getCaseCount() || installed_ctrls != ccs.getControlCount())
        return false;
    const ushort *cs = ccs.stream_case_begin(), *ct = ccs.stream_control_begin();
    const uint blocks = (uint)installed_case_mask.size();
    if (whole || check_every_call)
        return std::memcmp(cs, installed_case_mask.data(), blocks * sizeof(ushort)) == 0 &&
               std::memcmp(ct, installed_ctrl_mask.data(), blocks * sizeof(ushort)) == 0;
    // between two whole comparisons: pointer and counts on every call (above), the sampled words on every 16th --
    // 30 ns that the per-row walk (100 ns per row, maf_func.cpp:304-314) would otherwise pay on each row
    if (++calls_since_sample & 15u) return true;
    return mask_sample(cs, blocks) == installed_sample[0] && mask_sample(ct, blocks) == installed_sample[1];
}

void HipGenotypeTable::install(CaseControlSet &ccs, bool whole) {
    if (masks_unchanged(ccs, whole)) return;
    drop_prefetch();
    run.valid = false;
    // the set's own block count (case_control_set.cpp:35-52); the C-ABI refuses one that is not the table's
    const uint blocks = gw_stream_blocks(ccs.getMaximumIndex() + 1u);
    const ushort *cs = ccs.stream_case_begin(), *ct = ccs.stream_control_begin();
    die_unless_ok(gw_table_set_case_control(tbl, cs, ct, blocks, ccs.getCaseCount(), ccs.getControlCount()),
                  "gw_table_set_case_control");
    installed = &ccs;
    installed_cases = ccs.getCaseCount();
    installed_ctrls = ccs.getControlCount();
    installed_case_mask.assign(cs, cs + blocks);
    installed_ctrl_mask.assign(ct, ct + blocks);
    installed_sample[0] = mask_sample(cs, blocks);
    installed_sample[1] = mask_sample(ct, blocks);
    nCaseCount = ccs.getCaseCount();
    nControlCount = ccs.getControlCount();
    nIndivids = nCaseCount + nControlCount;
    cc_count = 0;
    mar_cache.clear();
}

void HipGenotypeTable::selectCaseControl(CaseControlSet &ccs) {
    // the reference compacts cases and controls into dense planes here
    // (table5.cpp:443-575, 6.7 s for 10k x 50k); on the GPU the masks are applied
    // inline at full bandwidth, so "select" only installs them
    install(ccs, true);
}

void HipGenotypeTable::fill_cc_cache(uint rIdx) {
    if (!installed) {
        std::fprintf(stderr, "HipGenotypeTable: no CaseControlSet selected\n");
        std::abort();
    }
    const ulong begin = (rIdx / window) * (ulong)window;
    const ulong n = std::min<ulong>(window, (ulong)max_row - begin);
    if (!take_prefetch(1, begin)) {
        drop_prefetch();
        cc_cache.resize(n * GW_N_COUNTS);
        stat_cache.resize(n * GW_N_STATS);
        die_unless_ok(gw_scan_case_control(tbl, begin, n, cc_cache.data(), stat_cache.data()), "gw_scan_case_control");
        cc_begin = begin;
        cc_count = n;
    }
    start_prefetch(1, begin + n);  // the callers walk r = 0 .. M-1 in order (maf_func.cpp:304-314)
}

void HipGenotypeTable::getCaseControlGenotypeDistribution(uint rIdx, CaseControlGenotypeDistribution &ccgd) {
    if (rIdx < cc_begin || rIdx >= cc_begin + cc_count) fill_cc_cache(rIdx);
    frequency_table ca, co;
    const uint *c = &cc_cache[(rIdx - cc_begin) * GW_N_COUNTS];
    to_freq(c, ca);
    to_freq(c + 4, co);
    ccgd.setCaseDistribution(ca);
    ccgd.setControlDistribution(co);
}

void HipGenotypeTable::getCaseControlGenotypeDistribution(uint rIdx, CaseControlSet &ccs,
                                                          CaseControlGenotypeDistribution &ccgd) {
    // Every mask byte is compared unless this call is the strict successor of the previous one inside a scanned window
    // (the walk of maf_func.cpp:304-314): a new window, a jump, and in particular a second walk that starts over at
    // row 0 -- what a caller does after relabelling the same CaseControlSet in place (GeneticData::setCaseControlSet,
    // genetic_data.cpp:115-157), also with equal group sizes -- all get the whole comparison.  Only
    // inside a strictly sequential walk the sampled check stands in (object, counts, 32 words on every 16th call).
    const bool sequential = rIdx == last_cc_row + 1u && rIdx >= cc_begin && rIdx < cc_begin + cc_count;
    install(ccs, !sequential);
    last_cc_row = rIdx;
    getCaseControlGenotypeDistribution(rIdx, ccgd);
}

void HipGenotypeTable::getCaseControlGenotypeDistribution(uint rIdx, CaseControlGenotypeDistribution &ccgd,
                                                          marginal_information &m) {
    // table5.cpp:703-747: the counts of the selected table + computeMarginalInformation, here computed on the
    // device for a whole window of rows at once (gw_scan_margins); computeMargins (epistasis_func.cpp:706-721)
    // walks r = 0 .. M-1 in order
    static_assert(sizeof(marginal_information) == sizeof(gw_marginal_information), "marginal_information layout");
    if (rIdx < mar_begin || rIdx >= mar_begin + mar_cache.size()) {
        if (!installed) {
            std::fprintf(stderr, "HipGenotypeTable: no CaseControlSet selected\n");
            std::abort();
        }
        settle();
        const ulong begin = (rIdx / window) * (ulong)window;
        const ulong n = std::min<ulong>(window, (ulong)max_row - begin);
        mar_cache.resize(n);
        die_unless_ok(gw_scan_margins(tbl, begin, n, mar_cache.data()), "gw_scan_margins");
        mar_begin = begin;
    }
    std::memcpy(&m, &mar_cache[rIdx - mar_begin], sizeof m);
    ccgd.setCaseDistribution(m.cases);
    ccgd.setControlDistribution(m.controls);
}

const double *HipGenotypeTable::getAssociationStatistics(uint rIdx) {
    if (rIdx < cc_begin || rIdx >= cc_begin + cc_count) fill_cc_cache(rIdx);
    return &stat_cache[(rIdx - cc_begin) * GW_N_STATS];
}

// ----------------------------------------------------------------------- pairs

// The pair (r1, r2) out of a run (r1, r2 .. r2 + count - 1) fetched in one launch.  The reference's pair loops
// (computeBoost :393-486, ContingencyPerformance, contingency_debug) keep the first marker and advance the
// second: a hit extends nothing, a miss right after the run's end fetches a run 4 x as long (up to 4096),
// anything else starts over with a single pair -- random access costs what it did before.
const uint *HipGenotypeTable::pair_run(uint r1, uint r2, bool masked, const uint **ctrl_out) {
    const bool hit = run.valid && run.masked == masked && run.first == r1 && r2 >= run.second_begin &&
                     r2 < run.second_begin + run.count;
    if (!hit) {
        settle();
        const bool sequential = run.valid && run.masked == masked && run.first == r1 &&
                                r2 == run.second_begin + run.count;
        uint len = sequential ? std::min<uint>(run.next_len, 4096u) : 1u;
        len = (uint)std::min<ulong>(len, (ulong)max_row - r2);
        if (len == 0) len = 1;
        std::vector<uint> pairs(2u * len);
        for (uint k = 0; k < len; ++k) {
            pairs[2u * k] = r1;
            pairs[2u * k + 1u] = r2 + k;
        }
        if (masked) {
            run.cases.resize(16u * len);
            run.ctrls.resize(16u * len);
            die_unless_ok(gw_pair_case_control_contingency(tbl, pairs.data(), len, run.cases.data(), run.ctrls.data()),
                          "gw_pair_case_control_contingency");
        } else {
            run.plain.resize(16u * len);
            die_unless_ok(gw_pair_contingency(tbl, pairs.data(), len, run.plain.data()), "gw_pair_contingency");
        }
        run.valid = true;
        run.masked = masked;
        run.first = r1;
        run.second_begin = r2;
        run.count = len;
        run.next_len = sequential ? len * 4u : 4u;
    }
    const size_t off = 16u * (size_t)(r2 - run.second_begin);
    if (masked) {
        *ctrl_out = run.ctrls.data() + off;
        return run.cases.data() + off;
    }
    return run.plain.data() + off;
}

void HipGenotypeTable::getContingencyTable(uint rIdx1, uint rIdx2, ContingencyTable &ct) {
    CONTIN_TABLE_T c;
    std::memcpy(c.contin, pair_run(rIdx1, rIdx2, false, nullptr), sizeof c.contin);
    ct.setMarkerAIndex(rIdx1);
    ct.setMarkerBIndex(rIdx2);
    ct.setContingency(c);
}

void HipGenotypeTable::pair_cc(uint r1, uint r2, CONTIN_TABLE_T &cs, CONTIN_TABLE_T &ct) {
    const uint *ctrl = nullptr;
    const uint *cases = pair_run(r1, r2, true, &ctrl);
    std::memcpy(cs.contin, cases, sizeof cs.contin);
    std::memcpy(ct.contin, ctrl, sizeof ct.contin);
}

void HipGenotypeTable::getCaseControlContingencyTable(uint rIdx1, uint rIdx2, CaseControlSet &ccs,
                                                      CaseControlContingencyTable &ccct) {
    install(ccs, !(run.valid && run.masked && run.first == rIdx1 && rIdx2 >= run.second_begin &&
                   rIdx2 < run.second_begin + run.count));  // whole comparison whenever a launch follows
    CONTIN_TABLE_T cs, ct;
    pair_cc(rIdx1, rIdx2, cs, ct);  // masks before decoding: table5.cpp:823-829
    ccct.setMarkerAIndex(rIdx1);
    ccct.setMarkerBIndex(rIdx2);
    ccct.updateContingencyTables(cs, ct);
}

void HipGenotypeTable::getCaseControlContingencyTable(uint rIdx1, uint rIdx2, CaseControlContingencyTable &ccct) {
    if (!installed) std::abort();
    CONTIN_TABLE_T cs, ct;
    pair_cc(rIdx1, rIdx2, cs, ct);
    // the reference works on compacted rows here (table5.cpp:896-987): only the
    // group's members plus the pad bits of the compacted planes reach xx_xx
    const uint case_words = (((nCaseCount / 16u + 1u) + 3u) & ~3u) / 4u;
    const uint ctrl_words = (((nControlCount / 16u + 1u) + 3u) & ~3u) / 4u;
    cs.xx_xx = cs.xx_xx - ref_words * 64u + case_words * 64u;
    ct.xx_xx = ct.xx_xx - ref_words * 64u + ctrl_words * 64u;
    ccct.setMarkerAIndex(rIdx1);
    ccct.setMarkerBIndex(rIdx2);
    ccct.updateContingencyTables(cs, ct);
}

void HipGenotypeTable::getCaseControlContingencyTable(uint rIdx1, uint rIdx2, const marginal_information &m1,
                                                      const marginal_information &m2,
                                                      CaseControlContingencyTable &ccct) {
    (void)m1; (void)m2;  // the margins only save popcounts on the CPU (table5.cpp:1066-1143)
    if (!installed) std::abort();
    CONTIN_TABLE_T cs, ct;
    pair_cc(rIdx1, rIdx2, cs, ct);
    // exact tables: xx_xx = members missing at both markers (table5.cpp:1031,1062)
    cs.xx_xx -= ref_words * 64u - nCaseCount;
    ct.xx_xx -= ref_words * 64u - nControlCount;
    ccct.setMarkerAIndex(rIdx1);
    ccct.setMarkerBIndex(rIdx2);
    ccct.updateContingencyTables(cs, ct);
}

void HipGenotypeTable::getCaseControlContingencyTables(const uint *pairs, size_t n_pairs, CaseControlSet &ccs,
                                                       std::vector<uint> &cases, std::vector<uint> &ctrls) {
    settle();
    install(ccs, true);
    cases.resize(n_pairs * 16);
    ctrls.resize(n_pairs * 16);
    die_unless_ok(gw_pair_case_control_contingency(tbl, pairs, n_pairs, cases.data(), ctrls.data()),
                  "gw_pair_case_control_contingency");
}

void HipGenotypeTable::getContingencyTables(const uint *pairs, size_t n_pairs, std::vector<uint> &tables) {
    settle();
    tables.resize(n_pairs * 16);
    die_unless_ok(gw_pair_contingency(tbl, pairs, n_pairs, tables.data()), "gw_pair_contingency");
}

void HipGenotypeTable::getSelectedCaseControlContingencyTables(const uint *pairs, size_t n_pairs,
                                                               std::vector<CONTIN_TABLE_T> &cases,
                                                               std::vector<CONTIN_TABLE_T> &ctrls) {
    settle();
    if (!installed) std::abort();
    std::vector<uint> cs(n_pairs * 16), ct(n_pairs * 16);
    if (n_pairs)
        die_unless_ok(gw_pair_case_control_contingency(tbl, pairs, n_pairs, cs.data(), ct.data()),
                      "gw_pair_case_control_contingency");
    cases.resize(n_pairs);
    ctrls.resize(n_pairs);
    for (size_t k = 0; k < n_pairs; ++k) {
        for (int c = 0; c < 16; ++c) {
            cases[k].contin[c] = cs[k * 16 + c];
            ctrls[k].contin[c] = ct[k * 16 + c];
        }
        // exact tables: xx_xx = members missing at both markers (table5.cpp:1031,1062)
        cases[k].xx_xx -= ref_words * 64u - nCaseCount;
        ctrls[k].xx_xx -= ref_words * 64u - nControlCount;
    }
}

void HipGenotypeTable::boostScreen(double threshold, std::vector<std::pair<std::pair<uint, uint>, double> > &passing) {
    settle();
    if (!installed) std::abort();
    passing.clear();
    // (2^20 records = 16 MB, not value-initialised: only what a screen delivers is ever touched; a list that turns out too
    // short costs a second screen)
    size_t room = (size_t)1u << 20;
    std::unique_ptr<gw_boost_hit[]> buf(new gw_boost_hit[room]);
    uint64_t n_hits = 0;
    for (;;) {
        die_unless_ok(gw_boost_screen(tbl, 0, (uint64_t)max_row, 0, (uint64_t)max_row, threshold, buf.get(), room,
                                      &n_hits, nullptr, nullptr, nullptr),
                      "gw_boost_screen");
        if (n_hits <= room) break;
        room = (size_t)n_hits;  // the list was capped: run again with room for all of it
        buf.reset(new gw_boost_hit[room]);
    }
    gw_boost_hit *const hits_begin = buf.get(), *const hits_end = buf.get() + n_hits;
    std::sort(hits_begin, hits_end, [](const gw_boost_hit &a, const gw_boost_hit &b) {
        return a.snp_a != b.snp_a ? a.snp_a < b.snp_a : a.snp_b < b.snp_b;
    });
    passing.reserve((size_t)n_hits);
    for (const gw_boost_hit *ph = hits_begin; ph != hits_end; ++ph) {
        const gw_boost_hit &h = *ph;
        passing.push_back(std::make_pair(std::make_pair((uint)h.snp_a, (uint)h.snp_b), h.measure));
    }
}

void HipGenotypeTable::boostGTest(std::vector<std::pair<std::pair<uint, uint>, double> > &pairs, std::vector<double> &zval) {
    settle();
    if (!installed) std::abort();
    const size_t n = pairs.size();
    std::vector<gw_boost_hit> recs(n);
    for (size_t k = 0; k < n; ++k) {
        recs[k].snp_a = pairs[k].first.first;
        recs[k].snp_b = pairs[k].first.second;
        recs[k].measure = pairs[k].second;
    }
    std::vector<double> measure(n), z(n);
    uint64_t unsettled = 0;
    die_unless_ok(gw_boost_gtest(tbl, recs.data(), n, measure.data(), z.data(), &unsettled), "gw_boost_gtest");
    if (unsettled)   // the reference would still be iterating (its loop has no bound)
        std::cerr << "HipGenotypeTable: " << unsettled << " pair(s) whose proportional fitting did not settle in 2^24 sweeps"
                  << std::endl;
    for (size_t k = 0; k < n; ++k) {
        pairs[k].second = measure[k];
        zval.push_back(z[k]);
    }
}

}  // namespace genetics
}  // namespace libgwaspp
