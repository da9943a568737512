// gw_kernels.h -- launch interface between the C-ABI (gw_capi.cpp) and the
// gfx950 kernels (gw_kernels.hip).  Internal; the public surface is
// include/gwas_scan.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gw {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// HBM layout of the genotype table (DESIGN.md "Data layout in HBM"):
//   row r        : base + r * row_stride_u            (16-byte units)
//   plane 0 (lo) : units [0, units)                   bit set for codes 1,3
//   plane 1 (hi) : units [plane_stride_u, +units)     bit set for codes 2,3
// sample s lives at bit (s & 127) of unit (s >> 7), little-endian, i.e. the
// reference's stream packing (table5.cpp:333-344) with the header block moved
// to a side array and each plane padded to a 128-byte multiple.
struct TableView {
    u32x4 *base;
    uint64_t row_stride_u;
    uint32_t plane_stride_u;
    uint32_t units;      // 16-byte units per plane that hold samples
    uint32_t n_samples;
    uint64_t n_snps;
};

struct ScanShape {
    int unroll;         // units per lane in flight per plane: 1,2,4,8
    int mask_in_lds;    // stage masks in LDS
    int nontemporal;    // nt loads on the genotype stream
    int block_threads;  // 256/512/1024
    int complement;     // 1: control mask == ~case mask: single-mask kernel (k_scan_cc_compl)
    int pipeline;       // 1: continuous-stream kernel (k_scan_cc_pipe), 0: row-at-a-time (k_scan_cc)
    int short_rows;     // 1: rows of <= 256 units, a sub-wave of 16 or 32 lanes per row (k_scan_cc_sub; 256 threads, no LDS)
    int grid_blocks;    // persistent workgroups (k_scan_cc_sub: 0 = as many as fit)
    int compute_units;
    int reserve_cus;    // CUs the persistent grids leave free (GW_OPT_RESERVE_CUS)
    int defer;          // 1: records written in one burst at the end of the launch (k_scan_cc / k_scan_cc_sub <.., DEFER>)
    size_t lds_bytes;   // the masks' share of the dynamic LDS
    size_t lds_cap;     // dynamic LDS a workgroup may ask for on this device
};

// case/control scan: counts[n_rows][8], stats[n_rows][8] (stats may be null)
hipError_t launch_scan_cc(const TableView &tv, const u32x4 *d_case, const u32x4 *d_ctrl,
                          uint32_t case_count, uint32_t ctrl_count, uint64_t row_begin,
                          uint32_t n_rows, uint32_t *d_counts, double *d_stats,
                          const ScanShape &shape, hipStream_t stream,
                          // column-segmented scan: tv / masks describe one segment of the columns and the six
                          // sums per row go to d_raw[row][6] (added to it when raw_accumulate) -- not with
                          // shape.complement or shape.short_rows
                          uint32_t *d_raw = nullptr, bool raw_accumulate = false,
                          // char[96] or null: receives the kernel instantiation the launch went to, spelled as
                          // rocprofv3 prints it (e.g. "gw::k_scan_cc_pipe<2, true, true>")
                          char *kernel_name = nullptr);
// ... and its last step: counts[n][8] and stats[n][8] from the accumulated sums
hipError_t launch_finish_raw(const uint32_t *d_raw, uint32_t n_rows, uint32_t case_count, uint32_t ctrl_count,
                             uint32_t *d_counts, double *d_stats, hipStream_t stream);
// sets hipFuncAttributeMaxDynamicSharedMemorySize on every scan instantiation
hipError_t prepare_scan_kernels(size_t max_lds_bytes);

// plain distribution: dist[n_rows][4] = aa,ab,bb,xx(exact)
hipError_t launch_scan_dist(const TableView &tv, uint64_t row_begin, uint32_t n_rows,
                            uint32_t *d_dist, int grid_blocks, hipStream_t stream,
                            // > 0 (the CU count): rows of <= 256 units take the sub-wave kernel (k_scan_dist_sub)
                            int short_rows_cus = 0);

// pair contingency: pairs[n][2] (row indices), out[n][16]; with masks:
// out_case/out_ctrl (mask ANDed before decoding, table5.cpp:823-829)
hipError_t launch_pair_contingency(const TableView &tv, const uint32_t *d_pairs, uint32_t n_pairs,
                                   uint32_t *d_out, int grid_blocks, hipStream_t stream);
hipError_t launch_pair_cc_contingency(const TableView &tv, const u32x4 *d_case, const u32x4 *d_ctrl,
                                      const uint32_t *d_pairs, uint32_t n_pairs,
                                      uint32_t *d_out_case, uint32_t *d_out_ctrl, int grid_blocks,
                                      hipStream_t stream);

// reference packing <-> HBM layout.  src/dst = device staging buffer holding
// n_rows reference rows `ref_stride_bytes` apart (header block first).
hipError_t launch_restride_in(const TableView &tv, uint64_t row_begin, uint32_t n_rows,
                              const uint8_t *d_src, size_t ref_stride_bytes, int layout,
                              uint32_t ref_blocks, hipStream_t stream);
hipError_t launch_restride_out(const TableView &tv, uint64_t row_begin, uint32_t n_rows,
                               uint8_t *d_dst, size_t ref_stride_bytes, int layout,
                               uint32_t ref_blocks, hipStream_t stream);
// phenotype masks: reference stream packing (uint16 blocks) -> u32x4[units]
hipError_t launch_mask_in(const uint16_t *d_src, uint32_t n_blocks, u32x4 *d_mask, uint32_t units,
                          uint32_t n_samples, hipStream_t stream);

// synthetic input (restates data/simulate_data.cpp, see gw_kernels.hip)
hipError_t launch_simulate(const TableView &tv, uint64_t seed, uint64_t snp_global_begin,
                           uint32_t missing_q16, int grid_blocks, hipStream_t stream);
// ... with exactly floor(p * 2N) minor alleles per SNP, as the reference simulator's shuffle gives (simulate_data.cpp:191-205);
// *d_overflow is bumped for a row whose selection list did not fit (cohorts beyond ~6 M samples)
hipError_t launch_simulate_exact(const TableView &tv, uint64_t seed, uint64_t snp_global_begin, uint32_t missing_q16,
                                 int grid_blocks, uint32_t *d_overflow, hipStream_t stream);
hipError_t launch_simulate_phenotype(uint64_t seed, u32x4 *d_case, u32x4 *d_ctrl, uint32_t units,
                                     uint32_t n_samples, uint32_t *d_counts2, hipStream_t stream);

// streaming-read probe: the scan's traversal of the table without the arithmetic
hipError_t launch_stream_probe(const TableView &tv, uint32_t *d_sink, int grid_blocks, hipStream_t stream);

// ---- pairwise path (gw_boost.hip) ------------------------------------------
// marginal_information of one SNP (genotype/common_genotype.h:101-106)
struct SnpMargin {
    uint32_t margins[4], cases[4], ctrls[4];
    double entropy, entropy_y;
    double pbc[8];  // P(genotype | status): cases[4], controls[4]
    double pca[8];  // P(status | genotype)
};
// single-precision companions of a SnpMargin for the pre-screen's FP32 bound (k_boost<.., true>):
// index = 3 * group + genotype, group 0 = cases, 1 = controls; zero where the count is zero
struct SnpMarginF {
    float qb[6];  // margins[g] * n_group / group[g]   = 1 / (P(g | status) / margins[g])
    float qa[6];  // margins[g] / group[g]             = 1 / P(status | g)
    float rb[6];  // P(g | status) / margins[g]
    float pa[6];  // P(status | g)
    // the SNP's share of sum_cells k log2(qb qa / n) when the table has no missing call (boost_bound_split):
    //   ki = sum_g group[g] log2(qa[g])          the SNP as the pair's first marker
    //   kj = sum_g group[g] log2(qb[g] / n)      ... as its second
    float ki, kj;
    // round 4, tables without a missing call: tao - 1 from the four corners (gw_boost_dev.h boost_bound_quad_f).  With
    // d[c][g] = P(status c | genotype g) / P(c) - 1 (0 for an empty genotype class), per group c (0 cases, 1 controls):
    //   ua[3 c + {0, 1, 2}] = d[c][0] - d[c][1],  d[c][2] - d[c][1],  n d[c][1]                  the SNP as first marker
    //   vb[3 c + {0, 1, 2}] = P(c) / n x (d[c][0] - d[c][1],  d[c][2] - d[c][1],  d[c][1])       ... as second marker
    //   ea = max_c max(|ua[3c]|, |ua[3c + 1]|, |d[c][1]|),   eb = sum_c (C_0 |vb[3c]| + C_2 |vb[3c + 1]| + n |vb[3c + 2]|)
    //   (C_g = everyone with genotype g): ea x eb bounds the sum of the magnitudes of the terms tao - 1 is made of
    float ea, eb;
    float ua[6], vb[6];
    // round 4, tables WITH missing calls (boost_bound_quad9_f): lqa[3 c + g] = log2(qa[3 c + g]), lqb[3 c + g] = log2(qb[3 c + g] / n),
    // 0 where the count is 0 -- the pair's sum_cells k log2(qb qa / n) is then twelve products with the row / column sums of
    // its counted tables
    float lqa[6], lqb[6];
    float ma, mb;   // max |lqa[.]|, max |lqb[.]|, rounded up: they scale the rounding error of those twelve products
};
struct BoostHit {  // == gw_boost_hit of include/gwas_scan.h
    uint32_t snp_a, snp_b;
    double measure;
};
// What the matrix-core pair kernels hand to k_boost_finish: the pairs their in-kernel bound could not rule out, with the
// counted cells.  A record is {snp_a, snp_b, cells of the cases[NP], cells of the controls[NP]} as uint32, NP = 4 (corners,
// tables without a missing call) or 9; `count` is bumped once per candidate and may exceed `capacity` (the surplus is
// dropped and reported through d_tally[2], which makes the host repeat the block with every pair in FP64).
struct BoostCandidateList {
    uint32_t *records = nullptr;
    unsigned long long *count = nullptr;
    unsigned long long capacity = 0;   // in records of the kernel's own size
    // a block pair too large for the list at its cap is screened in row blocks of this many rows (a multiple of 128; 0 = one
    // launch): counter zeroed, counting kernel, k_boost_finish per row block, all on the launch's stream
    uint32_t rows_per_launch = 0;
};
inline size_t boost_candidate_bytes(bool nine_cells) { return (2u + 2u * (nine_cells ? 9u : 4u)) * sizeof(uint32_t); }

// selectCaseControl compaction (table5.cpp:443-575) of the whole table
// d_plan: 16 uint32 per 32-sample word of a row (n_words of them) -- per group the mask word, the five masks of its
// compress network, the bit offset of the word's first member in the group's planes and its member count (gw_boost.hip)
hipError_t launch_select_cc(const TableView &tv, const uint32_t *d_plan, uint32_t n_words, uint32_t wc, uint32_t wt,
                            uint32_t *d_cc, uint64_t cc_stride_w, int compute_units, hipStream_t stream);
hipError_t launch_margins(const uint32_t *d_counts, uint32_t n_rows, uint32_t n_individs, SnpMargin *d_mar,
                          SnpMarginF *d_marf, hipStream_t stream);
// BOOST pre-screen of the SNP block pair [a_begin,+n_a) x [b_begin,+n_b), pairs i < j
// d_tally: [0] hits, [1] pairs evaluated, [2] ordered max, [3] ordered min
hipError_t launch_boost_tables(uint32_t n_entries, uint32_t n_individs, double *d_frac, double *d_xlogx,
                               hipStream_t stream);
// One pair screen: everything launch_boost needs, by name (round 4: this was a list of ~35 positional arguments).
struct BoostLaunch {
    // the compacted rows (selectCaseControl), their block-tiled copy for the matrix-core kernels (or null), the margins
    const uint32_t *d_cc = nullptr;
    uint64_t cc_stride_w = 0;
    uint32_t wc = 0, wt = 0;
    const u32x4 *d_cct = nullptr;
    const SnpMargin *d_mar = nullptr;
    const SnpMarginF *d_marf = nullptr;            // needed by the pre-filter and by every matrix-core kernel but k_boost_mfma
    const double *d_frac = nullptr, *d_xlogx = nullptr;
    uint32_t n_individs = 0, n_cases = 0;          // of the compacted rows
    bool no_missing_call = false;                  // the table has no missing call: four corners + margins (table5.cpp:1066-1143)
    // the block pair [a_begin, +n_a) x [b_begin, +n_b), pairs i < j
    uint32_t a_begin = 0, n_a = 0, b_begin = 0, n_b = 0;
    double threshold = 0.0;
    // results: d_tally [0] hits, [1] pairs evaluated, [2] ordered max / (pre-filter) pairs outside the bound's error
    // model or candidates lost, [3] ordered min
    BoostHit *d_hits = nullptr;
    unsigned long long max_hits = 0;
    unsigned long long *d_tally = nullptr;
    double *d_dense = nullptr;                     // n_a x n_b measures (or, diagnostic, bounds), or null
    // FP32 bound first (prefilter), the FP64 measure only for pairs that bound + error bound cannot prove to be below
    // `bound_threshold` (the threshold rounded down to float); no extremes in d_tally then
    bool prefilter = false;
    float bound_threshold = 0.0f;
    bool first_level = false;                      // ... with the logarithm-free first level ahead of the bound proper
    bool dense_err = false;                        // diagnostic (prefilter + d_dense): write the error bound, not the bound
    // the pairs the in-kernel bound could not rule out (FP4 kernels): records of boost_candidate_bytes() each, a counter
    BoostCandidateList candidates;
    // kernel family: matrix_cores 0 = vector ALU, 1 = default, 2 = force the kernels that keep both groups' accumulators;
    // fp4 = count on FP4 operands where the family has such a kernel
    int matrix_cores = 1;
    bool fp4 = true;
    int compute_units = 0;
    hipStream_t stream = nullptr;
    char *kernel_name = nullptr;                   // char[96] or null: the instantiation the launch went to, as rocprofv3 prints it
};
hipError_t launch_boost(const BoostLaunch &l);
// BOOST stage 2 (computeGTest, epistasis_func.cpp:508-704) for the pairs {d_pairs[k * pair_stride], d_pairs[k * pair_stride + 1]},
// k < min(*d_n, n_max) (d_n null: n_max): d_tabs = 18 uint32 per pair of scratch, d_out[2k] = measure, [2k + 1] = z,
// d_status[0] += pairs whose fit did not settle within the sweep cap
hipError_t launch_boost_gtest(const uint32_t *d_cc, uint64_t cc_stride_w, uint32_t wc, uint32_t wt, const SnpMargin *d_mar,
                              uint32_t n_individs, const uint32_t *d_pairs, uint32_t pair_stride,
                              const unsigned long long *d_n, unsigned long long n_max, uint32_t *d_tabs, double *d_out,
                              unsigned long long *d_status, int compute_units, hipStream_t stream);
// bytes: ((n_snps + 63) / 64) * 64 * ((wc + wt) / 4) * 32
hipError_t launch_retile_cc(const uint32_t *d_cc, uint64_t cc_stride_w, uint32_t wc, uint32_t wt, uint32_t n_snps,
                            u32x4 *d_cct, hipStream_t stream);

}  // namespace gw
