/*
 * gw_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the libgwaspp per-SNP association-scan hot path,
 * written from scratch to follow the reference's algorithm step by step so
 * that the HIP path can be checked against it.  Every function cites the
 * reference file:line (relative to /root/reference/src/libgwaspp/) it
 * restates.
 *
 * Who may use this: tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg -- as the checker / reported baseline only.  Nothing under
 * libgwaspp_amd/ links, imports or executes it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - integer genotype counts: PINNED by the reference's own golden generator
 *     scripts/perl/genotype_set_builder.pl (tests/golden/perl/) and, for the
 *     phenotype masks, by the real reference CaseControlSet compiled in place
 *     (oracle/_ref/libref_ccs.so, built by oracle/Makefile when
 *     /root/reference is present).
 *   - MinorAlleleFrequency / computeMarginalInformation: restated from source;
 *     the reference ships no expected values for them and its table
 *     translation units cannot be built here (they include Rmath.h, which the
 *     image lacks) -> "parity unpinned" beyond the integer inputs.
 *   - chi-square / HWE statistics: NOT IN THE REFERENCE (build-defined, see
 *     gwo_assoc_stats) -> "parity unpinned"; GPU is compared with this file.
 */
#ifndef GW_ORACLE_H
#define GW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- data contract (genotype/common_genotype.h:67-75,182-192,101-106) ---- */

/* frequency_table: uint32 {aa, ab, bb, xx} */
typedef struct { uint32_t aa, ab, bb, xx; } gwo_freq;

/* marginal_information (common_genotype.h:101-106) */
typedef struct {
    gwo_freq margins, cases, controls;
    double marginal_entropy, marginal_entropy_y;
    double pbc[8]; /* cases[4]; controls[4] */
    double pca[8];
} gwo_marginal;

/* Row geometry of the two 2-bit layouts.
 *   stream (CompressedGenotypeTable5, compressed_genotype_table5.cpp:44-72):
 *      plane_blocks = N/16+1 rounded up to x4; row = 1 header + 2*plane_blocks
 *   block  (CompressedGenotypeTable3, compressed_genotype_table3.cpp:44-67):
 *      data_blocks  = N/8+1  rounded up to x4; row = 1 header + data_blocks
 * blocks are uint16 (DataBlock, libgwaspp.h:44-45); words are 64-bit PWORDs. */
typedef struct {
    uint32_t n_samples;
    uint32_t data_blocks;    /* per plane (stream) or per row (block), no header */
    uint32_t blocks_per_row; /* including the header block                     */
    uint32_t words;          /* data_blocks / 4                                 */
} gwo_geom;

void gwo_geometry_stream(uint32_t n_samples, gwo_geom *g);
void gwo_geometry_block(uint32_t n_samples, gwo_geom *g);

/* PopCount via a 64 Ki-entry byte table, four lookups per 64-bit word
 * (common_genotype_func.h:56-90, common_genotype_func.cpp:157-171). */
int gwo_popcount16(uint16_t v);
int gwo_popcount64(uint64_t v);

/* ---- ingest (compressed_genotype_table5.cpp:277-365, table3.cpp:260-325,
 *      HeaderStateMachine common_genotype.h:257-304) ----
 * text is "AC<d>AA<d>00<d>..." exactly as TpedGenotypeFile hands it over
 * (tped_genotype_file.cpp:127-190): two allele characters then one delimiter.
 * row must hold g->blocks_per_row uint16.  Returns 0, or -1 where the
 * reference would have hit an assert (a second heterozygote spelling, a
 * fourth genotype). */
int gwo_pack_row_stream(const char *begin, const char *end, const gwo_geom *g, uint16_t *row);
int gwo_pack_row_block(const char *begin, const char *end, const gwo_geom *g, uint16_t *row);

/* ParseHeader (common_genotype.h:306-334): out = {xx, aa, ab, bb} genotype
 * encodings (4*a1+a2 over "ACGT"), 0xFFFF where unset. */
void gwo_parse_header(uint16_t header, uint16_t out[4]);

/* operator()(r, c) (table5.cpp:383-417): 2-bit code of one sample: 0 missing,
 * 1 first-seen homozygote, 2 heterozygote, 3 other homozygote. */
int gwo_stream_code_at(const uint16_t *row, const gwo_geom *g, uint32_t c);
int gwo_block_code_at(const uint16_t *row, const gwo_geom *g, uint32_t c);

/* ---- CaseControlSet masks (analyzable/case_control_set.cpp:34-150) ----
 * block_mask : uint16[data_blocks of block geometry], 0b11 per member
 * stream_mask: uint16[data_blocks of stream geometry], 1 bit per member
 * either may be NULL.  idx need not be sorted; duplicates count once (the
 * reference takes a std::set<int>).  Returns the member count, -1 if an index
 * is out of range. */
int gwo_ccs_set(const int32_t *idx, size_t n_idx, uint32_t n_samples,
                uint16_t *block_mask, uint16_t *stream_mask);

/* ---- per-SNP distributions, stream layout (CGT5) ---- */
/* getGenotypeDistribution (table5.cpp:577-607): xx counts pad bits too (Q3). */
void gwo_stream_dist(const uint16_t *row, const gwo_geom *g, gwo_freq *out);
/* getCaseControlGenotypeDistribution(r, ccs, ccgd) (table5.cpp:609-657). */
void gwo_stream_cc_dist(const uint16_t *row, const gwo_geom *g,
                        const uint16_t *case_stream, const uint16_t *ctrl_stream,
                        uint32_t case_count, uint32_t ctrl_count,
                        gwo_freq *cases, gwo_freq *ctrls);

/* selectCaseControl compaction (table5.cpp:443-575) of ONE row into
 * out[2*(case_blocks+ctrl_blocks)]; the *_blocks come from
 * gwo_compact_geometry (table5.cpp:449-468). */
void gwo_compact_geometry(uint32_t case_count, uint32_t ctrl_count,
                          uint32_t *case_blocks, uint32_t *ctrl_blocks);
void gwo_stream_compact_row(const uint16_t *row, const gwo_geom *g,
                            const uint16_t *case_stream, const uint16_t *ctrl_stream,
                            uint32_t case_blocks, uint32_t ctrl_blocks, uint16_t *out);
/* getCaseControlGenotypeDistribution(r, ccgd) on a compacted row
 * (table5.cpp:659-701). */
void gwo_stream_cc_dist_compacted(const uint16_t *crow, uint32_t case_blocks, uint32_t ctrl_blocks,
                                  uint32_t case_count, uint32_t ctrl_count,
                                  gwo_freq *cases, gwo_freq *ctrls);

/* ---- per-SNP distributions, block layout (CGT3) ---- */
/* getGenotypeDistribution (table3.cpp:556-594): xx left 0 (Q3). */
void gwo_block_dist(const uint16_t *row, const gwo_geom *g, gwo_freq *out);
/* getCaseControlGenotypeDistribution(r, ccs, ccgd) (table3.cpp:596-640). */
void gwo_block_cc_dist(const uint16_t *row, const gwo_geom *g,
                       const uint16_t *case_block, const uint16_t *ctrl_block,
                       gwo_freq *cases, gwo_freq *ctrls);

/* ---- pairwise contingency, stream layout ---- */
/* getContingencyTable(i, j, ct) (table5.cpp:749-800): out[16] row-major 4x4,
 * rows = marker A {aa,ab,bb,xx}, columns = marker B {aa,ab,bb,xx}
 * (common_genotype.h:173-192). */
void gwo_stream_contingency(const uint16_t *row_a, const uint16_t *row_b, const gwo_geom *g,
                            uint32_t out[16]);
/* getCaseControlContingencyTable(i, j, ccs, ccct) (table5.cpp:806-895). */
void gwo_stream_cc_contingency(const uint16_t *row_a, const uint16_t *row_b, const gwo_geom *g,
                               const uint16_t *case_stream, const uint16_t *ctrl_stream,
                               uint32_t out_case[16], uint32_t out_ctrl[16]);

/* ---- pairwise contingency, block layout ----
 * getCaseControlContingencyTable(i, j, ccs, ccct) of the 2-bit block table
 * (table3.cpp:733-849): the nine joint genotype cells only, the xx row/column
 * stays zero.  (CGT3's plain getContingencyTable(i,j,ct) starts four blocks
 * into the row -- quirk Q4, table3.cpp:671-672 -- and is not restated.)
 * Together with gwo_stream_cc_contingency this is the cross-compression-level
 * check that scripts/perl/contin_output_comparer.pl performs on --epi-debug
 * dumps. */
void gwo_block_cc_contingency(const uint16_t *row_a, const uint16_t *row_b, const gwo_geom *g,
                              const uint16_t *case_block, const uint16_t *ctrl_block,
                              uint32_t out_case[16], uint32_t out_ctrl[16]);

/* ---- arithmetic on the counts ---- */
/* MinorAlleleFrequency (algorithms/maf_func.h:46-54), quirk Q5 kept. */
void gwo_minor_allele_frequency(const gwo_freq *ft, double *tot, double *maf);
/* computeMarginalInformation (common_genotype_func.cpp:173-219).  Entries the
 * reference leaves unwritten (zero-count classes, Q6) are set to 0 here.
 * Returns -1 where the reference asserts nCases+nControls == nIndivids. */
int gwo_marginal_information(const gwo_freq *cases, const gwo_freq *ctrls, uint32_t n_individs,
                             gwo_marginal *m);

/* Build-defined association statistics -- NOT IN THE REFERENCE (SURVEY.md
 * section 8a row 13): parity unpinned, GPU <-> this function only.
 * out[8]:
 *  [0] MinorAlleleFrequency(cases)   (reference formula, Q5)
 *  [1] MinorAlleleFrequency(controls)
 *  [2] conventional MAF over cases+controls: min(p,1-p), p=(2aa+ab)/(2n)
 *  [3] HWE chi-square (df 1) on controls
 *  [4] HWE chi-square (df 1) on cases+controls
 *  [5] genotypic Pearson chi-square, 2x3 {case,ctrl}x{aa,ab,bb} (df 2)
 *  [6] allelic Pearson chi-square, 2x2 (df 1)
 *  [7] call rate: non-missing / (case_count + ctrl_count)
 * missing calls (xx) are excluded everywhere; empty margins give 0. */
#define GWO_N_STATS 8
void gwo_assoc_stats(const gwo_freq *cases, const gwo_freq *ctrls, double out[GWO_N_STATS]);

/* ---- whole-table scan drivers (used for parity at moderate sizes and as the
 *      cpu_baseline).  table = M rows in the reference memory layout (row
 *      stride g->blocks_per_row uint16, header block first).
 *      counts[M][8] = case {aa,ab,bb,xx}, ctrl {aa,ab,bb,xx};
 *      stats[M][8] as gwo_assoc_stats (may be NULL). ---- */
/* faithful: one thread, LUT popcount, the word loop of table5.cpp:609-657
 * inside the SNP loop of maf_func.cpp:292-316. */
void gwo_scan_stream_cc(const uint16_t *table, uint32_t n_snps, const gwo_geom *g,
                        const uint16_t *case_stream, const uint16_t *ctrl_stream,
                        uint32_t case_count, uint32_t ctrl_count,
                        uint32_t *counts, double *stats);
/* same for the block layout (table3.cpp:596-640); xx computed as
 * group_count - aa - ab - bb so that both layouts return the same record. */
void gwo_scan_block_cc(const uint16_t *table, uint32_t n_snps, const gwo_geom *g,
                       const uint16_t *case_block, const uint16_t *ctrl_block,
                       uint32_t case_count, uint32_t ctrl_count,
                       uint32_t *counts, double *stats);
/* best effort: hardware popcount, n_threads pthreads over SNP ranges. */
void gwo_scan_stream_cc_mt(const uint16_t *table, uint32_t n_snps, const gwo_geom *g,
                           const uint16_t *case_stream, const uint16_t *ctrl_stream,
                           uint32_t case_count, uint32_t ctrl_count,
                           uint32_t *counts, double *stats, int n_threads);

/* ---- synthetic input (restates data/simulate_data.cpp:150-235 with a
 *      counter-based RNG; see DESIGN.md "Synthetic input") ----
 * Writes one row in the stream layout (header 0x7000|codes A/C) for global
 * SNP index snp.  missing_q16 = missing-call probability * 65536. */
void gwo_sim_row_stream(uint64_t seed, uint64_t snp, const gwo_geom *g, uint32_t missing_q16,
                        uint16_t *row);
/* phenotype masks (simulate_data.cpp:219-235: fair coin per sample). */
/* exact minor-allele count per SNP, as simulate_data.cpp:191-205 (0 on success) */
int gwo_sim_row_stream_exact(uint64_t seed, uint64_t snp, const gwo_geom *g, uint32_t missing_q16, uint16_t *row);
void gwo_sim_phenotype_stream(uint64_t seed, const gwo_geom *g,
                              uint16_t *case_stream, uint16_t *ctrl_stream,
                              uint32_t *case_count, uint32_t *ctrl_count);
/* the per-SNP minor-allele probability the simulator drew, as p*1e5. */
uint32_t gwo_sim_snp_p_e5(uint64_t seed, uint64_t snp);

/* ---- BOOST / KSA pre-screen (algorithms/epistasis_func.cpp:349-506; NEXT row) ----
 * The interaction measure of one SNP pair as computeBoost evaluates it
 * (:424-470) from the margin-assisted 4x4 case / control tables
 * (table5.cpp:989-1150) and the two SNPs' marginal_information.  Entries the
 * reference leaves uninitialised (quirk Q6) are zero here. */
double gwo_boost_measure(const uint32_t case_tab[16], const uint32_t ctrl_tab[16],
                         const gwo_marginal *m1, const gwo_marginal *m2, uint32_t n_individs);
/* All pairs i < j of a stream-layout table: measures[M*(M-1)/2] in the order
 * (0,1),(0,2),...,(1,2),...  Tables are the exact member tables the
 * margin-assisted getter returns. */
void gwo_boost_all_pairs(const uint16_t *table, uint32_t n_snps, const gwo_geom *g,
                         const uint16_t *case_stream, const uint16_t *ctrl_stream,
                         uint32_t case_count, uint32_t ctrl_count, double *measures);

/* Stage 2 of computeBoost on one pair that passed the screen: computeGTest
 * (algorithms/epistasis_func.cpp:508-704) -- iterative proportional fitting of
 * the no-three-way-interaction model, the exact measure (:650-683) and the
 * allelic odds-ratio z value (:686-701).  Parity unpinned (the reference holds
 * no expected values); restated with its quirks: the initial error is 18
 * (:552-556 never advances its pointers), uint32 products in the odds ratios. */
void gwo_gtest(const uint32_t case_tab[16], const uint32_t ctrl_tab[16], const gwo_marginal *m1,
               const gwo_marginal *m2, uint32_t n_individs, double *measure, double *zval);

/* ---- pairwise LLR known-answer test (src/test/pairwise.c:50-133; NEXT row) ---- */
double gwo_pairwise_llr(const uint32_t case_tab[9], const uint32_t ctrl_tab[9]);

#ifdef __cplusplus
}
#endif
#endif /* GW_ORACLE_H */
