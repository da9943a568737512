/*
 * gwas_scan.h -- C-ABI of the MI355X-native per-SNP association scan.
 *
 * This is the drop-in boundary for ONE path of putnampp/libgwaspp: the
 * per-SNP scan over the 2-bit-packed genotype table (genotype counts ->
 * case/control contingency -> MAF / HWE / chi-square).  Each entry point
 * names the reference interface it replaces; paths are relative to
 * /root/reference/src/libgwaspp/.  The reference is single-process C++ with
 * no FFI of its own, so "what its FFI for this path would bind" is the set of
 * virtuals of genetics::GenoTable that the scan harnesses in
 * algorithms/maf_func.cpp call, batched over SNP ranges.
 * INTEGRATION.md shows the GenoTable subclass a maintainer would add on top.
 *
 * Conventions
 *   - plain pointers and sizes only; handles are opaque.
 *   - every call returns 0 (GW_OK) or a negative gw_status; gw_last_error()
 *     gives the message of the calling thread's last failure.  The reference
 *     aborts via assert() on the same conditions (CMakeLists.txt:34 keeps
 *     asserts on at -O3); the C++ mirror in libgwaspp_amd/host turns a
 *     non-zero status back into an abort.
 *   - a handle is thread-compatible, not thread-safe (the reference's tables
 *     are not re-entrant either: SURVEY.md section 5).
 *   - "stream packing" / "block packing" are the reference's two 2-bit row
 *     layouts: CompressedGenotypeTable5 (two bit-planes per row,
 *     genotype/compressed_genotype_table5.cpp:44-72) and
 *     CompressedGenotypeTable3 (2 bits per genotype, table3.cpp:44-67), both
 *     with one uint16 header block in front of each row.
 *   - there is NO CPU fallback: without a HIP device every compute entry
 *     point fails with GW_E_HIP.
 */
#ifndef GWAS_SCAN_H
#define GWAS_SCAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gw_status {
    GW_OK = 0,
    GW_E_INVALID = -1, /* bad argument                                     */
    GW_E_HIP = -2,     /* HIP runtime / no device / kernel launch failure   */
    GW_E_NOMEM = -3,   /* host or device allocation failed                  */
    GW_E_STATE = -4,   /* call order (e.g. scan before masks were set)      */
    GW_E_RANGE = -5,   /* SNP / sample index outside the table              */
    GW_E_FORMAT = -6   /* genotype text the reference would assert on       */
} gw_status;

/* row layouts of the reference, incl. the uint16 header block */
typedef enum gw_layout {
    GW_LAYOUT_STREAM = 5, /* CompressedGenotypeTable5 (e2BitStream)           */
    GW_LAYOUT_BLOCK = 3   /* CompressedGenotypeTable3 (e2BitBlockCompression) */
} gw_layout;

typedef struct gw_table gw_table; /* opaque */

/* per-SNP result records */
#define GW_N_COUNTS 8 /* case {aa,ab,bb,xx}, control {aa,ab,bb,xx}: two
                         frequency_table (genotype/common_genotype.h:67-75) */
#define GW_N_STATS 8  /* doubles, see gw_scan_case_control                  */
#define GW_N_CONTIN 16 /* CONTIN_TABLE_T (common_genotype.h:182-192)         */

typedef struct gw_geometry {
    uint32_t n_samples;
    uint64_t n_snps;
    uint32_t stream_blocks;     /* uint16 blocks per plane, reference stream packing */
    uint32_t block_blocks;      /* uint16 data blocks per row, reference block packing */
    uint32_t stream_row_bytes;  /* reference row stride, stream packing (header incl.) */
    uint32_t block_row_bytes;   /* reference row stride, block packing (header incl.)  */
    uint32_t device_units;      /* 16-byte units per plane in HBM                       */
    uint32_t device_plane_bytes;/* plane stride in HBM (128-byte multiple)              */
    uint64_t device_row_bytes;  /* row stride in HBM = 2 planes                         */
    uint64_t device_table_bytes;
    uint64_t algorithmic_bytes_per_snp; /* n_samples/4, SURVEY.md 8d              */
} gw_geometry;

typedef struct gw_device_info {
    char name[128];
    char arch[32];
    int compute_units;
    int lds_bytes_per_cu;
    uint64_t hbm_bytes;
    uint64_t hbm_free_bytes;
} gw_device_info;

/* ---- library / device ------------------------------------------------- */
const char *gw_last_error(void);
const char *gw_version(void);
int gw_device_count(int *count);
int gw_device_info_get(int device, gw_device_info *out);

/* ---- table lifetime ------------------------------------------------------
 * replaces: new CompressedGenotypeTable5(markers, individs) chosen by
 * GeneticData::updateGenotypeTable (genetics/genetic_data.cpp:46-80) and the
 * table destructors.  n_snps = rows->included_size(), n_samples =
 * columns->included_size() (util/table/table.h:49-74).  Allocates the whole
 * table in HBM, zero-filled ("unknown" everywhere, as table5.cpp:82-83). */
int gw_table_create(int device, uint32_t n_samples, uint64_t n_snps, gw_table **out);
int gw_table_destroy(gw_table *t);
int gw_table_geometry(const gw_table *t, gw_geometry *out);

/* ---- ingest --------------------------------------------------------------
 * gw_table_add_genotype_row replaces GenoTable::addGenotypeRow(int r, const
 * char *b, const char *e, char delim) (genotype/geno_table.h:53;
 * table5.cpp:277-365): text "AC<d>AA<d>..." -> 2-bit codes with first-seen
 * labelling and the row header (HeaderStateMachine,
 * common_genotype.h:257-304).  Rows are packed on the host and staged; they
 * reach HBM at gw_table_flush() or at the next scan.  GW_E_FORMAT where the
 * reference asserts (second heterozygote spelling, fourth genotype). */
int gw_table_add_genotype_row(gw_table *t, uint64_t snp, const char *begin, const char *end,
                              char delim);
int gw_table_flush(gw_table *t);
/* The same for n_rows consecutive SNPs at once: rows[k] = [begins[k], ends[k]) is the genotype text
 * of SNP snp_begin + k.  Rows are packed by n_threads host threads (0 = one per available core) and
 * uploaded in chunks of up to 256 MB of packed rows; on GW_E_FORMAT the chunk holding the bad row and
 * everything after it are not stored. */
int gw_table_add_genotype_rows(gw_table *t, uint64_t snp_begin, uint64_t n_rows, const char *const *begins,
                               const char *const *ends, char delim, int n_threads);

/* Bulk hand-over of rows ALREADY in a reference packing (the bytes at
 * Table<DataBlock>::data, header block included, `row_stride_bytes` apart):
 * re-strided on the device into the HBM layout. */
int gw_table_upload_rows(gw_table *t, uint64_t snp_begin, uint64_t n_rows, const void *rows,
                         size_t row_stride_bytes, int layout);
/* The reverse (used by operator()(r,c) in the C++ mirror, by tests and by the
 * CPU-baseline sample): rows come back in a reference packing, header incl. */
int gw_table_download_rows(gw_table *t, uint64_t snp_begin, uint64_t n_rows, void *rows,
                           size_t row_stride_bytes, int layout);
/* row header blocks (uint16 per SNP), as table5.cpp:363 stores them */
int gw_table_get_headers(gw_table *t, uint64_t snp_begin, uint64_t n_rows, uint16_t *headers);

/* Synthetic input generated in HBM: restates data/simulate_data.cpp:150-235
 * (MAF-spectrum driven genotypes in HWE, coin-flip phenotype) with a
 * counter-based RNG so that any row can be regenerated anywhere.  Row r of
 * the table gets global SNP index snp_global_begin + r.  missing_q16 =
 * missing-call probability * 65536 (0 = none, like the reference simulator).
 * gw_table_simulate_phenotype also installs the case/control masks. */
int gw_table_simulate(gw_table *t, uint64_t seed, uint64_t snp_global_begin, uint32_t missing_q16);
/* The same, with the reference simulator's exact allele count: a SNP gets exactly floor(p * 2 n_samples) minor alleles
 * in a uniformly random arrangement (simulate_data.cpp:191-205 fills s 'B's and shuffles the 2N alleles), where
 * gw_table_simulate draws every allele independently with probability p (binomial count).  Same p per SNP, same missing
 * calls, same phenotype.  GW_E_RANGE for cohorts beyond about 6 million samples. */
int gw_table_simulate_exact(gw_table *t, uint64_t seed, uint64_t snp_global_begin, uint32_t missing_q16);
int gw_table_simulate_phenotype(gw_table *t, uint64_t seed);

/* ---- phenotype masks -----------------------------------------------------
 * replaces handing a CaseControlSet to the table: the arguments are
 * CaseControlSet::stream_case_begin() / stream_control_begin()
 * (analyzable/case_control_set.h:64-74; one bit per sample, n_blocks =
 * stream_block_count uint16) and getCaseCount()/getControlCount(). */
int gw_table_set_case_control(gw_table *t, const uint16_t *case_stream, const uint16_t *ctrl_stream,
                              uint32_t n_blocks, uint32_t case_count, uint32_t ctrl_count);
int gw_table_get_case_control(gw_table *t, uint16_t *case_stream, uint16_t *ctrl_stream,
                              uint32_t n_blocks, uint32_t *case_count, uint32_t *ctrl_count);

/* ---- the scan ------------------------------------------------------------
 * gw_scan_case_control replaces the loop of inline_cc_maf
 * (algorithms/maf_func.cpp:292-316):
 *     for i in [snp_begin, snp_begin+n):
 *         gt.getCaseControlGenotypeDistribution(i, ccs, ccgd)   table5.cpp:609-657
 *         MinorAlleleFrequency(cases) / (controls)              maf_func.h:46-54
 * counts[n][8]: case {aa,ab,bb,xx}, control {aa,ab,bb,xx}; xx = group count
 *   - aa - ab - bb (table5.cpp:649,653).  Bit-exact integers.
 * stats[n][8] (may be NULL: counts only):
 *   [0] MinorAlleleFrequency(cases)    reference formula (quirk Q5)
 *   [1] MinorAlleleFrequency(controls)
 *   [2] conventional MAF over cases+controls, min(p,1-p), p=(2aa+ab)/(2n)
 *   [3] HWE chi-square (df 1) on controls
 *   [4] HWE chi-square (df 1) on cases+controls
 *   [5] genotypic 2x3 Pearson chi-square (df 2)
 *   [6] allelic 2x2 Pearson chi-square (df 1)
 *   [7] call rate
 *   [3..7] are NOT in the reference (it has no single-SNP test); they are
 *   defined in DESIGN.md and computed in FP64 in the same kernel.
 * Host-pointer form: results are copied back before returning. */
int gw_scan_case_control(gw_table *t, uint64_t snp_begin, uint64_t n_snps, uint32_t *counts,
                         double *stats);

/* Device-pointer form for callers that keep results in HBM (bench, the
 * multi-GPU gather): d_counts / d_stats are device pointers with room for
 * n_snps records (d_stats may be NULL), the launch goes to `hip_stream`
 * (a hipStream_t; NULL = the handle's own stream) and is NOT synchronised. */
int gw_scan_case_control_device(gw_table *t, uint64_t snp_begin, uint64_t n_snps, void *d_counts,
                                void *d_stats, void *hip_stream);

/* gw_scan_distribution replaces the loop of genotype_dist_performance /
 * inline_maf_print (maf_func.cpp:271-290,318-335) over
 * getGenotypeDistribution(i, dist) (table5.cpp:577-607):
 * dist[n][4] = {aa, ab, bb, xx}, xx = n_samples - aa - ab - bb (the exact
 * missing count; the reference's stream table adds its pad bits here, quirk
 * Q3 -- the C++ mirror re-adds them). */
int gw_scan_distribution(gw_table *t, uint64_t snp_begin, uint64_t n_snps, uint32_t *dist);
int gw_scan_distribution_device(gw_table *t, uint64_t snp_begin, uint64_t n_snps, void *d_dist,
                                void *hip_stream);

/* gw_scan_margins replaces the loop of computeMargins (algorithms/epistasis_func.cpp:706-721) over
 * getCaseControlGenotypeDistribution(i, ccgd, m) (table5.cpp:703-747), i.e. the case/control counts of
 * the selected table followed by computeMarginalInformation(cases, controls, nCases + nControls, m)
 * (genotype/common_genotype_func.cpp:173-219), computed on the device.  The record has the layout of
 * the reference's marginal_information (genotype/common_genotype.h:101-106, 192 bytes):
 *   margins = cases + controls per genotype class {aa, ab, bb, xx};
 *   marginal_entropy   = -sum p ln p over the 4 classes,      p = margin / n
 *   marginal_entropy_y = -sum p ln p over 8 (class x status), p = count / n
 *   pbc[g] / pbc[4+g]  = P(class | case) / P(class | control) = count / group size
 *   pca[g] / pca[4+g]  = P(case | class) / P(control | class) = count / margin
 * Entries of pbc / pca that belong to a class nobody has are ZERO here; the reference leaves them
 * unwritten (quirk Q6), so compare only the written ones.  As in selectCaseControl
 * (table5.cpp:531-548) a sample that is in both masks counts as a case only. */
typedef struct gw_marginal_information {
    uint32_t margins[4], cases[4], controls[4];
    double marginal_entropy, marginal_entropy_y;
    double pbc[8], pca[8];
} gw_marginal_information;
int gw_scan_margins(gw_table *t, uint64_t snp_begin, uint64_t n_snps, gw_marginal_information *out);

/* ---- pairwise contingency (callers of the scan) -----
 * replaces getContingencyTable(i, j, ct) (table5.cpp:749-800) and
 * getCaseControlContingencyTable(i, j, ccs, ccct) (table5.cpp:806-895) for a
 * list of SNP pairs.  pairs[n][2]; out[n][16] row-major 4x4
 * {aa,ab,bb,xx} x {aa,ab,bb,xx}; for the case/control form out_case and
 * out_ctrl are both required. */
int gw_pair_contingency(gw_table *t, const uint32_t *pairs, uint64_t n_pairs, uint32_t *out);
int gw_pair_case_control_contingency(gw_table *t, const uint32_t *pairs, uint64_t n_pairs,
                                     uint32_t *out_case, uint32_t *out_ctrl);

/* ---- BOOST / KSA pre-screen over SNP pairs ----
 * replaces the O(M^2) pre-screening loop of computeBoost
 * (algorithms/epistasis_func.cpp:349-506): selectCaseControl (table5.cpp:443-575),
 * computeMargins (:706-721), then for every pair i < j with i in
 * [a_begin, a_begin+n_a) and j in [b_begin, b_begin+n_b) the margin-assisted
 * case/control tables (table5.cpp:989-1150) and the interaction measure
 * (:424-470); pairs whose measure exceeds `threshold` (30.0 in the reference,
 * :390,482) are listed.  The block form lets callers shard the pair space.
 * hits[max_hits] receives at most max_hits entries in no particular order;
 * *n_hits is the total found (may exceed max_hits), *n_pairs the pairs
 * evaluated, *max_measure / *min_measure the extremes (:472-480; the
 * reference tracks but never prints them).  Any of the four may be NULL; with
 * both extremes NULL the screen skips the FP64 measure of pairs that a
 * single-precision bound plus its error bound proves to be below the threshold
 * (GW_OPT_BOOST_PREFILTER) -- same hit list, ~1.5x faster at N = 10 k. */
typedef struct gw_boost_hit {
    uint32_t snp_a, snp_b;
    double measure;
} gw_boost_hit;
int gw_boost_screen(gw_table *t, uint64_t a_begin, uint64_t n_a, uint64_t b_begin, uint64_t n_b,
                    double threshold, gw_boost_hit *hits, uint64_t max_hits, uint64_t *n_hits,
                    uint64_t *n_pairs, double *max_measure, double *min_measure);
/* Device-resident form, for callers that walk the pair space block by block and keep the hits in HBM (the
 * sharded screen: every rank appends the hits of its row blocks to one list and the lists are all-gathered,
 * SURVEY.md 8e "C5").  d_list is a caller-owned device buffer of GW_BOOST_LIST_HEADER_BYTES + capacity *
 * sizeof(gw_boost_hit) bytes:
 *     uint64 n_hits      hits found by all calls since gw_boost_list_init (may exceed capacity: the list is full)
 *     uint64 n_pairs     pairs evaluated
 *     uint64 aux[2]      pre-filter on (default): aux[0] = pairs whose FP64 measure fell outside the single-precision
 *                        bound's error model -- must be 0, otherwise repeat with GW_OPT_BOOST_PREFILTER 0 (the host form
 *                        does that by itself); pre-filter off: the ordered maximum / minimum of the measures
 *     gw_boost_hit hits[capacity]   in no particular order
 * gw_boost_list_init zeroes the header (it needs gw_table_set_case_control: which meaning slots 2 / 3 have follows from the
 * cohort, as in the screen itself; like the screen, the first call after a mask or data change runs selectCaseControl's
 * compaction and computeMargins, synchronously, on the handle's stream).  gw_boost_screen_device enqueues the screen of the block
 * on `hip_stream` (NULL = the handle's stream) and does not synchronise -- except that a call which needs a larger candidate
 * list than any before it (GW_OPT_BOOST_CANDIDATES) re-allocates it, which waits for the device. */
#define GW_BOOST_LIST_HEADER_BYTES 32
int gw_boost_list_init(gw_table *t, void *d_list, void *hip_stream);
int gw_boost_screen_device(gw_table *t, uint64_t a_begin, uint64_t n_a, uint64_t b_begin, uint64_t n_b,
                           double threshold, void *d_list, uint64_t capacity, void *hip_stream);
/* BOOST stage 2 on the device: computeGTest (algorithms/epistasis_func.cpp:508-704) for n pairs -- per pair the two
 * 3 x 3 case / control tables from the compacted rows (what getCaseControlContingencyTable(i, j, margins_i, margins_j, ccct)
 * hands the reference, table5.cpp:989-1150), the homogeneous-association model fitted by iterative proportional fitting in
 * FP64 exactly as the reference iterates it, measure[k] = the exact interaction measure (the value the reference writes back
 * into the pair, :684-685) and z[k] = the difference of the allelic log odds ratios over its standard error (:686-701).
 * Only snp_a / snp_b of pairs[k] are read.  The reference's fitting loop has no iteration bound; this one stops after 2^24
 * sweeps and counts such pairs in *n_unsettled (may be NULL) -- tables with missing calls need 10^4 sweeps at the median
 * and millions at worst in the reference's formulation, tables without a missing call tens.  Needs gw_table_set_case_control. */
int gw_boost_gtest(gw_table *t, const gw_boost_hit *pairs, uint64_t n, double *measure, double *z, uint64_t *n_unsettled);
/* ... for a hit list in HBM as gw_boost_screen_device leaves it (header + hits; capacity as given there): the pairs are the
 * first min(n_hits, capacity) records, d_measure_z receives {measure, z} per record (2 * capacity doubles of device memory),
 * d_status (8 bytes of device memory the caller has zeroed, or NULL) is incremented per unsettled pair.  Enqueued on
 * `hip_stream` behind the screens, nothing returns to the host (the first call with a larger capacity than any before it
 * allocates its table scratch, which waits for the device). */
int gw_boost_gtest_device(gw_table *t, const void *d_list, uint64_t capacity, void *d_measure_z, void *d_status,
                          void *hip_stream);
/* the kernel instantiation the handle's last pair screen was dispatched to, as rocprofv3 prints it */
const char *gw_last_boost_kernel(const gw_table *t);
/* every measure of a (small) block, row-major n_a x n_b; NaN where i >= j */
int gw_boost_measures(gw_table *t, uint64_t a_begin, uint64_t n_a, uint64_t b_begin, uint64_t n_b,
                      double *measures);
/* selectCaseControl's compacted rows (table5.cpp:443-575) as the reference
 * stores them in m_cases_controls: per row 2*(case_blocks+ctrl_blocks) uint16,
 * [case plane0 | case plane1 | ctrl plane0 | ctrl plane1], blocks = count/16+1
 * rounded up to x4 (:449-468).  rows may be NULL to query the block counts. */
int gw_table_download_selected(gw_table *t, uint64_t snp_begin, uint64_t n_rows, uint16_t *rows,
                               uint32_t *case_blocks, uint32_t *ctrl_blocks);

/* ---- multi-GPU: SNP-range shards + one gather -----------------
 * Every SNP row is independent (the loop of algorithms/maf_func.cpp:304-314 touches row i and the shared masks
 * only), so a job shards by contiguous SNP ranges, one process per GPU, each with its own gw_table holding
 * its rows; the phenotype masks are replicated.  The one exchange step is the gather of the per-SNP records
 * to one rank.  These entry points let a C or C++ host do all of that without Python: librccl is loaded at
 * run time (dlopen) the first time a gw_comm_* function is called.
 *   gw_shard_range   rank r of `world` owns SNPs [begin, end) of n_snps_total (sizes differ by at most one)
 *   gw_comm_new_id   one rank makes the rendezvous id (ncclGetUniqueId, GW_COMM_ID_BYTES bytes) and hands it
 *                    to the others by any means (a file, MPI, the launcher's environment)
 *   gw_comm_create   ncclCommInitRank on `device`; collective over all ranks
 *   gw_comm_gather   every rank gives `bytes` at d_send, the root receives rank p's slice at d_recv + p * bytes:
 *                    grouped ncclSend / ncclRecv, i.e. every peer's slice over its own xGMI link; enqueued on
 *                    `hip_stream` (NULL = the communicator's own stream) behind whatever produced d_send there,
 *                    not synchronised
 *   gw_comm_allgather  every rank gives `bytes` at d_send and receives rank p's slice at d_recv + p * bytes
 *                    (ncclAllGather; a world of one copies).  The pair screen's exchange step: every rank ends up
 *                    with every rank's fixed-capacity hit list (SURVEY.md 8e "C5")
 *   gw_comm_broadcast  `bytes` at d_buf on the root reach d_buf on every rank (ncclBroadcast): replicates packed
 *                    rows, e.g. the 250 MB table of the pair screen, from the rank that ingested them
 *   gw_table_stream  the handle's own stream (what NULL means in the *_device scan forms), to enqueue the gather
 *                    behind a scan
 * gw_device_malloc / _free / _download: device buffers for hosts without a HIP toolchain of their own. */
typedef struct gw_comm gw_comm;
#define GW_COMM_ID_BYTES 128
int gw_shard_range(uint64_t n_snps_total, int rank, int world, uint64_t *begin, uint64_t *end);
/* The pair space {(i, j): i < j} of the BOOST screen (computeBoost's loop, algorithms/epistasis_func.cpp:393-486)
 * sharded over `world` ranks: the rows i are cut into blocks of `block` SNPs dealt in snake order (0..W-1, W-1..0,
 * ...; block k holds about block x (n_snps - k block) pairs, so the triangle is balanced to < 1 %).  The rank
 * screens [a_begin[k], +n_a[k]) x [a_begin[k], n_snps) for each of its entries (adjacent blocks come merged).
 * *n_blocks = entries the rank has; at most `capacity` are written.  Every rank holds the whole table
 * (gw_table_replicate). */
int gw_shard_pair_blocks(uint64_t n_snps, uint32_t block, int rank, int world, uint64_t *a_begin, uint64_t *n_a,
                         size_t capacity, size_t *n_blocks);
/* rows (and row headers) of the root's table to the same-sized tables of all ranks: one ncclBroadcast of the HBM
 * rows over xGMI -- the replication step of the pair screen (250 MB at 100 k SNPs x 10 k samples).  Collective;
 * synchronises `hip_stream` (NULL = the handle's). */
int gw_table_replicate(gw_table *t, gw_comm *c, int root, void *hip_stream);
int gw_comm_new_id(void *id);
int gw_comm_create(int device, int rank, int world, const void *id, gw_comm **out);
int gw_comm_destroy(gw_comm *c);
int gw_comm_gather(gw_comm *c, const void *d_send, size_t bytes, void *d_recv, int root, void *hip_stream);
int gw_comm_allgather(gw_comm *c, const void *d_send, size_t bytes, void *d_recv, void *hip_stream);
int gw_comm_broadcast(gw_comm *c, void *d_buf, size_t bytes, int root, void *hip_stream);
int gw_comm_synchronize(gw_comm *c, void *hip_stream);
int gw_table_stream(gw_table *t, void **hip_stream);
int gw_device_malloc(int device, size_t bytes, void **d_ptr);
int gw_device_free(int device, void *d_ptr);
int gw_device_memset(int device, void *d_ptr, int value, size_t bytes, void *hip_stream);
int gw_device_download(int device, void *host_dst, const void *d_src, size_t bytes, void *hip_stream);

/* ---- measurement ----------------------------------------------------------
 * HIP events on the stream the kernels are launched on (NULL = the handle's
 * stream).  gw_timer_stop synchronises on the stop event and returns the
 * elapsed milliseconds since the matching gw_timer_start. */
int gw_timer_start(gw_table *t, void *hip_stream);
int gw_timer_stop(gw_table *t, void *hip_stream, float *elapsed_ms);
/* A plain 16-byte-per-lane nt read of the whole table allocation (flat
 * grid-stride, no arithmetic): the empirical streaming-read ceiling SURVEY.md 8d
 * asks to report next to the nominal 8 TB/s.  Returns the bytes one launch
 * loads (the rows with their padding). */
int gw_stream_read_probe(gw_table *t, void *hip_stream, uint64_t *bytes_read);
int gw_synchronize(gw_table *t);
/* The kernel instantiation the handle's last case/control scan was dispatched to, spelled the way
 * rocprofv3's kernel trace prints it (e.g. "gw::k_scan_cc_pipe<2, true, true>"); "" before the first
 * scan.  bench.py reports it as roofline.kernel and matches it against the stamp of the committed PMC
 * traffic figure. */
const char *gw_last_scan_kernel(const gw_table *t);

/* kernel-shape knobs for A/B measurement (defaults are chosen at create) */
typedef enum gw_option {
    GW_OPT_UNROLL = 1,       /* 16-byte units per lane in flight per plane: 0 (auto),1,2,4,8 */
    GW_OPT_MASK_IN_LDS = 2,  /* 1: stage masks in LDS (default when they fit); 0: read via L2 */
    GW_OPT_BLOCKS_PER_CU = 3,/* persistent workgroups per CU (0 = auto)          */
    GW_OPT_NONTEMPORAL = 4,  /* 1: nt loads for the genotype stream (default)     */
    GW_OPT_BLOCK_THREADS = 5,/* threads per workgroup: 256, 512, 1024 (0 = auto) */
    GW_OPT_COMPLEMENT = 7,   /* when the control mask is exactly the complement of the case mask the scan
                                can stage one mask in LDS and take controls = all - cases: 0 never,
                                1 (default) only where two masks do not fit the LDS (650 k < N <= 1.3 M),
                                2 whenever possible; get returns whether the current masks/shape use it */
    GW_OPT_BOOST_PREFILTER = 8, /* gw_boost_screen without max_measure / min_measure: 1 (default) evaluate a
                                single-precision bound of each pair's measure first, together with a running
                                error bound err with |bound - measure| <= err (derived in
                                libgwaspp_amd/csrc/gw_boost_dev.h: ~6e-8 * (145 n + 30 |s| + ...), 0.09 at
                                n = 10 k), and run the FP64 measure only where bound + err >= threshold -- the hit
                                list and its FP64 measures are those of the all-FP64 screen; 0 never;
                                2 / 3 diagnostic: gw_boost_measures returns the bound / its error bound
                                (4 / 5: those of the first level, GW_OPT_BOOST_FIRST_LEVEL) */
    GW_OPT_BOOST_PREFILTER_TRIPS = 12, /* get: how many pre-filtered screens of this handle found a pair whose FP64
                                measure lay outside [bound - err, bound + err] (the error model's assumptions on
                                v_log_f32 / v_rcp_f32 failing) and were therefore repeated with every pair in
                                FP64; set 1: treat every screen as such (test hook), 0: normal */
    GW_OPT_BOOST_MATRIX_CORES = 9, /* pair tables as int8 MFMA dot products of class vectors (1, default: four
                                products per pair when no call of the table is missing, nine otherwise)
                                or AND + popcount on the vector ALU (0); 2 = as 1 but always the kernels
                                that keep both groups' accumulators (what case groups >= 65 536 members
                                get) */
    GW_OPT_SHORT_ROWS = 10,  /* rows of at most 256 units (N <= 32 k) with every other shape option on auto:
                                1 (default) several rows per wave at a time, masks in registers; 0 the
                                row-at-a-time kernel; get returns whether the current table uses it */
    GW_OPT_COLUMN_SEGMENT = 11, /* column-segmented scan: the samples are walked in segments whose two masks fit
                                the LDS, six partial sums per SNP accumulate in a scratch array and a last
                                kernel finishes the rows.  0 (default) auto = only where one pass would read
                                the masks through L2 (N > ~650 k with arbitrary masks, > ~1.3 M with
                                complementary ones), segments of 4096 units (524 288 samples); -1 never;
                                > 0 that many 16-byte units per segment; get returns the segment in use */
    GW_OPT_RESERVE_CUS = 14, /* compute units the persistent scan grid leaves free (default 0).  A sharded job whose
                                gather runs beside the next scan sets a few: RCCL's kernel (37 664 B of LDS, up to
                                512 threads of 248 VGPRs) cannot share a CU with a long-row scan workgroup, and a
                                scan workgroup that has to wait for one finishes its static share of the rows late */
    GW_OPT_BOOST_FP4 = 18,   /* pair screen: 1 (default) the pair tables counted by v_mfma_scale_f32_32x32x64_f8f6f4 on FP4
                                operands -- an operand dword carries 8 samples (one per nibble) and an MFMA 64 per lane,
                                twice the int8 form's, exact counts in FP32 accumulators: k_boost_mfma_fp4 for case groups
                                < 65 536 (four corners, or nine cells for tables with missing calls), k_boost_mfma_fp4_both
                                for larger case groups of tables without a missing call (cohorts < 2^24); 0 the int8
                                forms (k_boost_mfma_seq, k_boost_mfma) */
    GW_OPT_BOOST_FIRST_LEVEL = 19, /* pre-filtered pair screen on the FP4 kernels (four corners; since round 4 also the nine-cell
                                form for tables with missing calls, boost_bound_quad9_f): 1 (default) a logarithm-free
                                upper bound of the measure first (the binary entropy of every cell replaced by a quadratic
                                that lies below it: nine v_rcp_f32, a third of the vector work; measure <= bound1 + err1,
                                derived in gw_boost_dev.h boost_bound_quad), the bound proper only for the ~3 pairs in
                                10 000 it lets through, the FP64 measure for what passes both -- the hit list is that of
                                the all-FP64 screen; 0 the bound proper for every pair.  GW_OPT_BOOST_PREFILTER 4 / 5
                                (diagnostic): gw_boost_measures returns the first-level bound / its error bound */
    GW_OPT_BOOST_LUT = 17,   /* RETIRED in round 4 (number kept): the LDS-table pair kernel of round 3, measured slower
                                (123 vs 117 ms on config 5), is no longer in the library (tools/experiments/); setting a
                                non-zero value is GW_E_INVALID, get returns 0 */
    GW_OPT_BOOST_STAGGER_US = 16, /* RETIRED in round 4 (number kept): the start-up stagger of co-resident workgroups
                                measured no effect; setting a positive value is GW_E_INVALID, get returns 0 */
    GW_OPT_BOOST_CANDIDATES = 20, /* pre-filtered pair screen on the FP4 kernels: the pairs the in-kernel bound cannot rule out
                                go to a candidate list in HBM (40 B each; 80 B for tables with missing calls) and get their
                                second-level bound / FP64 measure from a finishing kernel.  0 (default) auto = one record
                                per 64 pairs of the block pair + 65 536, at least 2^20 (never more than one per pair), at
                                most 32 M; > 0 that many records.  A block
                                whose candidates do not fit is repeated with every pair in FP64 (counted by
                                GW_OPT_BOOST_PREFILTER_TRIPS), so the value bounds cost, never results */
    GW_OPT_BOOST_ROWS_PER_LAUNCH = 21, /* pre-filtered pair screen: a block pair [a, a + n_a) x [b, b + n_b) is screened
                                n rows of its first range at a time (rounded up to a multiple of 128), every row block
                                through the same candidate list, hits and tallies accumulating -- same results as one launch.
                                0 (default) auto = one launch, unless the list's cap of 32 M records would leave the block pair
                                less than a record per 512 pairs (n_a x n_b > 1.7e10: all pairs of ~130 000 SNPs and more in
                                one call): then (32 M - 65 536) x 64 / n_b rows per launch, a record per 64 pairs again */
    GW_OPT_DEFER_RECORDS = 15, /* rows up to ~100 k samples: 1 = nothing is written while the table is being read; a wave
                                parks the six sums of its finished rows in LDS and, after its last row, runs the FP64
                                epilogue over all of them and writes its records in one burst (launches are cut
                                into pieces that fit the LDS).  Same results bit for bit.  -1 (default) auto = rows of
                                at most 80 units (N <= 10 k) when one call's records exceed the 256 MB Infinity
                                Cache; 0 never.  get returns whether a whole-table scan with statistics uses it */
    GW_OPT_PIPELINE = 6      /* continuous-stream kernel (next item's loads in flight while the current one
                                is consumed, across row boundaries; inline-asm loads + counted waits):
                                -1 (default) auto = where the masks allow one workgroup per CU only
                                (N > ~320 k), 0 never, 1 whenever the masks are in LDS */
} gw_option;
int gw_set_option(gw_table *t, int option, int64_t value);
int gw_get_option(gw_table *t, int option, int64_t *value);

/* ---- host-only helpers (no device needed) ---------------------------------
 * The packer behind gw_table_add_genotype_row, exposed so that the host logic
 * can be checked without a GPU: packs one text row into a reference packing
 * (row must hold the layout's blocks-per-row uint16, header first). */
int gw_pack_genotype_row(const char *begin, const char *end, char delim, uint32_t n_samples,
                         int layout, uint16_t *row);
/* One step of the row labeller for GenoTable::addGenotype(r, c, gt)
 * (table5.cpp:156-194): *header is the row's current header block, encoding =
 * 4*a1+a2 over "ACGT"; *code gets the 2-bit code and *header is updated when
 * the genotype is new to the row.  GW_E_FORMAT where the reference asserts. */
int gw_header_assign(uint16_t *header, uint32_t encoding, int *code);
/* CaseControlSet::setCases / setControls (case_control_set.cpp:77-150) in the
 * stream packing: idx[n_idx] column indices -> mask of gw_stream_blocks(n)
 * uint16; *count = members (duplicates once). */
int gw_build_stream_mask(const int32_t *idx, size_t n_idx, uint32_t n_samples, uint16_t *mask,
                         uint32_t *count);
uint32_t gw_stream_blocks(uint32_t n_samples); /* N/16+1 -> x4 (table5.cpp:52-60) */
uint32_t gw_block_blocks(uint32_t n_samples);  /* N/8+1  -> x4 (table3.cpp:52-60) */

#ifdef __cplusplus
}
#endif
#endif /* GWAS_SCAN_H */
