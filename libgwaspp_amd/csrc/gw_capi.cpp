// gw_capi.cpp -- the C-ABI of include/gwas_scan.h on top of the gfx950
// kernels.  Compiled with hipcc into libgwaspp_amd/libgwscan.so.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <atomic>
#include <cmath>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gwas_scan.h"
#include "gw_kernels.h"
#include "gw_pack.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define GW_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            (void)hipGetLastError(); /* reset the runtime's sticky "last error" */         \
            return fail(e__ == hipErrorOutOfMemory ? GW_E_NOMEM : GW_E_HIP, "%s: %s", #expr, \
                        hipGetErrorString(e__));                                           \
        }                                                                                  \
    } while (0)

constexpr size_t kStagingBytes = 64u << 20;

}  // namespace

struct gw_table {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    gw::TableView tv{};
    gw_geometry geo{};
    int compute_units = 0;
    size_t lds_cap = 0;
    bool zero_pending = false;  // the table's zero fill has not happened yet (see use_device)

    gw::u32x4 *d_case = nullptr, *d_ctrl = nullptr;
    uint32_t case_count = 0, ctrl_count = 0;
    bool masks_set = false;
    bool masks_complementary = false;  // every sample is a case or a control, none both
    bool masks_overlap = false;        // some sample is in both masks (selectCaseControl then keeps it as a case only)
    std::vector<uint16_t> h_case, h_ctrl;  // reference stream packing, as handed in

    std::vector<uint16_t> headers;  // row header blocks (table5.cpp:363)

    uint8_t *d_staging = nullptr;  // reference-packed rows in flight to/from HBM
    uint8_t *h_pinned = nullptr;   // pinned host twin of d_staging (pitched sources / destinations)
    size_t staging_bytes = 0;
    uint32_t *d_scratch32 = nullptr;  // results of the host-pointer scan forms
    size_t scratch32_bytes = 0;
    double *d_scratch64 = nullptr;
    size_t scratch64_bytes = 0;
    uint32_t *d_misc = nullptr;  // 16 x uint32 (probe sink, phenotype counts)

    // pairwise path: selectCaseControl's compacted planes + per-SNP margins
    uint32_t *d_cc = nullptr;
    uint64_t cc_stride_w = 0;
    uint32_t wc = 0, wt = 0;
    gw::SnpMargin *d_mar = nullptr;
    gw::SnpMarginF *d_marf = nullptr;
    gw::u32x4 *d_cct = nullptr;  // block-tiled copy of d_cc for the MFMA pair kernel (tables without missing calls)
    double *d_frac = nullptr, *d_xlogx = nullptr;  // k/n and (k/n) log(k/n), k = 0..n
    bool selected_valid = false, any_missing = true;
    uint64_t prefilter_trips = 0;  // screens repeated in FP64 because a pair fell outside the bound's error model
    uint32_t sel_individs = 0;  // nCases + nControls of the compacted rows
    uint32_t sel_cases = 0;     // ... and nCases
    unsigned long long *d_tally = nullptr;
    void *d_hits_buf = nullptr;    // hit list / dense block of the pair screens, kept between calls (grow-only)
    size_t hits_buf_bytes = 0;
    double *d_dense_buf = nullptr;
    size_t dense_buf_bytes = 0;
    uint32_t *d_cand = nullptr;    // the pair screen's candidate records (gw_kernels.h BoostCandidateList), grow-only
    size_t cand_bytes = 0;

    // rows packed by gw_table_add_genotype_row, waiting for the next flush
    std::vector<uint16_t> pending;
    uint64_t pending_begin = 0;
    uint32_t pending_rows = 0, pending_cap = 0;

    // kernel shape
    int opt_unroll = 0, opt_mask_lds = -1, opt_blocks_per_cu = 0, opt_nt = 1, opt_threads = 0, opt_pipeline = -1, opt_complement = 1, opt_prefilter = 1, opt_boost_mfma = 1, opt_short_rows = 1, opt_segment = 0, opt_prefilter_selftest = 0, opt_reserve_cus = 0, opt_defer = -1, opt_boost_fp4 = 1, opt_boost_first_level = 1;
    long long opt_boost_candidates = 0;  // records the candidate list may hold; 0 = auto (from the size of the block pair)
    long long opt_boost_rows = 0;        // rows of a block pair per launch of the pre-filtered screen; 0 = auto (whole, unless the list's cap says otherwise)
    char last_boost_kernel[96] = "";  // ... and the last pair screen (gw_last_boost_kernel)
    char last_scan_kernel[96] = "";  // instantiation the last case/control scan launched (gw_last_scan_kernel)
    uint32_t *d_raw = nullptr;  // [row][6] partial sums of a column-segmented scan
    size_t raw_bytes = 0;
};

namespace {

// Every entry point that touches the table comes through here.  The zero fill a new table owes its callers
// ("unknown" everywhere, table5.cpp:82-83) happens at the first such call instead of at create, so that a table
// which is generated in place (gw_table_simulate overwrites every byte, padding included) never pays for it --
// 20.7 ms of memset for the 125 GB of config 3.
int use_device(gw_table *t, bool overwrites_whole_table = false) {
    GW_HIP(hipSetDevice(t->device));
    if (overwrites_whole_table) t->zero_pending = false;
    if (t->zero_pending) {
        t->zero_pending = false;
        GW_HIP(hipMemsetAsync(t->tv.base, 0, t->geo.device_table_bytes, t->stream));
        GW_HIP(hipStreamSynchronize(t->stream));  // callers may launch on streams of their own
    }
    return GW_OK;
}

hipStream_t pick_stream(gw_table *t, void *s) { return s ? (hipStream_t)s : t->stream; }

// `seg_units` > 0: the shape for one column segment of that many units (row-at-a-time or continuous-stream
// kernel with both masks in LDS; no single-mask or sub-wave form)
gw::ScanShape scan_shape(const gw_table *t, uint32_t seg_units = 0, uint64_t n_rows = 0, bool with_stats = true) {
    const uint32_t units = seg_units ? seg_units : t->tv.units;
    // Defaults from the round-1 sweeps on MI355X (profiles/r01_sweep_*.txt):
    //   masks fill more than half the LDS (N > ~320 k): one 256-thread workgroup per CU (one wave per
    //     SIMD) running the continuous-stream kernel with two 2-unit buffers per lane and plane --
    //     fewer, fatter streams read HBM best (without it: 512 threads, 2 units in flight);
    //   masks fit several times: 1024-thread workgroups, 1 unit in flight, up to 4 per CU;
    //   masks do not fit (N > ~650 k): read through L2, 4 units in flight;
    //   short rows (N <= ~32 k): 256-thread workgroups, 8 per CU.
    gw::ScanShape sh{};
    sh.compute_units = t->compute_units;
    sh.reserve_cus = std::min(std::max(t->opt_reserve_cus, 0), t->compute_units - 1);
    sh.lds_cap = t->lds_cap;
    sh.nontemporal = t->opt_nt;
    const bool want_lds = (t->opt_mask_lds < 0) ? true : (t->opt_mask_lds != 0);
    // single-mask kernel when the control mask is the complement of the case mask
    const size_t one_mask = (size_t)units * sizeof(gw::u32x4);
    // (auto: only where two masks do not fit the LDS -- 650 k < N <= 1.3 M, 86 % vs 75 % of the HBM
    // peak measured at N = 1 M; where both fit the two kernels run at the same speed)
    const bool compl_wanted = t->opt_complement == 2 || (t->opt_complement == 1 && 2 * one_mask > t->lds_cap);
    const bool compl_ok = !seg_units && compl_wanted && t->masks_complementary && want_lds && t->opt_nt && t->opt_pipeline != 1 &&
                          one_mask <= t->lds_cap;
    const size_t raw_need = compl_ok ? one_mask : 2 * one_mask;
    bool lds = want_lds && raw_need <= t->lds_cap;
    const bool one_per_cu = lds && 2 * raw_need > t->lds_cap;
    // continuous-stream (asm-pipelined) kernel: auto = where one workgroup per CU is all the LDS allows
    // (N > ~320 k): 4 waves per CU, one per SIMD, two 2-unit buffers in flight -- 87 % vs 85 % at c3
    sh.pipeline = t->opt_pipeline < 0 ? ((one_per_cu && !compl_ok && lds) ? 1 : 0) : (t->opt_pipeline && lds && !compl_ok);
    int unroll = t->opt_unroll ? t->opt_unroll : (!lds ? 4 : (one_per_cu ? 2 : 1));
    // the continuous-stream kernel exists for 1, 2, 4 units and keeps one zero entry behind each mask in LDS
    if (sh.pipeline && unroll > 4) unroll = 4;
    size_t need = raw_need;
    if (sh.pipeline) {
        need = (size_t)2 * (units + 1u) * sizeof(gw::u32x4);
        if (need > t->lds_cap) { sh.pipeline = 0; need = raw_need; }
    }
    if (lds && need > t->lds_cap) lds = false;
    sh.complement = (compl_ok && lds) ? 1 : 0;
    sh.mask_in_lds = lds ? 1 : 0;
    sh.lds_bytes = lds ? need : 0;
    sh.unroll = unroll;
    const int lds_limit = lds ? (int)(t->lds_cap / need) : 8;
    const bool short_rows = units <= 256u;  // N <= ~32 k: rows of one or two wave-steps
    int threads = t->opt_threads ? t->opt_threads
                                 : (one_per_cu ? (sh.pipeline ? 256 : 512) : (short_rows ? 256 : 1024));
    if (sh.pipeline && threads > 512) threads = 512;  // k_scan_cc_pipe is built for <= 512 threads
    int per_cu = t->opt_blocks_per_cu ? t->opt_blocks_per_cu : (one_per_cu ? 1 : (short_rows ? 8 : 4));
    per_cu = std::max(1, std::min(per_cu, lds_limit));
    sh.block_threads = threads;
    // GW_OPT_RESERVE_CUS: CUs the persistent grid leaves free (for a collective's kernel running beside the scan)
    // GW_OPT_DEFER_RECORDS: row-at-a-time kernel with both masks in LDS and room for at least one parked group of 64
    // rows per wave beside them (fewer workgroups per CU where that is what it takes)
    const size_t with_park = need + (size_t)(threads / 64) * 1536u;
    sh.defer = (!seg_units && t->opt_defer > 0 && lds && !sh.complement && !sh.pipeline && sh.nontemporal && (unroll == 1 || unroll == 2) &&
                with_park <= t->lds_cap) ? 1 : 0;
    if (sh.defer) per_cu = std::max(1, std::min(per_cu, (int)(t->lds_cap / with_park)));
    sh.grid_blocks = (t->compute_units - sh.reserve_cus) * per_cu;
    // short rows with every shape option left on auto: several rows per wave at a time, masks in registers
    if (!seg_units && short_rows && t->opt_short_rows && !t->opt_unroll && !t->opt_threads && t->opt_mask_lds < 0 && t->opt_pipeline < 0 &&
        t->opt_nt && t->opt_complement != 2) {
        sh.short_rows = 1;
        sh.complement = 0;
        sh.pipeline = 0;
        sh.mask_in_lds = 0;
        sh.lds_bytes = 0;
        sh.unroll = 1;
        sh.block_threads = 256;
        sh.grid_blocks = (t->compute_units - sh.reserve_cus) * t->opt_blocks_per_cu;  // 0 = as many as the kernel's registers allow
        // Records deferred to the end of the launch (profiles/r03_defer_records_ab.txt): auto = the 16-lane kernels
        // (N <= 10 k) when one launch's records exceed the 256 MB Infinity Cache -- there the writes between the
        // reads cost 11 points at N = 10 k (62.5 -> 74.0 % at 4 M SNPs), while records that stay cache-resident
        // are cheaper written as they are finished (77.6 vs 71.4 % at 1 M SNPs)
        const uint64_t rows = n_rows ? n_rows : t->tv.n_snps;
        const uint64_t record_bytes = rows * (uint64_t)(with_stats ? 96u : 32u);
        sh.defer = t->opt_defer < 0 ? (units <= 80u && record_bytes > (256ull << 20)) : (t->opt_defer ? 1 : 0);
    }
    return sh;
}

int ensure_scratch(gw_table *t, size_t bytes32, size_t bytes64) {
    if (bytes32 > t->scratch32_bytes) {
        if (t->d_scratch32) GW_HIP(hipFree(t->d_scratch32));
        t->d_scratch32 = nullptr;
        t->scratch32_bytes = 0;
        GW_HIP(hipMalloc((void **)&t->d_scratch32, bytes32));
        t->scratch32_bytes = bytes32;
    }
    if (bytes64 > t->scratch64_bytes) {
        if (t->d_scratch64) GW_HIP(hipFree(t->d_scratch64));
        t->d_scratch64 = nullptr;
        t->scratch64_bytes = 0;
        GW_HIP(hipMalloc((void **)&t->d_scratch64, bytes64));
        t->scratch64_bytes = bytes64;
    }
    return GW_OK;
}

int check_range(const gw_table *t, uint64_t begin, uint64_t n) {
    if (begin > t->tv.n_snps || n > t->tv.n_snps - begin)
        return fail(GW_E_RANGE, "SNP range [%llu, +%llu) outside the table (%llu SNPs)",
                    (unsigned long long)begin, (unsigned long long)n,
                    (unsigned long long)t->tv.n_snps);
    if (n > 0xFFFFFFFFull) return fail(GW_E_RANGE, "at most 2^32-1 SNPs per call");
    return GW_OK;
}

bool overlapping(const std::vector<uint16_t> &cs, const std::vector<uint16_t> &ct) {
    for (size_t b = 0; b < cs.size(); ++b)
        if (cs[b] & ct[b]) return true;
    return false;
}

bool complementary(const std::vector<uint16_t> &cs, const std::vector<uint16_t> &ct, uint32_t n_samples) {
    for (uint32_t b = 0; b < cs.size(); ++b) {
        const uint32_t first = b * 16u;
        uint16_t valid = 0;
        if (first < n_samples) valid = (n_samples - first >= 16u) ? 0xFFFFu : (uint16_t)((1u << (n_samples - first)) - 1u);
        if ((uint16_t)(cs[b] & valid) != (uint16_t)(~ct[b] & valid)) return false;
        if ((cs[b] | ct[b]) & ~valid) return false;  // pad bits set (setAllAsCases): keep the two-mask kernel
    }
    return true;
}

}  // namespace

extern "C" {

const char *gw_last_error(void) { return g_err.c_str(); }
// for the library's other translation units (gw_comm.cpp): sets the calling thread's message, returns `code`
int gw_set_error_message(int code, const char *msg) {
    g_err = msg ? msg : "";
    return code;
}
const char *gw_version(void) { return "libgwscan 0.1 (gfx950)"; }

int gw_device_count(int *count) {
    if (!count) return fail(GW_E_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(GW_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return GW_OK;
}

int gw_device_info_get(int device, gw_device_info *out) {
    if (!out) return fail(GW_E_INVALID, "out is NULL");
    hipDeviceProp_t p;
    GW_HIP(hipGetDeviceProperties(&p, device));
    std::memset(out, 0, sizeof *out);
    std::snprintf(out->name, sizeof out->name, "%s", p.name);
    std::snprintf(out->arch, sizeof out->arch, "%s", p.gcnArchName);
    out->compute_units = p.multiProcessorCount;
    out->lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    out->hbm_bytes = p.totalGlobalMem;
    GW_HIP(hipSetDevice(device));
    size_t fr = 0, tot = 0;
    GW_HIP(hipMemGetInfo(&fr, &tot));
    out->hbm_free_bytes = fr;
    return GW_OK;
}

uint32_t gw_stream_blocks(uint32_t n_samples) { return gw::stream_blocks(n_samples); }
uint32_t gw_block_blocks(uint32_t n_samples) { return gw::block_blocks(n_samples); }

int gw_table_create(int device, uint32_t n_samples, uint64_t n_snps, gw_table **out) {
    if (!out) return fail(GW_E_INVALID, "out is NULL");
    *out = nullptr;
    if (n_samples == 0) return fail(GW_E_INVALID, "n_samples must be > 0");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(GW_E_HIP, "no HIP device: this library has no CPU fallback (%s)",
                    e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(GW_E_INVALID, "device %d of %d", device, ndev);
    GW_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    GW_HIP(hipGetDeviceProperties(&prop, device));

    gw_table *t = new (std::nothrow) gw_table();
    if (!t) return fail(GW_E_NOMEM, "out of host memory");
    t->device = device;
    t->compute_units = prop.multiProcessorCount;
    // LDS a single workgroup may take: 160 KiB on gfx950 (64 KiB without opt-in)
    t->lds_cap = std::max<size_t>(prop.sharedMemPerBlock, prop.maxSharedMemoryPerMultiProcessor);
    if (t->lds_cap > (160u << 10)) t->lds_cap = 160u << 10;

    gw_geometry &g = t->geo;
    g.n_samples = n_samples;
    g.n_snps = n_snps;
    g.stream_blocks = gw::stream_blocks(n_samples);
    g.block_blocks = gw::block_blocks(n_samples);
    g.stream_row_bytes = gw::stream_row_blocks(n_samples) * 2u;
    g.block_row_bytes = gw::block_row_blocks(n_samples) * 2u;
    // 16-byte units per plane: enough for the reference's words (gbo/4 x 8 bytes)
    const uint32_t words = g.stream_blocks / 4u;
    g.device_units = (words + 1u) / 2u;
    g.device_plane_bytes = ((g.device_units * 16u) + 127u) & ~127u;
    g.device_row_bytes = 2ull * g.device_plane_bytes;
    g.device_table_bytes = g.device_row_bytes * n_snps;
    g.algorithmic_bytes_per_snp = n_samples / 4u;

    t->tv.units = g.device_units;
    t->tv.plane_stride_u = g.device_plane_bytes / 16u;
    t->tv.row_stride_u = g.device_row_bytes / 16u;
    t->tv.n_samples = n_samples;
    t->tv.n_snps = n_snps;

    auto bail = [&](int code) {
        gw_table_destroy(t);
        return code;
    };
#define GW_HIP_T(expr)                                                                       \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            (void)hipGetLastError();                                                         \
            return bail(fail(e__ == hipErrorOutOfMemory ? GW_E_NOMEM : GW_E_HIP, "%s: %s",   \
                             #expr, hipGetErrorString(e__)));                                \
        }                                                                                    \
    } while (0)
    GW_HIP_T(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    GW_HIP_T(hipEventCreate(&t->ev_start));
    GW_HIP_T(hipEventCreate(&t->ev_stop));
    if (g.device_table_bytes) {
        GW_HIP_T(hipMalloc((void **)&t->tv.base, g.device_table_bytes));
        t->zero_pending = true;  // filled at the first use that does not overwrite the whole table (use_device)
    }
    // masks are zero-padded to a whole number of the widest scan step (8 x 64 units)
    const size_t mask_bytes = (size_t)((t->tv.units + 511u) / 512u) * 512u * 16u;
    GW_HIP_T(hipMalloc((void **)&t->d_case, mask_bytes));
    GW_HIP_T(hipMalloc((void **)&t->d_ctrl, mask_bytes));
    GW_HIP_T(hipMemsetAsync(t->d_case, 0, mask_bytes, t->stream));
    GW_HIP_T(hipMemsetAsync(t->d_ctrl, 0, mask_bytes, t->stream));
    GW_HIP_T(hipMalloc((void **)&t->d_misc, 16 * sizeof(uint32_t)));
    GW_HIP_T(hipMemsetAsync(t->d_misc, 0, 16 * sizeof(uint32_t), t->stream));
    const size_t max_ref_row = std::max(g.stream_row_bytes, g.block_row_bytes);
    t->staging_bytes = std::max(kStagingBytes, max_ref_row);
    GW_HIP_T(hipMalloc((void **)&t->d_staging, t->staging_bytes));
    GW_HIP_T(gw::prepare_scan_kernels(t->lds_cap));
    GW_HIP_T(hipStreamSynchronize(t->stream));
#undef GW_HIP_T
    try {
        t->headers.assign(n_snps, 0);
        t->pending_cap = (uint32_t)std::max<size_t>(1, std::min<size_t>(4096, (16u << 20) / g.stream_row_bytes));
        t->pending.resize((size_t)t->pending_cap * (g.stream_row_bytes / 2u));
    } catch (const std::bad_alloc &) {
        return bail(fail(GW_E_NOMEM, "out of host memory"));
    }
    *out = t;
    return GW_OK;
}

int gw_table_destroy(gw_table *t) {
    if (!t) return GW_OK;
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    if (t->tv.base) (void)hipFree(t->tv.base);
    if (t->d_case) (void)hipFree(t->d_case);
    if (t->d_ctrl) (void)hipFree(t->d_ctrl);
    if (t->d_staging) (void)hipFree(t->d_staging);
    if (t->h_pinned) (void)hipHostFree(t->h_pinned);
    if (t->d_scratch32) (void)hipFree(t->d_scratch32);
    if (t->d_scratch64) (void)hipFree(t->d_scratch64);
    if (t->d_misc) (void)hipFree(t->d_misc);
    if (t->d_cc) (void)hipFree(t->d_cc);
    if (t->d_mar) (void)hipFree(t->d_mar);
    if (t->d_marf) (void)hipFree(t->d_marf);
    if (t->d_cct) (void)hipFree(t->d_cct);
    if (t->d_raw) (void)hipFree(t->d_raw);
    if (t->d_frac) (void)hipFree(t->d_frac);
    if (t->d_xlogx) (void)hipFree(t->d_xlogx);
    if (t->d_cand) (void)hipFree(t->d_cand);
    if (t->d_tally) (void)hipFree(t->d_tally);
    if (t->d_hits_buf) (void)hipFree(t->d_hits_buf);
    if (t->d_dense_buf) (void)hipFree(t->d_dense_buf);
    if (t->ev_start) (void)hipEventDestroy(t->ev_start);
    if (t->ev_stop) (void)hipEventDestroy(t->ev_stop);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
    return GW_OK;
}

int gw_table_geometry(const gw_table *t, gw_geometry *out) {
    if (!t || !out) return fail(GW_E_INVALID, "NULL argument");
    *out = t->geo;
    return GW_OK;
}

// ---------------------------------------------------------------- ingest

int gw_table_upload_rows(gw_table *t, uint64_t snp_begin, uint64_t n_rows, const void *rows,
                         size_t row_stride_bytes, int layout) {
    if (!t || (!rows && n_rows)) return fail(GW_E_INVALID, "NULL argument");
    if (layout != GW_LAYOUT_STREAM && layout != GW_LAYOUT_BLOCK)
        return fail(GW_E_INVALID, "layout must be GW_LAYOUT_STREAM or GW_LAYOUT_BLOCK");
    int rc = check_range(t, snp_begin, n_rows);
    if (rc) return rc;
    const size_t ref_bytes = layout == GW_LAYOUT_STREAM ? t->geo.stream_row_bytes : t->geo.block_row_bytes;
    const uint32_t ref_blocks = layout == GW_LAYOUT_STREAM ? t->geo.stream_blocks : t->geo.block_blocks;
    if (row_stride_bytes < ref_bytes || (row_stride_bytes & 1u))
        return fail(GW_E_INVALID, "row stride %zu: need an even value >= %zu", row_stride_bytes, ref_bytes);
    if ((rc = use_device(t))) return rc;
    const uint64_t chunk_rows = std::max<uint64_t>(1, t->staging_bytes / ref_bytes);
    const uint8_t *src = static_cast<const uint8_t *>(rows);
    for (uint64_t done = 0; done < n_rows; done += chunk_rows) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(chunk_rows, n_rows - done);
        // rows that lie back to back go over as one transfer; a pitched source is gathered row by row
        // (a 2-D copy from pageable memory crawls at well under 1 GB/s)
        if (row_stride_bytes == ref_bytes) {
            GW_HIP(hipMemcpyAsync(t->d_staging, src + done * row_stride_bytes, (size_t)n * ref_bytes,
                                  hipMemcpyHostToDevice, t->stream));
        } else {
            if (!t->h_pinned) GW_HIP(hipHostMalloc((void **)&t->h_pinned, t->staging_bytes, hipHostMallocDefault));
            for (uint32_t r = 0; r < n; ++r)
                std::memcpy(t->h_pinned + (size_t)r * ref_bytes, src + (done + r) * row_stride_bytes, ref_bytes);
            GW_HIP(hipMemcpyAsync(t->d_staging, t->h_pinned, (size_t)n * ref_bytes, hipMemcpyHostToDevice, t->stream));
        }
        t->selected_valid = false;
        GW_HIP(gw::launch_restride_in(t->tv, snp_begin + done, n, t->d_staging, ref_bytes, layout,
                                      ref_blocks, t->stream));
        GW_HIP(hipStreamSynchronize(t->stream));
        for (uint32_t r = 0; r < n; ++r) {
            uint16_t h;
            std::memcpy(&h, src + (done + r) * row_stride_bytes, 2);
            t->headers[snp_begin + done + r] = h;
        }
    }
    return GW_OK;
}

int gw_table_download_rows(gw_table *t, uint64_t snp_begin, uint64_t n_rows, void *rows,
                           size_t row_stride_bytes, int layout) {
    if (!t || (!rows && n_rows)) return fail(GW_E_INVALID, "NULL argument");
    if (layout != GW_LAYOUT_STREAM && layout != GW_LAYOUT_BLOCK)
        return fail(GW_E_INVALID, "layout must be GW_LAYOUT_STREAM or GW_LAYOUT_BLOCK");
    int rc = check_range(t, snp_begin, n_rows);
    if (rc) return rc;
    if ((rc = gw_table_flush(t))) return rc;
    const size_t ref_bytes = layout == GW_LAYOUT_STREAM ? t->geo.stream_row_bytes : t->geo.block_row_bytes;
    const uint32_t ref_blocks = layout == GW_LAYOUT_STREAM ? t->geo.stream_blocks : t->geo.block_blocks;
    if (row_stride_bytes < ref_bytes || (row_stride_bytes & 1u))
        return fail(GW_E_INVALID, "row stride %zu: need an even value >= %zu", row_stride_bytes, ref_bytes);
    if ((rc = use_device(t))) return rc;
    const uint64_t chunk_rows = std::max<uint64_t>(1, t->staging_bytes / ref_bytes);
    uint8_t *dst = static_cast<uint8_t *>(rows);
    for (uint64_t done = 0; done < n_rows; done += chunk_rows) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(chunk_rows, n_rows - done);
        GW_HIP(gw::launch_restride_out(t->tv, snp_begin + done, n, t->d_staging, ref_bytes, layout,
                                       ref_blocks, t->stream));
        if (row_stride_bytes == ref_bytes) {
            GW_HIP(hipMemcpyAsync(dst + done * row_stride_bytes, t->d_staging, (size_t)n * ref_bytes,
                                  hipMemcpyDeviceToHost, t->stream));
            GW_HIP(hipStreamSynchronize(t->stream));
        } else {
            if (!t->h_pinned) GW_HIP(hipHostMalloc((void **)&t->h_pinned, t->staging_bytes, hipHostMallocDefault));
            GW_HIP(hipMemcpyAsync(t->h_pinned, t->d_staging, (size_t)n * ref_bytes, hipMemcpyDeviceToHost, t->stream));
            GW_HIP(hipStreamSynchronize(t->stream));
            for (uint32_t r = 0; r < n; ++r)
                std::memcpy(dst + (done + r) * row_stride_bytes, t->h_pinned + (size_t)r * ref_bytes, ref_bytes);
        }
        for (uint32_t r = 0; r < n; ++r)
            std::memcpy(dst + (done + r) * row_stride_bytes, &t->headers[snp_begin + done + r], 2);
    }
    return GW_OK;
}

int gw_table_get_headers(gw_table *t, uint64_t snp_begin, uint64_t n_rows, uint16_t *headers) {
    if (!t || (!headers && n_rows)) return fail(GW_E_INVALID, "NULL argument");
    int rc = check_range(t, snp_begin, n_rows);
    if (rc) return rc;
    if ((rc = gw_table_flush(t))) return rc;
    std::memcpy(headers, t->headers.data() + snp_begin, n_rows * sizeof(uint16_t));
    return GW_OK;
}

int gw_table_flush(gw_table *t) {
    if (!t) return fail(GW_E_INVALID, "NULL table");
    if (t->pending_rows == 0) return GW_OK;
    const uint32_t n = t->pending_rows;
    t->pending_rows = 0;
    return gw_table_upload_rows(t, t->pending_begin, n, t->pending.data(), t->geo.stream_row_bytes,
                                GW_LAYOUT_STREAM);
}

int gw_table_add_genotype_row(gw_table *t, uint64_t snp, const char *begin, const char *end, char delim) {
    (void)delim;  // the reference skips one character after each allele pair, whatever it is
    if (!t || !begin || !end) return fail(GW_E_INVALID, "NULL argument");
    if (snp >= t->tv.n_snps) return fail(GW_E_RANGE, "SNP %llu outside the table", (unsigned long long)snp);
    if (t->pending_rows && (snp != t->pending_begin + t->pending_rows || t->pending_rows == t->pending_cap)) {
        int rc = gw_table_flush(t);
        if (rc) return rc;
    }
    if (t->pending_rows == 0) t->pending_begin = snp;
    uint16_t *row = t->pending.data() + (size_t)t->pending_rows * (t->geo.stream_row_bytes / 2u);
    if (gw::pack_genotype_row(begin, end, t->geo.n_samples, false, row) != 0)
        return fail(GW_E_FORMAT, "SNP %llu: genotype row the reference asserts on (second heterozygote "
                                 "spelling, more than three genotypes, or too many samples)",
                    (unsigned long long)snp);
    ++t->pending_rows;
    return GW_OK;
}

int gw_table_add_genotype_rows(gw_table *t, uint64_t snp_begin, uint64_t n_rows, const char *const *begins,
                               const char *const *ends, char delim, int n_threads) {
    (void)delim;
    if (!t || ((!begins || !ends) && n_rows)) return fail(GW_E_INVALID, "NULL argument");
    int rc = check_range(t, snp_begin, n_rows);
    if (rc) return rc;
    if ((rc = gw_table_flush(t))) return rc;  // keep single-row additions ordered before the batch
    if (n_rows == 0) return GW_OK;
    unsigned workers = n_threads > 0 ? (unsigned)n_threads : std::max(1u, std::thread::hardware_concurrency());
    workers = (unsigned)std::min<uint64_t>(workers, n_rows);
    const size_t row_blocks = t->geo.stream_row_bytes / 2u;
    const uint64_t chunk_rows = std::max<uint64_t>(1, std::min<uint64_t>(n_rows, (256u << 20) / t->geo.stream_row_bytes));
    std::vector<uint16_t> packed;
    try {
        packed.resize((size_t)chunk_rows * row_blocks);
    } catch (const std::bad_alloc &) {
        return fail(GW_E_NOMEM, "out of host memory");
    }
    for (uint64_t done = 0; done < n_rows; done += chunk_rows) {
        const uint64_t n = std::min<uint64_t>(chunk_rows, n_rows - done);
        std::atomic<uint64_t> next(0);
        std::atomic<long long> bad(-1);
        auto work = [&]() {
            for (;;) {
                const uint64_t k = next.fetch_add(1);
                if (k >= n) return;
                if (!begins[done + k] || !ends[done + k] ||
                    gw::pack_genotype_row(begins[done + k], ends[done + k], t->geo.n_samples, false,
                                          packed.data() + (size_t)k * row_blocks) != 0) {
                    long long expect = -1;
                    bad.compare_exchange_strong(expect, (long long)(done + k));
                }
            }
        };
        std::vector<std::thread> pool;
        for (unsigned w = 1; w < workers; ++w) pool.emplace_back(work);
        work();
        for (std::thread &th : pool) th.join();
        if (bad.load() >= 0)
            return fail(GW_E_FORMAT, "SNP %llu: genotype row the reference asserts on",
                        (unsigned long long)(snp_begin + (uint64_t)bad.load()));
        rc = gw_table_upload_rows(t, snp_begin + done, n, packed.data(), t->geo.stream_row_bytes, GW_LAYOUT_STREAM);
        if (rc) return rc;
    }
    return GW_OK;
}

int gw_pack_genotype_row(const char *begin, const char *end, char delim, uint32_t n_samples, int layout,
                         uint16_t *row) {
    (void)delim;
    if (!begin || !end || !row || n_samples == 0) return fail(GW_E_INVALID, "NULL argument");
    if (layout != GW_LAYOUT_STREAM && layout != GW_LAYOUT_BLOCK)
        return fail(GW_E_INVALID, "layout must be GW_LAYOUT_STREAM or GW_LAYOUT_BLOCK");
    if (gw::pack_genotype_row(begin, end, n_samples, layout == GW_LAYOUT_BLOCK, row) != 0)
        return fail(GW_E_FORMAT, "genotype row the reference asserts on");
    return GW_OK;
}

int gw_header_assign(uint16_t *header, uint32_t encoding, int *code) {
    if (!header || !code || encoding > 15u) return fail(GW_E_INVALID, "bad argument");
    const int c = gw::label_from_header(*header, encoding);
    if (c == 0) return fail(GW_E_FORMAT, "genotype the reference asserts on for this row");
    *code = c;
    return GW_OK;
}

int gw_build_stream_mask(const int32_t *idx, size_t n_idx, uint32_t n_samples, uint16_t *mask,
                         uint32_t *count) {
    if ((!idx && n_idx) || !mask || n_samples == 0) return fail(GW_E_INVALID, "NULL argument");
    long m = gw::build_stream_mask(idx, n_idx, n_samples, mask);
    if (m < 0) return fail(GW_E_RANGE, "sample index outside [0, %u)", n_samples);
    if (count) *count = (uint32_t)m;
    return GW_OK;
}

int gw_table_simulate(gw_table *t, uint64_t seed, uint64_t snp_global_begin, uint32_t missing_q16) {
    if (!t) return fail(GW_E_INVALID, "NULL table");
    if (missing_q16 > 65535u) return fail(GW_E_INVALID, "missing_q16 must be < 65536");
    int rc = use_device(t, true);
    if (rc) return rc;
    t->pending_rows = 0;
    t->selected_valid = false;
    GW_HIP(gw::launch_simulate(t->tv, seed, snp_global_begin, missing_q16, t->compute_units * 8, t->stream));
    GW_HIP(hipStreamSynchronize(t->stream));
    // alleles A (major) / C (minor), all three genotypes: order 7, AA=0, AC=1, CC=5
    std::fill(t->headers.begin(), t->headers.end(), (uint16_t)0x7015);
    return GW_OK;
}

int gw_table_simulate_exact(gw_table *t, uint64_t seed, uint64_t snp_global_begin, uint32_t missing_q16) {
    if (!t) return fail(GW_E_INVALID, "NULL table");
    if (missing_q16 > 65535u) return fail(GW_E_INVALID, "missing_q16 must be < 65536");
    int rc = use_device(t, true);
    if (rc) return rc;
    t->pending_rows = 0;
    t->selected_valid = false;
    GW_HIP(hipMemsetAsync(t->d_misc + 8, 0, sizeof(uint32_t), t->stream));
    GW_HIP(gw::launch_simulate_exact(t->tv, seed, snp_global_begin, missing_q16, t->compute_units * 4, t->d_misc + 8, t->stream));
    uint32_t overflow = 0;
    GW_HIP(hipMemcpyAsync(&overflow, t->d_misc + 8, sizeof overflow, hipMemcpyDeviceToHost, t->stream));
    GW_HIP(hipStreamSynchronize(t->stream));
    if (overflow)
        return fail(GW_E_RANGE, "gw_table_simulate_exact: %u rows of %u samples exceeded the selection list (cohort too large)",
                    overflow, t->geo.n_samples);
    std::fill(t->headers.begin(), t->headers.end(), (uint16_t)0x7015);
    return GW_OK;
}

int gw_table_simulate_phenotype(gw_table *t, uint64_t seed) {
    if (!t) return fail(GW_E_INVALID, "NULL table");
    int rc = use_device(t);
    if (rc) return rc;
    GW_HIP(hipMemsetAsync(t->d_misc, 0, 2 * sizeof(uint32_t), t->stream));
    GW_HIP(gw::launch_simulate_phenotype(seed, t->d_case, t->d_ctrl, t->tv.units, t->tv.n_samples,
                                         t->d_misc, t->stream));
    uint32_t counts[2];
    GW_HIP(hipMemcpyAsync(counts, t->d_misc, sizeof counts, hipMemcpyDeviceToHost, t->stream));
    const uint32_t nb = t->geo.stream_blocks;
    t->h_case.assign(nb, 0);
    t->h_ctrl.assign(nb, 0);
    // masks back in the reference stream packing: units hold the same bytes
    const size_t bytes = std::min<size_t>((size_t)nb * 2u, (size_t)t->tv.units * 16u);
    GW_HIP(hipMemcpyAsync(t->h_case.data(), t->d_case, bytes, hipMemcpyDeviceToHost, t->stream));
    GW_HIP(hipMemcpyAsync(t->h_ctrl.data(), t->d_ctrl, bytes, hipMemcpyDeviceToHost, t->stream));
    GW_HIP(hipStreamSynchronize(t->stream));
    t->case_count = counts[0];
    t->ctrl_count = counts[1];
    t->masks_set = true;
    t->masks_complementary = complementary(t->h_case, t->h_ctrl, t->geo.n_samples);
    t->masks_overlap = overlapping(t->h_case, t->h_ctrl);
    t->selected_valid = false;
    return GW_OK;
}

int gw_table_set_case_control(gw_table *t, const uint16_t *case_stream, const uint16_t *ctrl_stream,
                              uint32_t n_blocks, uint32_t case_count, uint32_t ctrl_count) {
    if (!t || !case_stream || !ctrl_stream) return fail(GW_E_INVALID, "NULL argument");
    if (n_blocks != t->geo.stream_blocks)
        return fail(GW_E_INVALID, "n_blocks %u: CaseControlSet stream masks of %u samples have %u blocks",
                    n_blocks, t->geo.n_samples, t->geo.stream_blocks);
    int rc = use_device(t);
    if (rc) return rc;
    t->h_case.assign(case_stream, case_stream + n_blocks);
    t->h_ctrl.assign(ctrl_stream, ctrl_stream + n_blocks);
    uint16_t *d_src = reinterpret_cast<uint16_t *>(t->d_staging);
    GW_HIP(hipMemcpyAsync(d_src, t->h_case.data(), (size_t)n_blocks * 2u, hipMemcpyHostToDevice, t->stream));
    GW_HIP(hipMemcpyAsync(d_src + n_blocks, t->h_ctrl.data(), (size_t)n_blocks * 2u, hipMemcpyHostToDevice,
                          t->stream));
    GW_HIP(gw::launch_mask_in(d_src, n_blocks, t->d_case, t->tv.units, t->tv.n_samples, t->stream));
    GW_HIP(gw::launch_mask_in(d_src + n_blocks, n_blocks, t->d_ctrl, t->tv.units, t->tv.n_samples, t->stream));
    GW_HIP(hipStreamSynchronize(t->stream));
    t->case_count = case_count;
    t->ctrl_count = ctrl_count;
    t->masks_set = true;
    t->masks_complementary = complementary(t->h_case, t->h_ctrl, t->geo.n_samples);
    t->masks_overlap = overlapping(t->h_case, t->h_ctrl);
    t->selected_valid = false;
    return GW_OK;
}

int gw_table_get_case_control(gw_table *t, uint16_t *case_stream, uint16_t *ctrl_stream, uint32_t n_blocks,
                              uint32_t *case_count, uint32_t *ctrl_count) {
    if (!t) return fail(GW_E_INVALID, "NULL table");
    if (!t->masks_set) return fail(GW_E_STATE, "no case/control masks installed");
    if (n_blocks != t->geo.stream_blocks) return fail(GW_E_INVALID, "n_blocks must be %u", t->geo.stream_blocks);
    if (case_stream) std::memcpy(case_stream, t->h_case.data(), (size_t)n_blocks * 2u);
    if (ctrl_stream) std::memcpy(ctrl_stream, t->h_ctrl.data(), (size_t)n_blocks * 2u);
    if (case_count) *case_count = t->case_count;
    if (ctrl_count) *ctrl_count = t->ctrl_count;
    return GW_OK;
}

// ------------------------------------------------------------------ scans

// Units per column segment, or 0 for the one-pass scan.  Auto: segment only where the one-pass kernels
// would have to read the masks through L2 (two masks beyond the LDS and no single-mask form: N > ~650 k
// with arbitrary masks, N > ~1.3 M with complementary ones) -- 4096 units = 524 288 samples, both masks
// 128 KiB, the shape of the c3 scan.
static uint32_t column_segment(const gw_table *t) {
    if (t->opt_segment > 0) return (uint32_t)t->opt_segment < t->tv.units ? (uint32_t)t->opt_segment : 0u;
    if (t->opt_segment < 0 || t->opt_mask_lds == 0) return 0;
    const gw::ScanShape sh = scan_shape(t);
    if (sh.mask_in_lds || sh.short_rows) return 0;
    return t->tv.units > 4096u ? 4096u : 0u;
}

int gw_scan_case_control_device(gw_table *t, uint64_t snp_begin, uint64_t n_snps, void *d_counts,
                                void *d_stats, void *hip_stream) {
    if (!t || (!d_counts && n_snps)) return fail(GW_E_INVALID, "NULL argument");
    if (!t->masks_set)
        return fail(GW_E_STATE, "gw_table_set_case_control must be called before a case/control scan");
    int rc = check_range(t, snp_begin, n_snps);
    if (rc) return rc;
    if ((rc = gw_table_flush(t))) return rc;
    if ((rc = use_device(t))) return rc;
    hipStream_t stream = pick_stream(t, hip_stream);
    const uint32_t seg = column_segment(t);
    if (!seg) {
        const gw::ScanShape sh = scan_shape(t, 0, n_snps, d_stats != nullptr);
        GW_HIP(gw::launch_scan_cc(t->tv, t->d_case, t->d_ctrl, t->case_count, t->ctrl_count, snp_begin,
                                  (uint32_t)n_snps, (uint32_t *)d_counts, (double *)d_stats, sh, stream, nullptr, false,
                                  t->last_scan_kernel));
        return GW_OK;
    }
    // Column-segmented scan: the masks of a segment fit the LDS, every segment streams its slice of all
    // rows at the full-speed kernel's rate and adds six sums per row to a scratch array; a last small
    // kernel turns the sums into counts and statistics.
    const size_t raw_bytes = n_snps * 6u * sizeof(uint32_t);
    if (raw_bytes > t->raw_bytes) {
        if (t->d_raw) GW_HIP(hipFree(t->d_raw));
        t->d_raw = nullptr;
        t->raw_bytes = 0;
        GW_HIP(hipMalloc((void **)&t->d_raw, raw_bytes));
        t->raw_bytes = raw_bytes;
    }
    for (uint32_t u0 = 0; u0 < t->tv.units; u0 += seg) {
        gw::TableView view = t->tv;
        view.base = t->tv.base + u0;
        view.units = std::min(seg, t->tv.units - u0);
        const gw::ScanShape sh = scan_shape(t, view.units);
        GW_HIP(gw::launch_scan_cc(view, t->d_case + u0, t->d_ctrl + u0, t->case_count, t->ctrl_count, snp_begin,
                                  (uint32_t)n_snps, nullptr, nullptr, sh, stream, t->d_raw, u0 != 0, t->last_scan_kernel));
    }
    GW_HIP(gw::launch_finish_raw(t->d_raw, (uint32_t)n_snps, t->case_count, t->ctrl_count, (uint32_t *)d_counts,
                                 (double *)d_stats, stream));
    return GW_OK;
}

int gw_scan_case_control(gw_table *t, uint64_t snp_begin, uint64_t n_snps, uint32_t *counts, double *stats) {
    if (!t || (!counts && n_snps)) return fail(GW_E_INVALID, "NULL argument");
    if (n_snps == 0) return check_range(t, snp_begin, 0);
    int rc = use_device(t);
    if (rc) return rc;
    const size_t b32 = n_snps * GW_N_COUNTS * sizeof(uint32_t), b64 = n_snps * GW_N_STATS * sizeof(double);
    if ((rc = ensure_scratch(t, b32, stats ? b64 : 0))) return rc;
    rc = gw_scan_case_control_device(t, snp_begin, n_snps, t->d_scratch32, stats ? t->d_scratch64 : nullptr,
                                     nullptr);
    if (rc) return rc;
    GW_HIP(hipMemcpyAsync(counts, t->d_scratch32, b32, hipMemcpyDeviceToHost, t->stream));
    if (stats) GW_HIP(hipMemcpyAsync(stats, t->d_scratch64, b64, hipMemcpyDeviceToHost, t->stream));
    GW_HIP(hipStreamSynchronize(t->stream));
    return GW_OK;
}

int gw_scan_distribution_device(gw_table *t, uint64_t snp_begin, uint64_t n_snps, void *d_dist,
                                void *hip_stream) {
    if (!t || (!d_dist && n_snps)) return fail(GW_E_INVALID, "NULL argument");
    int rc = check_range(t, snp_begin, n_snps);
    if (rc) return rc;
    if ((rc = gw_table_flush(t))) return rc;
    if ((rc = use_device(t))) return rc;
    // 512-thread workgroups: one per CU for long rows (few, fat streams), more for short ones
    const int per_cu = t->tv.units <= 256u ? 4 : (t->tv.units < 2000u ? 2 : 1);
    const int cus = std::max(1, t->compute_units - std::max(t->opt_reserve_cus, 0));  // GW_OPT_RESERVE_CUS
    GW_HIP(gw::launch_scan_dist(t->tv, snp_begin, (uint32_t)n_snps, (uint32_t *)d_dist, cus * per_cu,
                                pick_stream(t, hip_stream), t->opt_short_rows ? cus : 0));
    return GW_OK;
}

int gw_scan_distribution(gw_table *t, uint64_t snp_begin, uint64_t n_snps, uint32_t *dist) {
    if (!t || (!dist && n_snps)) return fail(GW_E_INVALID, "NULL argument");
    if (n_snps == 0) return check_range(t, snp_begin, 0);
    int rc = use_device(t);
    if (rc) return rc;
    const size_t b32 = n_snps * 4 * sizeof(uint32_t);
    if ((rc = ensure_scratch(t, b32, 0))) return rc;
    if ((rc = gw_scan_distribution_device(t, snp_begin, n_snps, t->d_scratch32, nullptr))) return rc;
    GW_HIP(hipMemcpyAsync(dist, t->d_scratch32, b32, hipMemcpyDeviceToHost, t->stream));
    GW_HIP(hipStreamSynchronize(t->stream));
    return GW_OK;
}

static int pair_common(gw_table *t, const uint32_t *pairs, uint64_t n_pairs, uint32_t *out_a, uint32_t *out_b,
                       bool masked) {
    if (!t || ((!pairs || !out_a) && n_pairs) || (masked && !out_b && n_pairs))
        return fail(GW_E_INVALID, "NULL argument");
    if (masked && !t->masks_set)
        return fail(GW_E_STATE, "gw_table_set_case_control must be called before a case/control table");
    if (n_pairs == 0) return GW_OK;
    if (n_pairs > 0x0FFFFFFFull) return fail(GW_E_RANGE, "too many pairs in one call");
    for (uint64_t i = 0; i < 2 * n_pairs; ++i)
        if (pairs[i] >= t->tv.n_snps) return fail(GW_E_RANGE, "pair %llu names SNP %u outside the table",
                                                  (unsigned long long)(i / 2), pairs[i]);
    int rc = gw_table_flush(t);
    if (rc) return rc;
    if ((rc = use_device(t))) return rc;
    const size_t pair_bytes = n_pairs * 2 * sizeof(uint32_t), tab_bytes = n_pairs * 16 * sizeof(uint32_t);
    if ((rc = ensure_scratch(t, pair_bytes + tab_bytes * (masked ? 2 : 1), 0))) return rc;
    uint32_t *d_pairs = t->d_scratch32;
    uint32_t *d_a = d_pairs + n_pairs * 2, *d_b = d_a + n_pairs * 16;
    GW_HIP(hipMemcpyAsync(d_pairs, pairs, pair_bytes, hipMemcpyHostToDevice, t->stream));
    const int grid = t->compute_units * 8;
    if (masked)
        GW_HIP(gw::launch_pair_cc_contingency(t->tv, t->d_case, t->d_ctrl, d_pairs, (uint32_t)n_pairs, d_a, d_b,
                                              grid, t->stream));
    else
        GW_HIP(gw::launch_pair_contingency(t->tv, d_pairs, (uint32_t)n_pairs, d_a, grid, t->stream));
    GW_HIP(hipMemcpyAsync(out_a, d_a, tab_bytes, hipMemcpyDeviceToHost, t->stream));
    if (masked) GW_HIP(hipMemcpyAsync(out_b, d_b, tab_bytes, hipMemcpyDeviceToHost, t->stream));
    GW_HIP(hipStreamSynchronize(t->stream));
    return GW_OK;
}

int gw_pair_contingency(gw_table *t, const uint32_t *pairs, uint64_t n_pairs, uint32_t *out) {
    return pair_common(t, pairs, n_pairs, out, nullptr, false);
}

int gw_pair_case_control_contingency(gw_table *t, const uint32_t *pairs, uint64_t n_pairs, uint32_t *out_case,
                                     uint32_t *out_ctrl) {
    return pair_common(t, pairs, n_pairs, out_case, out_ctrl, true);
}


// ------------------------------------------------------- pairwise / BOOST path

namespace {

// members of a stream mask in ascending sample order
void mask_members(const std::vector<uint16_t> &mask, uint32_t n_samples, const std::vector<uint16_t> *exclude,
                  std::vector<uint32_t> &out) {
    out.clear();
    for (uint32_t s = 0; s < n_samples; ++s) {
        const uint16_t bit = (uint16_t)(1u << (s & 15u));
        if (!(mask[s >> 4] & bit)) continue;
        if (exclude && ((*exclude)[s >> 4] & bit)) continue;  // a case wins over a control (table5.cpp:531-548)
        out.push_back(s);
    }
}

// selectCaseControl + computeMargins on the device (idempotent until data or masks change)
int ensure_selected(gw_table *t) {
    if (!t->masks_set)
        return fail(GW_E_STATE, "gw_table_set_case_control must be called before a case/control pair analysis");
    int rc = gw_table_flush(t);
    if (rc) return rc;
    if (t->selected_valid) return GW_OK;
    if ((rc = use_device(t))) return rc;
    const uint64_t m = t->tv.n_snps;
    std::vector<uint32_t> cases, ctrls;
    mask_members(t->h_case, t->geo.n_samples, nullptr, cases);
    mask_members(t->h_ctrl, t->geo.n_samples, &t->h_case, ctrls);
    // dense planes zero-padded to whole 32-word chunks (1024 members): k_boost stages a chunk at a time
    // with unconditional 16-byte loads
    const uint32_t wc = std::max<uint32_t>(32u, (uint32_t)((cases.size() + 1023u) / 1024u) * 32u);
    const uint32_t wt = std::max<uint32_t>(32u, (uint32_t)((ctrls.size() + 1023u) / 1024u) * 32u);
    // the compaction plan: per 32-sample word of a row and per group, the mask word with its compress network
    // ("Hacker's Delight" 7-4), the bit offset of its first member in the group's dense planes, its member count
    const uint32_t n_words = (t->geo.n_samples + 31u) / 32u;
    std::vector<uint32_t> src((size_t)n_words * 16u, 0u);
    for (int g = 0; g < 2; ++g) {
        const std::vector<uint32_t> &members = g == 0 ? cases : ctrls;
        for (uint32_t s : members) src[(size_t)(s >> 5) * 16u + 8u * g] |= 1u << (s & 31u);
        uint32_t offset = 0;
        for (uint32_t w = 0; w < n_words; ++w) {
            uint32_t *rec = &src[(size_t)w * 16u + 8u * g];
            uint32_t mask = rec[0], mk = ~mask << 1;
            for (int i = 0; i < 5; ++i) {
                uint32_t mp = mk ^ (mk << 1);
                mp ^= mp << 2; mp ^= mp << 4; mp ^= mp << 8; mp ^= mp << 16;
                const uint32_t mv = mp & mask;
                rec[1 + i] = mv;
                mask = (mask ^ mv) | (mv >> (1u << i));
                mk &= ~mp;
            }
            const uint32_t count = (uint32_t)__builtin_popcount(rec[0]);
            rec[6] = offset;
            rec[7] = count;
            offset += count;
        }
    }
    const uint64_t stride_w = 2ull * (wc + wt);
    // the compacted planes and their tiled copy depend on the group sizes: kept while those stay the same (new
    // masks of the same cohort -- another phenotype -- re-use them); the per-SNP margins never change size
    const bool same_geometry = t->d_cc && t->wc == wc && t->wt == wt;
    if (!same_geometry) {
        if (t->d_cc) { GW_HIP(hipFree(t->d_cc)); t->d_cc = nullptr; }
        if (t->d_cct) { GW_HIP(hipFree(t->d_cct)); t->d_cct = nullptr; }
    }
    // [0..3] the kernels' tally, [4] the candidate counter of the pair screen
    if (!t->d_tally) GW_HIP(hipMalloc((void **)&t->d_tally, 8 * sizeof(unsigned long long)));
    if (m) {
        if (!t->d_cc) GW_HIP(hipMalloc((void **)&t->d_cc, m * stride_w * sizeof(uint32_t)));
        if (!t->d_mar) GW_HIP(hipMalloc((void **)&t->d_mar, m * sizeof(gw::SnpMargin)));
        if (!t->d_marf) GW_HIP(hipMalloc((void **)&t->d_marf, m * sizeof(gw::SnpMarginF)));
    }
    uint32_t *d_src = nullptr;
    GW_HIP(hipMalloc((void **)&d_src, src.size() * sizeof(uint32_t)));
    hipError_t e = hipMemcpyAsync(d_src, src.data(), src.size() * sizeof(uint32_t), hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess) e = gw::launch_select_cc(t->tv, d_src, n_words, wc, wt, t->d_cc, stride_w, t->compute_units, t->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    (void)hipFree(d_src);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? GW_E_NOMEM : GW_E_HIP, "selectCaseControl on the device: %s",
                    hipGetErrorString(e));
    }
    t->wc = wc;
    t->wt = wt;
    t->cc_stride_w = stride_w;
    // computeMargins: the per-SNP case/control counts of the scan kernel -> marginal_information.
    // A sample in both masks is a case only here (as in the compacted rows), so the counts come
    // from masks with the overlap removed.
    std::vector<uint32_t> counts(m * GW_N_COUNTS);
    if (m) {
        std::vector<uint16_t> ctrl_only(t->h_ctrl);
        for (size_t b = 0; b < ctrl_only.size(); ++b) ctrl_only[b] = (uint16_t)(ctrl_only[b] & ~t->h_case[b]);
        const std::vector<uint16_t> keep_case(t->h_case), keep_ctrl(t->h_ctrl);  // copies: set_case_control
        const uint32_t keep_nc = t->case_count, keep_nt = t->ctrl_count;         // re-assigns the members
        const bool overlap = ctrl_only != keep_ctrl;
        if (overlap) {
            rc = gw_table_set_case_control(t, keep_case.data(), ctrl_only.data(), (uint32_t)ctrl_only.size(),
                                           (uint32_t)cases.size(), (uint32_t)ctrls.size());
            if (rc) return rc;
        }
        rc = gw_scan_case_control(t, 0, m, counts.data(), nullptr);
        if (overlap) {
            int rc2 = gw_table_set_case_control(t, keep_case.data(), keep_ctrl.data(), (uint32_t)keep_ctrl.size(),
                                                keep_nc, keep_nt);
            if (!rc) rc = rc2;
        }
        if (rc) return rc;
        GW_HIP(gw::launch_margins(t->d_scratch32, (uint32_t)m, (uint32_t)(cases.size() + ctrls.size()), t->d_mar,
                                  t->d_marf, t->stream));
        GW_HIP(hipStreamSynchronize(t->stream));
    }
    const uint32_t individs = (uint32_t)(cases.size() + ctrls.size());
    if (!t->d_frac || individs != t->sel_individs) {  // the k/n and (k/n) log(k/n) tables depend on n only
        if (t->d_frac) { GW_HIP(hipFree(t->d_frac)); t->d_frac = nullptr; }
        if (t->d_xlogx) { GW_HIP(hipFree(t->d_xlogx)); t->d_xlogx = nullptr; }
        GW_HIP(hipMalloc((void **)&t->d_frac, ((size_t)individs + 1u) * sizeof(double)));
        GW_HIP(hipMalloc((void **)&t->d_xlogx, ((size_t)individs + 1u) * sizeof(double)));
    }
    t->sel_individs = individs;
    t->sel_cases = (uint32_t)cases.size();
    GW_HIP(gw::launch_boost_tables(t->sel_individs + 1u, t->sel_individs, t->d_frac, t->d_xlogx, t->stream));
    GW_HIP(hipStreamSynchronize(t->stream));
    t->any_missing = false;
    for (uint64_t r = 0; r < m && !t->any_missing; ++r)
        t->any_missing = counts[r * 8 + 3] != 0 || counts[r * 8 + 7] != 0;
    if (m) {  // the MFMA pair kernels read a block-tiled copy
        const size_t bytes = ((m + 63u) / 64u) * 64u * (size_t)((wc + wt) / 4u) * 32u;
        if (!t->d_cct) GW_HIP(hipMalloc((void **)&t->d_cct, bytes));
        GW_HIP(hipMemsetAsync(t->d_cct, 0, bytes, t->stream));
        GW_HIP(gw::launch_retile_cc(t->d_cc, t->cc_stride_w, wc, wt, (uint32_t)m, t->d_cct, t->stream));
        GW_HIP(hipStreamSynchronize(t->stream));
    }
    t->selected_valid = true;
    return GW_OK;
}

double from_ordered(unsigned long long u) {
    u = (u & 0x8000000000000000ull) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
    double d;
    std::memcpy(&d, &u, sizeof d);
    return d;
}

// should this block pair go through the FP32 bound first?  One predicate for every form of the screen (host list, device
// list, list header): the bound's error model takes float(count) to be exact, so cohorts of 2^24 members or more get every
// pair in FP64
bool boost_prefilter_on(const gw_table *t) { return t->opt_prefilter != 0 && t->sel_individs < (1u << 24); }

// the candidate list of a screen over `pairs_upper` pairs at most: grow-only, owned by the handle.  Auto size: one
// record per 64 pairs + 65 536 but at least 2^20, at most one per pair and 32 M records (the logarithm-free first level lets ~3 pairs in 10 000 through
// on unlinked SNPs; a block that overflows it is repeated with every pair in FP64, so the size only bounds the cost)
int boost_candidates(gw_table *t, uint64_t n_a, uint64_t n_b, gw::BoostCandidateList *out) {
    const size_t rec = gw::boost_candidate_bytes(t->any_missing);
    const unsigned long long pairs_upper = (unsigned long long)n_a * n_b, cap = 32ull << 20;
    // (a block pair of up to 2^20 pairs gets a record per pair: whatever the threshold, it cannot overflow)
    unsigned long long want = t->opt_boost_candidates > 0
                                  ? (unsigned long long)t->opt_boost_candidates
                                  : std::min<unsigned long long>(std::max<unsigned long long>(pairs_upper / 64u + 65536u, 1ull << 20),
                                                                 std::min<unsigned long long>(std::max<unsigned long long>(pairs_upper, 1ull), cap));
    // a block pair that would be left with less than a record per 512 pairs at the cap (n_a x n_b > 1.7e10: all pairs of ~130 000
    // SNPs and more in one call; the first level lets ~1 pair in 3 300 through) is screened in row blocks that have the full
    // record-per-64-pairs allowance: (cap - 65 536) x 64 / n_b rows, a multiple of the 128-row tile
    out->rows_per_launch = 0;
    if (t->opt_boost_rows > 0)
        out->rows_per_launch = (uint32_t)std::min<long long>(((t->opt_boost_rows + 127) / 128) * 128, 1ll << 30);
    else if (t->opt_boost_candidates <= 0 && pairs_upper / 512u + 65536u > cap && n_b)
        out->rows_per_launch = (uint32_t)std::max<unsigned long long>(128ull, (((cap - 65536ull) * 64ull) / n_b) & ~127ull);
    const size_t need = (size_t)want * rec;
    if (need > t->cand_bytes) {
        if (t->d_cand) GW_HIP(hipFree(t->d_cand));
        t->d_cand = nullptr;
        t->cand_bytes = 0;
        GW_HIP(hipMalloc((void **)&t->d_cand, need));
        t->cand_bytes = need;
    }
    out->records = t->d_cand;
    out->count = t->d_tally + 4;
    out->capacity = want;
    return GW_OK;
}

// everything about a launch that comes from the handle (ensure_selected has run)
void boost_launch_of(gw_table *t, uint64_t a_begin, uint64_t n_a, uint64_t b_begin, uint64_t n_b, double threshold,
                     hipStream_t stream, gw::BoostLaunch &l) {
    l.d_cc = t->d_cc; l.cc_stride_w = t->cc_stride_w; l.wc = t->wc; l.wt = t->wt; l.d_cct = t->d_cct;
    l.d_mar = t->d_mar; l.d_marf = t->d_marf; l.d_frac = t->d_frac; l.d_xlogx = t->d_xlogx;
    l.n_individs = t->sel_individs; l.n_cases = t->sel_cases; l.no_missing_call = !t->any_missing;
    l.a_begin = (uint32_t)a_begin; l.n_a = (uint32_t)n_a; l.b_begin = (uint32_t)b_begin; l.n_b = (uint32_t)n_b;
    l.threshold = threshold;
    l.matrix_cores = t->opt_boost_mfma; l.fp4 = t->opt_boost_fp4 != 0; l.compute_units = t->compute_units;
    l.stream = stream; l.kernel_name = t->last_boost_kernel;
}

int boost_common(gw_table *t, uint64_t a_begin, uint64_t n_a, uint64_t b_begin, uint64_t n_b, double threshold,
                 gw_boost_hit *hits, uint64_t max_hits, uint64_t *n_hits, uint64_t *n_pairs, double *vmax,
                 double *vmin, double *dense) {
    if (!t) return fail(GW_E_INVALID, "NULL table");
    int rc = check_range(t, a_begin, n_a);
    if (rc) return rc;
    if ((rc = check_range(t, b_begin, n_b))) return rc;
    if (t->tv.n_snps > 0xFFFFFFFFull) return fail(GW_E_RANGE, "pair analysis supports up to 2^32-1 SNPs");
    if ((rc = ensure_selected(t))) return rc;
    if ((rc = use_device(t))) return rc;
    const unsigned long long init[4] = {0ull, 0ull, 0ull, ~0ull};
    GW_HIP(hipMemcpyAsync(t->d_tally, init, sizeof init, hipMemcpyHostToDevice, t->stream));
    // result buffers live in the handle and only ever grow: a caller that walks the pair space block by block
    // (sharded screens) pays for the allocation once
    gw::BoostHit *d_hits = nullptr;
    double *d_dense = nullptr;
    if (max_hits) {
        const size_t need = max_hits * sizeof(gw::BoostHit);
        if (need > t->hits_buf_bytes) {
            if (t->d_hits_buf) GW_HIP(hipFree(t->d_hits_buf));
            t->d_hits_buf = nullptr;
            t->hits_buf_bytes = 0;
            GW_HIP(hipMalloc(&t->d_hits_buf, need));
            t->hits_buf_bytes = need;
        }
        d_hits = static_cast<gw::BoostHit *>(t->d_hits_buf);
    }
    hipError_t e = hipSuccess;
    if (dense && n_a && n_b) {
        const size_t need = n_a * n_b * sizeof(double);
        if (need > t->dense_buf_bytes) {
            if (t->d_dense_buf) GW_HIP(hipFree(t->d_dense_buf));
            t->d_dense_buf = nullptr;
            t->dense_buf_bytes = 0;
            GW_HIP(hipMalloc((void **)&t->d_dense_buf, need));
            t->dense_buf_bytes = need;
        }
        d_dense = t->d_dense_buf;
        e = hipMemsetAsync(d_dense, 0xFF, need, t->stream);  // NaN
    }
    // FP32 bound first where nobody asked for the extremes: a pair is dropped without its FP64 measure only if
    // bound + err < threshold, err = the bound's own running error bound (gw_boost_dev.h boost_bound).
    // Options 2 / 3 = diagnostic: the dense form returns the bound / its error bound instead of the measure.
    // (the bound's error model takes float(count) to be exact: cohorts of 2^24 members or more get every pair in FP64)
    const bool diag = dense && t->opt_prefilter >= 2;
    const bool pre = diag || (!dense && !vmax && !vmin && boost_prefilter_on(t));
    float bound_thr = 0.0f;
    if (pre && !diag) {
        bound_thr = (float)threshold;
        if (!((double)bound_thr <= threshold)) bound_thr = std::nextafter(bound_thr, -INFINITY);  // round down; NaN stays NaN
    }
    gw::BoostLaunch l;
    boost_launch_of(t, a_begin, n_a, b_begin, n_b, threshold, t->stream, l);
    l.d_hits = d_hits; l.max_hits = max_hits; l.d_tally = t->d_tally; l.d_dense = d_dense;
    l.bound_threshold = bound_thr;
    l.dense_err = t->opt_prefilter == 3 || t->opt_prefilter == 5;
    l.first_level = diag ? t->opt_prefilter >= 4 : t->opt_boost_first_level != 0;
    if (pre && !dense && (rc = boost_candidates(t, n_a, n_b, &l.candidates))) return rc;
    auto launch = [&](bool prefilter) {
        l.prefilter = prefilter;
        return gw::launch_boost(l);
    };
    if (e == hipSuccess) e = launch(pre);
    unsigned long long tally[4] = {0, 0, 0, ~0ull};
    if (e == hipSuccess) e = hipMemcpyAsync(tally, t->d_tally, sizeof tally, hipMemcpyDeviceToHost, t->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    if (e == hipSuccess && pre && !diag && (tally[2] != 0 || t->opt_prefilter_selftest)) {
        // a pair whose FP64 measure lies further from its bound than the bound's error model allows: the model's
        // hardware assumptions do not hold here, so nothing the bound ruled out can be trusted -- the whole block
        // again with every pair in FP64 (never seen on gfx950; GW_OPT_BOOST_PREFILTER_TRIPS counts it)
        ++t->prefilter_trips;
        e = hipMemcpyAsync(t->d_tally, init, sizeof init, hipMemcpyHostToDevice, t->stream);
        if (e == hipSuccess) e = launch(false);
        if (e == hipSuccess) e = hipMemcpyAsync(tally, t->d_tally, sizeof tally, hipMemcpyDeviceToHost, t->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    }
    if (e == hipSuccess && hits && max_hits) {
        const uint64_t n = std::min<uint64_t>(tally[0], max_hits);
        if (n) e = hipMemcpy(hits, d_hits, n * sizeof(gw::BoostHit), hipMemcpyDeviceToHost);
    }
    if (e == hipSuccess && dense && d_dense) e = hipMemcpy(dense, d_dense, n_a * n_b * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? GW_E_NOMEM : GW_E_HIP, "BOOST pre-screen: %s", hipGetErrorString(e));
    }
    if (n_hits) *n_hits = tally[0];
    if (n_pairs) *n_pairs = tally[1];
    if (vmax) *vmax = tally[1] ? from_ordered(tally[2]) : 0.0;
    if (vmin) *vmin = tally[1] ? from_ordered(tally[3]) : 0.0;
    // the result buffers stay with the handle for the next block of a walk, but only up to 256 MB each: a one-off
    // large call (a dense block of 20 k x 20 k measures is 3.2 GB) must not sit next to a 125 GB table for ever
    const size_t keep = (size_t)256u << 20;
    if (t->dense_buf_bytes > keep) { (void)hipFree(t->d_dense_buf); t->d_dense_buf = nullptr; t->dense_buf_bytes = 0; }
    // (the candidate list of a whole-triangle screen is 1.3 GB at its 32 M-record cap -- next to the 0.25 - 3 GB tables that
    // get pair screens; allocating and freeing it around every call cost a computeBoost ~5 ms of its ~85)
    if (t->cand_bytes > 8u * keep) { (void)hipFree(t->d_cand); t->d_cand = nullptr; t->cand_bytes = 0; }
    if (t->hits_buf_bytes > keep) { (void)hipFree(t->d_hits_buf); t->d_hits_buf = nullptr; t->hits_buf_bytes = 0; }
    return GW_OK;
}

}  // namespace

int gw_boost_screen(gw_table *t, uint64_t a_begin, uint64_t n_a, uint64_t b_begin, uint64_t n_b, double threshold,
                    gw_boost_hit *hits, uint64_t max_hits, uint64_t *n_hits, uint64_t *n_pairs,
                    double *max_measure, double *min_measure) {
    static_assert(sizeof(gw_boost_hit) == sizeof(gw::BoostHit), "hit record layout");
    if (max_hits && !hits) return fail(GW_E_INVALID, "hits is NULL");
    return boost_common(t, a_begin, n_a, b_begin, n_b, threshold, hits, max_hits, n_hits, n_pairs, max_measure,
                        min_measure, nullptr);
}

// ---- the device-resident form of the screen: hits appended to a caller-owned list in HBM -----------------
int gw_boost_list_init(gw_table *t, void *d_list, void *hip_stream) {
    if (!t || !d_list) return fail(GW_E_INVALID, "NULL argument");
    int rc = use_device(t);
    if (rc) return rc;
    // the kernels' tally: hits, pairs, then (pre-filter on) pairs outside the bound's error model / unused,
    // or (pre-filter off) the ordered maximum / minimum of the measures
    // (the same predicate as the screen itself: with a cohort of 2^24 or more the screen runs without the pre-filter and
    // slots 2 / 3 are the extremes; the cohort is known once the masks are set, which a screen needs anyway)
    if ((rc = ensure_selected(t))) return rc;
    const bool pre = boost_prefilter_on(t);
    const unsigned long long init[4] = {0ull, 0ull, 0ull, pre ? 0ull : ~0ull};
    GW_HIP(hipMemcpyAsync(d_list, init, sizeof init, hipMemcpyHostToDevice, pick_stream(t, hip_stream)));
    GW_HIP(hipStreamSynchronize(pick_stream(t, hip_stream)));  // `init` lives on this stack
    return GW_OK;
}

int gw_boost_screen_device(gw_table *t, uint64_t a_begin, uint64_t n_a, uint64_t b_begin, uint64_t n_b, double threshold,
                           void *d_list, uint64_t capacity, void *hip_stream) {
    static_assert(GW_BOOST_LIST_HEADER_BYTES == 4 * sizeof(unsigned long long), "list header = the kernels' tally");
    if (!t || !d_list) return fail(GW_E_INVALID, "NULL argument");
    int rc = check_range(t, a_begin, n_a);
    if (rc) return rc;
    if ((rc = check_range(t, b_begin, n_b))) return rc;
    if (t->tv.n_snps > 0xFFFFFFFFull) return fail(GW_E_RANGE, "pair analysis supports up to 2^32-1 SNPs");
    if ((rc = ensure_selected(t))) return rc;  // compaction + margins, once per mask / data change (synchronous)
    if ((rc = use_device(t))) return rc;
    const bool pre = boost_prefilter_on(t);  // as in the host form and in gw_boost_list_init
    float bound_thr = 0.0f;
    if (pre) {
        bound_thr = (float)threshold;
        if (!((double)bound_thr <= threshold)) bound_thr = std::nextafter(bound_thr, -INFINITY);
    }
    gw::BoostLaunch l;
    boost_launch_of(t, a_begin, n_a, b_begin, n_b, threshold, pick_stream(t, hip_stream), l);
    l.d_tally = static_cast<unsigned long long *>(d_list);
    l.d_hits = reinterpret_cast<gw::BoostHit *>(static_cast<char *>(d_list) + GW_BOOST_LIST_HEADER_BYTES);
    l.max_hits = capacity;
    l.prefilter = pre; l.bound_threshold = bound_thr; l.first_level = t->opt_boost_first_level != 0;
    if (pre && (rc = boost_candidates(t, n_a, n_b, &l.candidates))) return rc;
    GW_HIP(gw::launch_boost(l));
    return GW_OK;
}

// ---- BOOST stage 2 on the device: computeGTest (epistasis_func.cpp:508-704) --------------------------------------------
namespace {
// scratch for n pairs: 18 uint32 of tables + 2 doubles of results each, + the status word (grow-only, in the handle's
// 64-bit scratch so that the doubles are aligned)
int gtest_scratch(gw_table *t, uint64_t n, uint32_t **tabs, double **out, unsigned long long **status) {
    const size_t out_bytes = (size_t)n * 2u * sizeof(double), tab_bytes = (size_t)n * 18u * sizeof(uint32_t);
    int rc = ensure_scratch(t, tab_bytes, out_bytes + 16u);
    if (rc) return rc;
    *tabs = t->d_scratch32;
    *status = reinterpret_cast<unsigned long long *>(t->d_scratch64);
    *out = t->d_scratch64 + 2;
    return GW_OK;
}
}  // namespace

int gw_boost_gtest(gw_table *t, const gw_boost_hit *pairs, uint64_t n, double *measure, double *z, uint64_t *n_unsettled) {
    if (!t || ((!pairs || !measure || !z) && n)) return fail(GW_E_INVALID, "NULL argument");
    if (n_unsettled) *n_unsettled = 0;
    if (n == 0) return GW_OK;
    if (n > 0x0FFFFFFFull) return fail(GW_E_RANGE, "too many pairs in one call");
    for (uint64_t k = 0; k < n; ++k)
        if (pairs[k].snp_a >= t->tv.n_snps || pairs[k].snp_b >= t->tv.n_snps)
            return fail(GW_E_RANGE, "pair %llu names a SNP outside the table", (unsigned long long)k);
    int rc = ensure_selected(t);
    if (rc) return rc;
    if ((rc = use_device(t))) return rc;
    uint32_t *d_tabs;
    double *d_out;
    unsigned long long *d_status;
    if ((rc = gtest_scratch(t, n, &d_tabs, &d_out, &d_status))) return rc;
    // the pairs travel as the hit records they are (16 bytes: snp_a, snp_b, the screen's measure)
    const size_t need = (size_t)n * sizeof(gw_boost_hit);
    if (need > t->hits_buf_bytes) {
        if (t->d_hits_buf) GW_HIP(hipFree(t->d_hits_buf));
        t->d_hits_buf = nullptr;
        t->hits_buf_bytes = 0;
        GW_HIP(hipMalloc(&t->d_hits_buf, need));
        t->hits_buf_bytes = need;
    }
    GW_HIP(hipMemcpyAsync(t->d_hits_buf, pairs, need, hipMemcpyHostToDevice, t->stream));
    GW_HIP(hipMemsetAsync(d_status, 0, 16, t->stream));
    GW_HIP(gw::launch_boost_gtest(t->d_cc, t->cc_stride_w, t->wc, t->wt, t->d_mar, t->sel_individs,
                                  static_cast<const uint32_t *>(t->d_hits_buf), 4u, nullptr, n, d_tabs, d_out, d_status,
                                  t->compute_units, t->stream));
    std::vector<double> both((size_t)n * 2u);
    unsigned long long status[2] = {0, 0};
    GW_HIP(hipMemcpyAsync(both.data(), d_out, both.size() * sizeof(double), hipMemcpyDeviceToHost, t->stream));
    GW_HIP(hipMemcpyAsync(status, d_status, sizeof status, hipMemcpyDeviceToHost, t->stream));
    GW_HIP(hipStreamSynchronize(t->stream));
    for (uint64_t k = 0; k < n; ++k) { measure[k] = both[2 * k]; z[k] = both[2 * k + 1]; }
    if (n_unsettled) *n_unsettled = status[0];
    return GW_OK;
}

int gw_boost_gtest_device(gw_table *t, const void *d_list, uint64_t capacity, void *d_measure_z, void *d_status,
                          void *hip_stream) {
    if (!t || !d_list || !d_measure_z) return fail(GW_E_INVALID, "NULL argument");
    if (capacity == 0) return GW_OK;
    if (capacity > 0x0FFFFFFFull) return fail(GW_E_RANGE, "too many pairs in one call");
    int rc = ensure_selected(t);
    if (rc) return rc;
    if ((rc = use_device(t))) return rc;
    uint32_t *d_tabs;
    double *d_unused;
    unsigned long long *d_own_status;
    if ((rc = gtest_scratch(t, capacity, &d_tabs, &d_unused, &d_own_status))) return rc;
    const unsigned long long *d_n = static_cast<const unsigned long long *>(d_list);   // header slot 0: hits found
    const uint32_t *d_hits = reinterpret_cast<const uint32_t *>(static_cast<const char *>(d_list) + GW_BOOST_LIST_HEADER_BYTES);
    GW_HIP(gw::launch_boost_gtest(t->d_cc, t->cc_stride_w, t->wc, t->wt, t->d_mar, t->sel_individs, d_hits, 4u, d_n, capacity,
                                  d_tabs, static_cast<double *>(d_measure_z), static_cast<unsigned long long *>(d_status),
                                  t->compute_units, pick_stream(t, hip_stream)));
    return GW_OK;
}

// ---- replication of a table's rows over the ranks of a communicator (the pair screen needs every row on every GPU)
int gw_table_replicate(gw_table *t, gw_comm *c, int root, void *hip_stream) {
    if (!t || !c) return fail(GW_E_INVALID, "NULL argument");
    int rc = gw_table_flush(t);
    if (rc) return rc;
    if ((rc = use_device(t))) return rc;   // (a receiver's pending zero fill happens here, before the rows land)
    hipStream_t s = pick_stream(t, hip_stream);
    if ((rc = gw_comm_broadcast(c, t->tv.base, t->geo.device_table_bytes, root, s))) return rc;
    // the row header blocks live on the host: through the staging buffer
    const size_t hb = t->headers.size() * sizeof(uint16_t);
    if (hb) {
        void *d_h = nullptr;
        GW_HIP(hipMalloc(&d_h, hb));
        hipError_t e = hipMemcpyAsync(d_h, t->headers.data(), hb, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            rc = gw_comm_broadcast(c, d_h, hb, root, s);
            if (!rc) e = hipMemcpyAsync(t->headers.data(), d_h, hb, hipMemcpyDeviceToHost, s);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void)hipFree(d_h);
        if (rc) return rc;
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(GW_E_HIP, "gw_table_replicate: %s", hipGetErrorString(e)); }
    } else {
        GW_HIP(hipStreamSynchronize(s));
    }
    t->selected_valid = false;
    return GW_OK;
}

int gw_shard_pair_blocks(uint64_t n_snps, uint32_t block, int rank, int world, uint64_t *a_begin, uint64_t *n_a,
                         size_t capacity, size_t *n_blocks) {
    if (!n_blocks || ((!a_begin || !n_a) && capacity)) return fail(GW_E_INVALID, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(GW_E_INVALID, "rank %d outside a world of %d", rank, world);
    if (block == 0) return fail(GW_E_INVALID, "block must be > 0");
    // row blocks dealt in snake order 0 .. W-1, W-1 .. 0, ...: block k holds ~ block x (n - k block) pairs, the
    // snake balances the triangle to well under 1 % (libgwaspp_amd/shard.py pair_row_blocks is the same cut)
    const uint64_t total = (n_snps + block - 1u) / block;
    size_t n = 0;
    for (uint64_t k = 0; k < total; ++k) {
        const uint64_t rnd = k / (uint64_t)world, pos = k % (uint64_t)world;
        const uint64_t owner = (rnd % 2u == 0u) ? pos : (uint64_t)world - 1u - pos;
        if (owner != (uint64_t)rank) continue;
        const uint64_t a = k * block, len = std::min<uint64_t>(block, n_snps - a);
        if (n && n <= capacity && a_begin[n - 1] + n_a[n - 1] == a) {  // adjacent row blocks: one launch
            n_a[n - 1] += len;
            continue;
        }
        if (n < capacity) { a_begin[n] = a; n_a[n] = len; }
        ++n;
    }
    *n_blocks = n;
    return GW_OK;
}

int gw_boost_measures(gw_table *t, uint64_t a_begin, uint64_t n_a, uint64_t b_begin, uint64_t n_b, double *measures) {
    if (!measures && n_a && n_b) return fail(GW_E_INVALID, "measures is NULL");
    if (n_a * n_b > (1ull << 28)) return fail(GW_E_RANGE, "block too large for the dense form");
    return boost_common(t, a_begin, n_a, b_begin, n_b, 1.0e300, nullptr, 0, nullptr, nullptr, nullptr, nullptr,
                        measures);
}

int gw_table_download_selected(gw_table *t, uint64_t snp_begin, uint64_t n_rows, uint16_t *rows, uint32_t *case_blocks,
                               uint32_t *ctrl_blocks) {
    if (!t) return fail(GW_E_INVALID, "NULL table");
    int rc = check_range(t, snp_begin, n_rows);
    if (rc) return rc;
    if ((rc = ensure_selected(t))) return rc;
    // reference geometry of the compacted rows (table5.cpp:449-468); a sample in both sets is a case
    std::vector<uint32_t> cases, ctrls;
    mask_members(t->h_case, t->geo.n_samples, nullptr, cases);
    mask_members(t->h_ctrl, t->geo.n_samples, &t->h_case, ctrls);
    const uint32_t cb = (((uint32_t)cases.size() / 16u + 1u) + 3u) & ~3u;
    const uint32_t tb = (((uint32_t)ctrls.size() / 16u + 1u) + 3u) & ~3u;
    if (case_blocks) *case_blocks = cb;
    if (ctrl_blocks) *ctrl_blocks = tb;
    if (!rows || !n_rows) return GW_OK;
    std::vector<uint32_t> buf(n_rows * t->cc_stride_w);
    GW_HIP(hipMemcpy(buf.data(), t->d_cc + snp_begin * t->cc_stride_w, buf.size() * sizeof(uint32_t),
                     hipMemcpyDeviceToHost));
    const size_t out_stride = 2u * (size_t)(cb + tb);
    std::memset(rows, 0, n_rows * out_stride * sizeof(uint16_t));
    auto copy_plane = [](uint16_t *dst, uint32_t dst_blocks, const uint32_t *src, uint32_t src_words) {
        const uint32_t n = std::min<uint32_t>(dst_blocks, src_words * 2u);
        std::memcpy(dst, src, (size_t)n * sizeof(uint16_t));  // little-endian: same bit order
    };
    for (uint64_t r = 0; r < n_rows; ++r) {
        const uint32_t *src = buf.data() + r * t->cc_stride_w;
        uint16_t *dst = rows + r * out_stride;
        copy_plane(dst, cb, src, t->wc);
        copy_plane(dst + cb, cb, src + t->wc, t->wc);
        copy_plane(dst + 2u * cb, tb, src + 2u * t->wc, t->wt);
        copy_plane(dst + 2u * cb + tb, tb, src + 2u * t->wc + t->wt, t->wt);
    }
    return GW_OK;
}

// marginal_information of SNPs [snp_begin, +n) as getCaseControlGenotypeDistribution(r, ccgd, m) returns it
// after selectCaseControl (table5.cpp:703-747 -> common_genotype_func.cpp:173-219)
int gw_scan_margins(gw_table *t, uint64_t snp_begin, uint64_t n_snps, gw_marginal_information *out) {
    static_assert(sizeof(gw_marginal_information) == sizeof(gw::SnpMargin) && sizeof(gw::SnpMargin) == 192,
                  "marginal_information record layout");
    if (!t || (!out && n_snps)) return fail(GW_E_INVALID, "NULL argument");
    if (!t->masks_set)
        return fail(GW_E_STATE, "gw_table_set_case_control must be called before gw_scan_margins");
    int rc = check_range(t, snp_begin, n_snps);
    if (rc) return rc;
    if (n_snps == 0) return GW_OK;
    if ((rc = gw_table_flush(t))) return rc;
    if ((rc = use_device(t))) return rc;
    const size_t bytes = n_snps * sizeof(gw::SnpMargin);
    // A sample in both masks is a case only once the table is "selected" (table5.cpp:531-548): that rare
    // case goes through the compaction's own margins; otherwise one scan of the range is all it takes.
    if (t->selected_valid || t->masks_overlap) {
        if ((rc = ensure_selected(t))) return rc;
        GW_HIP(hipMemcpyAsync(out, t->d_mar + snp_begin, bytes, hipMemcpyDeviceToHost, t->stream));
        GW_HIP(hipStreamSynchronize(t->stream));
        return GW_OK;
    }
    if ((rc = ensure_scratch(t, n_snps * GW_N_COUNTS * sizeof(uint32_t), bytes))) return rc;
    if ((rc = gw_scan_case_control_device(t, snp_begin, n_snps, t->d_scratch32, nullptr, nullptr))) return rc;
    GW_HIP(gw::launch_margins(t->d_scratch32, (uint32_t)n_snps, t->case_count + t->ctrl_count,
                              reinterpret_cast<gw::SnpMargin *>(t->d_scratch64), nullptr, t->stream));
    GW_HIP(hipMemcpyAsync(out, t->d_scratch64, bytes, hipMemcpyDeviceToHost, t->stream));
    GW_HIP(hipStreamSynchronize(t->stream));
    return GW_OK;
}

// ------------------------------------------------------------ measurement

int gw_timer_start(gw_table *t, void *hip_stream) {
    if (!t) return fail(GW_E_INVALID, "NULL table");
    int rc = use_device(t);
    if (rc) return rc;
    GW_HIP(hipEventRecord(t->ev_start, pick_stream(t, hip_stream)));
    return GW_OK;
}

int gw_timer_stop(gw_table *t, void *hip_stream, float *elapsed_ms) {
    if (!t || !elapsed_ms) return fail(GW_E_INVALID, "NULL argument");
    int rc = use_device(t);
    if (rc) return rc;
    GW_HIP(hipEventRecord(t->ev_stop, pick_stream(t, hip_stream)));
    GW_HIP(hipEventSynchronize(t->ev_stop));
    GW_HIP(hipEventElapsedTime(elapsed_ms, t->ev_start, t->ev_stop));
    return GW_OK;
}

int gw_stream_read_probe(gw_table *t, void *hip_stream, uint64_t *bytes_read) {
    if (!t) return fail(GW_E_INVALID, "NULL table");
    int rc = use_device(t);
    if (rc) return rc;
    GW_HIP(gw::launch_stream_probe(t->tv, t->d_misc + 8, t->compute_units, pick_stream(t, hip_stream)));
    // bytes the probe loads: the whole table allocation (rows with their padding)
    if (bytes_read) *bytes_read = t->tv.n_snps * t->tv.row_stride_u * 16ull;
    return GW_OK;
}

const char *gw_last_scan_kernel(const gw_table *t) { return t ? t->last_scan_kernel : ""; }
const char *gw_last_boost_kernel(const gw_table *t) { return t ? t->last_boost_kernel : ""; }

int gw_table_stream(gw_table *t, void **hip_stream) {
    if (!t || !hip_stream) return fail(GW_E_INVALID, "NULL argument");
    *hip_stream = (void *)t->stream;
    return GW_OK;
}

int gw_synchronize(gw_table *t) {
    if (!t) return fail(GW_E_INVALID, "NULL table");
    int rc = use_device(t);
    if (rc) return rc;
    GW_HIP(hipStreamSynchronize(t->stream));
    return GW_OK;
}

int gw_set_option(gw_table *t, int option, int64_t v) {
    if (!t) return fail(GW_E_INVALID, "NULL table");
    switch (option) {
    case GW_OPT_UNROLL:
        if (v != 0 && v != 1 && v != 2 && v != 4 && v != 8)
            return fail(GW_E_INVALID, "unroll must be 0 (auto), 1, 2, 4 or 8");
        t->opt_unroll = (int)v;
        return GW_OK;
    case GW_OPT_MASK_IN_LDS:
        if (v < -1 || v > 1) return fail(GW_E_INVALID, "mask_in_lds must be -1 (auto), 0 or 1");
        t->opt_mask_lds = (int)v;
        return GW_OK;
    case GW_OPT_BLOCKS_PER_CU:
        if (v < 0 || v > 8) return fail(GW_E_INVALID, "blocks_per_cu must be 0 (auto) .. 8");
        t->opt_blocks_per_cu = (int)v;
        return GW_OK;
    case GW_OPT_NONTEMPORAL:
        t->opt_nt = v ? 1 : 0;
        return GW_OK;
    case GW_OPT_BLOCK_THREADS:
        if (v != 0 && v != 256 && v != 512 && v != 1024)
            return fail(GW_E_INVALID, "block_threads must be 0 (auto), 256, 512 or 1024");
        t->opt_threads = (int)v;
        return GW_OK;
    case GW_OPT_PIPELINE:
        if (v < -1 || v > 1) return fail(GW_E_INVALID, "pipeline must be -1 (auto), 0 or 1");
        t->opt_pipeline = (int)v;
        return GW_OK;
    case GW_OPT_COMPLEMENT:
        if (v < 0 || v > 2) return fail(GW_E_INVALID, "complement must be 0 (never), 1 (auto) or 2 (whenever possible)");
        t->opt_complement = (int)v;
        return GW_OK;
    case GW_OPT_BOOST_MATRIX_CORES:
        if (v < 0 || v > 2) return fail(GW_E_INVALID, "boost_matrix_cores must be 0, 1 or 2");
        t->opt_boost_mfma = (int)v;
        return GW_OK;
    case GW_OPT_BOOST_LUT:
        if (v) return fail(GW_E_INVALID, "boost_lut: the LUT pair kernel was an experiment of round 3 and left the library in round 4 (tools/experiments/)");
        return GW_OK;
    case GW_OPT_BOOST_FIRST_LEVEL:
        t->opt_boost_first_level = v ? 1 : 0;
        return GW_OK;
    case GW_OPT_BOOST_FP4:
        t->opt_boost_fp4 = v ? 1 : 0;
        return GW_OK;
    case GW_OPT_BOOST_STAGGER_US:
        if (v > 0) return fail(GW_E_INVALID, "boost_stagger_us: the start-up stagger was an experiment of round 3 (no effect) and left the library in round 4");
        return GW_OK;
    case GW_OPT_BOOST_CANDIDATES:
        if (v < 0 || v > (1ll << 28)) return fail(GW_E_INVALID, "boost_candidates must be 0 (auto) or a record count up to 2^28");
        t->opt_boost_candidates = v;
        return GW_OK;
    case GW_OPT_BOOST_ROWS_PER_LAUNCH:
        if (v < 0 || v > (1ll << 30)) return fail(GW_E_INVALID, "boost_rows_per_launch must be 0 (auto) or a row count up to 2^30");
        t->opt_boost_rows = v;
        return GW_OK;
    case GW_OPT_DEFER_RECORDS:
        if (v < -1 || v > 1) return fail(GW_E_INVALID, "defer_records must be -1 (auto), 0 or 1");
        t->opt_defer = (int)v;
        return GW_OK;
    case GW_OPT_SHORT_ROWS:
        t->opt_short_rows = v ? 1 : 0;
        return GW_OK;
    case GW_OPT_RESERVE_CUS:
        if (v < 0 || v >= 128) return fail(GW_E_INVALID, "reserve_cus must be 0 .. 127");
        t->opt_reserve_cus = (int)v;
        return GW_OK;
    case GW_OPT_COLUMN_SEGMENT:
        if (v < -1 || v > (1 << 20)) return fail(GW_E_INVALID, "column_segment must be -1 (never), 0 (auto) or a unit count");
        t->opt_segment = (int)v;
        return GW_OK;
    case GW_OPT_BOOST_PREFILTER:
        if (v < 0 || v > 5) return fail(GW_E_INVALID, "boost_prefilter must be 0 (never), 1 (default) or 2 ... 5 (diagnostic)");
        t->opt_prefilter = (int)v;
        return GW_OK;
    case GW_OPT_BOOST_PREFILTER_TRIPS:  // set: 1 = treat every pre-filtered screen as tripped (exercises the FP64 repeat), 0 = normal
        t->opt_prefilter_selftest = v ? 1 : 0;
        return GW_OK;
    default:
        return fail(GW_E_INVALID, "unknown option %d", option);
    }
}

int gw_get_option(gw_table *t, int option, int64_t *v) {
    if (!t || !v) return fail(GW_E_INVALID, "NULL argument");
    const gw::ScanShape sh = scan_shape(t);
    switch (option) {
    case GW_OPT_UNROLL: *v = sh.unroll; return GW_OK;
    case GW_OPT_MASK_IN_LDS: *v = sh.mask_in_lds; return GW_OK;
    case GW_OPT_BLOCKS_PER_CU: *v = t->compute_units ? sh.grid_blocks / t->compute_units : 0; return GW_OK;
    case GW_OPT_NONTEMPORAL: *v = sh.nontemporal; return GW_OK;
    case GW_OPT_BLOCK_THREADS: *v = sh.block_threads; return GW_OK;
    case GW_OPT_PIPELINE: *v = sh.pipeline; return GW_OK;
    case GW_OPT_COMPLEMENT: *v = sh.complement; return GW_OK;
    case GW_OPT_BOOST_PREFILTER: *v = t->opt_prefilter; return GW_OK;
    case GW_OPT_BOOST_PREFILTER_TRIPS: *v = (int64_t)t->prefilter_trips; return GW_OK;
    case GW_OPT_SHORT_ROWS: *v = sh.short_rows; return GW_OK;
    case GW_OPT_DEFER_RECORDS: *v = sh.defer; return GW_OK;
    case GW_OPT_BOOST_STAGGER_US: *v = 0; return GW_OK;  // retired (round 4)
    case GW_OPT_BOOST_LUT: *v = 0; return GW_OK;         // retired (round 4)
    case GW_OPT_BOOST_CANDIDATES: *v = t->opt_boost_candidates; return GW_OK;
    case GW_OPT_BOOST_ROWS_PER_LAUNCH: *v = t->opt_boost_rows; return GW_OK;
    case GW_OPT_BOOST_FP4: *v = t->opt_boost_fp4; return GW_OK;
    case GW_OPT_BOOST_FIRST_LEVEL: *v = t->opt_boost_first_level; return GW_OK;
    case GW_OPT_COLUMN_SEGMENT: *v = column_segment(t); return GW_OK;
    case GW_OPT_RESERVE_CUS: *v = t->opt_reserve_cus; return GW_OK;
    case GW_OPT_BOOST_MATRIX_CORES: *v = t->opt_boost_mfma; return GW_OK;
    default: return fail(GW_E_INVALID, "unknown option %d", option);
    }
}

}  // extern "C"
