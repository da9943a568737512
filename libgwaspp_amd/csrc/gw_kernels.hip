// gw_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the per-SNP association
// scan.  Hand-written HIP; no MFMA: this is a popcount / HBM-bandwidth
// problem (about 0.4 integer ops per byte), so everything here is about
// 16-byte-per-lane coalesced streaming, the right number of bytes in flight per
// CU, and keeping the phenotype masks out of the HBM stream.
//
// Reference algorithm restated by each kernel (paths relative to
// /root/reference/src/libgwaspp/):
//   k_scan_cc        getCaseControlGenotypeDistribution(r, ccs, ccgd)
//   k_scan_cc_pipe   genotype/compressed_genotype_table5.cpp:609-657, inside the
//   k_scan_cc_compl  SNP loop of inline_cc_maf, algorithms/maf_func.cpp:292-316,
//                    + MinorAlleleFrequency, algorithms/maf_func.h:46-54, + the
//                    build-defined HWE / chi-square statistics.
//   k_scan_cc_sub    Four forms of the same scan: row-at-a-time; continuous
//                    stream (inline-asm loads, counted waits; the c3 default);
//                    single mask for complementary phenotypes (N up to 1.3 M);
//                    a sub-wave of 16 / 32 lanes per row for short rows (N <= 32 k).
//   k_scan_dist(_sub) getGenotypeDistribution, table5.cpp:577-607
//   k_pair           getContingencyTable / getCaseControlContingencyTable for a
//                    list of SNP pairs, table5.cpp:749-800 / 806-895
//   k_restride_*     the row packings of table5.cpp:277-365 / table3.cpp:260-325
//   k_simulate*      data/simulate_data.cpp:150-235
//   k_stream_probe   plain read of the table allocation (the empirical HBM ceiling)
// (the all-pairs BOOST path is in gw_boost*.hip)
#include "gw_kernels.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>

namespace gw {

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------

template <bool NT>
__device__ __forceinline__ u32x4 ld_stream(const u32x4 *p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over the 64 lanes without touching LDS: four DPP adds fold each 16-lane
// row (quad swaps, half mirror, full mirror), four v_readlane pick the row
// sums.  The result is wave-uniform (lives in an SGPR).
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// one 16-byte unit of both planes against both masks: 6 running popcounts
// (table5.cpp:629-646: aa += pc(p0&m), ab += pc(p1&m), bb += pc(p0&p1&m))
__device__ __forceinline__ void acc_unit(const u32x4 a, const u32x4 b, const u32x4 mc,
                                         const u32x4 mt, uint32_t (&c)[6]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t x = a[k] & mc[k], y = b[k] & mc[k];
        c[0] += __popc(x);
        c[1] += __popc(y);
        c[2] += __popc(x & y);
        x = a[k] & mt[k];
        y = b[k] & mt[k];
        c[3] += __popc(x);
        c[4] += __popc(y);
        c[5] += __popc(x & y);
    }
}

// ---- build-defined statistics (DESIGN.md "Statistics"); the operation order
// is the oracle's (oracle/gw_oracle.c gwo_assoc_stats), contraction is off.
__device__ __forceinline__ double ref_maf(double aa, double ab, double bb) {
    // MinorAlleleFrequency, algorithms/maf_func.h:46-54 (quirk Q5 kept)
    double t = aa;
    double m = 2.0 * t;
    t += ab;
    m += ab;
    t += bb;
    m /= t;
    if (m < 0.5) m = 1.0 - m;
    return m;
}

__device__ __forceinline__ double hwe_chi2(double aa, double ab, double bb) {
    double n = aa + ab + bb;
    if (n <= 0.0) return 0.0;
    double p = (2.0 * aa + ab) / (2.0 * n);
    double q = 1.0 - p;
    double e0 = n * p * p, e1 = 2.0 * n * p * q, e2 = n * q * q;
    double x = 0.0, d;
    if (e0 > 0.0) { d = aa - e0; x += d * d / e0; }
    if (e1 > 0.0) { d = ab - e1; x += d * d / e1; }
    if (e2 > 0.0) { d = bb - e2; x += d * d / e2; }
    return x;
}

__device__ __forceinline__ void assoc_stats(const uint32_t ca[4], const uint32_t co[4],
                                            double *out) {
    const double caa = ca[0], cab = ca[1], cbb = ca[2];
    const double taa = co[0], tab = co[1], tbb = co[2];
    const double r1 = caa + cab + cbb, r2 = taa + tab + tbb, n = r1 + r2;
    out[0] = (r1 > 0.0) ? ref_maf(caa, cab, cbb) : 0.0;
    out[1] = (r2 > 0.0) ? ref_maf(taa, tab, tbb) : 0.0;
    if (n > 0.0) {
        double p = (2.0 * (caa + taa) + (cab + tab)) / (2.0 * n);
        out[2] = (p > 0.5) ? 1.0 - p : p;
    } else {
        out[2] = 0.0;
    }
    out[3] = hwe_chi2(taa, tab, tbb);
    out[4] = hwe_chi2(caa + taa, cab + tab, cbb + tbb);
    {
        const double o[2][3] = {{caa, cab, cbb}, {taa, tab, tbb}};
        const double r[2] = {r1, r2};
        const double c[3] = {caa + taa, cab + tab, cbb + tbb};
        double x = 0.0;
        if (n > 0.0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double e = r[i] * c[j] / n;
                    if (e > 0.0) { double d = o[i][j] - e; x += d * d / e; }
                }
        }
        out[5] = x;
    }
    {
        double a = 2.0 * caa + cab, b = 2.0 * cbb + cab;
        double c = 2.0 * taa + tab, d = 2.0 * tbb + tab;
        double den = (a + b) * (c + d) * (a + c) * (b + d);
        double det = a * d - b * c;
        out[6] = (den > 0.0) ? (a + b + c + d) * det * det / den : 0.0;
    }
    {
        double members = r1 + (double)ca[3] + r2 + (double)co[3];
        out[7] = (members > 0.0) ? n / members : 0.0;
    }
}

// ---------------------------------------------------------------------------
// the scan: one wavefront owns one SNP row at a time
// ---------------------------------------------------------------------------

struct ScanArgs {
    const u32x4 *table;
    uint64_t row_stride_u;
    uint32_t plane_stride_u;
    uint32_t units;
    const u32x4 *case_mask;
    const u32x4 *ctrl_mask;
    uint64_t row_begin;
    uint32_t n_rows;
    uint32_t case_count, ctrl_count;
    uint32_t *counts;
    double *stats;
    uint32_t *raw;        // column-segmented scan: the six sums of this segment go here instead ([row][6])
    int raw_accumulate;   // ... added to what earlier segments left
    uint32_t defer_flushes;  // DEFER kernels: 64-row groups a wave can park in LDS (the launch is sized to fit)
    uint32_t defer_lds_off;  // ... and where the parking area starts in the dynamic LDS (u32 index, behind the masks)
};

// ---- row epilogue shared by both scan kernels (table5.cpp:647-653 + section 5 of DESIGN.md)
__device__ __forceinline__ void finish_row(const ScanArgs &a, uint32_t r, const uint32_t (&s)[6]) {
    if (a.raw) {  // one segment of a column-segmented scan: partial sums only
        uint32_t *o = a.raw + (size_t)r * 6u;
#pragma unroll
        for (int j = 0; j < 6; ++j) o[j] = a.raw_accumulate ? o[j] + s[j] : s[j];
        return;
    }
    uint32_t ca[4], co[4];
    ca[2] = s[2]; ca[0] = s[0] - s[2]; ca[1] = s[1] - s[2];
    ca[3] = a.case_count - ca[0] - ca[1] - ca[2];
    co[2] = s[5]; co[0] = s[3] - s[5]; co[1] = s[4] - s[5];
    co[3] = a.ctrl_count - co[0] - co[1] - co[2];
    u32x4 *out = reinterpret_cast<u32x4 *>(a.counts + (size_t)r * 8u);
    u32x4 v0 = {ca[0], ca[1], ca[2], ca[3]};
    u32x4 v1 = {co[0], co[1], co[2], co[3]};
    out[0] = v0;
    out[1] = v1;
    if (a.stats) {
        double st[8];
        assoc_stats(ca, co, st);
        double *so = a.stats + (size_t)r * 8u;
#pragma unroll
        for (int j = 0; j < 8; ++j) so[j] = st[j];
    }
}

// Persistent workgroups: the grid is (CUs x workgroups-per-CU); each wave
// walks rows gwave, gwave + total_waves, ...  Per row a lane streams 16-byte
// units lane, lane+64, ... of both bit-planes (two 1-KiB coalesced
// wave-loads per step, UNROLL steps in flight), ANDs them with the case and
// control masks (staged once per workgroup in LDS, or read through L2 when
// they do not fit) and keeps six 32-bit popcounts; a DPP wave reduction ends
// the row, and the FP64 epilogue runs for 64 finished rows at a time.
// DEFER: nothing is written while the table is being read.  A wave's finished rows (six sums each, parked one per
// lane as before) go to LDS 64 at a time -- [wave][flush][6][64] u32, each lane reads back only what it wrote, no
// barrier -- and when the wave has read its last row it runs the FP64 epilogue over everything it parked and emits
// its records in one burst.  Why: profiles/r02_short_rows_store_ablation.txt -- a record line written back from L2
// between the reads of the genotype stream costs about six read bytes per byte; the same stores are free while the
// lines stay in L2.  The launch is sized by the host so that a wave never has more than a.defer_flushes groups.
template <int UNROLL, bool MASK_LDS, bool NT, bool DEFER = false>
__global__ void __launch_bounds__(1024) k_scan_cc(ScanArgs a) {
    extern __shared__ u32x4 lds_masks[];  // [2][units] when MASK_LDS
    const uint32_t units = a.units;
    const u32x4 *mcase = a.case_mask;
    const u32x4 *mctrl = a.ctrl_mask;
    if (MASK_LDS) {
        for (uint32_t u = threadIdx.x; u < units; u += blockDim.x) {
            lds_masks[u] = a.case_mask[u];
            lds_masks[units + u] = a.ctrl_mask[u];
        }
        __syncthreads();
        mcase = lds_masks;
        mctrl = lds_masks + units;
    }

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t gwave = blockIdx.x * waves_per_block + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t total_waves = gridDim.x * waves_per_block;
    constexpr uint32_t STEP = 64u * UNROLL;
    const uint32_t full = (units / STEP) * STEP;

    // Row results are parked one per lane (lane k keeps the k-th row this wave
    // finished) and the FP64 epilogue runs once per 64 rows with every lane
    // busy, instead of once per row on lane 0 while the wave's loads drain.
    uint32_t mine[6] = {0, 0, 0, 0, 0, 0};
    uint32_t my_row = 0xFFFFFFFFu, slot = 0, flush = 0;

    for (uint32_t r = gwave; r < a.n_rows; r += total_waves) {
        const u32x4 *p0 = a.table + (a.row_begin + r) * a.row_stride_u;
        const u32x4 *p1 = p0 + a.plane_stride_u;
        uint32_t c0[6] = {0, 0, 0, 0, 0, 0}, c1[6] = {0, 0, 0, 0, 0, 0};

        for (uint32_t base = 0; base < full; base += STEP) {
            u32x4 A[UNROLL], B[UNROLL];
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                A[k] = ld_stream<NT>(p0 + base + 64u * k + lane);
                B[k] = ld_stream<NT>(p1 + base + 64u * k + lane);
            }
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                const u32x4 mc = mcase[base + 64u * k + lane];
                const u32x4 mt = mctrl[base + 64u * k + lane];
                if (k & 1) acc_unit(A[k], B[k], mc, mt, c1);
                else acc_unit(A[k], B[k], mc, mt, c0);
            }
        }
        for (uint32_t u = full + lane; u < units; u += 64u) {
            const u32x4 A = ld_stream<NT>(p0 + u), B = ld_stream<NT>(p1 + u);
            acc_unit(A, B, mcase[u], mctrl[u], c0);
        }

        const bool park = lane == slot;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const uint32_t total = wave_sum_dpp(c0[j] + c1[j]);  // wave-uniform
            mine[j] = park ? total : mine[j];
        }
        my_row = park ? r : my_row;
        if (++slot == 64u) {
            if (DEFER) {
                uint32_t *pk = reinterpret_cast<uint32_t *>(lds_masks) + a.defer_lds_off +
                               ((threadIdx.x >> 6) * a.defer_flushes + flush) * 384u + lane;
#pragma unroll
                for (int j = 0; j < 6; ++j) pk[64 * j] = mine[j];
                ++flush;
            } else {
                finish_row(a, my_row, mine);  // all 64 lanes hold a row
            }
            my_row = 0xFFFFFFFFu;
            slot = 0;
        }
    }
    if (DEFER) {
        // row of (flush f, lane l) = gwave + (64 f + l) * total_waves: the order the wave walked them in
        for (uint32_t f = 0; f < flush; ++f) {
            const uint32_t *pk = reinterpret_cast<const uint32_t *>(lds_masks) + a.defer_lds_off +
                                 ((threadIdx.x >> 6) * a.defer_flushes + f) * 384u + lane;
            uint32_t sums[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) sums[j] = pk[64 * j];
            finish_row(a, gwave + (64u * f + lane) * total_waves, sums);
        }
    }
    if (my_row != 0xFFFFFFFFu) finish_row(a, my_row, mine);
}

// Complementary-mask form of the scan, for the usual phenotype where every
// sample is either a case or a control (control mask == ~case mask over the
// N samples; data/simulate_data.cpp:219-235 produces exactly that): only the
// case mask is staged in LDS (N/8 bytes -- fits up to N ~ 1.3 M), the counts of
// ALL samples are taken without a mask and controls = all - cases.  Same six
// integers per row, 4 AND + 6 popcount per word instead of 6 + 6.
__device__ __forceinline__ void acc_unit_compl(const u32x4 a, const u32x4 b, const u32x4 mc, uint32_t (&c)[6]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t x = a[k] & mc[k], y = b[k] & mc[k];
        c[0] += __popc(x);
        c[1] += __popc(y);
        c[2] += __popc(x & y);
        c[3] += __popc(a[k]);
        c[4] += __popc(b[k]);
        c[5] += __popc(a[k] & b[k]);
    }
}

template <int UNROLL>
__global__ void __launch_bounds__(1024) k_scan_cc_compl(ScanArgs a) {
    extern __shared__ u32x4 lds_masks[];  // [units]: the case mask
    const uint32_t units = a.units;
    for (uint32_t u = threadIdx.x; u < units; u += blockDim.x) lds_masks[u] = a.case_mask[u];
    __syncthreads();
    const u32x4 *mcase = lds_masks;

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t gwave = blockIdx.x * waves_per_block + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t total_waves = gridDim.x * waves_per_block;
    constexpr uint32_t STEP = 64u * UNROLL;
    const uint32_t full = (units / STEP) * STEP;

    uint32_t mine[6] = {0, 0, 0, 0, 0, 0};
    uint32_t my_row = 0xFFFFFFFFu, slot = 0;
    for (uint32_t r = gwave; r < a.n_rows; r += total_waves) {
        const u32x4 *p0 = a.table + (a.row_begin + r) * a.row_stride_u;
        const u32x4 *p1 = p0 + a.plane_stride_u;
        uint32_t c0[6] = {0, 0, 0, 0, 0, 0}, c1[6] = {0, 0, 0, 0, 0, 0};
        for (uint32_t base = 0; base < full; base += STEP) {
            u32x4 A[UNROLL], B[UNROLL];
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                A[k] = ld_stream<true>(p0 + base + 64u * k + lane);
                B[k] = ld_stream<true>(p1 + base + 64u * k + lane);
            }
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                const u32x4 mc = mcase[base + 64u * k + lane];
                if (k & 1) acc_unit_compl(A[k], B[k], mc, c1);
                else acc_unit_compl(A[k], B[k], mc, c0);
            }
        }
        for (uint32_t u = full + lane; u < units; u += 64u) {
            const u32x4 A = ld_stream<true>(p0 + u), B = ld_stream<true>(p1 + u);
            acc_unit_compl(A, B, mcase[u], c0);
        }
        const bool park = lane == slot;
        uint32_t tot[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) tot[j] = wave_sum_dpp(c0[j] + c1[j]);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            mine[j] = park ? tot[j] : mine[j];                    // cases
            mine[3 + j] = park ? tot[3 + j] - tot[j] : mine[3 + j];  // controls = all - cases
        }
        my_row = park ? r : my_row;
        if (++slot == 64u) {
            finish_row(a, my_row, mine);
            my_row = 0xFFFFFFFFu;
            slot = 0;
        }
    }
    if (my_row != 0xFFFFFFFFu) finish_row(a, my_row, mine);
}

// ---------------------------------------------------------------------------
// Continuous-stream form of the scan (GW_OPT_PIPELINE): the loads of the NEXT
// (row, step) item are in flight while the current one is consumed, also across
// the row boundary.  hipcc cannot be made to do this with ordinary loads (it
// re-uses in-flight destination registers as temporaries and falls back to
// vmcnt(0)), so the genotype stream is loaded with inline asm and waited for
// with hand-counted s_waitcnt vmcnt(N):
//   * a load's destination is unprotected until the wait that covers it; the
//     wait statement takes the buffer as a read-write operand, so every use of
//     the data depends on it;
//   * loads are never predicated (a lane past the row's end re-reads the last
//     unit; its mask words are zero) and the item after the wave's last one
//     re-reads the current row, so the number of loads in flight is exact;
//   * vmcnt counts in issue order, stores included: a wait written for "the
//     other buffer's 2*UNROLL loads may remain" can only over-wait when the
//     64-row epilogue's stores sit in between, never under-wait.
// ---------------------------------------------------------------------------
template <bool NT>
__device__ __forceinline__ void asm_load(u32x4 &dst, const u32x4 *p) {
    if (NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p));
    else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p));
}

template <int UNROLL, bool NT>
__device__ __forceinline__ void asm_load_step(const u32x4 *p0, const u32x4 *p1, uint32_t first, uint32_t lane,
                                              uint32_t last_unit, u32x4 (&A)[UNROLL], u32x4 (&B)[UNROLL]) {
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
        const uint32_t u = min(first + 64u * k + lane, last_unit);
        asm_load<NT>(A[k], p0 + u);
        asm_load<NT>(B[k], p1 + u);
    }
}

// wait until at most 2*UNROLL vector-memory operations are outstanding, i.e.
// until buffer (A, B) -- issued before the other buffer's 2*UNROLL loads -- has landed
template <int UNROLL>
__device__ __forceinline__ void asm_wait_older(u32x4 (&A)[UNROLL], u32x4 (&B)[UNROLL]) {
    static_assert(UNROLL == 1 || UNROLL == 2 || UNROLL == 4, "pipelined scan: 1, 2 or 4 units per lane");
    if (UNROLL == 1) asm volatile("s_waitcnt vmcnt(2)" : "+v"(A[0]), "+v"(B[0]));
    if (UNROLL == 2) asm volatile("s_waitcnt vmcnt(4)" : "+v"(A[0]), "+v"(B[0]), "+v"(A[1]), "+v"(B[1]));
    if (UNROLL == 4)
        asm volatile("s_waitcnt vmcnt(8)"
                     : "+v"(A[0]), "+v"(B[0]), "+v"(A[1]), "+v"(B[1]), "+v"(A[2]), "+v"(B[2]), "+v"(A[3]), "+v"(B[3]));
}

template <int UNROLL, bool MASK_LDS, bool NT>
__global__ void __launch_bounds__(512) k_scan_cc_pipe(ScanArgs a) {  // <= 512 threads: up to 256 VGPRs for the two buffers
    extern __shared__ u32x4 lds_masks[];  // [2][units + 1] when MASK_LDS: entry `units` of each mask is zero
    const uint32_t units = a.units;
    constexpr uint32_t STEP = 64u * UNROLL;
    const uint32_t steps = (units + STEP - 1u) / STEP;
    // A lane past the row's end re-reads the row's last unit against a ZERO mask.  In LDS that is one shared zero
    // entry behind each mask (index min(u, units)), not a whole zero step: at N = 500 k the two masks then take
    // 125 056 B of the 160 KiB, which leaves a CU room for a second, small-LDS workgroup next to the scan -- RCCL's
    // ncclDevKernel_Generic needs 37 664 B.  The global masks are zero-padded to 512 units anyway.
    const uint32_t mlen = units + 1u;
    const u32x4 *mcase = a.case_mask;
    const u32x4 *mctrl = a.ctrl_mask;
    if (MASK_LDS) {
        for (uint32_t u = threadIdx.x; u < mlen; u += blockDim.x) {  // zero at `units` (a column segment's masks
            const u32x4 zero = {0, 0, 0, 0};                           // continue with the next segment's)
            lds_masks[u] = u < units ? a.case_mask[u] : zero;
            lds_masks[mlen + u] = u < units ? a.ctrl_mask[u] : zero;
        }
        __syncthreads();
        mcase = lds_masks;
        mctrl = lds_masks + mlen;
    }
    const uint32_t lane = threadIdx.x & 63u;
    // the row walk lives in SGPRs: row indices, base addresses and branches are scalar
    const uint32_t waves_per_block = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t total_waves = __builtin_amdgcn_readfirstlane(gridDim.x * waves_per_block);
    const uint32_t last_unit = units - 1u;

    uint32_t r = __builtin_amdgcn_readfirstlane(blockIdx.x * waves_per_block + wave);
    if (r >= a.n_rows) return;  // no barrier below
    uint32_t s = 0;
    u32x4 A0[UNROLL], B0[UNROLL], A1[UNROLL], B1[UNROLL];
    uint32_t c0[6] = {0, 0, 0, 0, 0, 0}, c1[6] = {0, 0, 0, 0, 0, 0};
    uint32_t mine[6] = {0, 0, 0, 0, 0, 0};  // finished rows parked one per lane, epilogue per 64 rows
    uint32_t my_row = 0xFFFFFFFFu, slot = 0;

    auto consume = [&](const u32x4 (&A)[UNROLL], const u32x4 (&B)[UNROLL], uint32_t row, uint32_t step) {
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const uint32_t u0 = step * STEP + 64u * k + lane;
            const uint32_t u = MASK_LDS ? min(u0, units) : u0;  // past the end: the zero entry (global masks are zero-padded)
            const u32x4 mc = mcase[u];
            const u32x4 mt = mctrl[u];
            if (k & 1) acc_unit(A[k], B[k], mc, mt, c1);
            else acc_unit(A[k], B[k], mc, mt, c0);
        }
        if (step + 1u == steps) {  // wave-uniform: the row is complete
            const bool park = lane == slot;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const uint32_t total = wave_sum_dpp(c0[j] + c1[j]);
                mine[j] = park ? total : mine[j];
                c0[j] = 0;
                c1[j] = 0;
            }
            my_row = park ? row : my_row;
            if (++slot == 64u) {
                finish_row(a, my_row, mine);
                my_row = 0xFFFFFFFFu;
                slot = 0;
            }
        }
    };

    {
        const u32x4 *p0 = a.table + (a.row_begin + r) * a.row_stride_u;
        asm_load_step<UNROLL, NT>(p0, p0 + a.plane_stride_u, 0, lane, last_unit, A0, B0);
    }
    for (;;) {
        // item after (r, s); past the wave's last row the loads re-read the current row
        uint32_t r1 = r, s1 = s + 1u;
        if (s1 == steps) { r1 = r + total_waves; s1 = 0; }
        r1 = __builtin_amdgcn_readfirstlane(r1);
        s1 = __builtin_amdgcn_readfirstlane(s1);
        const bool have1 = r1 < a.n_rows;
        {
            const u32x4 *q0 = a.table + (a.row_begin + (have1 ? r1 : r)) * a.row_stride_u;
            asm_load_step<UNROLL, NT>(q0, q0 + a.plane_stride_u, s1 * STEP, lane, last_unit, A1, B1);
        }
        asm_wait_older<UNROLL>(A0, B0);
        consume(A0, B0, r, s);
        if (!have1) break;
        // item after (r1, s1)
        uint32_t r2 = r1, s2 = s1 + 1u;
        if (s2 == steps) { r2 = r1 + total_waves; s2 = 0; }
        r2 = __builtin_amdgcn_readfirstlane(r2);
        s2 = __builtin_amdgcn_readfirstlane(s2);
        const bool have2 = r2 < a.n_rows;
        {
            const u32x4 *t0 = a.table + (a.row_begin + (have2 ? r2 : r1)) * a.row_stride_u;
            asm_load_step<UNROLL, NT>(t0, t0 + a.plane_stride_u, s2 * STEP, lane, last_unit, A0, B0);
        }
        asm_wait_older<UNROLL>(A1, B1);
        consume(A1, B1, r1, s1);
        if (!have2) break;
        r = r2;
        s = s2;
    }
    // the loads issued for the item that does not exist must land before the registers die
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (my_row != 0xFFFFFFFFu) finish_row(a, my_row, mine);
}

// ---------------------------------------------------------------------------
// Short rows (rows of at most 256 units, N <= 32 767).  With one wave per row the
// scan is not memory- but instruction-bound there (PMC at N = 10 k: 11 % of the
// wave cycles wait for memory, 160 VALU per row): a 79-unit row fills 62 % of its
// two wave-steps, and every row pays six cross-lane reductions with lane reads into
// SGPRs.  Here a row belongs to a SUB-WAVE of G = 16 or 32 lanes:
//   * the wave works on 64/G consecutive rows at once (ceil(units/G) steps, 99 % of
//     the lane slots carry data at N = 10 k), all loads of a batch issued before the
//     first popcount and the next batch's already in flight (two register buffers);
//   * a 16-lane group is a DPP row: four v_add_dpp leave the row's sum in every lane
//     of the group (G = 32: one ds_swizzle adds the other half) -- no lane reads,
//     no scalar adds;
//   * batch p of a wave parks its results in lane p of each group, so the 64
//     parked rows of the FP64 epilogue fill up without moving anything across lanes;
//   * masks live in registers (no LDS, no barrier); lanes past the row end re-read
//     its last unit against a zero mask, so no load is conditional;
//   * with N < 65 536 two counters share a register through the reduction, and the
//     popcounts accumulate in v_bcnt's own add operand.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void bcnt_acc(uint32_t &c, uint32_t x) {
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c) : "v"(x));
}
__device__ __forceinline__ void acc_unit_chain(const u32x4 a, const u32x4 b, const u32x4 mc, const u32x4 mt,
                                               uint32_t (&c)[6]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t x = a[k] & mc[k], y = b[k] & mc[k];
        bcnt_acc(c[0], x);
        bcnt_acc(c[1], y);
        bcnt_acc(c[2], x & y);
        x = a[k] & mt[k];
        y = b[k] & mt[k];
        bcnt_acc(c[3], x);
        bcnt_acc(c[4], y);
        bcnt_acc(c[5], x & y);
    }
}
// sum over each 16-lane DPP row, result in all of its lanes
__device__ __forceinline__ uint32_t row16_sum_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror
    return v;
}

template <int G, int STEPS, bool PACK, bool DEFER = false>
__global__ void __launch_bounds__(256) k_scan_cc_sub(ScanArgs a) {
    static_assert(G == 16 || G == 32, "sub-wave of 16 or 32 lanes per row");
    extern __shared__ uint32_t lds_park[];  // DEFER (see k_scan_cc): [wave][flush][NP][64], NP = 3 packed or 6 sums
    constexpr uint32_t NP = PACK ? 3u : 6u;
    uint32_t flush = 0;
    constexpr uint32_t RW = 64 / G;  // rows per wave and batch
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t grp = lane / G, li = lane % G;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t gwave = blockIdx.x * waves_per_block + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t total_waves = gridDim.x * waves_per_block;
    const uint32_t last_unit = a.units - 1u;

    u32x4 mc[STEPS], mt[STEPS];
    uint32_t uidx[STEPS];
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
        const uint32_t u = (uint32_t)G * k + li;
        uidx[k] = min(u, last_unit);
        const u32x4 zero = {0, 0, 0, 0};
        mc[k] = u <= last_unit ? a.case_mask[u] : zero;
        mt[k] = u <= last_unit ? a.ctrl_mask[u] : zero;
    }

    // parked results: batch p of this wave -> lane p of every group (all 64 lanes fill up in G batches)
    uint32_t mine[6] = {0, 0, 0, 0, 0, 0};
    uint32_t my_row = 0xFFFFFFFFu, slot = 0;
    const uint32_t n_batches = (a.n_rows + RW - 1u) / RW, last_row = a.n_rows - 1u;

    auto issue = [&](u32x4 (&A)[STEPS], u32x4 (&B)[STEPS], uint32_t b) {
        const uint32_t row = min(b * RW + grp, last_row);  // past the end: the last row again, result dropped
        const u32x4 *p0 = a.table + (a.row_begin + row) * a.row_stride_u;
        const u32x4 *p1 = p0 + a.plane_stride_u;
#pragma unroll
        for (int k = 0; k < STEPS; ++k) {
            A[k] = ld_stream<true>(p0 + uidx[k]);
            B[k] = ld_stream<true>(p1 + uidx[k]);
        }
    };
    auto consume = [&](const u32x4 (&A)[STEPS], const u32x4 (&B)[STEPS], uint32_t b) {
        uint32_t c[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < STEPS; ++k) acc_unit_chain(A[k], B[k], mc[k], mt[k], c);
        constexpr int NRED = PACK ? 3 : 6;
        uint32_t tot[NRED];
#pragma unroll
        for (int j = 0; j < NRED; ++j) {
            uint32_t v = row16_sum_dpp(PACK ? (c[j] | (c[3 + j] << 16)) : c[j]);  // packed halves < 65 536
            if (G == 32)  // + the other 16-lane half of the group (lane ^ 16, through the LDS crossbar): all 32 lanes
                v += (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);  // bit mode: and 0x1f, or 0, xor 0x10
            tot[j] = v;
        }
        const uint32_t row = b * RW + grp;
        const bool park = li == slot && row < a.n_rows;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            mine[j] = park ? (PACK ? (tot[j] & 0xFFFFu) : tot[j]) : mine[j];
            mine[3 + j] = park ? (PACK ? (tot[j] >> 16) : tot[(3 + j) % NRED]) : mine[3 + j];
        }
        my_row = park ? row : my_row;
        if (++slot == (uint32_t)G) {
            if (DEFER) {
                uint32_t *pk = lds_park + ((threadIdx.x >> 6) * a.defer_flushes + flush) * (NP * 64u) + lane;
#pragma unroll
                for (uint32_t j = 0; j < NP; ++j) pk[64u * j] = PACK ? (mine[j] | (mine[3 + j] << 16)) : mine[j];
                ++flush;
            } else if (my_row != 0xFFFFFFFFu) {
                finish_row(a, my_row, mine);
            }
            my_row = 0xFFFFFFFFu;
            slot = 0;
        }
    };

    uint32_t b = gwave;
    if (b < n_batches) {
        u32x4 A0[STEPS], B0[STEPS], A1[STEPS], B1[STEPS];
        issue(A0, B0, b);
        for (;;) {
            issue(A1, B1, b + total_waves);
            consume(A0, B0, b);
            b += total_waves;
            if (b >= n_batches) break;
            issue(A0, B0, b + total_waves);
            consume(A1, B1, b);
            b += total_waves;
            if (b >= n_batches) break;
        }
    }
    if (DEFER) {
        // (flush f, lane (grp, li)) holds batch gwave + (G f + li) * total_waves, row = batch * RW + grp
        for (uint32_t f = 0; f < flush; ++f) {
            const uint32_t *pk = lds_park + ((threadIdx.x >> 6) * a.defer_flushes + f) * (NP * 64u) + lane;
            uint32_t sums[6];
#pragma unroll
            for (uint32_t j = 0; j < 3u; ++j) {
                const uint32_t v = pk[64u * j];
                sums[j] = PACK ? (v & 0xFFFFu) : v;
                sums[3 + j] = PACK ? (v >> 16) : pk[64u * (3u + j)];
            }
            const uint32_t row = (gwave + ((uint32_t)G * f + li) * total_waves) * RW + grp;
            if (row < a.n_rows) finish_row(a, row, sums);
        }
    }
    if (my_row != 0xFFFFFFFFu) finish_row(a, my_row, mine);
}

// the instantiation a scan launch went to, as rocprofv3 prints it (launch_scan_cc's kernel_name)
static thread_local char *t_name_out = nullptr;
static void note_kernel(const char *fmt, ...) {
    if (!t_name_out) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_name_out, 96, fmt, ap);
    va_end(ap);
}
static const char *tf(bool b) { return b ? "true" : "false"; }

// workgroups of k_scan_cc_sub a CU holds (registers; <= 8), per instantiation and device
template <int G, int STEPS, bool PACK, bool DEFER>
static int sub_resident() {
    static int resident[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!resident[dev]) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_scan_cc_sub<G, STEPS, PACK, DEFER>, 256, 0) != hipSuccess || n < 1) n = 2;
        resident[dev] = n > 8 ? 8 : n;
    }
    return resident[dev];
}

template <int G, int STEPS, bool PACK>
static hipError_t launch_scan_sub_p(const ScanArgs &args, const ScanShape &sh, hipStream_t stream) {
    const int cus = sh.compute_units > sh.reserve_cus ? sh.compute_units - sh.reserve_cus : 1;
    if (!sh.defer) {
        int grid = sh.grid_blocks;  // 0 = auto: as many persistent workgroups as the registers let a CU hold
        if (grid <= 0) grid = cus * sub_resident<G, STEPS, PACK, false>();
        note_kernel("gw::k_scan_cc_sub<%d, %d, %s, false>", G, STEPS, tf(PACK));
        hipLaunchKernelGGL((k_scan_cc_sub<G, STEPS, PACK, false>), dim3(grid), dim3(256), 0, stream, args);
        return hipGetLastError();
    }
    // records deferred to the end of the launch: the wave's rows are parked in LDS in groups of 64; a launch takes
    // as many rows as the parking area of the resident workgroups holds, longer scans go in pieces
    constexpr uint32_t RW = 64 / G, GROUP_BYTES = (PACK ? 3u : 6u) * 256u;
    const int per_cu = sh.grid_blocks > 0 ? std::max(1, sh.grid_blocks / cus) : sub_resident<G, STEPS, PACK, true>();
    const int grid = cus * per_cu;
    const uint32_t total_waves = (uint32_t)grid * 4u;
    // (the parking area is plain dynamic LDS: at most 64 KB per workgroup -- hipFuncAttributeMaxDynamicSharedMemorySize is
    // not raised for these instantiations -- whatever GW_OPT_BLOCKS_PER_CU leaves per workgroup)
    const size_t per_wg = std::min<size_t>(sh.lds_cap / (size_t)per_cu, (size_t)64u << 10);
    const uint32_t f_max = std::max<uint32_t>(1u, (uint32_t)per_wg / (4u * GROUP_BYTES));
    const uint64_t piece_rows = (uint64_t)total_waves * 64u * f_max;
    note_kernel("gw::k_scan_cc_sub<%d, %d, %s, true>", G, STEPS, tf(PACK));
    for (uint64_t r0 = 0; r0 < args.n_rows; r0 += piece_rows) {
        ScanArgs a = args;
        a.n_rows = (uint32_t)std::min<uint64_t>(piece_rows, args.n_rows - r0);
        a.row_begin = args.row_begin + r0;
        a.counts = args.counts + r0 * 8u;
        a.stats = args.stats ? args.stats + r0 * 8u : nullptr;
        const uint32_t n_batches = (a.n_rows + RW - 1u) / RW;
        const uint32_t per_wave = (n_batches + total_waves - 1u) / total_waves;
        a.defer_flushes = std::max(1u, (per_wave + (uint32_t)G - 1u) / (uint32_t)G);   // <= f_max by construction
        hipLaunchKernelGGL((k_scan_cc_sub<G, STEPS, PACK, true>), dim3(grid), dim3(256),
                           (size_t)4u * a.defer_flushes * GROUP_BYTES, stream, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
template <int G, int STEPS>
static hipError_t launch_scan_sub(const ScanArgs &args, const ScanShape &sh, bool pack, hipStream_t stream) {
    return pack ? launch_scan_sub_p<G, STEPS, true>(args, sh, stream) : launch_scan_sub_p<G, STEPS, false>(args, sh, stream);
}

// k_scan_cc with the records deferred (see the kernel): pieces sized to the LDS left beside the masks
template <int UNROLL>
static hipError_t launch_scan_defer(const ScanArgs &args, const ScanShape &sh, hipStream_t stream) {
    const uint32_t waves = (uint32_t)sh.block_threads / 64u, total_waves = (uint32_t)sh.grid_blocks * waves;
    const int per_cu = std::max(1, sh.grid_blocks / std::max(1, sh.compute_units - sh.reserve_cus));
    const size_t room = sh.lds_cap / (size_t)per_cu > sh.lds_bytes ? sh.lds_cap / (size_t)per_cu - sh.lds_bytes : 0;
    const uint32_t f_max = (uint32_t)(room / ((size_t)waves * 1536u));
    if (f_max == 0) return hipErrorInvalidValue;   // scan_shape only asks for this where one group per wave fits
    const uint64_t piece_rows = (uint64_t)total_waves * 64u * f_max;
    note_kernel("gw::k_scan_cc<%d, true, true, true>", UNROLL);
    for (uint64_t r0 = 0; r0 < args.n_rows; r0 += piece_rows) {
        ScanArgs a = args;
        a.n_rows = (uint32_t)std::min<uint64_t>(piece_rows, args.n_rows - r0);
        a.row_begin = args.row_begin + r0;
        a.counts = args.counts + r0 * 8u;
        a.stats = args.stats ? args.stats + r0 * 8u : nullptr;
        const uint32_t per_wave = (a.n_rows + total_waves - 1u) / total_waves;
        a.defer_flushes = std::max(1u, (per_wave + 63u) / 64u);
        a.defer_lds_off = (uint32_t)(sh.lds_bytes / 4u);
        hipLaunchKernelGGL((k_scan_cc<UNROLL, true, true, true>), dim3(sh.grid_blocks), dim3(sh.block_threads),
                           sh.lds_bytes + (size_t)waves * a.defer_flushes * 1536u, stream, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <int UNROLL, bool MASK_LDS, bool NT>
static hipError_t launch_scan_inst(const ScanArgs &args, const ScanShape &sh, hipStream_t stream) {
    if (sh.pipeline && UNROLL <= 4) {
        note_kernel("gw::k_scan_cc_pipe<%d, %s, %s>", UNROLL <= 4 ? UNROLL : 4, tf(MASK_LDS), tf(NT));
        hipLaunchKernelGGL((k_scan_cc_pipe<(UNROLL <= 4 ? UNROLL : 4), MASK_LDS, NT>), dim3(sh.grid_blocks),
                           dim3(sh.block_threads), MASK_LDS ? sh.lds_bytes : 0, stream, args);
    } else if (sh.defer && MASK_LDS && NT && (UNROLL == 1 || UNROLL == 2)) {
        return launch_scan_defer<(UNROLL == 2 ? 2 : 1)>(args, sh, stream);
    } else {
        note_kernel("gw::k_scan_cc<%d, %s, %s, false>", UNROLL, tf(MASK_LDS), tf(NT));
        hipLaunchKernelGGL((k_scan_cc<UNROLL, MASK_LDS, NT>), dim3(sh.grid_blocks),
                           dim3(sh.block_threads), MASK_LDS ? sh.lds_bytes : 0, stream, args);
    }
    return hipGetLastError();
}

template <int UNROLL>
static hipError_t launch_scan_u(const ScanArgs &args, const ScanShape &sh, hipStream_t stream) {
    if (sh.mask_in_lds) {
        return sh.nontemporal ? launch_scan_inst<UNROLL, true, true>(args, sh, stream)
                              : launch_scan_inst<UNROLL, true, false>(args, sh, stream);
    }
    return sh.nontemporal ? launch_scan_inst<UNROLL, false, true>(args, sh, stream)
                          : launch_scan_inst<UNROLL, false, false>(args, sh, stream);
}

// the last step of a column-segmented scan: counts and statistics from the accumulated sums
__global__ void k_finish_raw(ScanArgs a, const uint32_t *raw) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_rows) return;
    uint32_t s[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) s[j] = raw[(size_t)r * 6u + j];
    finish_row(a, r, s);
}

hipError_t launch_finish_raw(const uint32_t *d_raw, uint32_t n_rows, uint32_t case_count, uint32_t ctrl_count,
                             uint32_t *d_counts, double *d_stats, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    ScanArgs a{};
    a.n_rows = n_rows;
    a.case_count = case_count;
    a.ctrl_count = ctrl_count;
    a.counts = d_counts;
    a.stats = d_stats;
    a.raw = nullptr;
    hipLaunchKernelGGL(k_finish_raw, dim3((n_rows + 255u) / 256u), dim3(256), 0, stream, a, d_raw);
    return hipGetLastError();
}

hipError_t launch_scan_cc(const TableView &tv, const u32x4 *d_case, const u32x4 *d_ctrl,
                          uint32_t case_count, uint32_t ctrl_count, uint64_t row_begin,
                          uint32_t n_rows, uint32_t *d_counts, double *d_stats,
                          const ScanShape &sh, hipStream_t stream, uint32_t *d_raw, bool raw_accumulate,
                          char *kernel_name) {
    if (n_rows == 0) return hipSuccess;
    struct NameScope {  // note_kernel() writes here for the duration of this call
        explicit NameScope(char *p) { t_name_out = p; }
        ~NameScope() { t_name_out = nullptr; }
    } name_scope(kernel_name);
    ScanArgs a;
    a.table = tv.base;
    a.row_stride_u = tv.row_stride_u;
    a.plane_stride_u = tv.plane_stride_u;
    a.units = tv.units;
    a.case_mask = d_case;
    a.ctrl_mask = d_ctrl;
    a.row_begin = row_begin;
    a.n_rows = n_rows;
    a.case_count = case_count;
    a.ctrl_count = ctrl_count;
    a.counts = d_counts;
    a.stats = d_stats;
    a.raw = d_raw;
    a.raw_accumulate = raw_accumulate ? 1 : 0;
    if (sh.short_rows) {  // rows of at most 256 units: a sub-wave of 16 (<= 80 units) or 32 lanes per row
        const bool pack = tv.n_samples < 65536u;
        if (tv.units <= 80u) {  // at most 5 steps of 16 lanes
            switch ((tv.units + 15u) / 16u) {
            case 1: return launch_scan_sub<16, 1>(a, sh, pack, stream);
            case 2: return launch_scan_sub<16, 2>(a, sh, pack, stream);
            case 3: return launch_scan_sub<16, 3>(a, sh, pack, stream);
            case 4: return launch_scan_sub<16, 4>(a, sh, pack, stream);
            default: return launch_scan_sub<16, 5>(a, sh, pack, stream);
            }
        }
        switch ((tv.units + 31u) / 32u) {
        case 3: return launch_scan_sub<32, 3>(a, sh, pack, stream);
        case 4: return launch_scan_sub<32, 4>(a, sh, pack, stream);
        case 5: return launch_scan_sub<32, 5>(a, sh, pack, stream);
        case 6: return launch_scan_sub<32, 6>(a, sh, pack, stream);
        case 7: return launch_scan_sub<32, 7>(a, sh, pack, stream);
        default: return launch_scan_sub<32, 8>(a, sh, pack, stream);
        }
    }
    if (sh.complement) {
        note_kernel("gw::k_scan_cc_compl<%d>", (sh.unroll == 1 || sh.unroll == 4 || sh.unroll == 8) ? sh.unroll : 2);
        switch (sh.unroll) {
        case 1: hipLaunchKernelGGL((k_scan_cc_compl<1>), dim3(sh.grid_blocks), dim3(sh.block_threads), sh.lds_bytes, stream, a); break;
        case 4: hipLaunchKernelGGL((k_scan_cc_compl<4>), dim3(sh.grid_blocks), dim3(sh.block_threads), sh.lds_bytes, stream, a); break;
        case 8: hipLaunchKernelGGL((k_scan_cc_compl<8>), dim3(sh.grid_blocks), dim3(sh.block_threads), sh.lds_bytes, stream, a); break;
        default: hipLaunchKernelGGL((k_scan_cc_compl<2>), dim3(sh.grid_blocks), dim3(sh.block_threads), sh.lds_bytes, stream, a); break;
        }
        return hipGetLastError();
    }
    switch (sh.unroll) {
    case 1: return launch_scan_u<1>(a, sh, stream);
    case 2: return launch_scan_u<2>(a, sh, stream);
    case 8: return launch_scan_u<8>(a, sh, stream);
    default: return launch_scan_u<4>(a, sh, stream);
    }
}

template <int UNROLL>
static hipError_t prepare_u(size_t bytes) {
    hipError_t e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scan_cc<UNROLL, true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scan_cc<UNROLL, true, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    constexpr int PU = UNROLL <= 4 ? UNROLL : 4;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scan_cc_pipe<PU, true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scan_cc_pipe<PU, true, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

hipError_t prepare_scan_kernels(size_t max_lds_bytes) {
    hipError_t e;
    const int b = (int)max_lds_bytes;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scan_cc_compl<1>), hipFuncAttributeMaxDynamicSharedMemorySize, b)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scan_cc_compl<2>), hipFuncAttributeMaxDynamicSharedMemorySize, b)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scan_cc_compl<4>), hipFuncAttributeMaxDynamicSharedMemorySize, b)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scan_cc_compl<8>), hipFuncAttributeMaxDynamicSharedMemorySize, b)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scan_cc<1, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, b)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scan_cc<2, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, b)) != hipSuccess) return e;
    if ((e = prepare_u<1>(max_lds_bytes)) != hipSuccess) return e;
    if ((e = prepare_u<2>(max_lds_bytes)) != hipSuccess) return e;
    if ((e = prepare_u<4>(max_lds_bytes)) != hipSuccess) return e;
    return prepare_u<8>(max_lds_bytes);
}

// ---------------------------------------------------------------------------
// plain genotype distribution (no masks), table5.cpp:577-607
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_scan_dist(const u32x4 *table, uint64_t row_stride_u,
                                                    uint32_t plane_stride_u, uint32_t units,
                                                    uint32_t n_samples, uint64_t row_begin,
                                                    uint32_t n_rows, uint32_t *dist) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t gwave = blockIdx.x * waves_per_block + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t total_waves = gridDim.x * waves_per_block;
    constexpr int U = 2;  // 512-thread workgroups, 2 units per lane and plane in flight (profiles/r01_read_patterns_*)
    constexpr uint32_t STEP = 64u * U;
    const uint32_t full = (units / STEP) * STEP;
    for (uint32_t r = gwave; r < n_rows; r += total_waves) {
        const u32x4 *p0 = table + (row_begin + r) * row_stride_u;
        const u32x4 *p1 = p0 + plane_stride_u;
        uint32_t c[3] = {0, 0, 0};
        for (uint32_t base = 0; base < full; base += STEP) {
            u32x4 A[U], B[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                A[k] = ld_stream<true>(p0 + base + 64u * k + lane);
                B[k] = ld_stream<true>(p1 + base + 64u * k + lane);
            }
#pragma unroll
            for (int k = 0; k < U; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c[0] += __popc(A[k][j]);
                    c[1] += __popc(B[k][j]);
                    c[2] += __popc(A[k][j] & B[k][j]);
                }
        }
        for (uint32_t u = full + lane; u < units; u += 64u) {
            const u32x4 A = ld_stream<true>(p0 + u), B = ld_stream<true>(p1 + u);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c[0] += __popc(A[j]);
                c[1] += __popc(B[j]);
                c[2] += __popc(A[j] & B[j]);
            }
        }
        const uint32_t s0 = wave_sum_dpp(c[0]), s1 = wave_sum_dpp(c[1]), s2 = wave_sum_dpp(c[2]);
        if (lane == 0) {
            const uint32_t aa = s0 - s2, ab = s1 - s2, bb = s2;
            u32x4 v = {aa, ab, bb, n_samples - aa - ab - bb};
            *reinterpret_cast<u32x4 *>(dist + (size_t)r * 4u) = v;
        }
    }
}

// Short rows: the sub-wave traversal of k_scan_cc_sub without masks (three counters; packed two and one
// through the reduction when N < 65 536).  Lanes past the row end re-read its last unit and are
// switched off by a per-lane factor instead of a mask.
template <int G, int STEPS>
__global__ void __launch_bounds__(256) k_scan_dist_sub(const u32x4 *table, uint64_t row_stride_u,
                                                       uint32_t plane_stride_u, uint32_t units, uint32_t n_samples,
                                                       uint64_t row_begin, uint32_t n_rows, uint32_t *dist) {
    constexpr uint32_t RW = 64 / G;
    const uint32_t lane = threadIdx.x & 63u, grp = lane / G, li = lane % G;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t gwave = blockIdx.x * waves_per_block + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t total_waves = gridDim.x * waves_per_block;
    const uint32_t last_unit = units - 1u, last_row = n_rows - 1u, n_batches = (n_rows + RW - 1u) / RW;
    uint32_t uidx[STEPS], live[STEPS];  // live: all ones where the lane's unit exists
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
        const uint32_t u = (uint32_t)G * k + li;
        uidx[k] = min(u, last_unit);
        live[k] = u <= last_unit ? 0xFFFFFFFFu : 0u;
    }
    uint32_t mine[3] = {0, 0, 0}, my_row = 0xFFFFFFFFu, slot = 0;
    auto flush = [&]() {
        if (my_row != 0xFFFFFFFFu) {
            const uint32_t aa = mine[0] - mine[2], ab = mine[1] - mine[2], bb = mine[2];
            u32x4 v = {aa, ab, bb, n_samples - aa - ab - bb};
            *reinterpret_cast<u32x4 *>(dist + (size_t)my_row * 4u) = v;
        }
        my_row = 0xFFFFFFFFu;
        slot = 0;
    };
    auto issue = [&](u32x4 (&A)[STEPS], u32x4 (&B)[STEPS], uint32_t b) {
        const uint32_t row = min(b * RW + grp, last_row);
        const u32x4 *p0 = table + (row_begin + row) * row_stride_u;
        const u32x4 *p1 = p0 + plane_stride_u;
#pragma unroll
        for (int k = 0; k < STEPS; ++k) {
            A[k] = ld_stream<true>(p0 + uidx[k]);
            B[k] = ld_stream<true>(p1 + uidx[k]);
        }
    };
    auto consume = [&](const u32x4 (&A)[STEPS], const u32x4 (&B)[STEPS], uint32_t b) {
        uint32_t c[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < STEPS; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t x = A[k][j] & live[k], y = B[k][j] & live[k];
                bcnt_acc(c[0], x);
                bcnt_acc(c[1], y);
                bcnt_acc(c[2], x & y);
            }
        auto group_sum = [](uint32_t v) {
            v = row16_sum_dpp(v);
            if (G == 32) v += (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);  // + the other half (lane ^ 16)
            return v;
        };
        uint32_t t0, t1;
        if (n_samples < 65536u) {  // wave-uniform: two counters share a register through the reduction
            const uint32_t t01 = group_sum(c[0] | (c[1] << 16));
            t0 = t01 & 0xFFFFu;
            t1 = t01 >> 16;
        } else {
            t0 = group_sum(c[0]);
            t1 = group_sum(c[1]);
        }
        const uint32_t t2 = group_sum(c[2]);
        const uint32_t row = b * RW + grp;
        const bool park = li == slot && row < n_rows;
        mine[0] = park ? t0 : mine[0];
        mine[1] = park ? t1 : mine[1];
        mine[2] = park ? t2 : mine[2];
        my_row = park ? row : my_row;
        if (++slot == (uint32_t)G) flush();
    };
    uint32_t b = gwave;
    if (b < n_batches) {
        u32x4 A0[STEPS], B0[STEPS], A1[STEPS], B1[STEPS];
        issue(A0, B0, b);
        for (;;) {
            issue(A1, B1, b + total_waves);
            consume(A0, B0, b);
            b += total_waves;
            if (b >= n_batches) break;
            issue(A0, B0, b + total_waves);
            consume(A1, B1, b);
            b += total_waves;
            if (b >= n_batches) break;
        }
    }
    flush();
}

template <int G, int STEPS>
static hipError_t launch_scan_dist_sub(const TableView &tv, uint64_t row_begin, uint32_t n_rows, uint32_t *d_dist,
                                       int compute_units, hipStream_t stream) {
    static int per_device[16] = {0};  // workgroups a CU holds (registers; <= 8), per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!per_device[dev]) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_scan_dist_sub<G, STEPS>, 256, 0) != hipSuccess || n < 1) n = 2;
        per_device[dev] = n > 8 ? 8 : n;
    }
    const int resident = per_device[dev];
    hipLaunchKernelGGL((k_scan_dist_sub<G, STEPS>), dim3(compute_units * resident), dim3(256), 0, stream, tv.base,
                       tv.row_stride_u, tv.plane_stride_u, tv.units, tv.n_samples, row_begin, n_rows, d_dist);
    return hipGetLastError();
}

hipError_t launch_scan_dist(const TableView &tv, uint64_t row_begin, uint32_t n_rows,
                            uint32_t *d_dist, int grid_blocks, hipStream_t stream, int short_rows_cus) {
    if (n_rows == 0) return hipSuccess;
    if (short_rows_cus > 0 && tv.units <= 256u) {
        const int cu = short_rows_cus;
        if (tv.units <= 80u) {
            switch ((tv.units + 15u) / 16u) {
            case 1: return launch_scan_dist_sub<16, 1>(tv, row_begin, n_rows, d_dist, cu, stream);
            case 2: return launch_scan_dist_sub<16, 2>(tv, row_begin, n_rows, d_dist, cu, stream);
            case 3: return launch_scan_dist_sub<16, 3>(tv, row_begin, n_rows, d_dist, cu, stream);
            case 4: return launch_scan_dist_sub<16, 4>(tv, row_begin, n_rows, d_dist, cu, stream);
            default: return launch_scan_dist_sub<16, 5>(tv, row_begin, n_rows, d_dist, cu, stream);
            }
        }
        switch ((tv.units + 31u) / 32u) {
        case 3: return launch_scan_dist_sub<32, 3>(tv, row_begin, n_rows, d_dist, cu, stream);
        case 4: return launch_scan_dist_sub<32, 4>(tv, row_begin, n_rows, d_dist, cu, stream);
        case 5: return launch_scan_dist_sub<32, 5>(tv, row_begin, n_rows, d_dist, cu, stream);
        case 6: return launch_scan_dist_sub<32, 6>(tv, row_begin, n_rows, d_dist, cu, stream);
        case 7: return launch_scan_dist_sub<32, 7>(tv, row_begin, n_rows, d_dist, cu, stream);
        default: return launch_scan_dist_sub<32, 8>(tv, row_begin, n_rows, d_dist, cu, stream);
        }
    }
    hipLaunchKernelGGL(k_scan_dist, dim3(grid_blocks), dim3(512), 0, stream, tv.base,
                       tv.row_stride_u, tv.plane_stride_u, tv.units, tv.n_samples, row_begin,
                       n_rows, d_dist);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// pairwise 4x4 contingency for a LIST of SNP pairs: one wave per pair (the getters of
// PairwiseMarkerAnalyzable; the all-pairs screen of BASELINE config 5 is gw_boost*.hip)
// ---------------------------------------------------------------------------
// classes of one 32-bit word: [aa, ab, bb, xx] (DecodeBitStreams2BitStream,
// compressed_genotype_table5.h:104-115)
__device__ __forceinline__ void decode32(uint32_t p0, uint32_t p1, uint32_t (&cls)[4]) {
    cls[3] = ~(p0 | p1);
    cls[2] = p0 & p1;
    cls[0] = p0 ^ cls[2];
    cls[1] = p1 ^ cls[2];
}

template <bool MASKED>
__global__ void __launch_bounds__(256) k_pair(const u32x4 *table, uint64_t row_stride_u,
                                              uint32_t plane_stride_u, uint32_t units,
                                              uint32_t ref_words, const u32x4 *mcase,
                                              const u32x4 *mctrl, const uint32_t *pairs,
                                              uint32_t n_pairs, uint32_t *out_a, uint32_t *out_b) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t gwave = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    const uint32_t total_waves = gridDim.x * waves_per_block;
    for (uint32_t p = gwave; p < n_pairs; p += total_waves) {
        const u32x4 *a0 = table + (uint64_t)pairs[2 * p] * row_stride_u;
        const u32x4 *b0 = table + (uint64_t)pairs[2 * p + 1] * row_stride_u;
        uint32_t ca[16], cb[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { ca[i] = 0; cb[i] = 0; }
        for (uint32_t u = lane; u < units; u += 64u) {
            const u32x4 A0 = a0[u], A1 = a0[plane_stride_u + u];
            const u32x4 B0 = b0[u], B1 = b0[plane_stride_u + u];
            u32x4 MC, MT;
            if (MASKED) { MC = mcase[u]; MT = mctrl[u]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // the reference walks gbo/4 64-bit words; the HBM row may hold
                // one more (all-zero) word, which must not count as xx_xx
                if (2u * u + (uint32_t)(k >> 1) >= ref_words) continue;
                uint32_t x[4], y[4];
                if (MASKED) {
                    decode32(A0[k] & MC[k], A1[k] & MC[k], x);
                    decode32(B0[k] & MC[k], B1[k] & MC[k], y);
                } else {
                    decode32(A0[k], A1[k], x);
                    decode32(B0[k], B1[k], y);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) ca[4 * i + j] += __popc(x[i] & y[j]);
                if (MASKED) {
                    decode32(A0[k] & MT[k], A1[k] & MT[k], x);
                    decode32(B0[k] & MT[k], B1[k] & MT[k], y);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) cb[4 * i + j] += __popc(x[i] & y[j]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            uint32_t s = wave_sum(ca[i]);
            if (lane == 0) out_a[(size_t)p * 16u + i] = s;
            if (MASKED) {
                s = wave_sum(cb[i]);
                if (lane == 0) out_b[(size_t)p * 16u + i] = s;
            }
        }
    }
}

static uint32_t ref_stream_words(uint32_t n_samples) {
    uint32_t blocks = n_samples / 16u + 1u;
    blocks = (blocks + 3u) & ~3u;
    return blocks / 4u;
}

hipError_t launch_pair_contingency(const TableView &tv, const uint32_t *d_pairs, uint32_t n_pairs,
                                   uint32_t *d_out, int grid_blocks, hipStream_t stream) {
    if (n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pair<false>, dim3(grid_blocks), dim3(256), 0, stream, tv.base,
                       tv.row_stride_u, tv.plane_stride_u, tv.units, ref_stream_words(tv.n_samples),
                       (const u32x4 *)nullptr, (const u32x4 *)nullptr, d_pairs, n_pairs, d_out,
                       (uint32_t *)nullptr);
    return hipGetLastError();
}

hipError_t launch_pair_cc_contingency(const TableView &tv, const u32x4 *d_case, const u32x4 *d_ctrl,
                                      const uint32_t *d_pairs, uint32_t n_pairs,
                                      uint32_t *d_out_case, uint32_t *d_out_ctrl, int grid_blocks,
                                      hipStream_t stream) {
    if (n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pair<true>, dim3(grid_blocks), dim3(256), 0, stream, tv.base,
                       tv.row_stride_u, tv.plane_stride_u, tv.units, ref_stream_words(tv.n_samples),
                       d_case, d_ctrl, d_pairs, n_pairs, d_out_case, d_out_ctrl);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// reference packings <-> HBM layout
// ---------------------------------------------------------------------------

// gather the even bits of v into the low 16 bits
__device__ __forceinline__ uint32_t even_bits(uint32_t v) {
    v &= 0x55555555u;
    v = (v | (v >> 1)) & 0x33333333u;
    v = (v | (v >> 2)) & 0x0F0F0F0Fu;
    v = (v | (v >> 4)) & 0x00FF00FFu;
    v = (v | (v >> 8)) & 0x0000FFFFu;
    return v;
}
// spread the low 16 bits of v to the even bit positions
__device__ __forceinline__ uint32_t spread_bits(uint32_t v) {
    v &= 0x0000FFFFu;
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// one workgroup per row (grid-strided), threads over the row's 16-byte units
__global__ void __launch_bounds__(256) k_restride_in(u32x4 *table, uint64_t row_stride_u,
                                                     uint32_t plane_stride_u, uint64_t row_begin,
                                                     uint32_t n_rows, const uint8_t *src,
                                                     size_t ref_stride, int block_layout,
                                                     uint32_t ref_blocks) {
    for (uint32_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const uint16_t *row = reinterpret_cast<const uint16_t *>(src + (size_t)r * ref_stride) + 1;
        u32x4 *d0 = table + (row_begin + r) * row_stride_u;
        u32x4 *d1 = d0 + plane_stride_u;
        for (uint32_t u = threadIdx.x; u < plane_stride_u; u += blockDim.x) {
            u32x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
            if (!block_layout) {
                // stream packing: plane p block i at row[p*ref_blocks + i]; 8 blocks per unit
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t i = 8u * u + j;
                    if (i < ref_blocks) {
                        lo[j >> 1] |= (uint32_t)row[i] << (16 * (j & 1));
                        hi[j >> 1] |= (uint32_t)row[ref_blocks + i] << (16 * (j & 1));
                    }
                }
            } else {
                // block packing: 8 samples x 2 bits per block; 16 blocks per unit;
                // low bit -> plane 0, high bit -> plane 1 (codes 1,3 / 2,3)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t i = 16u * u + j;
                    if (i < ref_blocks) {
                        const uint32_t v = row[i];
                        lo[j >> 2] |= even_bits(v) << (8 * (j & 3));
                        hi[j >> 2] |= even_bits(v >> 1) << (8 * (j & 3));
                    }
                }
            }
            d0[u] = lo;
            d1[u] = hi;
        }
    }
}

__global__ void __launch_bounds__(256) k_restride_out(const u32x4 *table, uint64_t row_stride_u,
                                                      uint32_t plane_stride_u, uint32_t units,
                                                      uint64_t row_begin, uint32_t n_rows,
                                                      uint8_t *dst, size_t ref_stride,
                                                      int block_layout, uint32_t ref_blocks) {
    for (uint32_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        uint16_t *row = reinterpret_cast<uint16_t *>(dst + (size_t)r * ref_stride) + 1;
        const u32x4 *s0 = table + (row_begin + r) * row_stride_u;
        const u32x4 *s1 = s0 + plane_stride_u;
        if (threadIdx.x == 0) row[-1] = 0;  // header: filled in by the host
        const uint32_t per_unit = block_layout ? 16u : 8u;
        const uint32_t need = (ref_blocks + per_unit - 1u) / per_unit;
        for (uint32_t u = threadIdx.x; u < need; u += blockDim.x) {
            u32x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
            if (u < units) { lo = s0[u]; hi = s1[u]; }
            if (!block_layout) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t i = 8u * u + j;
                    if (i < ref_blocks) {
                        row[i] = (uint16_t)(lo[j >> 1] >> (16 * (j & 1)));
                        row[ref_blocks + i] = (uint16_t)(hi[j >> 1] >> (16 * (j & 1)));
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t i = 16u * u + j;
                    if (i < ref_blocks) {
                        const uint32_t l8 = (lo[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                        const uint32_t h8 = (hi[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                        row[i] = (uint16_t)(spread_bits(l8) | (spread_bits(h8) << 1));
                    }
                }
            }
        }
    }
}

static int small_grid(uint32_t n_rows) { return (int)(n_rows < 4096u ? n_rows : 4096u); }

hipError_t launch_restride_in(const TableView &tv, uint64_t row_begin, uint32_t n_rows,
                              const uint8_t *d_src, size_t ref_stride_bytes, int layout,
                              uint32_t ref_blocks, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(k_restride_in, dim3(small_grid(n_rows)), dim3(256), 0, stream, tv.base,
                       tv.row_stride_u, tv.plane_stride_u, row_begin, n_rows, d_src,
                       ref_stride_bytes, layout == 3 ? 1 : 0, ref_blocks);
    return hipGetLastError();
}

hipError_t launch_restride_out(const TableView &tv, uint64_t row_begin, uint32_t n_rows,
                               uint8_t *d_dst, size_t ref_stride_bytes, int layout,
                               uint32_t ref_blocks, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(k_restride_out, dim3(small_grid(n_rows)), dim3(256), 0, stream, tv.base,
                       tv.row_stride_u, tv.plane_stride_u, tv.units, row_begin, n_rows, d_dst,
                       ref_stride_bytes, layout == 3 ? 1 : 0, ref_blocks);
    return hipGetLastError();
}

__global__ void k_mask_in(const uint16_t *src, uint32_t n_blocks, u32x4 *mask, uint32_t units) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units) return;
    u32x4 m = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t i = 8u * u + j;
        if (i < n_blocks) m[j >> 1] |= (uint32_t)src[i] << (16 * (j & 1));
    }
    mask[u] = m;
}

hipError_t launch_mask_in(const uint16_t *d_src, uint32_t n_blocks, u32x4 *d_mask, uint32_t units,
                          uint32_t n_samples, hipStream_t stream) {
    (void)n_samples;
    hipLaunchKernelGGL(k_mask_in, dim3((units + 255u) / 256u), dim3(256), 0, stream, d_src,
                       n_blocks, d_mask, units);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// synthetic input: data/simulate_data.cpp:150-235 restated with a
// counter-based generator (same constants and draw order as
// oracle/gw_oracle.c gwo_sim_*; DESIGN.md "Synthetic input")
// ---------------------------------------------------------------------------
__device__ __constant__ uint32_t k_affy6[51] = {
    // affy6 column of data/maf_spectrum.tab, MAF bins 0.00 .. 0.50
    23486, 18503, 16090, 16751, 15610, 33044, 17674, 16029, 16403, 15372, 29748, 16045, 14741,
    14874, 13997, 27101, 15027, 13233, 14199, 12800, 25738, 13828, 13188, 13368, 12272, 24398,
    12968, 12653, 12657, 11557, 23319, 12626, 11183, 12830, 11411, 22332, 11922, 10943, 12225,
    10944, 21966, 11463, 10873, 11675, 11194, 21596, 11478, 10541, 11511, 10658, 6014};
#define K_AFFY6_TOTAL 792058ull

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
#define SIM_SNP_MUL 0x632BE59BD9B4E019ull
#define SIM_WORD_MUL 0xD1342543DE82EF95ull
#define SIM_DRAW_MUL 0xA0761D6478BD642Full
#define SIM_PHENO_TAG 0x00CA5EC0117201ull

__device__ __forceinline__ uint64_t bernoulli64(uint64_t key, uint32_t base, uint32_t thr) {
    uint64_t lt = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
        const uint64_t r = splitmix64(key + SIM_DRAW_MUL * (uint64_t)(base + k + 1));
        lt = ((thr >> k) & 1u) ? (lt | r) : (lt & r);
    }
    return lt;
}

__device__ __forceinline__ uint64_t valid_bits(uint32_t word, uint32_t n_samples) {
    const uint64_t first = (uint64_t)word * 64ull;
    if (first >= n_samples) return 0ull;
    const uint64_t left = n_samples - first;
    return left >= 64ull ? ~0ull : ((1ull << left) - 1ull);
}

__global__ void __launch_bounds__(256) k_simulate(u32x4 *table, uint64_t row_stride_u,
                                                  uint32_t plane_stride_u, uint32_t n_samples,
                                                  uint64_t n_rows, uint64_t seed, uint64_t snp0,
                                                  uint32_t missing_q16) {
    for (uint64_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        // simulate_data.cpp:176-192: MAF bin ~ spectrum, p = bin/100 + U{0..999}/1e5
        const uint64_t key = splitmix64(seed + SIM_SNP_MUL * (snp0 + r + 1));
        uint64_t pick = splitmix64(key ^ 1u) % K_AFFY6_TOTAL;
        uint32_t bin = 0;
        while (pick >= k_affy6[bin]) { pick -= k_affy6[bin]; ++bin; }
        const uint32_t p_e5 = bin * 1000u + (uint32_t)(splitmix64(key ^ 2u) % 1000u);
        const uint32_t thr = (uint32_t)(((uint64_t)p_e5 * 65536ull) / 100000ull);
        u32x4 *d0 = table + r * row_stride_u;
        u32x4 *d1 = d0 + plane_stride_u;
        for (uint32_t u = threadIdx.x; u < plane_stride_u; u += blockDim.x) {
            uint64_t a[2], b[2];
#pragma unroll
            for (uint32_t h = 0; h < 2; ++h) {
                const uint32_t w = 2u * u + h;
                const uint64_t valid = valid_bits(w, n_samples);
                a[h] = 0; b[h] = 0;
                if (valid) {
                    const uint64_t wkey = key + SIM_WORD_MUL * (uint64_t)(w + 1);
                    const uint64_t m1 = bernoulli64(wkey, 0, thr);
                    const uint64_t m2 = bernoulli64(wkey, 16, thr);
                    uint64_t x = ~(m1 ^ m2), y = m1 | m2;
                    if (missing_q16) {
                        const uint64_t miss = bernoulli64(wkey, 32, missing_q16);
                        x &= ~miss;
                        y &= ~miss;
                    }
                    a[h] = x & valid;
                    b[h] = y & valid;
                }
            }
            u32x4 lo = {(uint32_t)a[0], (uint32_t)(a[0] >> 32), (uint32_t)a[1], (uint32_t)(a[1] >> 32)};
            u32x4 hi = {(uint32_t)b[0], (uint32_t)(b[0] >> 32), (uint32_t)b[1], (uint32_t)(b[1] >> 32)};
            d0[u] = lo;
            d1[u] = hi;
        }
    }
}

hipError_t launch_simulate(const TableView &tv, uint64_t seed, uint64_t snp_global_begin,
                           uint32_t missing_q16, int grid_blocks, hipStream_t stream) {
    if (tv.n_snps == 0) return hipSuccess;
    hipLaunchKernelGGL(k_simulate, dim3(grid_blocks), dim3(256), 0, stream, tv.base,
                       tv.row_stride_u, tv.plane_stride_u, tv.n_samples, tv.n_snps, seed,
                       snp_global_begin, missing_q16);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// The same generator with the reference simulator's EXACT minor-allele count (data/simulate_data.cpp:191-205: exactly
// s = floor(p * 2N) 'B' alleles among the 2N alleles of a SNP, shuffled; consecutive allele pairs are the genotypes).
// A uniformly random arrangement of s marked alleles among 2N = the s smallest of 2N independent random keys: allele
// `pos` of the row gets the 32-bit key below (one splitmix64 per sample gives the keys of its two alleles), and the
// minor alleles are the s positions with the smallest (key, pos).  One workgroup per row finds the s-th smallest pair
// by a 4096-bin histogram of the keys' top 12 bits, an exact rank inside the one bin the order statistic falls in
// (about 2N / 4096 keys, held in LDS), and then writes the planes: three passes over the row's keys, nothing stored.
// oracle/gw_oracle.c gwo_sim_row_stream_exact is the same rule with a sort.
// ---------------------------------------------------------------------------
#define SIM_ALLELE_MUL 0x9FB21C651E98DF25ull
constexpr uint32_t SIM_BINS = 4096, SIM_LIST_CAP = 4096;

__device__ __forceinline__ uint64_t allele_keys(uint64_t rowkey, uint32_t sample) {  // alleles 2c (high half), 2c + 1 (low half)
    return splitmix64(rowkey + SIM_ALLELE_MUL * (uint64_t)(sample + 1u));
}

__global__ void __launch_bounds__(512) k_simulate_exact(u32x4 *table, uint64_t row_stride_u, uint32_t plane_stride_u,
                                                        uint32_t n_samples, uint64_t n_rows, uint64_t seed, uint64_t snp0,
                                                        uint32_t missing_q16, uint32_t *overflow) {
    __shared__ uint32_t s_hist[SIM_BINS];
    __shared__ unsigned long long s_list[SIM_LIST_CAP];
    __shared__ uint32_t s_scan[512];
    __shared__ uint32_t s_bin, s_below, s_count;
    __shared__ unsigned long long s_thr;
    const uint32_t t = threadIdx.x;
    for (uint64_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const uint64_t key = splitmix64(seed + SIM_SNP_MUL * (snp0 + r + 1));
        uint64_t pick = splitmix64(key ^ 1u) % K_AFFY6_TOTAL;
        uint32_t bin = 0;
        while (pick >= k_affy6[bin]) { pick -= k_affy6[bin]; ++bin; }
        const uint32_t p_e5 = bin * 1000u + (uint32_t)(splitmix64(key ^ 2u) % 1000u);
        const uint32_t s = (uint32_t)(((uint64_t)p_e5 * 2ull * n_samples) / 100000ull);  // floor(p * 2N), simulate_data.cpp:193

        // ---- 1. histogram of the keys' top 12 bits
        __syncthreads();
        for (uint32_t k = t; k < SIM_BINS; k += 512u) s_hist[k] = 0;
        if (t == 0) { s_count = 0; s_bin = 0xFFFFFFFFu; s_below = 0; s_thr = 0; }
        __syncthreads();
        for (uint32_t c = t; c < n_samples; c += 512u) {
            const uint64_t h = allele_keys(key, c);
            atomicAdd(&s_hist[(uint32_t)(h >> 52)], 1u);
            atomicAdd(&s_hist[((uint32_t)h) >> 20], 1u);
        }
        __syncthreads();
        // ---- 2. the bin the s-th smallest key falls in (8 bins per thread, scan over the 512 partial sums)
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) mine += s_hist[8u * t + k];
        s_scan[t] = mine;
        __syncthreads();
        for (uint32_t d = 1; d < 512u; d <<= 1) {
            const uint32_t v = t >= d ? s_scan[t - d] : 0u;
            __syncthreads();
            s_scan[t] += v;
            __syncthreads();
        }
        const uint32_t incl = s_scan[t], excl = incl - mine;
        if (s >= 1u && excl < s && s <= incl) {
            uint32_t below = excl;
            for (uint32_t k = 0; k < 8; ++k) {
                const uint32_t hcount = s_hist[8u * t + k];
                if (s <= below + hcount) { s_bin = 8u * t + k; s_below = below; break; }
                below += hcount;
            }
        }
        __syncthreads();
        const uint32_t the_bin = s_bin, rank = s - s_below;  // the rank-th smallest (1-based) of that bin is the last minor allele
        // ---- 3. that bin's (key, pos) pairs, and the one of them with rank - 1 smaller ones
        if (the_bin != 0xFFFFFFFFu) {
            for (uint32_t c = t; c < n_samples; c += 512u) {
                const uint64_t h = allele_keys(key, c);
                const uint32_t k0 = (uint32_t)(h >> 32), k1 = (uint32_t)h;
                if ((k0 >> 20) == the_bin) {
                    const uint32_t slot = atomicAdd(&s_count, 1u);
                    if (slot < SIM_LIST_CAP) s_list[slot] = ((unsigned long long)k0 << 32) | (2u * c);
                }
                if ((k1 >> 20) == the_bin) {
                    const uint32_t slot = atomicAdd(&s_count, 1u);
                    if (slot < SIM_LIST_CAP) s_list[slot] = ((unsigned long long)k1 << 32) | (2u * c + 1u);
                }
            }
            __syncthreads();
            const uint32_t cnt = s_count;
            if (cnt > SIM_LIST_CAP) {
                if (t == 0) atomicAdd(overflow, 1u);  // cohorts beyond ~6 M samples: reported by the host, nothing is wrong silently
            } else {
                for (uint32_t e = t; e < cnt; e += 512u) {
                    const unsigned long long v = s_list[e];
                    uint32_t smaller = 0;
                    for (uint32_t q = 0; q < cnt; ++q) smaller += s_list[q] < v ? 1u : 0u;
                    if (smaller == rank - 1u) s_thr = v;
                }
            }
        }
        __syncthreads();
        const unsigned long long thr = s_thr;
        const bool any = the_bin != 0xFFFFFFFFu;
        // ---- 4. the planes: allele `pos` is the minor allele iff (key, pos) <= thr
        u32x4 *d0 = table + r * row_stride_u;
        u32x4 *d1 = d0 + plane_stride_u;
        for (uint32_t u = t; u < plane_stride_u; u += 512u) {
            uint64_t a[2], b[2];
#pragma unroll
            for (uint32_t hh = 0; hh < 2; ++hh) {
                const uint32_t w = 2u * u + hh;
                const uint64_t valid = valid_bits(w, n_samples);
                a[hh] = 0; b[hh] = 0;
                if (valid) {
                    uint64_t m1 = 0, m2 = 0;
                    for (uint32_t bit = 0; bit < 64u; ++bit) {
                        const uint32_t c = w * 64u + bit;
                        if (c >= n_samples) break;
                        const uint64_t h = allele_keys(key, c);
                        const unsigned long long e0 = ((h >> 32) << 32) | (2u * c), e1 = ((h & 0xFFFFFFFFull) << 32) | (2u * c + 1u);
                        m1 |= (uint64_t)(any && e0 <= thr) << bit;
                        m2 |= (uint64_t)(any && e1 <= thr) << bit;
                    }
                    uint64_t x = ~(m1 ^ m2), y = m1 | m2;
                    if (missing_q16) {
                        const uint64_t wkey = key + SIM_WORD_MUL * (uint64_t)(w + 1);
                        const uint64_t miss = bernoulli64(wkey, 32, missing_q16);
                        x &= ~miss;
                        y &= ~miss;
                    }
                    a[hh] = x & valid;
                    b[hh] = y & valid;
                }
            }
            u32x4 lo = {(uint32_t)a[0], (uint32_t)(a[0] >> 32), (uint32_t)a[1], (uint32_t)(a[1] >> 32)};
            u32x4 hi = {(uint32_t)b[0], (uint32_t)(b[0] >> 32), (uint32_t)b[1], (uint32_t)(b[1] >> 32)};
            d0[u] = lo;
            d1[u] = hi;
        }
    }
}

hipError_t launch_simulate_exact(const TableView &tv, uint64_t seed, uint64_t snp_global_begin, uint32_t missing_q16,
                                 int grid_blocks, uint32_t *d_overflow, hipStream_t stream) {
    if (tv.n_snps == 0) return hipSuccess;
    hipLaunchKernelGGL(k_simulate_exact, dim3(grid_blocks), dim3(512), 0, stream, tv.base, tv.row_stride_u,
                       tv.plane_stride_u, tv.n_samples, tv.n_snps, seed, snp_global_begin, missing_q16, d_overflow);
    return hipGetLastError();
}

// simulate_data.cpp:219-235: fair coin per sample, 1 = case, 0 = control
__global__ void k_simulate_phenotype(uint64_t seed, u32x4 *mcase, u32x4 *mctrl, uint32_t units,
                                     uint32_t n_samples, uint32_t *counts2) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units) return;
    uint64_t c[2], t[2];
#pragma unroll
    for (uint32_t h = 0; h < 2; ++h) {
        const uint32_t w = 2u * u + h;
        const uint64_t coin = splitmix64((seed ^ SIM_PHENO_TAG) + SIM_WORD_MUL * (uint64_t)(w + 1));
        const uint64_t valid = valid_bits(w, n_samples);
        c[h] = coin & valid;
        t[h] = ~coin & valid;
    }
    u32x4 vc = {(uint32_t)c[0], (uint32_t)(c[0] >> 32), (uint32_t)c[1], (uint32_t)(c[1] >> 32)};
    u32x4 vt = {(uint32_t)t[0], (uint32_t)(t[0] >> 32), (uint32_t)t[1], (uint32_t)(t[1] >> 32)};
    mcase[u] = vc;
    mctrl[u] = vt;
    atomicAdd(&counts2[0], (uint32_t)(__popcll(c[0]) + __popcll(c[1])));
    atomicAdd(&counts2[1], (uint32_t)(__popcll(t[0]) + __popcll(t[1])));
}

hipError_t launch_simulate_phenotype(uint64_t seed, u32x4 *d_case, u32x4 *d_ctrl, uint32_t units,
                                     uint32_t n_samples, uint32_t *d_counts2, hipStream_t stream) {
    hipLaunchKernelGGL(k_simulate_phenotype, dim3((units + 255u) / 256u), dim3(256), 0, stream,
                       seed, d_case, d_ctrl, units, n_samples, d_counts2);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// streaming-read probe: the empirical HBM read ceiling on this very buffer -- a
// plain 16-byte-per-lane nt read of the whole table allocation (rows incl. their
// padding), flat grid-stride, four loads in flight per lane, folded into an XOR.
// Of the traversals compared in tools/read_patterns.hip this one reads 7.0-7.15
// TB/s at every row length (a wave-per-row read drops to 4.4 TB/s at N = 10 k).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(512) k_stream_probe(const u32x4 *table, uint64_t n_units, uint32_t *sink) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + 3ull * stride < n_units; i += 4ull * stride) {
        const u32x4 a = ld_stream<true>(table + i), b = ld_stream<true>(table + i + stride),
                    c = ld_stream<true>(table + i + 2 * stride), d = ld_stream<true>(table + i + 3 * stride);
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n_units; i += stride) acc ^= ld_stream<true>(table + i);
    uint32_t x = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x ^= __shfl_xor(x, off, 64);
    if ((threadIdx.x & 63u) == 0 && x == 0xDEADBEEFu) atomicAdd(sink, 1u);  // keeps the loads alive
}

hipError_t launch_stream_probe(const TableView &tv, uint32_t *d_sink, int grid_blocks, hipStream_t stream) {
    if (tv.n_snps == 0) return hipSuccess;
    hipLaunchKernelGGL(k_stream_probe, dim3(grid_blocks), dim3(512), 0, stream, tv.base,
                       tv.n_snps * tv.row_stride_u, d_sink);
    return hipGetLastError();
}

}  // namespace gw
