// gw_boost.hip -- gfx950 kernels of the pairwise (SNP x SNP) path: the
// BOOST / KSA pre-screen of libgwaspp's computeBoost
// (algorithms/epistasis_func.cpp:349-506) over all SNP pairs of a table.
//
// Reference steps restated (paths relative to /root/reference/src/libgwaspp/):
//   k_select_cc   selectCaseControl: cases and controls compacted into dense
//                 bit-planes, genotype/compressed_genotype_table5.cpp:443-575
//   k_margins     computeMargins -> computeMarginalInformation,
//                 algorithms/epistasis_func.cpp:706-721,
//                 genotype/common_genotype_func.cpp:173-219
//   k_retile_cc   the block-tiled copy of the compacted rows the MFMA kernels read
//   k_boost       the O(M^2) loop of computeBoost (:393-486): margin-assisted
//                 case/control 3x3 tables (table5.cpp:989-1150) + the
//                 interaction measure (:424-470) + the threshold list (:482),
//                 vector-ALU form (AND + v_bcnt); launch_boost picks the form
//
// Unlike the per-SNP scan this path re-uses every row M times: it is compute-
// bound, and the pair tables are a binary GEMM.  The default forms run on the
// matrix cores: gw_boost_seq4.hip / gw_boost_seq9.hip (k_boost_mfma_seq, one group
// at a time) and gw_boost_mfma.hip / gw_boost_both9.hip (both groups at once, any
// group size); shared
// device code (measure, bound, operand expansion) in gw_boost_dev.h.
#include "gw_boost_dev.h"

#include <algorithm>
#include <cmath>
#include <cstdio>

namespace gw {
// ---------------------------------------------------------------------------
// selectCaseControl: the members' bits of a row gathered into dense planes (compressed_genotype_table5.cpp:443-575 does it
// bit by bit; so did this kernel until round 4, one output bit at a time through an index list: 2.0 ms for 100 000 rows x
// 10 000 samples).  Now word by word: a 32-sample word of a plane is compressed under the group's mask word by the
// five-stage network of "Hacker's Delight" 7-4 (the masks are the same for every row: the host prepares the network once
// per mask, `plan`), and the compressed bits are OR-ed into the row's output at the bit offset of the word's first member
// (LDS atomics: a word's bits land in at most two output words).
// plan: four u32x4 per input word -- {case mask, mv0, mv1, mv2} {mv3, mv4, bit offset, members} and the same for controls.
// Compact row layout (32-bit words): [case p0 | case p1 | ctrl p0 | ctrl p1], case planes wc words, control planes wt words.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t compress_bits(uint32_t x, const u32x4 &a, const u32x4 &b) {
    x &= a[0];
    uint32_t t;
    t = x & a[1]; x = (x ^ t) | (t >> 1);
    t = x & a[2]; x = (x ^ t) | (t >> 2);
    t = x & a[3]; x = (x ^ t) | (t >> 4);
    t = x & b[0]; x = (x ^ t) | (t >> 8);
    t = x & b[1]; x = (x ^ t) | (t >> 16);
    return x;
}
__device__ __forceinline__ void place_bits(uint32_t *plane, uint32_t bits, uint32_t off, uint32_t count) {
    const uint32_t sh = off & 31u;
    if (bits) atomicOr(&plane[off >> 5], bits << sh);
    if (sh + count > 32u && (bits >> (32u - sh))) atomicOr(&plane[(off >> 5) + 1u], bits >> (32u - sh));
}
__global__ void __launch_bounds__(256) k_select_cc(const u32x4 *table, uint64_t row_stride_u, uint32_t plane_stride_u,
                                                   uint32_t n_words, uint32_t n_rows, const u32x4 *plan, uint32_t wc,
                                                   uint32_t wt, uint32_t *cc, uint64_t cc_stride_w) {
    extern __shared__ uint32_t s_out[];  // one compact row
    const uint32_t total = 2u * (wc + wt);
    for (uint32_t k = threadIdx.x; k < total; k += blockDim.x) s_out[k] = 0u;
    for (uint32_t r = blockIdx.x; r < n_rows; r += gridDim.x) {  // (the same trip count for every thread of a workgroup)
        const uint32_t *p0 = reinterpret_cast<const uint32_t *>(table + (uint64_t)r * row_stride_u);
        const uint32_t *p1 = p0 + (uint64_t)plane_stride_u * 4u;
        __syncthreads();
        for (uint32_t w = threadIdx.x; w < n_words; w += blockDim.x) {
            const uint32_t x0 = p0[w], x1 = p1[w];
            const u32x4 ca = plan[4u * w], cb = plan[4u * w + 1u], ta = plan[4u * w + 2u], tb = plan[4u * w + 3u];
            if (cb[3]) {
                place_bits(s_out, compress_bits(x0, ca, cb), cb[2], cb[3]);
                place_bits(s_out + wc, compress_bits(x1, ca, cb), cb[2], cb[3]);
            }
            if (tb[3]) {
                place_bits(s_out + 2u * wc, compress_bits(x0, ta, tb), tb[2], tb[3]);
                place_bits(s_out + 2u * wc + wt, compress_bits(x1, ta, tb), tb[2], tb[3]);
            }
        }
        __syncthreads();
        uint32_t *out = cc + (uint64_t)r * cc_stride_w;
        for (uint32_t k = threadIdx.x; k < total; k += blockDim.x) {
            out[k] = s_out[k];
            s_out[k] = 0u;  // for the next row (the barrier at the top orders it before that row's atomics)
        }
    }
}

hipError_t launch_select_cc(const TableView &tv, const uint32_t *d_plan, uint32_t n_words, uint32_t wc, uint32_t wt,
                            uint32_t *d_cc, uint64_t cc_stride_w, int compute_units, hipStream_t stream) {
    if (tv.n_snps == 0) return hipSuccess;
    const size_t lds = (size_t)2u * (wc + wt) * sizeof(uint32_t);
    if (lds > (size_t)160u * 1024u - 256u) return hipErrorInvalidValue;  // a compact row has to fit the 160 KB of a CU's LDS
    if (lds > 48u * 1024u) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_select_cc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint64_t cap = (uint64_t)(compute_units > 0 ? compute_units : 256) * 8u;
    const uint32_t grid = (uint32_t)(tv.n_snps < cap ? tv.n_snps : cap);
    hipLaunchKernelGGL(k_select_cc, dim3(grid), dim3(256), lds, stream, tv.base, tv.row_stride_u, tv.plane_stride_u,
                       n_words, (uint32_t)tv.n_snps, reinterpret_cast<const u32x4 *>(d_plan), wc, wt, d_cc, cc_stride_w);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// marginal_information per SNP from the scan's counts
// (common_genotype_func.cpp:173-219; unwritten entries are zero here)
// ---------------------------------------------------------------------------
__global__ void k_margins(const uint32_t *counts, uint32_t n_rows, uint32_t n_individs, SnpMargin *mar,
                          SnpMarginF *marf) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t *c = counts + (size_t)r * 8u;
    SnpMargin m;
    uint32_t n_cases = 0, n_ctrls = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        m.cases[g] = c[g];
        m.ctrls[g] = c[4 + g];
        n_cases += c[g];
        n_ctrls += c[4 + g];
    }
    const double n = (double)n_individs;
    m.entropy = 0.0;
    m.entropy_y = 0.0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t ca = m.cases[g], co = m.ctrls[g], mg = ca + co;
        m.margins[g] = mg;
        m.pbc[g] = m.pbc[4 + g] = m.pca[g] = m.pca[4 + g] = 0.0;
        if (mg > 0) {
            const double t = (double)mg / n;
            m.entropy += -(t)*log(t);
        }
        if (ca > 0) {
            const double t = (double)ca / n;
            m.entropy_y += -(t)*log(t);
            m.pbc[g] = (double)ca / (double)n_cases;
            m.pca[g] = (double)ca / (double)mg;
        }
        if (co > 0) {
            const double t = (double)co / n;
            m.entropy_y += -(t)*log(t);
            m.pbc[4 + g] = (double)co / (double)n_ctrls;
            m.pca[4 + g] = (double)co / (double)mg;
        }
    }
    mar[r] = m;
    if (!marf) return;
    // single-precision factors of the pre-screen's bound (not part of the reference's record)
    SnpMarginF f;
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
        const double n_grp = (double)(grp == 0 ? n_cases : n_ctrls);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const uint32_t k = grp == 0 ? m.cases[g] : m.ctrls[g];
            const double mg = (double)m.margins[g];
            f.qb[3 * grp + g] = k ? (float)(mg * n_grp / (double)k) : 0.0f;
            f.qa[3 * grp + g] = k ? (float)(mg / (double)k) : 0.0f;
            f.rb[3 * grp + g] = k ? (float)(m.pbc[4 * grp + g] / mg) : 0.0f;
            f.pa[3 * grp + g] = (float)m.pca[4 * grp + g];
        }
    }
    double ki = 0.0, kj = 0.0;
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
        const double n_grp = (double)(grp == 0 ? n_cases : n_ctrls);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const uint32_t k = grp == 0 ? m.cases[g] : m.ctrls[g];
            if (!k) continue;
            const double mg = (double)m.margins[g], kd = (double)k;
            ki += kd * log2(mg / kd);
            kj += kd * log2(mg * n_grp / (kd * n));
        }
    }
    f.ki = (float)ki;
    f.kj = (float)kj;
    // tao - 1 from the four corners (SnpMarginF::ua, ::vb; the derivation is at boost_bound_quad_f)
    double ea = 0.0, eb = 0.0;
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
        const double n_grp = (double)(grp == 0 ? n_cases : n_ctrls), pc = n_grp / n;
        double d[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const uint32_t k = grp == 0 ? m.cases[g] : m.ctrls[g];
            d[g] = (m.margins[g] && n_grp > 0.0) ? ((double)k / (double)m.margins[g]) / pc - 1.0 : 0.0;
        }
        const double u0 = d[0] - d[1], u2 = d[2] - d[1];
        f.ua[3 * grp + 0] = (float)u0;
        f.ua[3 * grp + 1] = (float)u2;
        f.ua[3 * grp + 2] = (float)(n * d[1]);
        f.vb[3 * grp + 0] = (float)(pc / n * u0);
        f.vb[3 * grp + 1] = (float)(pc / n * u2);
        f.vb[3 * grp + 2] = (float)(pc / n * d[1]);
        ea = fmax(ea, fmax(fmax(fabs(u0), fabs(u2)), fabs(d[1])));
        eb += pc / n * ((double)m.margins[0] * fabs(u0) + (double)m.margins[2] * fabs(u2) + n * fabs(d[1]));
    }
    f.ea = (float)(ea * 1.000001);   // rounded up: they scale an error bound
    f.eb = (float)(eb * 1.000001);
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
        const double n_grp = (double)(grp == 0 ? n_cases : n_ctrls);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const uint32_t k = grp == 0 ? m.cases[g] : m.ctrls[g];
            const double mg = (double)m.margins[g], kd = (double)k;
            f.lqa[3 * grp + g] = k ? (float)log2(mg / kd) : 0.0f;
            f.lqb[3 * grp + g] = k ? (float)log2(mg * n_grp / (kd * n)) : 0.0f;
        }
    }
    float ma = 0.0f, mb = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) { ma = fmaxf(ma, fabsf(f.lqa[k])); mb = fmaxf(mb, fabsf(f.lqb[k])); }
    f.ma = ma * 1.000001f;
    f.mb = mb * 1.000001f;
    marf[r] = f;
}

hipError_t launch_margins(const uint32_t *d_counts, uint32_t n_rows, uint32_t n_individs, SnpMargin *d_mar,
                          SnpMarginF *d_marf, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(k_margins, dim3((n_rows + 255u) / 256u), dim3(256), 0, stream, d_counts, n_rows,
                       n_individs, d_mar, d_marf);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// BOOST stage 2 on the device: computeGTest (algorithms/epistasis_func.cpp:508-704) for a list of pairs.
//   k_gtest_tables  one wave per pair: the two 3 x 3 tables from the compacted rows (selectCaseControl's planes, so a sample
//                   in both masks is a case only, as in the reference's compacted tables table5.cpp:989-1064)
//   k_gtest         one wave per pair: iterative proportional fitting of the homogeneous-association model in FP64, the
//                   exact interaction measure, the difference of the allelic log odds ratios over its standard error --
//                   operation for operation the oracle's gwo_gtest (-ffp-contract=off on both sides)
// The reference's `while (muError > 0.001)` has no bound; a kernel must have one: GTEST_MAX_ITER sweeps, after which
// the pair is evaluated as it stands and counted in status[0].  The bound is far out because the reference's own loop is
// slow where the tables have missing calls: the pair table then does not add up to the SNPs' margins, the fit creeps
// (error ~ 1 / sweep) and takes 10^4 sweeps at the median and up to 3 x 10^6 for a pair at 1 % missing calls
// (tests/test_gpu_boost.py measures against the oracle there); tables without a missing call settle in tens of sweeps.
// ---------------------------------------------------------------------------
constexpr int GTEST_MAX_ITER = 1 << 24;

__global__ void __launch_bounds__(256) k_gtest_tables(const uint32_t *cc, uint64_t cc_stride_w, uint32_t wc, uint32_t wt,
                                                      const uint32_t *pairs, uint32_t pair_stride, const unsigned long long *n_ptr,
                                                      unsigned long long n_max, uint32_t *tabs) {
    const unsigned long long n = n_ptr ? (*n_ptr < n_max ? *n_ptr : n_max) : n_max;
    const uint32_t lane = threadIdx.x & 63u, wpb = blockDim.x >> 6;
    for (unsigned long long p = (unsigned long long)blockIdx.x * wpb + (threadIdx.x >> 6); p < n; p += (unsigned long long)gridDim.x * wpb) {
        const uint32_t i = pairs[p * pair_stride], j = pairs[p * pair_stride + 1];
        const uint32_t *ri = cc + (uint64_t)i * cc_stride_w, *rj = cc + (uint64_t)j * cc_stride_w;
        uint32_t cnt[18];
#pragma unroll
        for (int c = 0; c < 18; ++c) cnt[c] = 0;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint32_t words = g == 0 ? wc : wt, base = g == 0 ? 0u : 2u * wc;
            for (uint32_t w = lane; w < words; w += 64u) {
                const uint32_t a0 = ri[base + w], a1 = ri[base + words + w], b0 = rj[base + w], b1 = rj[base + words + w];
                const uint32_t ca[3] = {a0 & ~a1, a1 & ~a0, a0 & a1}, cb[3] = {b0 & ~b1, b1 & ~b0, b0 & b1};
#pragma unroll
                for (int x = 0; x < 3; ++x)
#pragma unroll
                    for (int y = 0; y < 3; ++y) cnt[9 * g + 3 * x + y] += (uint32_t)__popc(ca[x] & cb[y]);
            }
        }
#pragma unroll
        for (int c = 0; c < 18; ++c) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) cnt[c] += __shfl_xor(cnt[c], off, 64);
        }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 18; ++c) tabs[p * 18u + c] = cnt[c];
        }
    }
}

// One WAVE per pair.  The fit of a pair is a serial chain of sweeps (a table with an empty cell takes thousands of them,
// one with missing calls millions), so what a launch takes is the slowest pair's sweep count times the latency of one
// sweep.  Eighteen lanes share a sweep -- lane l < 18 owns cell (group l / 9, a = (l % 9) / 3, b = l % 3) -- so that a
// sweep is 3 divisions per lane instead of 30 in one lane; the sums the reference accumulates in a fixed order (its
// ik[], jk[] and the error it tests, 18 terms) are formed in that very order from the other lanes' values, which keeps
// every double bit-identical to the one-lane form (the oracle's gwo_gtest; 0.0 + x = x for the first addend).
// (a first version ran one lane per pair: config 5's 22 096 hits took 56 ms -- the slowest pair's ~20 000 sweeps at
// 2.7 us each; this form: see DESIGN.md)
__device__ __forceinline__ double gt_lane(double v, int src) { return __shfl(v, src, 64); }   // per-lane source: ds_bpermute
// the same value for every lane from a lane known at compile time: two v_readlane_b32, no LDS crossbar
__device__ __forceinline__ double gt_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(256) k_gtest(const uint32_t *tabs, const uint32_t *pairs, uint32_t pair_stride,
                                               const unsigned long long *n_ptr, unsigned long long n_max, const SnpMargin *mar,
                                               uint32_t n_individs, double *out, unsigned long long *status) {
    const unsigned long long n = n_ptr ? (*n_ptr < n_max ? *n_ptr : n_max) : n_max;
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t wpb = blockDim.x >> 6;
    const int l = lane < 18 ? lane : 17;                 // lanes 18 .. 63 shadow lane 17 (their values are never read)
    const int g = l / 9, i = l % 9, a = i / 3, b = i % 3;
    for (unsigned long long p = (unsigned long long)blockIdx.x * wpb + (threadIdx.x >> 6); p < n; p += (unsigned long long)gridDim.x * wpb) {
        const uint32_t *tab = tabs + p * 18u;
        const SnpMargin *m1 = mar + pairs[p * pair_stride], *m2 = mar + pairs[p * pair_stride + 1];
        const uint32_t cnt = tab[9 * g + i];
        const uint32_t tot = tab[i] + tab[9 + i];
        const uint32_t f1 = g == 0 ? m1->cases[a] : m1->ctrls[a], f2 = g == 0 ? m2->cases[b] : m2->ctrls[b];
        // One sweep of the fit (:571-645): every lane its cell, sums in the reference's order.
        auto sweep = [&](double mu) -> double {
            const double other = gt_lane(mu, g == 0 ? l + 9 : l - 9);
            const double mij = g == 0 ? mu + other : other + mu;          // mu_ca[i] + mu_co[i]
            mu = mij > 0 ? mu * (tot) / mij : 0.0;                        // :571-598
            const int row = 9 * g + 3 * a, col = 9 * g + b;
            double ik = 0.0 + gt_lane(mu, row);                           // ik[a] += mu[a][0], [a][1], [a][2] in this order
            ik += gt_lane(mu, row + 1);
            ik += gt_lane(mu, row + 2);
            double jk = 0.0 + gt_lane(mu, col);                           // jk[b] += mu[0][b], [1][b], [2][b]
            jk += gt_lane(mu, col + 3);
            jk += gt_lane(mu, col + 6);
            const double t1 = ik > 0 ? f1 / ik : 0.0;                     // :611-645
            const double t3 = jk > 0 ? f2 / jk : 0.0;
            return mu * (t1)*t3;
        };
        // The reference's loop is  while (err > 0.001) { sweep; err = sum over the 18 cells of |new - old| }  with err = 18 before
        // the first sweep (:552-556).  Here sweep k + 1 is computed WHILE the error of sweep k is summed -- the sum is a chain of
        // 18 dependent additions over broadcast lanes, a fifth of a sweep's latency, and the next sweep does not depend on it; when
        // the error says stop, the sweep that ran ahead is dropped.  Same values, same number of sweeps counted.
        // The error itself, in the reference's order, is a chain of 18 broadcasts and additions -- half of a sweep's instructions,
        // and the sweeps of the slowest pair (27 000 at config 5) are the kernel's run time.  The loop only needs to know on which
        // side of 0.001 it lies: the same 18 non-negative terms summed in FP32 by a shift-and-add tree (four v_add_f32 with a DPP
        // row shift, the two row totals broadcast) differ from the ordered FP64 sum by less than 2e-6 of it (rounding of the
        // terms to float 6e-8 each, of the sums 18 x 6e-8; a term too small for a float counts as 0, one too large as infinity), so
        // outside a band of 2e-5 around the threshold the FP32 sum decides, and inside it -- or when it is a NaN -- the ordered sum does.
        auto settled = [&](double d) -> bool {   // d = the lane's |new - old|; true <=> the reference's error sum is NOT > 0.001
            float f = lane < 18 ? (float)d : 0.0f;
            f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x111, 0xf, 0xf, true));  // row_shr:1
            f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x112, 0xf, 0xf, true));  // row_shr:2
            f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x114, 0xf, 0xf, true));  // row_shr:4
            f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x118, 0xf, 0xf, true));  // row_shr:8
            const float total = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, f), 15)) +
                                __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, f), 31));
            if (total > 0.00100002f) return false;
            if (total < 0.00099998f) return true;
            asm volatile("" : "+v"(d));   // (keeps the ordered sum on this side of the branches: it has no side effect to hold it here)
            double err = 0.0;
#pragma unroll
            for (int c = 0; c < 9; ++c) {                                 // err += |ca[c] - old|; err += |co[c] - old|, c = 0 .. 8
                err += gt_bcast(d, c);
                err += gt_bcast(d, 9 + c);
            }
            return !(err > 0.001);
        };
        double mu = sweep(1.0);
        double d = fabs(mu - 1.0);
        int iter = 1;
        bool done;
        for (;;) {
            const double ahead = sweep(mu);
            done = settled(d);
            if (done || iter >= GTEST_MAX_ITER) break;
            d = fabs(ahead - mu);
            mu = ahead;
            ++iter;
        }
        // :648-683 -- every lane its cell's terms, lane-uniform sums in the reference's order
        double t1 = 0.0, term1 = 0.0, term2 = 0.0, t2 = 0.0;
        if (cnt > 0) { t1 = (double)cnt / n_individs; term1 = t1 * log(t1); }
        const bool fitted = mu > 0;
        if (fitted) { t2 = mu / n_individs; term2 = -t1 * log(t2); }
        double tao = 0.0, im = 0.0;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int src = 9 * gg + c;
                const double x1 = gt_bcast(term1, src), x2 = gt_bcast(term2, src), x3 = gt_bcast(t2, src);
                const bool has = __builtin_amdgcn_readlane((int)(cnt > 0), src) != 0, fit = __builtin_amdgcn_readlane((int)fitted, src) != 0;
                if (has) im += x1;
                if (fit) { im += x2; tao += x3; }
            }
        }
        if (lane == 0) {
            if (!done && status) atomicAdd(&status[0], 1ull);
            out[2 * p] = (im + log(tao)) * n_individs * 2.0;
            // allelic joint distribution and the odds-ratio z value, :686-701 (uint arithmetic, as the reference)
            uint32_t aj[8];
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const uint32_t *t = tab + 9 * gg;  // 3 x 3 here: the reference's 4 x 4 indices 0,1,2,4,5,6,8,9,10
                aj[4 * gg + 0] = (t[0] << 2) + (t[1] << 1) + (t[3] << 1) + t[4];
                aj[4 * gg + 1] = (t[2] << 2) + (t[1] << 1) + (t[5] << 1) + t[4];
                aj[4 * gg + 2] = (t[6] << 2) + (t[7] << 1) + (t[3] << 1) + t[4];
                aj[4 * gg + 3] = (t[8] << 2) + (t[7] << 1) + (t[5] << 1) + t[4];
            }
            const double or_aff = log((double)(aj[0] * aj[3]) / (double)(aj[1] * aj[2]));
            const double v_aff = 1 / (double)aj[0] + 1 / (double)aj[1] + 1 / (double)aj[2] + 1 / (double)aj[3];
            const double or_unf = log((double)(aj[4] * aj[7]) / (double)(aj[5] * aj[6]));
            const double v_unf = 1 / (double)aj[4] + 1 / (double)aj[5] + 1 / (double)aj[6] + 1 / (double)aj[7];
#ifdef GW_GTEST_COUNT_SWEEPS   // measurement only (tools/build_variant_file.sh): the sweeps a pair took, in place of its z value
            out[2 * p + 1] = (double)iter;
#else
            out[2 * p + 1] = (or_aff - or_unf) / sqrt(v_aff + v_unf);
#endif
        }
    }
}

hipError_t launch_boost_gtest(const uint32_t *d_cc, uint64_t cc_stride_w, uint32_t wc, uint32_t wt, const SnpMargin *d_mar,
                              uint32_t n_individs, const uint32_t *d_pairs, uint32_t pair_stride,
                              const unsigned long long *d_n, unsigned long long n_max, uint32_t *d_tabs, double *d_out,
                              unsigned long long *d_status, int compute_units, hipStream_t stream) {
    if (n_max == 0) return hipSuccess;
    const unsigned waves = (unsigned)std::min<unsigned long long>(n_max, (unsigned long long)(compute_units > 0 ? compute_units : 256) * 32ull);
    hipLaunchKernelGGL(k_gtest_tables, dim3((waves + 3u) / 4u), dim3(256), 0, stream, d_cc, cc_stride_w, wc, wt, d_pairs, pair_stride,
                       d_n, n_max, d_tabs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_gtest, dim3((waves + 3u) / 4u), dim3(256), 0, stream, d_tabs, d_pairs, pair_stride, d_n, n_max,
                       d_mar, n_individs, d_out, d_status);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// the pair kernel
// ---------------------------------------------------------------------------
namespace {
constexpr int TA = 32;  // A-block rows (SNP i) per workgroup tile
constexpr int TB = 32;  // B-block rows (SNP j)
constexpr int CH = 32;  // 32-bit words of every row staged per step (1024 samples)
constexpr int LD = CH + 1;  // padded LDS row length

}  // namespace

// t1 = k/n and t1*log(t1) for every possible cell count k: the same expressions
// the measure evaluates inline (so the table entries are bit-identical), done
// once per table instead of 18 times per pair.
__global__ void k_boost_tables(uint32_t n_entries, int n, double *frac, double *xlogx) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_entries) return;
    const double t1 = (double)k / n;
    frac[k] = t1;
    xlogx[k] = k ? t1 * log(t1) : 0.0;
}

hipError_t launch_boost_tables(uint32_t n_entries, uint32_t n_individs, double *d_frac, double *d_xlogx,
                               hipStream_t stream) {
    hipLaunchKernelGGL(k_boost_tables, dim3((n_entries + 255u) / 256u), dim3(256), 0, stream, n_entries,
                       (int)n_individs, d_frac, d_xlogx);
    return hipGetLastError();
}

// One workgroup = a TA x TB tile of SNP pairs; 256 threads, each owns a 2 x 2
// sub-tile.  For each group (cases, then controls) the rows' class vectors
// (aa, ab, bb -- or only aa, bb when no SNP of the table has a missing call)
// are staged through LDS CH words at a time and every thread accumulates the
// joint counts of its four pairs with AND + v_bcnt.
//   FAST = the reference's no-missing shortcut (table5.cpp:1066-1143): only
//   the four homozygote corners are counted, the rest follows from the margins.
//   PRE = the FP32 bound decides first; the FP64 measure runs only for pairs it cannot rule out.
template <bool FAST, bool PRE>
__global__ void __launch_bounds__(256) k_boost(BoostArgs a) {
    constexpr int NV = FAST ? 2 : 3;   // class vectors per row
    constexpr int NC = FAST ? 4 : 9;   // joint counts per pair and group
    __shared__ uint32_t sA[NV][TA][LD];
    __shared__ uint32_t sB[NV][TB][LD];

    const uint32_t a0 = a.a_begin + blockIdx.y * TA;  // first SNP i of the tile
    const uint32_t b0 = a.b_begin + blockIdx.x * TB;  // first SNP j
    const uint32_t a_end = a.a_begin + a.n_a, b_end = a.b_begin + a.n_b;
    // computeBoost only visits i < j (:396-400): tiles wholly on or below the diagonal do nothing
    if (a0 + 1u >= min(b0 + (uint32_t)TB, b_end)) return;

    const uint32_t t = threadIdx.x;
    const uint32_t ti = t >> 4, tj = t & 15u;  // rows 2ti, 2ti+1 of A; 2tj, 2tj+1 of B

    uint32_t cnt[2][4][NC];  // [group][pair][cell]
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int c = 0; c < NC; ++c) cnt[g][p][c] = 0;

    // Staging: thread t carries words 4q .. 4q+3 (q = t & 7) of A-row t >> 3 and of B-row t >> 3, both
    // planes, as four 16-byte loads.  The loads of chunk k+1 are issued before chunk k is consumed and
    // land in registers while the popcounts run; rows past the block end repeat its last row (their
    // pairs are never reported) and the planes are zero-padded to whole chunks (gw_capi.cpp), so no
    // load is conditional.
    static_assert(TA == 32 && TB == 32 && CH == 32, "staging map: 256 threads x (A row, B row) x 4 words");
    const uint32_t srow = t >> 3, sq = t & 7u;
    const uint32_t *rowA = a.cc + (uint64_t)min(a0 + srow, a_end - 1u) * a.cc_stride_w + 4u * sq;
    const uint32_t *rowB = a.cc + (uint64_t)min(b0 + srow, b_end - 1u) * a.cc_stride_w + 4u * sq;
    u32x4 nA0, nA1, nB0, nB1;  // next chunk: plane 0 / plane 1 of the A row and of the B row
    auto fetch = [&](uint32_t off, uint32_t wn, uint32_t w0) {
        nA0 = *reinterpret_cast<const u32x4 *>(rowA + off + w0);
        nA1 = *reinterpret_cast<const u32x4 *>(rowA + off + wn + w0);
        nB0 = *reinterpret_cast<const u32x4 *>(rowB + off + w0);
        nB1 = *reinterpret_cast<const u32x4 *>(rowB + off + wn + w0);
    };
    auto stash = [&]() {  // class vectors derived on the way in
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t pa0 = nA0[e], pa1 = nA1[e], pb0 = nB0[e], pb1 = nB1[e];
            sA[0][srow][4u * sq + e] = pa0 & ~pa1;       // aa
            sA[NV - 1][srow][4u * sq + e] = pa0 & pa1;   // bb
            sB[0][srow][4u * sq + e] = pb0 & ~pb1;
            sB[NV - 1][srow][4u * sq + e] = pb0 & pb1;
            if (!FAST) {
                sA[1][srow][4u * sq + e] = pa1 & ~pa0;   // ab
                sB[1][srow][4u * sq + e] = pb1 & ~pb0;
            }
        }
    };
    fetch(0u, a.wc, 0u);

#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const uint32_t wn = g == 0 ? a.wc : a.wt;              // words per plane of this group (x CH)
        const uint32_t off = g == 0 ? 0u : 2u * a.wc;          // plane 0 of the group inside the row
        for (uint32_t w0 = 0; w0 < wn; w0 += CH) {
            __syncthreads();
            stash();
            __syncthreads();
            if (w0 + CH < wn) fetch(off, wn, w0 + CH);
            else if (g == 0) fetch(2u * a.wc, a.wt, 0u);
#pragma unroll 4
            for (int w = 0; w < CH; ++w) {
                uint32_t x[2][NV], y[2][NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    x[0][v] = sA[v][2 * ti][w];
                    x[1][v] = sA[v][2 * ti + 1][w];
                    y[0][v] = sB[v][2 * tj][w];
                    y[1][v] = sB[v][2 * tj + 1][w];
                }
#pragma unroll
                for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                    for (int pj = 0; pj < 2; ++pj)
#pragma unroll
                        for (int va = 0; va < NV; ++va)
#pragma unroll
                            for (int vb = 0; vb < NV; ++vb)
                                cnt[g][2 * pi + pj][NV * va + vb] += __popc(x[pi][va] & y[pj][vb]);
            }
        }
    }

    // per pair: tables -> measure -> outputs
    unsigned long long n_eval = 0;
    double vmax = -1.0e300, vmin = 1.0e300;
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
#pragma unroll
        for (int pj = 0; pj < 2; ++pj) {
            const uint32_t i = a0 + 2 * ti + pi, j = b0 + 2 * tj + pj;
            if (i >= a_end || j >= b_end) continue;
            double measure = __longlong_as_double(0x7FF8000000000000ll);  // NaN: not a visited pair
            if (i < j) {
                const SnpMargin m1 = a.mar[i], m2 = a.mar[j];
                uint32_t ca[9], co[9];
                if (FAST) {
                    // table5.cpp:1088-1106 / 1125-1143: corners counted, the rest from the margins
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const uint32_t *c4 = cnt[g][2 * pi + pj];
                        const uint32_t *f1 = g == 0 ? m1.cases : m1.ctrls;
                        const uint32_t *f2 = g == 0 ? m2.cases : m2.ctrls;
                        uint32_t *o = g == 0 ? ca : co;
                        o[0] = c4[0]; o[2] = c4[1]; o[6] = c4[2]; o[8] = c4[3];
                        o[1] = f1[0] - o[0] - o[2];        // AA_Bb
                        o[7] = f1[2] - o[8] - o[6];        // aa_Bb
                        o[3] = f2[0] - o[0] - o[6];        // Aa_BB
                        o[5] = f2[2] - o[2] - o[8];        // Aa_bb
                        o[4] = f2[1] - o[1] - o[7];        // Aa_Bb
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 9; ++c) {
                        ca[c] = cnt[0][2 * pi + pj][c];
                        co[c] = cnt[1][2 * pi + pj][c];
                    }
                }
                ++n_eval;
                float bound = 0.0f, err = 0.0f;
                if (PRE) {
                    bound = boost_bound(ca, co, a.marf[i], a.marf[j], (float)a.n_individs,
                                        1.0f / (float)a.n_individs, err);
                    if (a.dense) {  // diagnostic form: the bound itself, or its error bound
                        a.dense[(size_t)(i - a.a_begin) * a.n_b + (j - a.b_begin)] = (double)(a.dense_err ? err : bound);
                        continue;
                    }
                    if (bound_rules_out(bound, err, a.bound_threshold)) continue;
                }
                measure = boost_measure(ca, co, m1, m2, a.n_individs, a.frac, a.xlogx);
                if (PRE) check_bound(measure, bound, err, a.tally);
                if (!PRE) {
                    vmax = fmax(vmax, measure);
                    vmin = fmin(vmin, measure);
                }
                if (measure > a.threshold) {  // :482-484
                    const unsigned long long slot = atomicAdd(&a.tally[0], 1ull);
                    if (slot < a.max_hits) {
                        BoostHit h;
                        h.snp_a = i;
                        h.snp_b = j;
                        h.measure = measure;
                        a.hits[slot] = h;
                    }
                }
            }
            if (a.dense) a.dense[(size_t)(i - a.a_begin) * a.n_b + (j - a.b_begin)] = measure;
        }
    }
    // tile tallies: wave reduce, one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n_eval += __shfl_xor(n_eval, off, 64);
        vmax = fmax(vmax, __shfl_xor(vmax, off, 64));
        vmin = fmin(vmin, __shfl_xor(vmin, off, 64));
    }
    if ((t & 63u) == 0 && n_eval) {
        atomicAdd(&a.tally[1], n_eval);
        if (!PRE) {
            atomicMax(&a.tally[2], ordered(vmax));
            atomicMin(&a.tally[3], ordered(vmin));
        }
    }
}

// ---------------------------------------------------------------------------
// Block-tiled copy of the compacted rows for k_boost_mfma:
//   unit (snp, chunk c, half h) = {plane0 words 4c+2h, 4c+2h+1, plane1 words 4c+2h, 4c+2h+1}
//   at 16-byte index (((snp >> 6) * nchunk + c) * 64 + (snp & 63)) * 2 + h,
// chunks of the case planes first, then the control planes.
// ---------------------------------------------------------------------------
__global__ void k_retile_cc(const uint32_t *cc, uint64_t cc_stride_w, uint32_t wc, uint32_t wt, uint32_t n_snps,
                            u32x4 *cct) {
    const uint32_t nchunk_c = wc / 4u, nchunk = nchunk_c + wt / 4u;
    const uint64_t total = (uint64_t)n_snps * nchunk * 2u;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t h = (uint32_t)(idx & 1u);
        const uint64_t rc = idx >> 1;
        const uint32_t c = (uint32_t)(rc % nchunk), snp = (uint32_t)(rc / nchunk);
        const bool ctrl = c >= nchunk_c;
        const uint32_t off = (ctrl ? 2u * wc + (c - nchunk_c) * 4u : c * 4u) + 2u * h;
        const uint32_t wn = ctrl ? wt : wc;
        const uint32_t *src = cc + (uint64_t)snp * cc_stride_w + off;
        u32x4 v;
        v[0] = src[0]; v[1] = src[1]; v[2] = src[wn]; v[3] = src[wn + 1u];
        cct[(((uint64_t)(snp >> 6) * nchunk + c) * 64u + (snp & 63u)) * 2u + h] = v;
    }
}

hipError_t launch_retile_cc(const uint32_t *d_cc, uint64_t cc_stride_w, uint32_t wc, uint32_t wt, uint32_t n_snps,
                            u32x4 *d_cct, hipStream_t stream) {
    if (n_snps == 0) return hipSuccess;
    hipLaunchKernelGGL(k_retile_cc, dim3(4096), dim3(256), 0, stream, d_cc, cc_stride_w, wc, wt, n_snps, d_cct);
    return hipGetLastError();
}

hipError_t launch_boost_valu(const BoostArgs &a, bool fast, bool pre, hipStream_t stream) {
    const dim3 grid((a.n_b + TB - 1) / TB, (a.n_a + TA - 1) / TA);
    if (pre) {
        if (fast) hipLaunchKernelGGL((k_boost<true, true>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_boost<false, true>), grid, dim3(256), 0, stream, a);
    } else {
        if (fast) hipLaunchKernelGGL((k_boost<true, false>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_boost<false, false>), grid, dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

// The constants of boost_bound_quad (gw_boost_dev.h) for a cohort of n with n_cases cases, in double, rounded so that
// the bound stays an upper bound: p = float(n_cases / n) is the expansion point BY DEFINITION (so it carries no error),
// c_lo = max(g(0), g(p)), c_hi = max(g(1), g(p)) with a 1e-6 allowance, k0 = -n h(p) - h'(p)(n_cases - p n) rounded up;
// c_m / c_d as floats such that c_m - c_d >= c_lo and c_m + c_d >= c_hi.  c_m == 0 (no first level) for a cohort with
// one group only.
QuadBound quad_bound_constants(uint32_t n_cases, uint32_t n) {
    QuadBound q{};
    if (n_cases == 0 || n_cases >= n) return q;
    const float pf = (float)((double)n_cases / (double)n);
    if (!(pf > 0.0f && pf < 1.0f)) return q;
    const double p = (double)pf, h0 = -(p * std::log2(p) + (1.0 - p) * std::log2(1.0 - p)), h1 = std::log2((1.0 - p) / p);
    const double g0 = (h0 - h1 * p) / (p * p), g1 = (h0 + h1 * (1.0 - p)) / ((1.0 - p) * (1.0 - p));
    const double gp = 1.0 / (2.0 * std::log(2.0) * p * (1.0 - p));
    const double c_lo = std::max(g0, gp) * (1.0 + 1.0e-6), c_hi = std::max(g1, gp) * (1.0 + 1.0e-6);
    const double k0 = -(double)n * h0 - h1 * ((double)n_cases - p * (double)n);
    q.p = pf;
    const float u = 5.9604645e-8f;  // 2^-24: the products below are exact
    q.n48 = u * std::nextafter(48.0f * (float)n, INFINITY);
    q.n30 = u * std::nextafter(30.0f * (float)n, INFINITY);
    q.n1 = (float)((double)n * 1.4426950408889634);
    q.n1t = (float)((double)n * 1.4426950408889634 * (double)n / (double)(n - n_cases));
    // tables with missing calls (boost_bound_quad9_f): the linear part of the quadratic is no constant there
    q.kx = (float)(-h1);
    q.ks = (float)(p * h1 - h0);
    {
        const double big_h = 2.0 * std::fabs(h1) + 1.0;
        q.nk = u * std::nextafter((float)((double)n * (10.0 * big_h + 26.0)), INFINITY);
        q.n22 = u * std::nextafter(22.0f * (float)n, INFINITY);
    }
    if (std::fabs(c_hi - c_lo) <= 0.02 * std::max(c_hi, c_lo)) {  // balanced cohort: one constant, the larger (LV = 1)
        q.c_d = 0.0f;
        q.c_m = std::nextafter((float)std::max(c_lo, c_hi), INFINITY);
        q.k0 = std::nextafter((float)k0, INFINITY);
        q.c27 = u * (27.0f * q.c_m * 1.000001f);
        // the sum as  sum x^2 / s  (gw_boost_dev.h): its constant / linear remainder with the float c and p actually used, in FP64,
        // rounded up; the rounding of a sum of magnitude p X instead of q is the constant u 24 c n, added to the error
        // bound's constant terms of both forms
        const double cm = (double)q.c_m;
        q.k0x = std::nextafter((float)(k0 - cm * (2.0 * p * (double)n_cases - p * p * (double)n)), INFINITY);
        q.kx9 = std::nextafter((float)(-h1 - 2.0 * p * cm), INFINITY);
        q.ks9 = std::nextafter((float)(p * h1 - h0 + cm * p * p), INFINITY);
        q.cx = u * std::nextafter(24.0f * q.c_m * (float)n, INFINITY);
        q.n48 += q.cx;
        q.nk += q.cx;
        return q;
    }
    q.c_d = (float)(0.5 * (c_hi - c_lo));
    // c_m - c_d >= c_lo and c_m + c_d >= c_hi for the float c_d actually used
    q.c_m = std::nextafter((float)std::max(c_lo + (double)q.c_d, c_hi - (double)q.c_d), INFINITY);
    q.k0 = std::nextafter((float)k0, INFINITY);
    q.c27 = u * (27.0f * (q.c_m + std::fabs(q.c_d)) * 1.000001f);
    return q;
}

hipError_t launch_boost(const BoostLaunch &l) {
    if (l.n_a == 0 || l.n_b == 0) return hipSuccess;
    // a screen whose candidates might not fit the list even at its cap (all pairs of more than ~130 000 SNPs in one call): row
    // block by row block through the same list -- hits, tally and the lost-candidates count accumulate across the launches
    if (l.prefilter && !l.d_dense && l.candidates.records && l.candidates.rows_per_launch && l.n_a > l.candidates.rows_per_launch) {
        for (uint32_t r0 = 0; r0 < l.n_a; r0 += l.candidates.rows_per_launch) {
            BoostLaunch part = l;
            part.a_begin = l.a_begin + r0;
            part.n_a = std::min(l.candidates.rows_per_launch, l.n_a - r0);
            part.candidates.rows_per_launch = 0;
            const hipError_t e = launch_boost(part);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    BoostArgs a;
    a.cc = l.d_cc; a.cc_stride_w = l.cc_stride_w; a.wc = l.wc; a.wt = l.wt; a.mar = l.d_mar; a.frac = l.d_frac; a.xlogx = l.d_xlogx;
    a.a_begin = l.a_begin; a.n_a = l.n_a; a.b_begin = l.b_begin; a.n_b = l.n_b;
    a.n_individs = (int)l.n_individs; a.threshold = l.threshold;
    a.hits = l.d_hits; a.max_hits = l.max_hits; a.tally = l.d_tally; a.dense = l.d_dense;
    a.marf = l.d_marf; a.bound_threshold = l.bound_threshold; a.dense_err = l.dense_err ? 1 : 0;
    a.cct = l.matrix_cores ? l.d_cct : nullptr;
    a.lg_n = (float)(std::log2((double)(l.n_individs ? l.n_individs : 1u)) * 1.000001) + 1.0e-5f;
    a.quad = QuadBound{};
    a.fp4_scale[0] = 126; a.fp4_scale[1] = 127; a.fp4_scale[2] = 128;
    a.cand = l.candidates.records; a.cand_count = l.candidates.count; a.cand_cap = l.candidates.capacity;
    const bool fast = l.no_missing_call;
    const bool pre = l.prefilter && l.d_marf;
    const bool exact_in_f32 = l.n_individs < (1u << 24);  // the FP4 kernels count in FP32 accumulators
    const bool small_case_group = l.wc * 32u < 65536u && l.matrix_cores != 2;  // case counts fit uint16
    // the instantiation the launch goes to, spelled as rocprofv3's kernel trace prints it (gw_last_boost_kernel)
    auto note = [&](const char *fmt, const char *x, const char *y) {
        if (l.kernel_name) std::snprintf(l.kernel_name, 96, fmt, x, y, l.d_dense ? "true" : "false");
    };
    const char *tp = pre ? "true" : "false", *tfast = fast ? "true" : "false";
    // the first level pays while it stops nearly every pair: it is loose by ~0.4 x a chi^2 with 8 degrees of freedom, so
    // below a threshold of ~25 more than one wave in five would run both levels (diagnostic form: whenever asked)
    // (tables with missing calls: the nine-cell form of the first level, on the FP4 kernel that parks its case counts)
    const bool first = pre && l.first_level && (l.threshold >= 25.0 || l.d_dense);
    // the pair tables counted on FP4 operands (8 samples per operand dword, 64 per MFMA): four corners / nine cells,
    // one group at a time with the case counts parked in LDS as uint16
    if (a.cct && l.d_marf && small_case_group && l.fp4 && exact_in_f32) {
        if (first) a.quad = quad_bound_constants(l.n_cases, l.n_individs);
        note(fast ? "gw::k_boost_mfma_fp4<%s, 2, 2, %s, %s>" : "gw::k_boost_mfma_fp4<%s, 3, 1, %s, %s>", tp,
             a.quad.c_m > 0.0f ? (a.quad.c_d != 0.0f ? "2" : "1") : "0");
        return launch_boost_mfma_fp4(a, pre, !fast, l.compute_units, l.stream);
    }
    if (a.cct && l.d_marf && small_case_group) {  // the same on int8 operands (exact in int32 for any cohort)
        note(fast ? "gw::k_boost_mfma_seq<%s, 2, 2, false>" : "gw::k_boost_mfma_seq<%s, 3, 1, false>", tp, "");
        return fast ? launch_boost_mfma_seq4(a, pre, l.compute_units, l.stream) : launch_boost_mfma_seq9(a, pre, l.compute_units, l.stream);
    }
    // tables without a missing call, case groups of any size: the FP4 kernel with both groups' accumulators in registers
    if (a.cct && l.d_marf && fast && l.fp4 && exact_in_f32) {
        if (first) a.quad = quad_bound_constants(l.n_cases, l.n_individs);  // (fast: four corners)
        note("gw::k_boost_mfma_fp4_both<%s, %s, %s>", tp, a.quad.c_m > 0.0f ? (a.quad.c_d != 0.0f ? "2" : "1") : "0");
        return launch_boost_mfma_fp4_both(a, pre, l.compute_units, l.stream);
    }
    if (a.cct && fast) {  // any group size, 2 waves per SIMD
        note("gw::k_boost_mfma<%s>", tp, "");
        return launch_boost_mfma_both4(a, pre, l.compute_units, l.stream);
    }
    if (a.cct && l.d_marf) {  // any group size, 1 wave per SIMD
        note("gw::k_boost_mfma_seq<%s, 3, 1, true>", tp, "");
        return launch_boost_mfma_both9(a, pre, l.compute_units, l.stream);
    }
    note("gw::k_boost<%s, %s>", tfast, tp);
    return launch_boost_valu(a, fast, pre, l.stream);
}

}  // namespace gw
