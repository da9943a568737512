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
//   k_boost       the O(M^2) loop of computeBoost (:393-486): margin-assisted
//                 case/control 3x3 tables (table5.cpp:989-1150) + the
//                 interaction measure (:424-470) + the threshold list (:482)
//
// Unlike the per-SNP scan this path is ALU-bound (rows are re-used M times
// from LDS): 9 (or 4) AND+popcount per 32 samples per pair, then ~20 FP64
// logarithms per pair.  Still no MFMA: AND-popcount has no matrix-core form
// on CDNA4 short of expanding every bit to a 4-bit element.
#include <type_traits>

#include "gw_kernels.h"

namespace gw {

// compile-time loop: the body sees its index as a constant
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---------------------------------------------------------------------------
// selectCaseControl: gather the members' bits into dense planes.
// src_idx[k] = sample index of the k-th member (cases first: [0, n_case), then
// controls: [case_pad_bits, case_pad_bits + n_ctrl)); 0xFFFFFFFF = pad bit.
// One workgroup per SNP row, the row's two planes staged in LDS.
// Compact row layout (32-bit words): [case p0 | case p1 | ctrl p0 | ctrl p1],
// case planes wc words, control planes wt words.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_select_cc(const u32x4 *table, uint64_t row_stride_u,
                                                   uint32_t plane_stride_u, uint32_t units, uint32_t n_rows,
                                                   const uint32_t *src_idx, uint32_t wc, uint32_t wt,
                                                   uint32_t *cc, uint64_t cc_stride_w) {
    extern __shared__ uint32_t s_row[];  // [2][units*4]
    const uint32_t words = units * 4u;
    for (uint32_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const uint32_t *p0 = reinterpret_cast<const uint32_t *>(table + (uint64_t)r * row_stride_u);
        const uint32_t *p1 = p0 + (uint64_t)plane_stride_u * 4u;
        __syncthreads();
        for (uint32_t w = threadIdx.x; w < words; w += blockDim.x) {
            s_row[w] = p0[w];
            s_row[words + w] = p1[w];
        }
        __syncthreads();
        uint32_t *out = cc + (uint64_t)r * cc_stride_w;
        const uint32_t total = wc + wt;  // output words per plane
        for (uint32_t k = threadIdx.x; k < total; k += blockDim.x) {
            uint32_t lo = 0, hi = 0;
#pragma unroll 8
            for (uint32_t b = 0; b < 32u; ++b) {
                const uint32_t s = src_idx[k * 32u + b];
                if (s != 0xFFFFFFFFu) {
                    lo |= ((s_row[s >> 5] >> (s & 31u)) & 1u) << b;
                    hi |= ((s_row[words + (s >> 5)] >> (s & 31u)) & 1u) << b;
                }
            }
            if (k < wc) { out[k] = lo; out[wc + k] = hi; }
            else { out[2u * wc + (k - wc)] = lo; out[2u * wc + wt + (k - wc)] = hi; }
        }
    }
}

hipError_t launch_select_cc(const TableView &tv, const uint32_t *d_src_idx, uint32_t wc, uint32_t wt,
                            uint32_t *d_cc, uint64_t cc_stride_w, hipStream_t stream) {
    if (tv.n_snps == 0) return hipSuccess;
    const uint32_t grid = (uint32_t)(tv.n_snps < 8192 ? tv.n_snps : 8192);
    hipLaunchKernelGGL(k_select_cc, dim3(grid), dim3(256), (size_t)tv.units * 32u, stream, tv.base,
                       tv.row_stride_u, tv.plane_stride_u, tv.units, (uint32_t)tv.n_snps, d_src_idx, wc, wt,
                       d_cc, cc_stride_w);
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
// the pair kernel
// ---------------------------------------------------------------------------
namespace {
constexpr int TA = 32;  // A-block rows (SNP i) per workgroup tile
constexpr int TB = 32;  // B-block rows (SNP j)
constexpr int CH = 32;  // 32-bit words of every row staged per step (1024 samples)
constexpr int LD = CH + 1;  // padded LDS row length

// order-preserving map double -> uint64 for atomicMax / atomicMin
__device__ __forceinline__ unsigned long long ordered(double v) {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
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

// computeBoost's measure for one pair from the two 3x3 tables
// (epistasis_func.cpp:424-470); operation order as the oracle's gwo_boost_measure
__device__ __forceinline__ double boost_measure(const uint32_t (&ca)[9], const uint32_t (&co)[9],
                                                const SnpMargin &m1, const SnpMargin &m2, int n,
                                                const double *__restrict__ frac,
                                                const double *__restrict__ xlogx) {
    double tao = 0.0, im = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const uint32_t x = ca[3 * a + b], y = co[3 * a + b];
            const double pab = (double)(x + y) / (double)m2.margins[b];
            const double t2 = (pab) * (m2.pbc[b]) * (m1.pca[a]);
            const double t3 = (pab) * (m2.pbc[4 + b]) * (m1.pca[4 + a]);
            tao += t2 + t3;
            if (x > 0) {
                const double t1 = frac[x];  // (double)x / n
                im += xlogx[x];             // t1 * log(t1)
                if (t2 > 0) im += -t1 * log(t2);
            }
            if (y > 0) {
                const double t1 = frac[y];
                im += xlogx[y];
                if (t3 > 0) im += -t1 * log(t3);
            }
        }
    }
    return (im + log(tao)) * n * 2.0;
}

// The same quantity in single precision, arranged for one v_log_f32 per cell and group:
//   measure = 2 * sum_cells [ x ln(r_case) + y ln(r_ctrl) ] + 2 n ln(tao),
//   r = (k/n) / (pab * pbc * pca) = k / (x+y) / n * qb[b] * qa[a]
// Its error against the FP64 measure stays far below the slack the caller subtracts from the
// threshold (tests/test_gpu_boost.py measures it), so "bound < threshold - slack" proves that the
// pair cannot pass; anything else (including NaN) is decided by the FP64 measure.
__device__ __forceinline__ float boost_bound(const uint32_t (&ca)[9], const uint32_t (&co)[9],
                                             const SnpMarginF &f1, const SnpMarginF &f2, float n, float inv_n) {
    float s = 0.0f, tao = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const uint32_t x = ca[3 * a + b], y = co[3 * a + b];
            const float fx = (float)x, fy = (float)y, fs = (float)(x + y);
            const float rs = __builtin_amdgcn_rcpf(fs) * inv_n;
            tao += fs * (f2.rb[b] * f1.pa[a] + f2.rb[3 + b] * f1.pa[3 + a]);
            const float lx = fx * __log2f(fx * rs * (f2.qb[b] * f1.qa[a]));
            const float ly = fy * __log2f(fy * rs * (f2.qb[3 + b] * f1.qa[3 + a]));
            s += x ? lx : 0.0f;
            s += y ? ly : 0.0f;
        }
    }
    return 1.3862943611198906f * (s + n * __log2f(tao));  // 2 ln 2
}

struct BoostArgs {
    const uint32_t *cc;
    uint64_t cc_stride_w;
    uint32_t wc, wt;
    const SnpMargin *mar;
    const double *frac, *xlogx;  // k/n and (k/n) log(k/n), k = 0 .. n
    uint32_t a_begin, n_a, b_begin, n_b;
    int n_individs;
    double threshold;
    BoostHit *hits;
    unsigned long long max_hits;
    unsigned long long *tally;  // [0] hits, [1] pairs evaluated, [2] ordered max, [3] ordered min
    double *dense;              // n_a x n_b, or null
    const SnpMarginF *marf;     // PRE only
    const u32x4 *cct;           // k_boost_mfma: the compacted rows in 64-SNP blocks, chunk-major (k_retile_cc)
    float bound_threshold;      // PRE: pairs whose FP32 bound is below this are not hits
};

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
                if (PRE) {
                    const float bound = boost_bound(ca, co, a.marf[i], a.marf[j], (float)a.n_individs,
                                                    1.0f / (float)a.n_individs);
                    if (a.dense) {  // diagnostic form: the bound itself
                        a.dense[(size_t)(i - a.a_begin) * a.n_b + (j - a.b_begin)] = (double)bound;
                        continue;
                    }
                    if (bound < a.bound_threshold) continue;  // false for NaN
                }
                measure = boost_measure(ca, co, m1, m2, a.n_individs, a.frac, a.xlogx);
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

// ---------------------------------------------------------------------------
// Matrix-core form of the same tile (tables without a missing call).
//
// The four counted corners of a pair are dot products of 0/1 class vectors over
// the samples of a group -- a binary GEMM, and unlike the per-SNP scan it re-uses
// every row M times, so it is compute- not HBM-bound.  v_bcnt_u32_b32 issues at
// half rate on gfx950 (tools/valu_rates.hip: 4.1 cycles; AND+BCNT 6.25 cycles per
// 64 x 32 sample pairs = 330 bit-MACs per cycle and SIMD); v_mfma_i32_32x32x32_i8
// does 1024 MACs per cycle and SIMD with exact int32 accumulation.  Price: every
// bit must become a byte.
//
// Wave = 32 x 32 SNP pairs: 4 corner products x 2 groups = 8 accumulator tiles
// (128 registers); workgroup = 2 x 2 waves that share nothing but the epilogue's
// margins.  No LDS and no barrier in the main loop: a first version that expanded
// each chunk once per workgroup into LDS was bound by LDS *stores* (ds_write_b128
// moves 79 B/clk/CU, the expanded operands of a 64 x 64 tile need 64 B/clk at the
// MFMA rate) and reached 48 % of the matrix peak.  Here every lane loads the 64
// sample bits it needs straight from the block-tiled table (k_retile_cc: lane
// (r, h) = row r, half h of the chunk, one 16-byte load, 1 KiB contiguous per wave)
// and expands them in registers.  The K order inside a chunk is ours to choose as
// long as A and B agree: lane half h owns samples 64h .. 64h+63, step s takes
// 16s .. 16s+15 of them.
//
// Expansion in ~1.3 VALU per 4 samples: the K order is free, so an operand dword
// takes one sample from each byte of the word -- a single v_and_b32 -- and carries
// a power-of-two weight that its partner compensates (expand_a / expand_b): every
// product is 64 and the accumulators hold 64 x count (exact; >> 6).
// ---------------------------------------------------------------------------
namespace {
constexpr int MT = 64;  // SNPs per workgroup tile side
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// Operand dword k (k = 0..7) of a 32-sample word = samples {k, 8+k, 16+k, 24+k}, one per byte: a single
// v_and_b32 with 0x01010101 << k isolates them, leaving weight 2^k in every byte.  B takes the same
// samples from the word with the bits of each byte reversed and shifted down by one, weight 2^(6-k):
// every product is 64.  Bit 7 of a byte would be the int8 sign, so k = 7 is done with a shift
// (A weight 1, B weight 64).  One fragment (step) = dwords 4hk .. 4hk+3 of one word.
__device__ __forceinline__ i32x4 expand_a(uint32_t w, int hk) {
    i32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 4 * hk + e;
        v[e] = k < 7 ? (int)(w & (0x01010101u << k)) : (int)((w >> 7) & 0x01010101u);
    }
    return v;
}
// rb = the B word with the bits of each byte reversed, >> 1 (bit 6-k of byte m = sample 8m+k, k <= 6)
__device__ __forceinline__ i32x4 expand_b(uint32_t w, uint32_t rb, int hk) {
    i32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 4 * hk + e;
        v[e] = k < 7 ? (int)(rb & (0x01010101u << (6 - k))) : (int)((w >> 1) & 0x40404040u);
    }
    return v;
}
}  // namespace

template <bool PRE>
__global__ void __launch_bounds__(256, 2) k_boost_mfma(BoostArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_mar[MT * 32];  // the A rows' margins (epilogue)

    const uint32_t a0 = a.a_begin + blockIdx.y * MT, b0 = a.b_begin + blockIdx.x * MT;
    const uint32_t a_end = a.a_begin + a.n_a, b_end = a.b_begin + a.n_b;
    if (a0 + 1u >= min(b0 + (uint32_t)MT, b_end)) return;  // nothing above the diagonal in this tile

    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t wy = wave >> 1, wx = wave & 1u;
    const uint32_t fr = lane & 31u, fh = lane >> 5;

    // this lane's A row and B row in the block-tiled table (rows past the block end repeat its last
    // row; their pairs are never reported)
    const uint32_t nchunk_c = a.wc / 4u, nchunk = nchunk_c + a.wt / 4u;  // planes are whole multiples of 8 chunks
    const uint32_t snp_a = min(a0 + wy * 32u + fr, a_end - 1u), snp_b = min(b0 + wx * 32u + fr, b_end - 1u);
    const u32x4 *ga = a.cct + ((uint64_t)(snp_a >> 6) * nchunk * 64u + (snp_a & 63u)) * 2u + fh;
    const u32x4 *gb = a.cct + ((uint64_t)(snp_b >> 6) * nchunk * 64u + (snp_b & 63u)) * 2u + fh;

    i32x16 acc[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][c][r] = 0;

    // the next RING chunks' bits wait in registers (the table of a large job sits in L2 / Infinity
    // Cache, 1-2 us away; a chunk takes ~0.25 us)
    constexpr int RING = 4;
    u32x4 ua[RING], ub[RING];  // {plane0 w0, plane0 w1, plane1 w0, plane1 w1} of the lane's 64 samples
#pragma unroll
    for (int k = 0; k < RING; ++k) {
        ua[k] = ga[(uint64_t)k * 128u];
        ub[k] = gb[(uint64_t)k * 128u];
    }
    const uint32_t last_chunk = nchunk - 1u;

    struct Frag { i32x4 xa, xb, ya, yb; };
    // one chunk = 4 steps of 16 samples per lane half: step s = half (s & 1) of word (s >> 1)
    auto chunk = [&](const u32x4 &va, const u32x4 &vb, i32x16 (&d)[4]) {
        uint32_t caa[2], cbb[2], baa[2], bbb[2], raa[2], rbb[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            caa[w] = va[w] & ~va[2 + w];  // A row: aa, bb
            cbb[w] = va[w] & va[2 + w];
            baa[w] = vb[w] & ~vb[2 + w];  // B row
            bbb[w] = vb[w] & vb[2 + w];
            raa[w] = __builtin_bswap32(__builtin_bitreverse32(baa[w])) >> 1;
            rbb[w] = __builtin_bswap32(__builtin_bitreverse32(bbb[w])) >> 1;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            Frag f;
            f.xa = expand_a(caa[s >> 1], s & 1);
            f.xb = expand_a(cbb[s >> 1], s & 1);
            f.ya = expand_b(baa[s >> 1], raa[s >> 1], s & 1);
            f.yb = expand_b(bbb[s >> 1], rbb[s >> 1], s & 1);
            d[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.xa, f.ya, d[0], 0, 0, 0);
            d[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.xa, f.yb, d[1], 0, 0, 0);
            d[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.xb, f.ya, d[2], 0, 0, 0);
            d[3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.xb, f.yb, d[3], 0, 0, 0);
            // issue order: an MFMA holds the matrix pipe for 32 cycles but the issue port for 8; the
            // ~6 VALU of expansion per MFMA go between them, not in front
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // 6 VALU
            }
        }
    };
    auto round = [&](uint32_t c0, i32x16 (&d)[4]) {
        static_for<0, RING>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            chunk(ua[k], ub[k], d);
            const uint64_t nxt = (uint64_t)min(c0 + k + RING, last_chunk) * 128u;  // past the end: re-read, unused
            ua[k] = ga[nxt];
            ub[k] = gb[nxt];
        });
    };
    for (uint32_t c0 = 0; c0 < nchunk_c; c0 += RING) round(c0, acc[0]);
    for (uint32_t c0 = nchunk_c; c0 < nchunk; c0 += RING) round(c0, acc[1]);

    // ---- per pair: corners -> tables -> bound / measure.  C layout of the 32x32 MFMA:
    // column (B index) = lane & 31, row (A index) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const uint32_t j = b0 + wx * 32u + fr, i_base = a0 + wy * 32u + 4u * fh;
    const bool j_ok = j < b_end;
    const SnpMargin *pm2 = a.mar + (j_ok ? j : b_end - 1u);
    uint32_t f2c[3], f2t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { f2c[k] = pm2->cases[k]; f2t[k] = pm2->ctrls[k]; }
    SnpMarginF mf2;
    if (PRE) mf2 = a.marf[j_ok ? j : b_end - 1u];
    const float nf = (float)a.n_individs, inv_n = 1.0f / (float)a.n_individs;

    // The 64 A rows' margins go through LDS: every lane needs 16 of them.
    // Per row 32 dwords: cases[3], ctrls[3], pad[2], then the 24 floats of SnpMarginF.
    for (uint32_t idx = t; idx < (uint32_t)MT * 32u; idx += 256u) {
        const uint32_t row = idx >> 5, k = idx & 31u;
        const uint32_t snp = min(a0 + row, a_end - 1u);
        uint32_t v = 0;
        if (k < 3u) v = a.mar[snp].cases[k];
        else if (k < 6u) v = a.mar[snp].ctrls[k - 3u];
        else if (k >= 8u && PRE) v = reinterpret_cast<const uint32_t *>(a.marf + snp)[k - 8u];
        s_mar[idx] = v;
    }
    __syncthreads();

    auto tables = [&](const uint32_t (&c4)[2][4], const uint32_t *f1c, const uint32_t *f1t, uint32_t (&ca)[9],
                      uint32_t (&co)[9]) {
        // table5.cpp:1088-1106 / 1125-1143: corners counted, the rest from the margins
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint32_t *f1 = g == 0 ? f1c : f1t;
            const uint32_t *f2 = g == 0 ? f2c : f2t;
            uint32_t *o = g == 0 ? ca : co;
            o[0] = c4[g][0]; o[2] = c4[g][1]; o[6] = c4[g][2]; o[8] = c4[g][3];
            o[1] = f1[0] - o[0] - o[2];
            o[7] = f1[2] - o[8] - o[6];
            o[3] = f2[0] - o[0] - o[6];
            o[5] = f2[2] - o[2] - o[8];
            o[4] = f2[1] - o[1] - o[7];
        }
    };

    uint32_t need = 0;  // bit r: pair of accumulator register r still wants its FP64 measure
    unsigned long long n_eval = 0;
    static_for<0, 16>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const uint32_t i = i_base + (r & 3) + 8 * (r >> 2);
        if (i < a_end && j_ok && i < j) {
            ++n_eval;
            if (PRE) {
                const uint32_t *row = s_mar + (i - a0) * 32u;
                uint32_t f1[8];
                *reinterpret_cast<u32x4 *>(f1) = *reinterpret_cast<const u32x4 *>(row);
                *reinterpret_cast<u32x4 *>(f1 + 4) = *reinterpret_cast<const u32x4 *>(row + 4);
                SnpMarginF mf1;
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    *reinterpret_cast<u32x4 *>(reinterpret_cast<uint32_t *>(&mf1) + 4 * q) =
                        *reinterpret_cast<const u32x4 *>(row + 8 + 4 * q);
                uint32_t c4[2][4], ca[9], co[9];
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int k = 0; k < 4; ++k) c4[g][k] = (uint32_t)acc[g][k][r] >> 6;  // every product was 64
                tables(c4, f1, f1 + 3, ca, co);
                const float bound = boost_bound(ca, co, mf1, mf2, nf, inv_n);
                if (a.dense) a.dense[(size_t)(i - a.a_begin) * a.n_b + (j - a.b_begin)] = (double)bound;
                else if (!(bound < a.bound_threshold)) need |= 1u << r;
            } else {
                need |= 1u << r;
            }
        }
    });

    double vmax = -1.0e300, vmin = 1.0e300;
    while (need) {  // FP64 measure, one accumulator register at a time (per lane its own)
        const int r = __ffs((int)need) - 1;
        need &= need - 1u;
        uint32_t c4[2][4];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t v = 0;
#pragma unroll
                for (int q = 0; q < 16; ++q) v = r == q ? (uint32_t)acc[g][k][q] : v;
                c4[g][k] = v >> 6;
            }
        const uint32_t i = i_base + (r & 3) + 8 * (r >> 2);
        const SnpMargin m1 = a.mar[i], m2 = *pm2;
        uint32_t ca[9], co[9];
        tables(c4, m1.cases, m1.ctrls, ca, co);
        const double measure = boost_measure(ca, co, m1, m2, a.n_individs, a.frac, a.xlogx);
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
        if (a.dense) a.dense[(size_t)(i - a.a_begin) * a.n_b + (j - a.b_begin)] = measure;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n_eval += __shfl_xor(n_eval, off, 64);
        if (!PRE) {
            vmax = fmax(vmax, __shfl_xor(vmax, off, 64));
            vmin = fmin(vmin, __shfl_xor(vmin, off, 64));
        }
    }
    if (lane == 0 && n_eval) {
        atomicAdd(&a.tally[1], n_eval);
        if (!PRE) {
            atomicMax(&a.tally[2], ordered(vmax));
            atomicMin(&a.tally[3], ordered(vmin));
        }
    }
}

// ---------------------------------------------------------------------------
// The same on tables WITH missing calls: all nine cells of both 3 x 3 tables are
// counted (table5.cpp:989-1064), nine products per pair and group.  Nine 32 x 32
// accumulator tiles per group would be 288 registers, so this form uses
// v_mfma_i32_16x16x64_i8 (4 registers per product): one wave = 16 A rows x 32 B
// rows (two 16-row B tiles) = 2 x 9 x 2 x 4 = 144 accumulators; workgroup =
// 2 x 2 waves = 32 x 64 pairs.  Lane (r, q) = row r of the tile, k-group q: it
// owns samples 32q .. 32q+31 of the chunk (one word per plane), two steps of 16.
// Operand expansion as in k_boost_mfma (expand_a / expand_b, products of 64).
// ---------------------------------------------------------------------------
namespace {
constexpr int M9A = 32, M9B = 64;  // workgroup tile: A rows x B rows
typedef int i32x4v __attribute__((ext_vector_type(4)));
}  // namespace

template <bool PRE>
__global__ void __launch_bounds__(256, 2) k_boost_mfma9(BoostArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_marf[(M9A + M9B) * 24];  // SnpMarginF of the tile's rows

    const uint32_t a0 = a.a_begin + blockIdx.y * M9A, b0 = a.b_begin + blockIdx.x * M9B;
    const uint32_t a_end = a.a_begin + a.n_a, b_end = a.b_begin + a.n_b;
    if (a0 + 1u >= min(b0 + (uint32_t)M9B, b_end)) return;  // nothing above the diagonal in this tile

    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t wy = wave >> 1, wx = wave & 1u;
    const uint32_t fr = lane & 15u, fq = lane >> 4;

    const uint32_t nchunk_c = a.wc / 4u, nchunk = nchunk_c + a.wt / 4u;
    auto unit = [&](uint32_t snp) {  // this lane's 16-byte unit of chunk 0: half fq >> 1 of the row
        return a.cct + ((uint64_t)(snp >> 6) * nchunk * 64u + (snp & 63u)) * 2u + (fq >> 1);
    };
    const u32x4 *ga = unit(min(a0 + wy * 16u + fr, a_end - 1u));
    const u32x4 *gb0 = unit(min(b0 + wx * 32u + fr, b_end - 1u));
    const u32x4 *gb1 = unit(min(b0 + wx * 32u + 16u + fr, b_end - 1u));
    const uint32_t ws = fq & 1u;  // which word of the half

    i32x4v acc[2][2][9];  // [group][B tile][3 * class of A + class of B]
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int c = 0; c < 9; ++c) acc[g][tb][c] = i32x4v{0, 0, 0, 0};

    constexpr int RING = 4;
    uint2 ua[RING], ub0[RING], ub1[RING];  // {plane 0 word, plane 1 word} of the lane's 32 samples
    auto fetch = [&](uint32_t c, uint2 &xa, uint2 &xb0, uint2 &xb1) {
        const u32x4 va = ga[(uint64_t)c * 128u], vb0 = gb0[(uint64_t)c * 128u], vb1 = gb1[(uint64_t)c * 128u];
        xa = ws ? make_uint2(va[1], va[3]) : make_uint2(va[0], va[2]);
        xb0 = ws ? make_uint2(vb0[1], vb0[3]) : make_uint2(vb0[0], vb0[2]);
        xb1 = ws ? make_uint2(vb1[1], vb1[3]) : make_uint2(vb1[0], vb1[2]);
    };
#pragma unroll
    for (int k = 0; k < RING; ++k) fetch(k, ua[k], ub0[k], ub1[k]);
    const uint32_t last_chunk = nchunk - 1u;

    auto chunk = [&](const uint2 &pa, const uint2 &pb0, const uint2 &pb1, i32x4v (&d)[2][9]) {
        // class words: aa = code 1, ab = code 2, bb = code 3 (plane0 = codes 1,3; plane1 = codes 2,3)
        const uint32_t ca3[3] = {pa.x & ~pa.y, pa.y & ~pa.x, pa.x & pa.y};
        uint32_t cb3[2][3], rb3[2][3];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const uint2 &pb = tb ? pb1 : pb0;
            cb3[tb][0] = pb.x & ~pb.y; cb3[tb][1] = pb.y & ~pb.x; cb3[tb][2] = pb.x & pb.y;
#pragma unroll
            for (int c = 0; c < 3; ++c) rb3[tb][c] = __builtin_bswap32(__builtin_bitreverse32(cb3[tb][c])) >> 1;
        }
#pragma unroll
        for (int hk = 0; hk < 2; ++hk) {
            i32x4 xa[3], yb[2][3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                xa[c] = expand_a(ca3[c], hk);
                yb[0][c] = expand_b(cb3[0][c], rb3[0][c], hk);
                yb[1][c] = expand_b(cb3[1][c], rb3[1][c], hk);
            }
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int x = 0; x < 3; ++x)
#pragma unroll
                    for (int y = 0; y < 3; ++y)
                        d[tb][3 * x + y] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xa[x], yb[tb][y], d[tb][3 * x + y], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 18; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA (16 cycles)
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // 3 VALU
            }
        }
    };
    auto round = [&](uint32_t c0, i32x4v (&d)[2][9]) {
        static_for<0, RING>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            chunk(ua[k], ub0[k], ub1[k], d);
            fetch(min(c0 + k + RING, last_chunk), ua[k], ub0[k], ub1[k]);  // past the end: re-read, unused
        });
    };
    for (uint32_t c0 = 0; c0 < nchunk_c; c0 += RING) round(c0, acc[0]);
    for (uint32_t c0 = nchunk_c; c0 < nchunk; c0 += RING) round(c0, acc[1]);

    // ---- per pair.  C layout of the 16x16 MFMA: column (B index) = lane & 15, row (A index) = 4 (lane >> 4) + reg
    if (PRE) {
        for (uint32_t idx = t; idx < (uint32_t)(M9A + M9B) * 24u; idx += 256u) {
            const uint32_t row = idx / 24u, k = idx % 24u;
            const uint32_t snp = row < (uint32_t)M9A ? min(a0 + row, a_end - 1u) : min(b0 + (row - M9A), b_end - 1u);
            s_marf[idx] = reinterpret_cast<const uint32_t *>(a.marf + snp)[k];
        }
        __syncthreads();
    }
    const float nf = (float)a.n_individs, inv_n = 1.0f / (float)a.n_individs;
    const uint32_t i_base = a0 + wy * 16u + 4u * fq;
    auto load_marf = [&](uint32_t row) {
        SnpMarginF m;
#pragma unroll
        for (int q = 0; q < 6; ++q)
            *reinterpret_cast<u32x4 *>(reinterpret_cast<uint32_t *>(&m) + 4 * q) =
                *reinterpret_cast<const u32x4 *>(s_marf + row * 24u + 4 * q);
        return m;
    };

    uint32_t need = 0;  // bit 4 tb + reg: that pair still wants its FP64 measure
    unsigned long long n_eval = 0;
    static_for<0, 8>([&](auto pc) {
        constexpr int tb = decltype(pc)::value >> 2, r = decltype(pc)::value & 3;
        const uint32_t i = i_base + r, j = b0 + wx * 32u + tb * 16u + fr;
        if (i < a_end && j < b_end && i < j) {
            ++n_eval;
            if (PRE) {
                uint32_t ca[9], co[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    ca[c] = (uint32_t)acc[0][tb][c][r] >> 6;  // every product was 64
                    co[c] = (uint32_t)acc[1][tb][c][r] >> 6;
                }
                const float bound = boost_bound(ca, co, load_marf(i - a0), load_marf(M9A + (j - b0)), nf, inv_n);
                if (a.dense) a.dense[(size_t)(i - a.a_begin) * a.n_b + (j - a.b_begin)] = (double)bound;
                else if (!(bound < a.bound_threshold)) need |= 1u << (4 * tb + r);
            } else {
                need |= 1u << (4 * tb + r);
            }
        }
    });

    double vmax = -1.0e300, vmin = 1.0e300;
    while (need) {
        const int p = __ffs((int)need) - 1;
        need &= need - 1u;
        uint32_t ca[9], co[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            uint32_t x = 0, y = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                x = p == q ? (uint32_t)acc[0][q >> 2][c][q & 3] : x;
                y = p == q ? (uint32_t)acc[1][q >> 2][c][q & 3] : y;
            }
            ca[c] = x >> 6;
            co[c] = y >> 6;
        }
        const uint32_t i = i_base + (p & 3), j = b0 + wx * 32u + (p >> 2) * 16u + fr;
        const SnpMargin m1 = a.mar[i], m2 = a.mar[j];
        const double measure = boost_measure(ca, co, m1, m2, a.n_individs, a.frac, a.xlogx);
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
        if (a.dense) a.dense[(size_t)(i - a.a_begin) * a.n_b + (j - a.b_begin)] = measure;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n_eval += __shfl_xor(n_eval, off, 64);
        if (!PRE) {
            vmax = fmax(vmax, __shfl_xor(vmax, off, 64));
            vmin = fmin(vmin, __shfl_xor(vmin, off, 64));
        }
    }
    if (lane == 0 && n_eval) {
        atomicAdd(&a.tally[1], n_eval);
        if (!PRE) {
            atomicMax(&a.tally[2], ordered(vmax));
            atomicMin(&a.tally[3], ordered(vmin));
        }
    }
}

// ---------------------------------------------------------------------------
// One group at a time (k_boost_mfma_seq): the accumulators are the register budget
// of these kernels, and half of them belong to the other group.  After the case
// planes a wave parks its case counts in LDS as uint16 (each lane reads back only
// what it wrote; used when the case group has < 65 536 members) and re-uses the
// registers for the controls.  That buys
//   NV = 3, TA = 1: all nine cells on 32 x 32 tiles (144 accumulators).  The
//       16x16x64 MFMA of k_boost_mfma9 holds the vector issue port for half of its
//       16 cycles, the 32x32x32 one for a quarter of its 32, and nine products
//       share six operand expansions: 3.3 VALU per MFMA, 88 % of the matrix peak.
//   NV = 2, TA = 2: the four corners on 64 x 32 pairs per wave (128 accumulators):
//       six expansions per eight MFMAs instead of four per four.
// Workgroup = 2 x 2 waves = (64 TA) x 64 pairs; everything else as k_boost_mfma.
// ---------------------------------------------------------------------------
template <bool PRE, int NV, int TA>
__global__ void __launch_bounds__(256, 2) k_boost_mfma_seq(BoostArgs a) {
    constexpr int NP = NV * NV;            // products per tile pair
    constexpr int NT = TA * NP;            // accumulator tiles
    constexpr int WA = 32 * TA;            // A rows per wave
    __shared__ uint32_t s_case[NT * 8 * 256];  // [tile][register pair][thread]: two uint16 case counts

    const uint32_t a0 = a.a_begin + blockIdx.y * (2 * WA), b0 = a.b_begin + blockIdx.x * MT;
    const uint32_t a_end = a.a_begin + a.n_a, b_end = a.b_begin + a.n_b;
    if (a0 + 1u >= min(b0 + (uint32_t)MT, b_end)) return;

    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t wy = wave >> 1, wx = wave & 1u;
    const uint32_t fr = lane & 31u, fh = lane >> 5;

    const uint32_t nchunk_c = a.wc / 4u, nchunk = nchunk_c + a.wt / 4u;
    auto unit = [&](uint32_t snp) {
        return a.cct + ((uint64_t)(snp >> 6) * nchunk * 64u + (snp & 63u)) * 2u + fh;
    };
    const u32x4 *ga[TA];
#pragma unroll
    for (int ta = 0; ta < TA; ++ta) ga[ta] = unit(min(a0 + wy * WA + ta * 32u + fr, a_end - 1u));
    const u32x4 *gb = unit(min(b0 + wx * 32u + fr, b_end - 1u));

    i32x16 acc[TA][NP];  // [A tile][NV * class of A + class of B], one group at a time
    auto clear = [&]() {
#pragma unroll
        for (int ta = 0; ta < TA; ++ta)
#pragma unroll
            for (int c = 0; c < NP; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ta][c][r] = 0;
    };
    clear();

    constexpr int RING = 4;
    u32x4 ua[RING][TA], ub[RING];
#pragma unroll
    for (int k = 0; k < RING; ++k) {
#pragma unroll
        for (int ta = 0; ta < TA; ++ta) ua[k][ta] = ga[ta][(uint64_t)k * 128u];
        ub[k] = gb[(uint64_t)k * 128u];
    }
    const uint32_t last_chunk = nchunk - 1u;

    // class v of a row's word w: NV = 2: aa, bb; NV = 3: aa, ab, bb (plane0 = codes 1,3; plane1 = codes 2,3)
    auto cls = [](const u32x4 &v, int c, int w) -> uint32_t {
        const uint32_t p0 = v[w], p1 = v[2 + w];
        if (c == 0) return p0 & ~p1;
        if (c == NV - 1) return p0 & p1;
        return p1 & ~p0;
    };
    auto chunk = [&](const u32x4 (&va)[TA], const u32x4 &vb) {
        uint32_t ca[TA][NV][2], cb[NV][2], rb[NV][2];
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int c = 0; c < NV; ++c) {
#pragma unroll
                for (int ta = 0; ta < TA; ++ta) ca[ta][c][w] = cls(va[ta], c, w);
                cb[c][w] = cls(vb, c, w);
                rb[c][w] = __builtin_bswap32(__builtin_bitreverse32(cb[c][w])) >> 1;
            }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            i32x4 xa[TA][NV], yb[NV];
#pragma unroll
            for (int c = 0; c < NV; ++c) {
#pragma unroll
                for (int ta = 0; ta < TA; ++ta) xa[ta][c] = expand_a(ca[ta][c][s >> 1], s & 1);
                yb[c] = expand_b(cb[c][s >> 1], rb[c][s >> 1], s & 1);
            }
#pragma unroll
            for (int ta = 0; ta < TA; ++ta)
#pragma unroll
                for (int x = 0; x < NV; ++x)
#pragma unroll
                    for (int y = 0; y < NV; ++y)
                        acc[ta][NV * x + y] =
                            __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[ta][x], yb[y], acc[ta][NV * x + y], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // 1 MFMA (32 cycles)
                __builtin_amdgcn_sched_group_barrier(0x002, NV == 3 ? 4 : 5, 0);  // its share of the VALU
            }
        }
    };
    auto round = [&](uint32_t c0) {
        static_for<0, RING>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            chunk(ua[k], ub[k]);
            const uint64_t nxt = (uint64_t)min(c0 + k + RING, last_chunk) * 128u;  // past the end: re-read, unused
#pragma unroll
            for (int ta = 0; ta < TA; ++ta) ua[k][ta] = ga[ta][nxt];
            ub[k] = gb[nxt];
        });
    };
    for (uint32_t c0 = 0; c0 < nchunk_c; c0 += RING) round(c0);
#pragma unroll
    for (int ta = 0; ta < TA; ++ta)
#pragma unroll
        for (int c = 0; c < NP; ++c)
#pragma unroll
            for (int r = 0; r < 16; r += 2)
                s_case[((ta * NP + c) * 8 + (r >> 1)) * 256 + t] =
                    ((uint32_t)acc[ta][c][r] >> 6) | (((uint32_t)acc[ta][c][r + 1] >> 6) << 16);  // products of 64
    clear();
    for (uint32_t c0 = nchunk_c; c0 < nchunk; c0 += RING) round(c0);

    // ---- per pair (C layout as in k_boost_mfma); pair p = 16 ta + reg
    const uint32_t j = b0 + wx * 32u + fr, i_base = a0 + wy * WA + 4u * fh;
    const bool j_ok = j < b_end;
    const SnpMargin *pm2 = a.mar + (j_ok ? j : b_end - 1u);
    uint32_t f2c[3], f2t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { f2c[k] = pm2->cases[k]; f2t[k] = pm2->ctrls[k]; }
    SnpMarginF mf2;
    if (PRE) mf2 = a.marf[j_ok ? j : b_end - 1u];
    const float nf = (float)a.n_individs, inv_n = 1.0f / (float)a.n_individs;
    auto case_count = [&](int tile, int r) {  // r may be a run-time value
        const uint32_t v = s_case[(tile * 8 + (r >> 1)) * 256 + t];
        return (r & 1) ? v >> 16 : v & 0xFFFFu;
    };
    auto row_of = [&](int p) { return i_base + (uint32_t)((p >> 4) * 32 + (p & 3) + 8 * ((p & 15) >> 2)); };
    // the two 3 x 3 tables of a pair from its counted cells (cells[0] cases, [1] controls)
    auto tables = [&](const uint32_t (&cells)[2][NP], const uint32_t *f1c, const uint32_t *f1t, uint32_t (&ca)[9],
                      uint32_t (&co)[9]) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            uint32_t *o = g == 0 ? ca : co;
            if (NV == 3) {
#pragma unroll
                for (int c = 0; c < 9; ++c) o[c] = cells[g][c % NP];
            } else {  // table5.cpp:1088-1106 / 1125-1143: corners counted, the rest from the margins
                const uint32_t *f1 = g == 0 ? f1c : f1t;
                const uint32_t *f2 = g == 0 ? f2c : f2t;
                o[0] = cells[g][0]; o[2] = cells[g][1]; o[6] = cells[g][2 % NP]; o[8] = cells[g][3 % NP];
                o[1] = f1[0] - o[0] - o[2];
                o[7] = f1[2] - o[8] - o[6];
                o[3] = f2[0] - o[0] - o[6];
                o[5] = f2[2] - o[2] - o[8];
                o[4] = f2[1] - o[1] - o[7];
            }
        }
    };

    uint32_t need = 0;
    unsigned long long n_eval = 0;
    static_for<0, 16 * TA>([&](auto pc) {
        constexpr int p = decltype(pc)::value, ta = p >> 4, r = p & 15;
        const uint32_t i = row_of(p);
        if (i < a_end && j_ok && i < j) {
            ++n_eval;
            if (PRE) {
                uint32_t cells[2][NP], ca[9], co[9];
#pragma unroll
                for (int c = 0; c < NP; ++c) {
                    cells[0][c] = case_count(ta * NP + c, r);
                    cells[1][c] = (uint32_t)acc[ta][c][r] >> 6;
                }
                const SnpMargin *pm1 = a.mar + i;
                tables(cells, pm1->cases, pm1->ctrls, ca, co);
                const float bound = boost_bound(ca, co, a.marf[i], mf2, nf, inv_n);
                if (a.dense) a.dense[(size_t)(i - a.a_begin) * a.n_b + (j - a.b_begin)] = (double)bound;
                else if (!(bound < a.bound_threshold)) need |= 1u << p;
            } else {
                need |= 1u << p;
            }
        }
    });

    double vmax = -1.0e300, vmin = 1.0e300;
    while (need) {
        const int p = __ffs((int)need) - 1;
        need &= need - 1u;
        uint32_t cells[2][NP], ca[9], co[9];
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            uint32_t y = 0;
#pragma unroll
            for (int q = 0; q < 16 * TA; ++q) y = p == q ? (uint32_t)acc[q >> 4][c][q & 15] : y;
            cells[0][c] = case_count((p >> 4) * NP + c, p & 15);
            cells[1][c] = y >> 6;
        }
        const uint32_t i = row_of(p);
        const SnpMargin m1 = a.mar[i], m2 = *pm2;
        tables(cells, m1.cases, m1.ctrls, ca, co);
        const double measure = boost_measure(ca, co, m1, m2, a.n_individs, a.frac, a.xlogx);
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
        if (a.dense) a.dense[(size_t)(i - a.a_begin) * a.n_b + (j - a.b_begin)] = measure;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n_eval += __shfl_xor(n_eval, off, 64);
        if (!PRE) {
            vmax = fmax(vmax, __shfl_xor(vmax, off, 64));
            vmin = fmin(vmin, __shfl_xor(vmin, off, 64));
        }
    }
    if (lane == 0 && n_eval) {
        atomicAdd(&a.tally[1], n_eval);
        if (!PRE) {
            atomicMax(&a.tally[2], ordered(vmax));
            atomicMin(&a.tally[3], ordered(vmin));
        }
    }
}

hipError_t launch_boost(const uint32_t *d_cc, uint64_t cc_stride_w, uint32_t wc, uint32_t wt,
                        const SnpMargin *d_mar, const double *d_frac, const double *d_xlogx,
                        uint32_t a_begin, uint32_t n_a, uint32_t b_begin,
                        uint32_t n_b, uint32_t n_individs, double threshold, BoostHit *d_hits,
                        unsigned long long max_hits, unsigned long long *d_tally, double *d_dense,
                        bool fast, hipStream_t stream, const SnpMarginF *d_marf, bool prefilter,
                        float bound_threshold, const u32x4 *d_cct, bool narrow_tiles) {
    if (n_a == 0 || n_b == 0) return hipSuccess;
    BoostArgs a;
    a.cc = d_cc; a.cc_stride_w = cc_stride_w; a.wc = wc; a.wt = wt; a.mar = d_mar; a.frac = d_frac; a.xlogx = d_xlogx;
    a.a_begin = a_begin; a.n_a = n_a; a.b_begin = b_begin; a.n_b = n_b;
    a.n_individs = (int)n_individs; a.threshold = threshold;
    a.hits = d_hits; a.max_hits = max_hits; a.tally = d_tally; a.dense = d_dense;
    a.marf = d_marf; a.bound_threshold = bound_threshold;
    a.cct = d_cct;
    const bool small_case_group = wc * 32u < 65536u && !narrow_tiles;  // case counts fit uint16
    if (d_cct && fast && small_case_group && d_marf) {  // four corners, 64 x 32 pairs per wave
        const dim3 mgrid((n_b + MT - 1) / MT, (n_a + 2 * MT - 1) / (2 * MT));
        if (prefilter) hipLaunchKernelGGL((k_boost_mfma_seq<true, 2, 2>), mgrid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_boost_mfma_seq<false, 2, 2>), mgrid, dim3(256), 0, stream, a);
        return hipGetLastError();
    }
    if (d_cct && fast) {
        const dim3 mgrid((n_b + MT - 1) / MT, (n_a + MT - 1) / MT);
        if (prefilter && d_marf) hipLaunchKernelGGL(k_boost_mfma<true>, mgrid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(k_boost_mfma<false>, mgrid, dim3(256), 0, stream, a);
        return hipGetLastError();
    }
    if (d_cct && d_marf && small_case_group) {  // missing calls: nine products per pair
        const dim3 mgrid((n_b + MT - 1) / MT, (n_a + MT - 1) / MT);
        if (prefilter) hipLaunchKernelGGL((k_boost_mfma_seq<true, 3, 1>), mgrid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_boost_mfma_seq<false, 3, 1>), mgrid, dim3(256), 0, stream, a);
        return hipGetLastError();
    }
    if (d_cct && d_marf) {  // ... and any group size
        const dim3 mgrid((n_b + M9B - 1) / M9B, (n_a + M9A - 1) / M9A);
        if (prefilter) hipLaunchKernelGGL(k_boost_mfma9<true>, mgrid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(k_boost_mfma9<false>, mgrid, dim3(256), 0, stream, a);
        return hipGetLastError();
    }
    const dim3 grid((n_b + TB - 1) / TB, (n_a + TA - 1) / TA);
    if (prefilter && d_marf) {
        if (fast) hipLaunchKernelGGL((k_boost<true, true>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_boost<false, true>), grid, dim3(256), 0, stream, a);
    } else {
        if (fast) hipLaunchKernelGGL((k_boost<true, false>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_boost<false, false>), grid, dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

}  // namespace gw
