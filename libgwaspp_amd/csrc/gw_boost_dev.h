// gw_boost_dev.h -- device-side pieces shared by the pair-path translation units
// (gw_boost.hip, gw_boost_mfma.hip, gw_boost_seq4.hip, gw_boost_seq9.hip, gw_boost_both9.hip, gw_boost_fp4*.hip; split so that
// they compile in parallel: the unrolled per-pair epilogues make each of them a minute of compile time).
#pragma once
#include <type_traits>

#include "gw_kernels.h"

namespace gw {


// v_mul_legacy_f32 (0 * anything = 0, also 0 * inf): clang has no builtin for it, the LLVM intrinsic is reached by name --
// as a real instruction to the compiler (scheduled, hazards handled), unlike an inline-asm statement
extern "C" __device__ float gw_fmul_legacy(float, float) __asm("llvm.amdgcn.fmul.legacy");

// compile-time loop: the body sees its index as a constant
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}


// order-preserving map double -> uint64 for atomicMax / atomicMin
__device__ __forceinline__ unsigned long long ordered(double v) {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
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
// together with a RUNNING ERROR BOUND `err` (Higham-style, first order in u = 2^-24, constants rounded up):
// |bound - measure| <= err for every pair, so "bound + err < threshold" PROVES that the pair cannot pass and
// anything else (including NaN / Inf) is decided by the FP64 measure.  Derivation, per operation:
//   * float(k), float(x+y) exact (counts < 2^24); every mul / add / float(double) rounds within u;
//     v_rcp_f32 within 2u (1 ulp); v_log_f32 within 2u |log2 z| + 4u (1 ulp, plus an absolute allowance of
//     2^-22 around z = 1) -- the two hardware assumptions; the screen verifies them at run time (below);
//   * argument of a cell's log: rcp 2u, inv_n u, rcp*inv_n u, qb u, qa u, qb*qa u, k*rs u, (.)*(qb qa) u = 9u
//     -> log2 off by 9u/ln2 = 13.2u, hardware log 2.01u |l| + 4.01u, product k*l u:
//     |t - k log2 z| <= 17.3u k + 3.02u |t|;  summing the 18 terms recursively adds 17.1u T, T = sum |t|:
//     |s - S| <= 17.3u n + 20.2u T                      (sum of all k <= n)
//   * tao: rb u, pa u, rb*pa u, sum of two u, *(x+y) u, eight more adds of non-negative terms: 13.1u relative
//     -> log2 off by 18.95u, hardware log 2.01u |L| + 4.01u, n*L u:  |nL - n log2 tao| <= 23u n + 3.02u n |L|
//   * s + nL, the factor 2 ln 2 (itself rounded) and the product: 3.1u |bound|
//   => |bound - measure| <= 2 ln 2 * u * (40.3 n + 20.2 T + 3.02 n |L|) + 3.1u |bound|
//   * T needs no per-term work: with S- the sum of the negative terms, T = S + 2 S-, and ln z >= 1 - 1/z gives
//     k ln z_k >= k - e_k with e_k = k / z_k = n * (the cell's share of tao), so S- <= sum e_k / ln 2 = n tao / ln 2:
//     T <= |s| + 2.886 n tao
//   => |bound - measure| <= u * (55.9 n + 28.1 |s| + 81 n tao + 4.19 n |L| + 3.1 |bound|);
//   err below uses (58, 30, 87, 4.5, 6): room for the second-order terms, its own float evaluation and the
//   rounding of bound + err.
// At n = 10 000 (tao ~ 1) that is ~0.09 -- twenty times below the empirical slack (2n * 1e-4 + 0.05) it replaces;
// measured |bound - measure| / err stays below 0.1 (tests/test_gpu_boost.py prints it).
// Run-time check of the assumptions: every pair that does get its FP64 measure is compared with its bound, and
// a pair with |measure - bound| > err bumps tally[2]; the host then repeats the call with every pair in FP64
// (gw_capi.cpp boost_common; GW_OPT_BOOST_PREFILTER_TRIPS counts such calls -- none has been seen).
__device__ __forceinline__ float boost_bound(const uint32_t (&ca)[9], const uint32_t (&co)[9],
                                             const SnpMarginF &f1, const SnpMarginF &f2, float n, float inv_n,
                                             float &err) {
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
    const float lt = __log2f(tao);
    const float r = 1.3862943611198906f * (s + n * lt);  // 2 ln 2
    err = 5.9604645e-8f * (n * (58.0f + 87.0f * tao + 4.5f * fabsf(lt)) + 30.0f * fabsf(s) + 6.0f * fabsf(r));
    return r;
}
// The bound for tables WITHOUT a missing call (the four-corner kernels), with the logarithms split:
//   z = k qb qa / (s n)  =>  sum_cells k log2 z = sum_cells k log2 k - sum_ab s log2 s + sum_cells k log2(qb qa / n),
// and with no call missing the cells of a row / column of the pair's table add up to the SNPs' own genotype counts,
// so the last sum is K_i + K_j, two numbers per SNP computed once in FP64 (SnpMarginF::ki, ::kj):
//   measure = 2 ln 2 * [ sum_ab (x lg x + y lg y - s lg s) + K_i + K_j + n lg(tao) ],   lg = log2, s = x + y.
// Per pair that is 27 v_log_f32 of integers, 27 multiply-adds and tao as 24 multiply-adds (tao = sum_a pa[a] * sum_b s_ab
// rb[b] per group) -- no reciprocal, no product of margin factors per cell: about half the vector instructions of
// boost_bound.  The price is conditioning: the three terms are each ~ n and cancel to O(1), so the error bound is
// the larger u * n * (8.5 lg n + 69 + ...) -- 0.11 at n = 10 000, still a hundred times below the threshold.
// Error bound, per operation (u = 2^-24; hardware assumptions as for boost_bound, checked at run time the same way):
//   * float(k) exact; l = v_log_f32(k): |l - lg k| <= 2u lg k + 4u <= 2u Lg + 4u with Lg = lg n (k <= n);
//     times k, summed over the 18 + 9 logarithms (sum of the k = 2n):            n u (4 Lg + 8)
//   * per (a, b): e = x lx (rounded: u x Lg), e = fma(y, ly, e) (|e| <= s Lg: u s Lg), e = fma(-s, ls, e)
//     (|e| <= s: u s):  u s (2 Lg + 1), summed:                                   n u (2 Lg + 1)
//   * A += e, nine times, |A| <= n:                                               9 n u
//   * K_i, K_j rounded to float (|K_i| <= n, |K_j| <= 1.07 n), (A + K_i) + K_j:   7.14 n u
//   * tao: inputs rb, pa within u, 3 + 6 multiply-adds of non-negative terms: <= 11u relative (boost_bound: 13.1u),
//     so as there n lt is within 23 u n + 3.02 u n |L|; R + n lt adds u (|final| + 2 n |L|)
//   => before the factor 2 ln 2:  n u (6 Lg + 48.2) + 5.02 n u |L| + u |final|;  times 1.3863 and the factor's own
//      two roundings (2.1 u |bound|):  u * [ n (8.32 Lg + 66.8 + 6.96 |L|) + 3.5 |bound| ];
//   err below uses (8.5, 69, 7.2, 6).
__device__ __forceinline__ float boost_bound_split(const uint32_t (&ca)[9], const uint32_t (&co)[9],
                                                   const SnpMarginF &f1, const SnpMarginF &f2, float n, float lg_n,
                                                   float &err) {
    float A = 0.0f, tao = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float in_c = 0.0f, in_t = 0.0f;  // sum_b s_ab rb[b], per group
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const uint32_t x = ca[3 * a + b], y = co[3 * a + b];
            const float fx = (float)x, fy = (float)y, fs = (float)(x + y);
            // k lg k with 0 lg 0 = 0: the logarithm of max(k, 1)
            float e = fx * __log2f(fmaxf(fx, 1.0f));
            e = __builtin_fmaf(fy, __log2f(fmaxf(fy, 1.0f)), e);
            e = __builtin_fmaf(-fs, __log2f(fmaxf(fs, 1.0f)), e);
            A += e;
            in_c = __builtin_fmaf(fs, f2.rb[b], in_c);
            in_t = __builtin_fmaf(fs, f2.rb[3 + b], in_t);
        }
        tao = __builtin_fmaf(f1.pa[a], in_c, tao);
        tao = __builtin_fmaf(f1.pa[3 + a], in_t, tao);
    }
    const float lt = __log2f(tao);
    const float r = 1.3862943611198906f * (((A + f1.ki) + f2.kj) + n * lt);  // 2 ln 2
    err = 5.9604645e-8f * (n * (8.5f * lg_n + 69.0f + 7.2f * fabsf(lt)) + 6.0f * fabsf(r));
    return r;
}
// A FIRST-level bound without any logarithm, for the same tables (no missing call).  With h the binary entropy in
// bits, A = sum_cells (x lg x + y lg y - s lg s) = - sum_cells s h(x / s).  For an expansion point p (the cases'
// share of the cohort, as a float) the two-piece quadratic
//   h(p) + h'(p)(z - p) - c(z) (z - p)^2,   c(z) = c_lo for z < p, c_hi for z > p
// lies below h on [0, 1] for  c_lo = max(g(0), g(p)), c_hi = max(g(1), g(p)),
//   g(z) = [h(p) + h'(p)(z - p) - h(z)] / (z - p)^2,   g(p) = -h''(p) / 2 = 1 / (2 ln 2 p (1 - p))
// (g is an average of the convex -h'' over the segment from p to z, hence convex in z, hence on either side of p
// largest at an end of that side), and because the x of a table add up to the case count,
//   A <= k0 + sum_cells c(x / s) (x - p s)^2 / s,   k0 = -n h(p) - h'(p)(n_case - p n).
// With d = x - p s, c_m = (c_lo + c_hi) / 2 and c_d = (c_hi - c_lo) / 2 the sum is c_m sum d^2 / s + c_d sum d |d| / s:
// nine v_rcp_f32 instead of 27 v_log_f32.  The bound is one-sided, measure <= bound1 + err1, and loose by
// (2 c ln 2 p (1 - p) - 1) times a cell's share of the pair's saturated-model G statistic -- 0.39 x a chi^2 with 8
// degrees of freedom for a balanced cohort, where c_lo ~ c_hi ~ 4; for a lopsided one the factor is 0 on the long
// side of p and at most 1 on the short one.  On the bench's input 3.4 pairs in 10 000 reach a threshold of 30
// (tools/bound_study.py) and go on to boost_bound_split.
//
// Round 4: the rest of the measure, K_i + K_j + n lg tao, without its logarithm and with a third of its multiply-adds.
//   * lg tao <= lg e (tao - 1) for every tao > 0, so  n lg tao <= N1 (tao - 1), N1 = n lg e: still an upper bound.  On
//     unlinked SNPs |tao - 1| < 1e-4 and the slack n (tao - 1 - ln tao) is below 1e-4 (tools/bound_study.py).
//   * tao - 1 from the four corners.  tao = sum_abc p(ab) p(ac) p(bc) / (p(a) p(b) p(c)) (epistasis_func.cpp:424-470; a, b the
//     genotypes of the two SNPs, c the status).  With d_ac = P(c | a) / P(c) - 1 for the first SNP and e_bc likewise for
//     the second, P(c | a) / P(c) = 1 + d_ac, sum_c P(c) d_ac = 0 = sum_c P(c) e_bc (every individual is a case or a
//     control) and sum_ab p(ab) = 1 (no missing call):
//         tao - 1 = sum_c P(c) sum_ab p(ab) d_ac e_bc = sum_c sum_ab S_ab u_a v_b,   u_a = d_ac, v_b = P(c) e_bc / n, S = s counts.
//     The rows and columns of S add up to the SNPs' own genotype counts R_a, C_b, with sum_a R_a u_a = 0 = sum_b C_b v_b, so
//     writing u_a = u_1 + u'_a, v_b = v_1 + v'_b (u'_1 = v'_1 = 0) leaves
//         sum_ab S_ab u_a v_b = sum_{a, b in {0, 2}} S_ab u'_a v'_b - n u_1 v_1
//     -- four products per group on the CORNER totals the kernel has anyway, and the result is tao - 1 itself (magnitude
//     1e-5) instead of tao (magnitude 1), so nothing cancels against the 1: 14 instead of 24 multiply-adds per pair.
//     u'_a, n u_1 (SnpMarginF::ua) and v'_b, v_1 (::vb) are per-SNP numbers computed in FP64 by k_margins.
//
// The constants are exact by definition (the host rounds c_lo, c_hi and k0 up), so only the arithmetic rounds; with
// c = max(c_lo, c_hi) = c_m + |c_d| and q = sum d^2 / s (u = 2^-24):
//   * d = fma(-p, s, x): u |d|; 1 / s: 2u (v_rcp_f32, 1 ulp; s == 0 has d == 0 and v_mul_legacy_f32 makes 0 x inf = 0);
//     t = d / s: 4u relative; q_m = fma(d, t, q_m), q_d = fma(|d|, t, q_d): 5u of the term + u q per step, nine steps: 14u q
//     each;  A1 = fma(c_d, q_d, fma(c_m, q_m, k0)): 14u c q + 2u (c q + |k0|),  |k0| <= 1.01 n:          u (16 c q + 2.02 n)
//   * K_i, K_j rounded (2.07 n u), (A1 + K_i) + K_j: u (2 |A1| + 2 |K_i| + |K_j|):                       u (2 c q + 7.2 n)
//   * tao - 1: every term S u' v' (S exact) carries 2u from its rounded factors and <= 3u from the multiply-adds it passes
//     through, so the computed sum is within 6u T of the exact one, T = the sum of the terms' magnitudes, and
//     T <= sum_c [max_a |u'_a| sum_b C_b |v'_b| + n |u_1| |v_1|] <= ea_i eb_j (SnpMarginF::ea, ::eb, rounded up).  Times N1
//     (rounded: u) and with |tao - 1| <= T:  N1 (tao - 1) is within 8u N1 T = 11.6 u n T.
//   * the final sum and the factor 2 ln 2 with its roundings:  u (25 c q + 13 n + 16.1 n T) + 3.5 u |bound|
//   err below uses (27, 48, 30, 6), the first three folded with u into QuadBound's constants on the host.  The screen checks  measure <= bound1 + err1  for every pair that gets its FP64
//   measure (check_upper_bound in k_boost_finish), as it checks the second level.
//
// One curvature constant (TWO = false, balanced cohorts), later in round 4: the sum without d.  The x of a table without a missing
// call add up to the case count X and the s to n, so  sum d^2 / s = sum x^2 / s - (2 p X - p^2 n)  and
//   k0 + c sum d^2 / s = c sum x^2 / s + k0x,   k0x = k0 - c (2 p X - p^2 n)   (the host, in FP64 from the float p and c, rounded up)
// -- per cell  t = x / s, qx = fma(x, t, qx)  and no  d = fma(-p, s, x): nine multiply-adds per pair less.  What it costs: qx ~ p X
// where q ~ 10, so the rounding of the sum is no longer relative to q.  t: 3u relative (reciprocal 2u, product u); qx: 3u of the
// term + u qx per step, nine steps: 12u qx, qx <= X;  A0 = fma(c, qx, k0x): 12u c X + u (c X + |k0x|), |k0x| <= 1.01 n + 2 c X:
// u (15 c X + 1.01 n).  err replaces its  27u c q  by the constant  u 24 c n  (QuadBound::cx, X <= n, rounded up; 0.04 at n = 10^4
// beside the 0.03 of u 48 n).  With tables that HAVE missing calls (boost_bound_quad9_f) X and S are the pair's own sums, which that
// form computes anyway:  c sum d^2 / s - h(p) S - h'(p)(X - p S) = c sum x^2 / s + (kx - 2 p c) X + (ks + c p^2) S  =: ... kx9 X + ks9 S.
#ifndef GW_QUAD_X2   // 0 (measurement only): the d form for one constant as well
#define GW_QUAD_X2 1
#endif
struct QuadBound {
    float p, c_m, c_d, k0, c27;  // c27 = u 27 max(c_lo, c_hi), u = 2^-24; c_m == 0: no first level
    float k0x, cx, kx9, ks9;     // one curvature constant: k0 - c (2 p X - p^2 n); u 24 c n; kx - 2 p c; ks + c p^2
    float n48, n30;              // u 48 n and u 30 n (rounded up): the error bound's constant terms
    float n1, n1t;               // N1 = n lg e; N1 n / n_controls (tao - 1 from the cases' term alone)
    float kx, ks, nk, n22;       // tables with missing calls: -h'(p), p h'(p) - h(p); u n (10 (2 |h'(p)| + 1) + 26) and u 22 n, rounded up
};
// (fx, fs: a cell's case count and its total as floats -- exact, counts are below 2^24)
// TWO = false: one curvature constant for both sides (the host sets c_m = max(c_lo, c_hi), c_d = 0 -- what a balanced
// cohort gets, where the two differ by a percent): the nine multiply-adds of q_d are not issued.
template <bool TWO = true>
__device__ __forceinline__ float boost_bound_quad_f(const float (&fx)[9], const float (&fs)[9],
                                                    const SnpMarginF &f1, const SnpMarginF &f2, float n,
                                                    const QuadBound &qb, float &err) {
    float q_m = 0.0f, q_d = 0.0f;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const float x = fx[c], s = fs[c];
        if constexpr (TWO || !GW_QUAD_X2) {
            const float d = __builtin_fmaf(-qb.p, s, x);
            const float t = gw_fmul_legacy(d, __builtin_amdgcn_rcpf(s));  // s == 0: d == 0 and 0 x inf = 0
            q_m = __builtin_fmaf(d, t, q_m);
            if constexpr (TWO) q_d = __builtin_fmaf(fabsf(d), t, q_d);
        } else {  // sum x^2 / s; the constant part of sum d^2 / s is in k0x
            const float t = gw_fmul_legacy(x, __builtin_amdgcn_rcpf(s));  // s == 0: x == 0 and 0 x inf = 0
            q_m = __builtin_fmaf(x, t, q_m);
        }
    }
    // tao - 1 = sum over the groups of  u'_0 (v'_0 S00 + v'_2 S02) + u'_2 (v'_0 S20 + v'_2 S22) - (n u_1) v_1
    // -- and the controls' term is the cases' times P(case) / P(control): everyone is one or the other, so P(0) d_a0 + P(1) d_a1 = 0
    // per genotype, i.e. u(controls) = -P(0) / P(1) u(cases), and v = P(c) e_bc / n gives v(controls) = -v(cases).  One group is
    // evaluated, the factor 1 + P(0) / P(1) = n / n_controls is in QuadBound::n1t.  (The error bound's T <= ea eb stays: with
    // rho = P(0) / P(1), (1 + rho) ea_0 eb_0 <= max(1, rho) ea_0 (2 eb_0) = ea eb.)
    float t1 = 0.0f;
#pragma unroll
    for (int g = 0; g < (GW_QUAD_X2 ? 1 : 2); ++g) {
        const float r0 = __builtin_fmaf(f2.vb[3 * g + 1], fs[2], f2.vb[3 * g] * fs[0]);
        const float r2 = __builtin_fmaf(f2.vb[3 * g + 1], fs[8], f2.vb[3 * g] * fs[6]);
        t1 = __builtin_fmaf(f1.ua[3 * g], r0, t1);
        t1 = __builtin_fmaf(f1.ua[3 * g + 1], r2, t1);
        t1 = __builtin_fmaf(-f1.ua[3 * g + 2], f2.vb[3 * g + 2], t1);
    }
    const float A0 = __builtin_fmaf(qb.c_m, q_m, (TWO || !GW_QUAD_X2) ? qb.k0 : qb.k0x);
    const float A1 = TWO ? __builtin_fmaf(qb.c_d, q_d, A0) : A0;
    const float r = 1.3862943611198906f * __builtin_fmaf(GW_QUAD_X2 ? qb.n1t : qb.n1, t1, (A1 + f1.ki) + f2.kj);  // 2 ln 2
    // (c27, n30, n48, cx arrive multiplied by u = 2^-24, exactly: a power of two; one constant: cx is part of n48)
    const float e0 = __builtin_fmaf(3.5762787e-7f, fabsf(r), __builtin_fmaf(qb.n30, f1.ea * f2.eb, qb.n48));
    err = (TWO || !GW_QUAD_X2) ? __builtin_fmaf(qb.c27, q_m, e0) : e0;
    return r;
}
// The first level for tables WITH missing calls (all nine cells counted; the rows / columns of a pair's table no longer add up
// to the SNPs' own counts).  From the measure itself (epistasis_func.cpp:424-470), with k = x (cases) / y (controls), s = x + y:
//   measure / (2 ln 2) = sum_cells (x lg x + y lg y - s lg s) + sum_cells [x lg(qb_b qa_a / n)]_cases + [y lg(qb_b qa_a / n)]_controls + n lg tao
// (qb = margins n_group / group of the second SNP, qa = margins / group of the first: SnpMarginF).  The middle sum is linear in
// the cells:  sum_{a,c} Row_c[a] lqa[a,c] + sum_{b,c} Col_c[b] lqb[b,c]  with the row / column sums of the pair's counted case and
// control tables -- twelve products (lqa = lg qa, lqb = lg(qb / n), per-SNP numbers from k_margins).  The first sum is A of the
// no-missing form, bounded by the same quadratic,  A <= c sum d^2 / s - h(p) S - h'(p)(X - p S)  (X, S = the table's case and
// overall totals; without missing calls that linear part was the constant k0), and  n lg tao <= N1 (tao - 1)  as there; tao from
// all nine cells (24 multiply-adds of non-negative terms).  ~130 vector instructions + 9 v_rcp_f32 instead of boost_bound's ~375
// with 27 v_log_f32.  Error bound, per operation (u = 2^-24; counts and their sums are exact floats below 2^24):
//   * q as in boost_bound_quad_f:                                                                          u 16 c q
//   * kx X + ks S: the two constants rounded (u each), two multiply-adds, |.| <= (2 |h'(p)| + 1) n =: H n:     u 4 H n
//   * the twelve products: lq rounded (u); the row sums of a group add up to at most n, so the sum of the terms' magnitudes is at
//     most n (M_i + M_j), M_i = max |lqa| of the first SNP, M_j = max |lqb| of the second (SnpMarginF::ma, ::mb -- ~1 - 3 for
//     common variants, not lg n); rounded inputs + twelve accumulations: u 13 n (M_i + M_j)
//   * tao: <= 11u relative; tao - 1 and the product with N1 (rounded): u n (15.9 + 18.8 |tao - 1|)
//   * the sums of the three parts (2u of their magnitudes: c q + H n + n (M_i + M_j)) and the factor 2 ln 2 with its roundings:
//   => u [25 c q + 8.4 H n + 20.8 n (M_i + M_j) + 22.1 n + 26.1 n |tao - 1|] + 3.5 u |bound|;  err below uses
//      (27, 10 H + 26, 22, 30, 6), the constants folded with u into QuadBound::nk, ::n22, ::n30 on the host.
template <bool TWO = true>
__device__ __forceinline__ float boost_bound_quad9_f(const float (&fx)[9], const float (&fy)[9], const SnpMarginF &f1,
                                                     const SnpMarginF &f2, const QuadBound &qb, float &err) {
    float q_m = 0.0f, q_d = 0.0f, tao = 0.0f;
    float rx[3] = {0.0f, 0.0f, 0.0f}, ry[3] = {0.0f, 0.0f, 0.0f}, cx[3] = {0.0f, 0.0f, 0.0f}, cy[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float in_c = 0.0f, in_t = 0.0f;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float x = fx[3 * a + b], y = fy[3 * a + b], s = x + y;
            if constexpr (TWO || !GW_QUAD_X2) {
                const float d = __builtin_fmaf(-qb.p, s, x);
                const float t = gw_fmul_legacy(d, __builtin_amdgcn_rcpf(s));  // s == 0: d == 0 and 0 x inf = 0
                q_m = __builtin_fmaf(d, t, q_m);
                if constexpr (TWO) q_d = __builtin_fmaf(fabsf(d), t, q_d);
            } else {  // sum x^2 / s; the rest of sum d^2 / s is linear in the pair's totals: kx9, ks9
                const float t = gw_fmul_legacy(x, __builtin_amdgcn_rcpf(s));
                q_m = __builtin_fmaf(x, t, q_m);
            }
            rx[a] += x; ry[a] += y; cx[b] += x; cy[b] += y;   // exact
            in_c = __builtin_fmaf(s, f2.rb[b], in_c);
            in_t = __builtin_fmaf(s, f2.rb[3 + b], in_t);
        }
        tao = __builtin_fmaf(f1.pa[a], in_c, tao);
        tao = __builtin_fmaf(f1.pa[3 + a], in_t, tao);
    }
    const float xs = (rx[0] + rx[1]) + rx[2], ss = xs + ((ry[0] + ry[1]) + ry[2]);  // exact
    float k = 0.0f;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        k = __builtin_fmaf(rx[g], f1.lqa[g], k);
        k = __builtin_fmaf(ry[g], f1.lqa[3 + g], k);
        k = __builtin_fmaf(cx[g], f2.lqb[g], k);
        k = __builtin_fmaf(cy[g], f2.lqb[3 + g], k);
    }
    const float t1 = tao - 1.0f;
    const float lin = (TWO || !GW_QUAD_X2) ? __builtin_fmaf(qb.kx, xs, qb.ks * ss) : __builtin_fmaf(qb.kx9, xs, qb.ks9 * ss);
    const float A0 = __builtin_fmaf(qb.c_m, q_m, lin);
    const float A1 = TWO ? __builtin_fmaf(qb.c_d, q_d, A0) : A0;
    const float r = 1.3862943611198906f * __builtin_fmaf(qb.n1, t1, A1 + k);  // 2 ln 2
    const float e0 = __builtin_fmaf(3.5762787e-7f, fabsf(r), __builtin_fmaf(qb.n30, fabsf(t1), __builtin_fmaf(qb.n22, f1.ma + f2.mb, qb.nk)));
    err = (TWO || !GW_QUAD_X2) ? __builtin_fmaf(qb.c27, q_m, e0) : e0;  // (one constant: the sum's rounding is the constant cx, part of nk)
    return r;
}
template <bool TWO = true>
__device__ __forceinline__ float boost_bound_quad9(const uint32_t (&ca)[9], const uint32_t (&co)[9], const SnpMarginF &f1,
                                                   const SnpMarginF &f2, const QuadBound &qb, float &err) {
    float fx[9], fy[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        fx[c] = (float)ca[c];
        fy[c] = (float)co[c];
    }
    return boost_bound_quad9_f<TWO>(fx, fy, f1, f2, qb, err);
}
template <bool TWO = true>
__device__ __forceinline__ float boost_bound_quad(const uint32_t (&ca)[9], const uint32_t (&co)[9],
                                                  const SnpMarginF &f1, const SnpMarginF &f2, float n,
                                                  const QuadBound &qb, float &err) {
    float fx[9], fs[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        fx[c] = (float)ca[c];
        fs[c] = (float)(ca[c] + co[c]);
    }
    return boost_bound_quad_f<TWO>(fx, fs, f1, f2, n, qb, err);
}
// true: the pair is provably below the threshold (NaN / Inf anywhere -> false -> FP64 decides)
__device__ __forceinline__ bool bound_rules_out(float bound, float err, float bound_threshold) {
    return bound + err < bound_threshold;
}
// the run-time check of the error model, for a pair that has both values
__device__ __forceinline__ void check_bound(double measure, float bound, float err, unsigned long long *tally) {
    if (fabs(measure - (double)bound) > (double)err) atomicAdd(&tally[2], 1ull);  // false for NaN
}
// ... and of the one-sided first level
__device__ __forceinline__ void check_upper_bound(double measure, float bound, float err, unsigned long long *tally) {
    if (measure > (double)bound + (double)err) atomicAdd(&tally[2], 1ull);  // false for NaN
}

// The two 3 x 3 tables of a pair from its counted cells (cells[0] cases, [1] controls).  NV = 3: all nine cells counted;
// NV = 2 (tables without a missing call): the four corners aa/aa, aa/bb, bb/aa, bb/bb counted, the rest from the SNPs' own
// genotype counts f1 (first marker) and f2 (table5.cpp:1088-1106 / 1125-1143)
template <int NV>
__device__ __forceinline__ void boost_tables_of(const uint32_t (&cells)[2][NV * NV], const uint32_t *f1c, const uint32_t *f1t,
                                                const uint32_t *f2c, const uint32_t *f2t, uint32_t (&ca)[9], uint32_t (&co)[9]) {
    constexpr int NP = NV * NV;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        uint32_t *o = g == 0 ? ca : co;
        if (NV == 3) {
#pragma unroll
            for (int c = 0; c < 9; ++c) o[c] = cells[g][c % NP];
        } else {
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
    unsigned long long *tally;  // [0] hits, [1] pairs evaluated, [2] ordered max (PRE: pairs outside the bound's error model), [3] ordered min
    double *dense;              // n_a x n_b, or null
    int dense_err;              // PRE + dense (diagnostic): 0 = write the bound, 1 = write its error bound
    const SnpMarginF *marf;     // PRE only
    const u32x4 *cct;           // k_boost_mfma: the compacted rows in 64-SNP blocks, chunk-major (k_retile_cc)
    float bound_threshold;      // PRE: the threshold rounded down to float; bound + err < this rules a pair out
    float lg_n;                 // PRE: log2(n_individs), rounded up (boost_bound_split's error bound)
    QuadBound quad;             // PRE, tables without a missing call: the first-level bound's constants (c_m == 0: none)
    int fp4_scale[3];           // k_boost_mfma_fp4: E8M0 block scales 2^-1, 2^0, 2^1 (126, 127, 128), as run-time values
    // PRE, the FP4 kernels: the pairs the in-kernel bound could not rule out go to this list (gw_kernels.h
    // BoostCandidateList) and get their second level / FP64 measure from k_boost_finish
    uint32_t *cand;
    unsigned long long *cand_count;
    unsigned long long cand_cap;
};

// a pair over the threshold (epistasis_func.cpp:482-484): appended while the list has room, counted either way
__device__ __forceinline__ void boost_append_hit(const BoostArgs &a, uint32_t i, uint32_t j, double measure) {
    const unsigned long long slot = atomicAdd(&a.tally[0], 1ull);
    if (slot < a.max_hits) {
        BoostHit h;
        h.snp_a = i;
        h.snp_b = j;
        h.measure = measure;
        a.hits[slot] = h;
    }
}

// a pair the in-kernel bound could not rule out leaves the kernel with its counted cells (BoostCandidateList): snp_a, snp_b,
// cells of the cases[NP], cells of the controls[NP]; k_boost_finish<NV> takes it from there
template <int NP>
__device__ __forceinline__ void boost_append_candidate(const BoostArgs &a, uint32_t i, uint32_t j, const uint32_t (&cells)[2][NP]) {
    const unsigned long long slot = atomicAdd(a.cand_count, 1ull);
    if (slot < a.cand_cap) {
        uint32_t *rec = a.cand + slot * (2u + 2u * NP);
        rec[0] = i;
        rec[1] = j;
#pragma unroll
        for (int c = 0; c < NP; ++c) { rec[2 + c] = cells[0][c]; rec[2 + NP + c] = cells[1][c]; }
    }
}

// ---- operand expansion of the int8 MFMA kernels
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

// ---- per-file launchers behind launch_boost (grid and <PRE> chosen inside).  A pre-filtered screen (pre, no dense output) of
// any matrix-core family is: boost_screen_begin (the candidate counter zeroed), the counting kernel, launch_boost_finish
hipError_t boost_screen_begin(const BoostArgs &a, hipStream_t stream);                            // gw_boost_fp4.hip
hipError_t launch_boost_finish(const BoostArgs &a, bool nine, int compute_units, hipStream_t stream);
hipError_t launch_boost_valu(const BoostArgs &a, bool fast, bool pre, hipStream_t stream);        // gw_boost.hip (everything in one kernel)
hipError_t launch_boost_mfma_both4(const BoostArgs &a, bool pre, int compute_units, hipStream_t stream);   // gw_boost_mfma.hip
hipError_t launch_boost_mfma_seq4(const BoostArgs &a, bool pre, int compute_units, hipStream_t stream);    // gw_boost_seq4.hip
hipError_t launch_boost_mfma_seq9(const BoostArgs &a, bool pre, int compute_units, hipStream_t stream);    // gw_boost_seq9.hip
hipError_t launch_boost_mfma_both9(const BoostArgs &a, bool pre, int compute_units, hipStream_t stream);   // gw_boost_both9.hip
// (the FP4 launchers enqueue k_boost_finish behind the screen when it is one with a pre-filter: compute_units sizes its grid)
hipError_t launch_boost_mfma_fp4(const BoostArgs &a, bool pre, bool nine, int compute_units, hipStream_t stream);  // gw_boost_fp4*.hip
hipError_t launch_boost_mfma_fp4_both(const BoostArgs &a, bool pre, int compute_units, hipStream_t stream);
// every pair in FP64 (pre = false) and the forms with a value per pair (a.dense): gw_boost_fp4_full.hip
hipError_t launch_boost_mfma_fp4_full(const BoostArgs &a, bool pre, bool nine, hipStream_t stream);
hipError_t launch_boost_mfma_fp4_both_full(const BoostArgs &a, bool pre, hipStream_t stream);

}  // namespace gw
