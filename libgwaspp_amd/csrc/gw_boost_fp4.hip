// gw_boost_fp4.hip -- the pre-filtered pair screens on FP4 operands (the kernels the default dispatch reaches) and
// k_boost_finish, which takes their candidates to the second level, the FP64 measure and the hit list.
// (gw_boost_fp4.inc has the tile code; gw_boost_fp4_full.hip the every-pair-in-FP64 and dense forms)
#include "gw_boost_fp4.inc"

namespace gw {

// One lane per candidate {snp_a, snp_b, cells of the cases, cells of the controls} (BoostCandidateList): the two 3 x 3
// tables, the bound proper (boost_bound_split / boost_bound) -- the SECOND level where the screen ran the logarithm-free
// first level, and then the gate of the FP64 measure --, computeBoost's measure (epistasis_func.cpp:424-470), the run-time
// checks of both bounds' error models (tally[2]) and the hit list (:482-484).  A persistent grid strides over the
// list; the count is read from device memory, so nothing about a screen returns to the host between the two kernels.
template <int NV>
__global__ void __launch_bounds__(256) k_boost_finish(BoostArgs a) {
    constexpr int NP = NV * NV, REC = 2 + 2 * NP;
    const unsigned long long total = *a.cand_count;
    const unsigned long long n = total < a.cand_cap ? total : a.cand_cap;
    // candidates that did not fit: reported like a violated error model -- the host repeats the block in FP64
    if (total > a.cand_cap && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&a.tally[2], total - a.cand_cap);
    const bool first_level = a.quad.c_m > 0.0f;
    const float nf = (float)a.n_individs, inv_n = 1.0f / (float)a.n_individs;
    for (unsigned long long k = (unsigned long long)blockIdx.x * 256u + threadIdx.x; k < n; k += (unsigned long long)gridDim.x * 256u) {
        const uint32_t *rec = a.cand + k * REC;
        const uint32_t i = rec[0], j = rec[1];
        uint32_t cells[2][NP], ca[9], co[9];
#pragma unroll
        for (int c = 0; c < NP; ++c) { cells[0][c] = rec[2 + c]; cells[1][c] = rec[2 + NP + c]; }
        const SnpMargin m1 = a.mar[i], m2 = a.mar[j];
        boost_tables_of<NV>(cells, m1.cases, m1.ctrls, m2.cases, m2.ctrls, ca, co);
        const SnpMarginF f1 = a.marf[i], f2 = a.marf[j];
        float err2;
        const float bound2 = NV == 2 ? boost_bound_split(ca, co, f1, f2, nf, a.lg_n, err2) : boost_bound(ca, co, f1, f2, nf, inv_n, err2);
        // what the first level let through meets the bound proper before any FP64 (without a first level this bound is
        // what sent the pair here, recomputed for the check below)
        if (first_level && bound_rules_out(bound2, err2, a.bound_threshold)) continue;
        const double measure = boost_measure(ca, co, m1, m2, a.n_individs, a.frac, a.xlogx);
        check_bound(measure, bound2, err2, a.tally);
        if (first_level) {
            float err1, bound1;
            if constexpr (NV == 2)
                bound1 = a.quad.c_d != 0.0f ? boost_bound_quad<true>(ca, co, f1, f2, nf, a.quad, err1)
                                            : boost_bound_quad<false>(ca, co, f1, f2, nf, a.quad, err1);
            else
                bound1 = a.quad.c_d != 0.0f ? boost_bound_quad9<true>(ca, co, f1, f2, a.quad, err1)
                                            : boost_bound_quad9<false>(ca, co, f1, f2, a.quad, err1);
            check_upper_bound(measure, bound1, err1, a.tally);
        }
        if (measure > a.threshold) boost_append_hit(a, i, j, measure);
    }
}

hipError_t launch_boost_finish(const BoostArgs &a, bool nine, int compute_units, hipStream_t stream) {
    const dim3 grid((unsigned)(compute_units > 0 ? compute_units : 256) * 4u);
    if (nine) hipLaunchKernelGGL((k_boost_finish<3>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((k_boost_finish<2>), grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}

// a screen = the candidate counter zeroed, the counting kernel, the finishing kernel, all on `stream`
hipError_t boost_screen_begin(const BoostArgs &a, hipStream_t stream) {
    if (!a.cand || !a.cand_count || !a.cand_cap) return hipErrorInvalidValue;  // launch_boost's caller sizes the list
    return hipMemsetAsync(a.cand_count, 0, sizeof(unsigned long long), stream);
}

hipError_t launch_boost_mfma_fp4_both(const BoostArgs &a, bool pre, int compute_units, hipStream_t stream) {
    if (!pre || a.dense) return launch_boost_mfma_fp4_both_full(a, pre, stream);
    hipError_t e = boost_screen_begin(a, stream);
    if (e != hipSuccess) return e;
    const dim3 grid = fp4_grid(a, MT);
    if (a.quad.c_m > 0.0f && a.quad.c_d != 0.0f) hipLaunchKernelGGL((k_boost_mfma_fp4_both<true, 2, false>), grid, dim3(256), 0, stream, a);
    else if (a.quad.c_m > 0.0f) hipLaunchKernelGGL((k_boost_mfma_fp4_both<true, 1, false>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((k_boost_mfma_fp4_both<true, 0, false>), grid, dim3(256), 0, stream, a);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return launch_boost_finish(a, false, compute_units, stream);
}

hipError_t launch_boost_mfma_fp4(const BoostArgs &a, bool pre, bool nine, int compute_units, hipStream_t stream) {
    if (!pre || a.dense) return launch_boost_mfma_fp4_full(a, pre, nine, stream);
    hipError_t e = boost_screen_begin(a, stream);
    if (e != hipSuccess) return e;
    if (nine) {
        const dim3 grid = fp4_grid(a, MT);
        if (a.quad.c_m > 0.0f && a.quad.c_d != 0.0f) hipLaunchKernelGGL((k_boost_mfma_fp4<true, 3, 1, 2, false>), grid, dim3(256), 0, stream, a);
        else if (a.quad.c_m > 0.0f) hipLaunchKernelGGL((k_boost_mfma_fp4<true, 3, 1, 1, false>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_boost_mfma_fp4<true, 3, 1, 0, false>), grid, dim3(256), 0, stream, a);
    } else {
        const dim3 grid = fp4_grid(a, 2 * MT);
        if (a.quad.c_m > 0.0f && a.quad.c_d != 0.0f) hipLaunchKernelGGL((k_boost_mfma_fp4<true, 2, 2, 2, false>), grid, dim3(256), 0, stream, a);
        else if (a.quad.c_m > 0.0f) hipLaunchKernelGGL((k_boost_mfma_fp4<true, 2, 2, 1, false>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_boost_mfma_fp4<true, 2, 2, 0, false>), grid, dim3(256), 0, stream, a);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return launch_boost_finish(a, nine, compute_units, stream);
}

}  // namespace gw
