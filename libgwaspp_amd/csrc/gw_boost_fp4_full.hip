// gw_boost_fp4_full.hip -- the FP4 pair kernels in the forms the default screen does NOT take: every pair's FP64 measure in
// the kernel (pre-filter off, extremes asked for, or the repeat after a violated error model), and the forms that write
// a value per pair (gw_boost_measures; the diagnostic bounds of GW_OPT_BOOST_PREFILTER 2 .. 5).  Same tile code
// (gw_boost_fp4.inc), a translation unit of its own so that it compiles beside gw_boost_fp4.hip.
#include "gw_boost_fp4.inc"

namespace gw {

hipError_t launch_boost_mfma_fp4_both_full(const BoostArgs &a, bool pre, hipStream_t stream) {
    const dim3 grid = fp4_grid(a, MT);
    if (pre) {  // diagnostic: the in-kernel bound (first level if its constants are set) per pair
        if (!a.dense) return hipErrorInvalidValue;
        if (a.quad.c_m > 0.0f && a.quad.c_d != 0.0f) hipLaunchKernelGGL((k_boost_mfma_fp4_both<true, 2, true>), grid, dim3(256), 0, stream, a);
        else if (a.quad.c_m > 0.0f) hipLaunchKernelGGL((k_boost_mfma_fp4_both<true, 1, true>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_boost_mfma_fp4_both<true, 0, true>), grid, dim3(256), 0, stream, a);
    } else if (a.dense) {
        hipLaunchKernelGGL((k_boost_mfma_fp4_both<false, 0, true>), grid, dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL((k_boost_mfma_fp4_both<false, 0, false>), grid, dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_boost_mfma_fp4_full(const BoostArgs &a, bool pre, bool nine, hipStream_t stream) {
    if (pre && !a.dense) return hipErrorInvalidValue;
    if (nine) {
        const dim3 grid = fp4_grid(a, MT);
        if (pre && a.quad.c_m > 0.0f && a.quad.c_d != 0.0f) hipLaunchKernelGGL((k_boost_mfma_fp4<true, 3, 1, 2, true>), grid, dim3(256), 0, stream, a);
        else if (pre && a.quad.c_m > 0.0f) hipLaunchKernelGGL((k_boost_mfma_fp4<true, 3, 1, 1, true>), grid, dim3(256), 0, stream, a);
        else if (pre) hipLaunchKernelGGL((k_boost_mfma_fp4<true, 3, 1, 0, true>), grid, dim3(256), 0, stream, a);
        else if (a.dense) hipLaunchKernelGGL((k_boost_mfma_fp4<false, 3, 1, 0, true>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_boost_mfma_fp4<false, 3, 1, 0, false>), grid, dim3(256), 0, stream, a);
    } else {
        const dim3 grid = fp4_grid(a, 2 * MT);
        if (pre && a.quad.c_m > 0.0f && a.quad.c_d != 0.0f) hipLaunchKernelGGL((k_boost_mfma_fp4<true, 2, 2, 2, true>), grid, dim3(256), 0, stream, a);
        else if (pre && a.quad.c_m > 0.0f) hipLaunchKernelGGL((k_boost_mfma_fp4<true, 2, 2, 1, true>), grid, dim3(256), 0, stream, a);
        else if (pre) hipLaunchKernelGGL((k_boost_mfma_fp4<true, 2, 2, 0, true>), grid, dim3(256), 0, stream, a);
        else if (a.dense) hipLaunchKernelGGL((k_boost_mfma_fp4<false, 2, 2, 0, true>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_boost_mfma_fp4<false, 2, 2, 0, false>), grid, dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

}  // namespace gw
