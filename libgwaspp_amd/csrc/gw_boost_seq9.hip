// gw_boost_seq9.hip -- k_boost_mfma_seq<PRE, 3, 1> (see gw_boost_seq.inc)
#include "gw_boost_dev.h"

namespace gw {

#include "gw_boost_seq.inc"

hipError_t launch_boost_mfma_seq9(const BoostArgs &a, bool pre, int compute_units, hipStream_t stream) {
    const dim3 grid((a.n_b + MT - 1) / MT, (a.n_a + 1 * MT - 1) / (1 * MT));
    const bool screen = pre && !a.dense;   // candidates out, k_boost_finish behind the kernel (gw_boost_fp4.hip)
    hipError_t e = screen ? boost_screen_begin(a, stream) : hipSuccess;
    if (e != hipSuccess) return e;
    if (pre) hipLaunchKernelGGL((k_boost_mfma_seq<true, 3, 1>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((k_boost_mfma_seq<false, 3, 1>), grid, dim3(256), 0, stream, a);
    if ((e = hipGetLastError()) != hipSuccess || !screen) return e;
    return launch_boost_finish(a, true, compute_units, stream);
}

}  // namespace gw
