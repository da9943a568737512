// mfma_rate_probe.hip -- what the matrix core sustains for the FP4 forms the pair kernels use, measured, not assumed:
//   v_mfma_scale_f32_32x32x64_f8f6f4 (scaled, what k_boost_mfma_fp4 issues) vs v_mfma_f32_32x32x64_f8f6f4 (unscaled),
//   FP4 operands, 8 independent accumulators per wave, 1 / 2 waves per SIMD, with 0 / 2 / 4 independent vector
//   instructions between two MFMAs (v_bitop3-like work on other registers).
// Prints TOP/s (2 * 32 * 32 * 64 per MFMA) and the share of the 10 POP/s FP4 peak.  build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool SCALED, int NVALU>
__global__ void __launch_bounds__(256, 2) k_rate(const uint32_t *in, float *out, int iters, int scale) {
    const uint32_t t = threadIdx.x;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
    uint32_t w[8];
    for (int j = 0; j < 8; ++j) w[j] = in[(t * 8 + j) & 4095];
    for (int j = 0; j < 4; ++j) { a[j] = (int)(w[j] & 0x22222222u); b[j] = (int)(w[4 + j] & 0x22222222u); }
    f32x16 acc[8];
    for (int m = 0; m < 8; ++m)
        for (int q = 0; q < 16; ++q) acc[m][q] = 0.0f;
    uint32_t x0 = w[0], x1 = w[1], x2 = w[2], x3 = w[3];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (SCALED) acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[m], 4, 4, 0, scale, 0, scale);
            else acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[m], 4, 4, 0, 0, 0, 0);  // zero scales: the unscaled instruction
            if (NVALU >= 1) x0 = (x0 & 0x11111111u) ^ (x1 >> 1);
            if (NVALU >= 2) x1 = (x1 & 0x22222222u) ^ (x2 >> 1);
            if (NVALU >= 3) x2 = (x2 & 0x44444444u) ^ (x3 >> 1);
            if (NVALU >= 4) x3 = (x3 & 0x11111111u) ^ (x0 >> 1);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (NVALU) __builtin_amdgcn_sched_group_barrier(0x002, 2 * NVALU, 0);
        }
    }
    float s = 0.0f;
    for (int m = 0; m < 8; ++m)
        for (int q = 0; q < 16; ++q) s += acc[m][q];
    out[blockIdx.x * 256 + t] = s + (float)(x0 ^ x1 ^ x2 ^ x3);
}

template <bool SCALED, int NVALU>
void run(const char *name, const uint32_t *din, float *dout, int wgs_per_cu, int cus) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = cus * wgs_per_cu;
    hipLaunchKernelGGL((k_rate<SCALED, NVALU>), dim3(grid), dim3(256), 0, 0, din, dout, 10, 127);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<SCALED, NVALU>), dim3(grid), dim3(256), 0, 0, din, dout, iters, 127);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)grid * 4 * iters * 8 * 2.0 * 32 * 32 * 64;
    printf("%-44s %d wave(s)/SIMD  %8.3f ms  %7.1f TOP/s  %5.1f %% of 10 POP/s   %.1f cycles per MFMA per SIMD at 2.4 GHz\n", name,
           wgs_per_cu, ms, ops / ms / 1e9, ops / ms / 1e9 / 100.0, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wgs_per_cu));
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    uint32_t *din; float *dout;
    hipMalloc(&din, 4096 * 4); hipMalloc(&dout, (size_t)cus * 2 * 256 * 4);
    hipMemset(din, 0x5a, 4096 * 4);
    printf("%d CUs, clock %.0f MHz\n", cus, p.clockRate / 1e3);
    for (int w = 1; w <= 2; ++w) {
        run<false, 0>("unscaled, MFMA only", din, dout, w, cus);
        run<true, 0>("scaled, MFMA only", din, dout, w, cus);
        run<false, 2>("unscaled + 4 vector instructions per MFMA", din, dout, w, cus);
        run<true, 2>("scaled + 4 vector instructions per MFMA", din, dout, w, cus);
        run<true, 3>("scaled + 6 vector instructions per MFMA", din, dout, w, cus);
        run<true, 4>("scaled + 8 vector instructions per MFMA", din, dout, w, cus);
    }
    return 0;
}
