// mfma_agpr_probe.hip -- does it matter to the shared vector issue port WHERE an MFMA's accumulators live?
// tools/mfma_rate_probe.hip found that a v_mfma_scale_f32_32x32x64_f8f6f4 (FP4 operands) keeps other vector instructions from
// issuing for ~16 of its 36 cycles; the compiler puts the pair kernels' accumulators in ordinary VGPRs (num_agpr = 0).  Here the same
// stream by name, once with the accumulators in VGPRs ("v") and once in accumulation registers ("a"), 8 independent
// accumulators per wave, with 0 / 4 / 6 / 8 independent vector instructions per MFMA, 1 / 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_agpr_probe.hip -o /tmp/mfma_agpr_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(CONS, ACC) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4" \
                                     : "+" CONS(ACC) : "v"(a), "v"(b), "v"(scale))
#define VALU2() do { asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x78" : "+v"(x0) : "v"(x1), "v"(x2)); \
                     asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x78" : "+v"(x3) : "v"(x1), "v"(x2)); } while (0)

template <bool AGPR, int NVALU>
__global__ void __launch_bounds__(256, 2) k_rate(const uint32_t *in, float *out, int iters, int scale) {
    const uint32_t t = threadIdx.x;
    i32x4 a, b;
    for (int j = 0; j < 4; ++j) { a[j] = (int)(in[(t * 8 + j) & 4095] & 0x22222222u); b[j] = (int)(in[(t * 8 + 4 + j) & 4095] & 0x22222222u); }
    f32x16 acc[8];
    for (int m = 0; m < 8; ++m)
        for (int q = 0; q < 16; ++q) acc[m][q] = 0.0f;
    uint32_t x0 = in[t & 4095], x1 = in[(t + 1) & 4095], x2 = in[(t + 2) & 4095], x3 = in[(t + 3) & 4095];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (AGPR) MFMA("a", acc[m]); else MFMA("v", acc[m]);
            if (NVALU >= 2) VALU2();
            if (NVALU >= 4) VALU2();
            if (NVALU >= 6) VALU2();
            if (NVALU >= 8) VALU2();
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float s = 0.0f;
    for (int m = 0; m < 8; ++m)
        for (int q = 0; q < 16; ++q) s += acc[m][q];
    out[blockIdx.x * 256 + t] = s + (float)(x0 ^ x3);
}

template <bool AGPR, int NVALU>
void run(const uint32_t *din, float *dout, int wgs_per_cu, int cus) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = cus * wgs_per_cu;
    hipLaunchKernelGGL((k_rate<AGPR, NVALU>), dim3(grid), dim3(256), 0, 0, din, dout, 10, 127);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<AGPR, NVALU>), dim3(grid), dim3(256), 0, 0, din, dout, iters, 127);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)grid * 4 * iters * 8 * 2.0 * 32 * 32 * 64;
    printf("accumulators in %s, %d vector instructions per MFMA   %d wave(s)/SIMD  %8.3f ms  %5.1f %% of 10 POP/s   %.1f nominal cycles per MFMA per SIMD\n",
           AGPR ? "AGPRs" : "VGPRs", NVALU, wgs_per_cu, ms, ops / ms / 1e9 / 100.0, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wgs_per_cu));
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
        run<false, 0>(din, dout, w, cus); run<true, 0>(din, dout, w, cus);
        run<false, 4>(din, dout, w, cus); run<true, 4>(din, dout, w, cus);
        run<false, 6>(din, dout, w, cus); run<true, 6>(din, dout, w, cus);
        run<false, 8>(din, dout, w, cus); run<true, 8>(din, dout, w, cus);
    }
    return 0;
}
