// pk_rate_probe.hip -- do the packed FP32 vector instructions (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32) issue at the rate of
// the scalar-per-lane forms on gfx950, i.e. twice the multiply-adds per instruction slot?  16 independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256, 2) k_rate(float *out, int iters, float a, float b) {
    float x[16]; f2 y[16];
    for (int i = 0; i < 16; ++i) { x[i] = (float)(threadIdx.x + i); y[i] = f2{x[i], x[i] + 0.5f}; }
    const f2 a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
            else y[i] = __builtin_elementwise_fma(y[i], a2, b2);
        }
    }
    float s = 0.0f;
    for (int i = 0; i < 16; ++i) s += MODE == 0 ? x[i] : y[i][0] + y[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, float *dout, int cus) {
    const int iters = 20000, grid = cus * 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate<MODE>), dim3(grid), dim3(256), 0, 0, dout, 10, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<MODE>), dim3(grid), dim3(256), 0, 0, dout, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instrs_per_simd = (double)iters * 16 * 2;  // 2 waves per SIMD
    printf("%-28s %8.3f ms   %.2f cycles per vector instruction and SIMD at 2.4 GHz   %.1f TFLOP/s\n", name, ms,
           ms * 1e-3 * 2.4e9 / instrs_per_simd, (double)grid * 256 * iters * 16 * (MODE ? 4.0 : 2.0) / ms / 1e9);
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    float *dout;
    (void)hipMalloc(&dout, (size_t)p.multiProcessorCount * 2 * 256 * 4);
    run<0>("v_fma_f32", dout, p.multiProcessorCount);
    run<1>("v_pk_fma_f32", dout, p.multiProcessorCount);
    return 0;
}
