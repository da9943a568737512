// valu_chain_probe.hip -- how many INDEPENDENT instructions a wave needs in flight for the vector ALU's issue rate (gfx950):
// a chain of dependent v_fma_f32 / v_mul_f32 / v_rcp_f32 / v_bitop3_b32, 1 .. 16 chains per lane, 1 and 2 waves per SIMD.
// tools/valu_rates.hip gives the rates with 16 chains; the per-pair bound of the pair screen is long dependent chains at
// 2 waves per SIMD, so this is the number that prices it.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_chain_probe tools/valu_chain_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int ITER = 2048;

template <int KIND, int CH>
__global__ void __launch_bounds__(256) k_chain(float *out, float seed) {
    float r[CH], x = seed + threadIdx.x * 1e-9f, y = 1.0f - seed * 1e-9f;
#pragma unroll
    for (int i = 0; i < CH; ++i) r[i] = seed + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int rep = 0; rep < 32 / CH; ++rep) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(x));
                else if (KIND == 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[i]) : "v"(y));
                else if (KIND == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
                else if (KIND == 3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r[i]) : "v"(y), "v"(x));
                else if (KIND == 4) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r[i]) : "v"(y));
                else if (KIND == 5) {  // one v_rcp_f32 and three v_fma_f32 on other registers: do they overlap?
                    if ((i & 3) == 0) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(x));
                } else if (KIND == 6) {  // one v_rcp_f32 and one v_fma_f32
                    if ((i & 1) == 0) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(x));
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += r[i];
    if (s == 123.456f) out[0] = s;
}

template <int KIND, int CH>
static void run(const char *name, int cus, double ghz) {
    float *d;
    CK(hipMalloc((void **)&d, 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int wps = 1; wps <= 2; ++wps) {
        const int grid = cus * wps * 8;
        hipLaunchKernelGGL((k_chain<KIND, CH>), dim3(grid), dim3(256), 0, 0, d, 1.0f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_chain<KIND, CH>), dim3(grid), dim3(256), 0, 0, d, 1.0f);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        const double instr_per_wave = (double)ITER * 32.0, waves = (double)grid * 4.0;
        const double cyc = ms * 1e-3 * ghz * 1e9 / (instr_per_wave * waves / (cus * 4.0));
        std::printf("%-14s %2d chain(s)  %d wave(s)/SIMD  %.2f cycles per instruction and SIMD\n", name, CH, wps, cyc);
    }
    CK(hipFree(d));
}

template <int KIND>
static void all(const char *name, int cus, double ghz) {
    run<KIND, 1>(name, cus, ghz);
    run<KIND, 2>(name, cus, ghz);
    run<KIND, 4>(name, cus, ghz);
    run<KIND, 8>(name, cus, ghz);
    run<KIND, 16>(name, cus, ghz);
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate / 1e6;
    std::printf("%s, %d CUs, %.2f GHz\n", p.name, p.multiProcessorCount, ghz);
    all<0>("v_fma_f32", p.multiProcessorCount, ghz);
    all<1>("v_mul_f32", p.multiProcessorCount, ghz);
    all<2>("v_rcp_f32", p.multiProcessorCount, ghz);
    all<3>("v_bitop3_b32", p.multiProcessorCount, ghz);
    all<4>("v_sub_f32", p.multiProcessorCount, ghz);
    run<5, 16>("rcp + 3 fma", p.multiProcessorCount, ghz);   // serial: (8.1 + 3 x 2.5) / 4 = 3.9 per instruction; overlapped: ~2.5
    run<6, 16>("rcp + 1 fma", p.multiProcessorCount, ghz);   // serial: (8.1 + 2.5) / 2 = 5.3; overlapped: ~4.05
    return 0;
}
