// valu_rates.hip -- issue rate of the VALU instructions the pair kernels live on (gfx950): the integer ops of the operand
// expansion, the FP32 ops of the per-pair bound (incl. the transcendentals), 64-bit shifts, packed forms; 1..2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_rates tools/valu_rates.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int ITER = 4096;

// 16 independent chains, 16 instructions per loop body x UNR
#define BODY16(OP)                                                                                         \
    OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(uint32_t *out, uint32_t seed) {
    uint32_t r[16], x = seed ^ threadIdx.x, y = seed * 2654435761u + threadIdx.x;
    uint64_t q[16], qx = ((uint64_t)x << 32) | y;  // the 64-bit / packed kinds
#pragma unroll
    for (int i = 0; i < 16; ++i) { r[i] = seed + i; q[i] = (uint64_t)(seed + i) * 0x100000001ull; }
    for (int it = 0; it < ITER; ++it) {
        if (KIND == 0) {
#define OP(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[i]) : "v"(x));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 1) {
#define OP(i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r[i]) : "v"(x));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 2) {
#define OP(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[i]) : "v"(x));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 3) {
#define OP(i) asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96" : "+v"(r[i]) : "v"(x), "v"(y));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 5) {
#define OP(i) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(r[i]));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 6) {
#define OP(i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(q[i]));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 7) {
#define OP(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(x), "v"(y));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 8) {
#define OP(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r[i]) : "v"(x));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 9) {
#define OP(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[i]) : "v"(x));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 10) {
#define OP(i) asm volatile("v_max_f32 %0, 1.0, %0" : "+v"(r[i]));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 11) {
#define OP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 12) {
#define OP(i) asm volatile("v_log_f32 %0, %0" : "+v"(r[i]));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 13) {
#define OP(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r[i]));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 14) {
#define OP(i) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(r[i]));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 15) {
#define OP(i) asm volatile("v_mul_legacy_f32 %0, %1, %0" : "+v"(r[i]) : "v"(x));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 16) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(q[i]) : "v"(qx));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 17) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(q[i]) : "v"(qx));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 18) {
#define OP(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(q[i]) : "v"(qx));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 19) {
#define OP(i) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(q[i]) : "v"(qx));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 20) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r[i]) : "v"(x));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 21) {
#define OP(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(x), "v"(y));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 22) {
#define OP(i) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(x), "v"(y));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 23) {
#define OP(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(r[i]) : "v"(x), "v"(y));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 24) {
#define OP(i) asm volatile("v_rcp_f16 %0, %0" : "+v"(r[i]));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 25) {
#define OP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 26) {
#define OP(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(r[i]));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 27) {
#define OP(i) asm volatile("v_div_fixup_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(x), "v"(y));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 28) {
#define OP(i) asm volatile("v_alignbit_b32 %0, %1, %0, 3" : "+v"(r[i]) : "v"(x));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 29) {
#define OP(i) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(x), "v"(y));
            BODY16(OP) BODY16(OP)
#undef OP
        } else if (KIND == 4) {  // the pair kernel's pattern: t = x & y_i ; acc_i += popc(t)
            uint32_t t;
#define OP(i) asm volatile("v_and_b32 %1, %2, %3\n\tv_bcnt_u32_b32 %0, %1, %0" : "+v"(r[i]), "=&v"(t) : "v"(x), "v"(y));
            BODY16(OP)
#undef OP
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i] + (uint32_t)q[i] + (uint32_t)(q[i] >> 32);
    if (s == 0xDEADBEEFu) out[0] = s;
}

template <int KIND>
static void run(const char *name, int cus, double clock_ghz) {
    uint32_t *d;
    CK(hipMalloc((void **)&d, 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        const int grid = cus * blocks_per_cu * 8;  // 8 rounds
        hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), 0, 0, d, 1u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), 0, 0, d, 1u);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        const double instr_per_wave = (double)ITER * 32.0;
        const double waves = (double)grid * 4.0;
        const double lane_ops = instr_per_wave * waves * 64.0;
        // cycles one SIMD spends per wave-instruction
        const double cyc = ms * 1e-3 * clock_ghz * 1e9 / (instr_per_wave * waves / (cus * 4.0));
        std::printf("%-28s %d wave(s)/SIMD  %7.3f ms  %6.2f T lane-ops/s  %.2f cyc/instr/SIMD (at %.1f GHz)\n", name,
                    blocks_per_cu, ms, lane_ops / ms / 1e9, cyc, clock_ghz);
    }
    CK(hipFree(d));
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate / 1e6;
    std::printf("%s, %d CUs, %.2f GHz\n", p.name, p.multiProcessorCount, ghz);
    run<0>("v_and_b32", p.multiProcessorCount, ghz);
    run<1>("v_bcnt_u32_b32", p.multiProcessorCount, ghz);
    run<2>("v_add_u32", p.multiProcessorCount, ghz);
    run<3>("v_bitop3_b32", p.multiProcessorCount, ghz);
    run<4>("v_and + v_bcnt (dependent)", p.multiProcessorCount, ghz);
    run<5>("v_lshrrev_b32", p.multiProcessorCount, ghz);
    run<6>("v_lshrrev_b64", p.multiProcessorCount, ghz);
    run<7>("v_fma_f32", p.multiProcessorCount, ghz);
    run<8>("v_sub_f32", p.multiProcessorCount, ghz);
    run<9>("v_mul_f32", p.multiProcessorCount, ghz);
    run<10>("v_max_f32", p.multiProcessorCount, ghz);
    run<11>("v_rcp_f32", p.multiProcessorCount, ghz);
    run<12>("v_log_f32", p.multiProcessorCount, ghz);
    run<13>("v_cvt_f32_u32", p.multiProcessorCount, ghz);
    run<14>("v_cvt_f32_u32 sdwa WORD_1", p.multiProcessorCount, ghz);
    run<15>("v_mul_legacy_f32", p.multiProcessorCount, ghz);
    run<16>("v_pk_fma_f32", p.multiProcessorCount, ghz);
    run<17>("v_pk_mul_f32", p.multiProcessorCount, ghz);
    run<18>("v_pk_add_f32", p.multiProcessorCount, ghz);
    run<19>("v_fma_f64", p.multiProcessorCount, ghz);
    run<20>("v_cndmask_b32", p.multiProcessorCount, ghz);
    run<21>("v_fma_f32 (2 sgpr-free, literal)", p.multiProcessorCount, ghz);
    run<22>("v_pk_fma_f16", p.multiProcessorCount, ghz);
    run<23>("v_mad_u32_u24", p.multiProcessorCount, ghz);
    run<24>("v_rcp_f16", p.multiProcessorCount, ghz);
    run<25>("v_exp_f32", p.multiProcessorCount, ghz);
    run<26>("v_rsq_f32", p.multiProcessorCount, ghz);
    run<27>("v_div_fixup_f32", p.multiProcessorCount, ghz);
    run<28>("v_alignbit_b32", p.multiProcessorCount, ghz);
    run<29>("v_and_or_b32", p.multiProcessorCount, ghz);
    return 0;
}
