// read_patterns.hip -- which traversal of a row-major 2-plane table reads HBM fastest on MI355X?
// Stand-alone experiment behind the scan kernel's design: pure 16-byte nt loads
// folded into an XOR, no masks, no popcounts.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/read_patterns tools/read_patterns.hip
//   gpurun_out/read_patterns [rows] [units_per_plane]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ u32x4 ldnt(const u32x4 *p) { return __builtin_nontemporal_load(p); }

__device__ __forceinline__ void sink_acc(u32x4 acc, uint32_t *sink) {
    uint32_t x = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    for (int off = 32; off > 0; off >>= 1) x ^= __shfl_xor(x, off, 64);
    if ((threadIdx.x & 63u) == 0 && x == 0xDEADBEEFu) atomicAdd(sink, 1u);
}

// V0: one wave per row (rows strided over all waves), both planes at once, U units per lane in flight
template <int U>
__global__ void v_wave_row(const u32x4 *t, uint64_t stride, uint32_t pstride, uint32_t units, uint32_t rows, uint32_t *sink) {
    const uint32_t lane = threadIdx.x & 63u, wpb = blockDim.x >> 6;
    const uint32_t g = blockIdx.x * wpb + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), tw = gridDim.x * wpb;
    const uint32_t step = 64u * U, full = (units / step) * step;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t r = g; r < rows; r += tw) {
        const u32x4 *p0 = t + (uint64_t)r * stride, *p1 = p0 + pstride;
        for (uint32_t b = 0; b < full; b += step) {
            u32x4 A[U], B[U];
#pragma unroll
            for (int k = 0; k < U; ++k) { A[k] = ldnt(p0 + b + 64u * k + lane); B[k] = ldnt(p1 + b + 64u * k + lane); }
#pragma unroll
            for (int k = 0; k < U; ++k) acc ^= A[k] ^ B[k];
        }
        for (uint32_t u = full + lane; u < units; u += 64u) acc ^= ldnt(p0 + u) ^ ldnt(p1 + u);
    }
    sink_acc(acc, sink);
}

// V1: one wave per row, the row read as ONE contiguous stream (plane 0 then plane 1)
template <int U>
__global__ void v_wave_row_seq(const u32x4 *t, uint64_t stride, uint32_t pstride, uint32_t units, uint32_t rows, uint32_t *sink) {
    const uint32_t lane = threadIdx.x & 63u, wpb = blockDim.x >> 6;
    const uint32_t g = blockIdx.x * wpb + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), tw = gridDim.x * wpb;
    const uint32_t step = 64u * U;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t r = g; r < rows; r += tw) {
        for (int pl = 0; pl < 2; ++pl) {
            const u32x4 *p = t + (uint64_t)r * stride + (pl ? pstride : 0);
            const uint32_t full = (units / step) * step;
            for (uint32_t b = 0; b < full; b += step) {
                u32x4 A[U];
#pragma unroll
                for (int k = 0; k < U; ++k) A[k] = ldnt(p + b + 64u * k + lane);
#pragma unroll
                for (int k = 0; k < U; ++k) acc ^= A[k];
            }
            for (uint32_t u = full + lane; u < units; u += 64u) acc ^= ldnt(p + u);
        }
    }
    sink_acc(acc, sink);
}

// V2: one WORKGROUP per row: wave w reads unit w*64 + lane of every (waves*64)-unit chunk
__global__ void v_block_row(const u32x4 *t, uint64_t stride, uint32_t pstride, uint32_t units, uint32_t rows, uint32_t *sink) {
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const u32x4 *p0 = t + (uint64_t)r * stride, *p1 = p0 + pstride;
        uint32_t u = threadIdx.x;
        for (; u + blockDim.x < units; u += 2u * blockDim.x) {
            const u32x4 a0 = ldnt(p0 + u), b0 = ldnt(p1 + u), a1 = ldnt(p0 + u + blockDim.x), b1 = ldnt(p1 + u + blockDim.x);
            acc ^= a0 ^ b0 ^ a1 ^ b1;
        }
        for (; u < units; u += blockDim.x) acc ^= ldnt(p0 + u) ^ ldnt(p1 + u);
    }
    sink_acc(acc, sink);
}

// V3: flat grid-stride over the whole allocation
__global__ void v_flat(const u32x4 *t, uint64_t n_units, uint32_t *sink) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + 3ull * stride < n_units; i += 4ull * stride) {
        const u32x4 a = ldnt(t + i), b = ldnt(t + i + stride), c = ldnt(t + i + 2 * stride), d = ldnt(t + i + 3 * stride);
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n_units; i += stride) acc ^= ldnt(t + i);
    sink_acc(acc, sink);
}

// V4: one wave per row, each wave owns a CONTIGUOUS range of rows
template <int U>
__global__ void v_wave_row_blocked(const u32x4 *t, uint64_t stride, uint32_t pstride, uint32_t units, uint32_t rows, uint32_t *sink) {
    const uint32_t lane = threadIdx.x & 63u, wpb = blockDim.x >> 6;
    const uint32_t g = blockIdx.x * wpb + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), tw = gridDim.x * wpb;
    const uint32_t per = (rows + tw - 1) / tw, r0 = g * per, r1 = min(rows, r0 + per);
    const uint32_t step = 64u * U, full = (units / step) * step;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t r = r0; r < r1; ++r) {
        const u32x4 *p0 = t + (uint64_t)r * stride, *p1 = p0 + pstride;
        for (uint32_t b = 0; b < full; b += step) {
            u32x4 A[U], B[U];
#pragma unroll
            for (int k = 0; k < U; ++k) { A[k] = ldnt(p0 + b + 64u * k + lane); B[k] = ldnt(p1 + b + 64u * k + lane); }
#pragma unroll
            for (int k = 0; k < U; ++k) acc ^= A[k] ^ B[k];
        }
        for (uint32_t u = full + lane; u < units; u += 64u) acc ^= ldnt(p0 + u) ^ ldnt(p1 + u);
    }
    sink_acc(acc, sink);
}

// V5: a sub-wave of G lanes per row (64/G consecutive rows per wave at a time), S steps per row, the next
// batch's loads issued before the current one is consumed (the short-row scan's traversal)
template <int G, int S>
__global__ void v_subwave(const u32x4 *t, uint64_t stride, uint32_t pstride, uint32_t units, uint32_t rows, uint32_t *sink) {
    constexpr uint32_t RW = 64 / G;
    const uint32_t lane = threadIdx.x & 63u, wpb = blockDim.x >> 6, grp = lane / G, li = lane % G;
    const uint32_t g = blockIdx.x * wpb + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), tw = gridDim.x * wpb;
    const uint32_t nb = (rows + RW - 1) / RW, last = units - 1;
    uint32_t ui[S];
#pragma unroll
    for (int k = 0; k < S; ++k) ui[k] = min((uint32_t)G * k + li, last);
    u32x4 acc = {0, 0, 0, 0};
    auto issue = [&](u32x4 (&A)[S], u32x4 (&B)[S], uint32_t b) {
        const uint32_t r = min(b * RW + grp, rows - 1);
        const u32x4 *p0 = t + (uint64_t)r * stride, *p1 = p0 + pstride;
#pragma unroll
        for (int k = 0; k < S; ++k) { A[k] = ldnt(p0 + ui[k]); B[k] = ldnt(p1 + ui[k]); }
    };
    auto eat = [&](u32x4 (&A)[S], u32x4 (&B)[S]) {
#pragma unroll
        for (int k = 0; k < S; ++k) acc ^= A[k] ^ B[k];
    };
    uint32_t b = g;
    if (b < nb) {
        u32x4 A0[S], B0[S], A1[S], B1[S];
        issue(A0, B0, b);
        for (;;) {
            issue(A1, B1, b + tw); eat(A0, B0); b += tw; if (b >= nb) break;
            issue(A0, B0, b + tw); eat(A1, B1); b += tw; if (b >= nb) break;
        }
    }
    sink_acc(acc, sink);
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    const uint32_t rows = argc > 1 ? std::atoi(argv[1]) : 400000;
    const uint32_t units = argc > 2 ? std::atoi(argv[2]) : 3907;          // N = 500 k
    const uint32_t pstride = (units * 16u + 127u) / 128u * 8u;             // 128-byte multiple, in units
    const uint64_t stride = 2ull * pstride, total_units = stride * rows;
    u32x4 *t;
    uint32_t *sink;
    CK(hipMalloc((void **)&t, total_units * 16ull));
    CK(hipMalloc((void **)&sink, 4));
    CK(hipMemset(t, 0x5A, total_units * 16ull));
    CK(hipMemset(sink, 0, 4));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double useful = (double)rows * 2.0 * units * 16.0;
    std::printf("%u rows x %u units/plane, %.2f GB useful, %d CUs\n", rows, units, useful / 1e9, cus);
    auto report = [&](const char *name, double ms, double bytes) { std::printf("%-52s %8.3f ms %8.0f GB/s\n", name, ms, bytes / ms / 1e6); };
#define RUN(name, kern, grid, block, ...) report(name, time_ms([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, __VA_ARGS__); }, 5), useful)
    RUN("V0 wave/row strided, planes together, U2, 512thr x1", (v_wave_row<2>), cus, 512, t, stride, pstride, units, rows, sink);
    RUN("V0 U4, 512thr x1", (v_wave_row<4>), cus, 512, t, stride, pstride, units, rows, sink);
    RUN("V0 U2, 256thr x1", (v_wave_row<2>), cus, 256, t, stride, pstride, units, rows, sink);
    RUN("V0 U4, 256thr x1", (v_wave_row<4>), cus, 256, t, stride, pstride, units, rows, sink);
    RUN("V0 U8, 256thr x1", (v_wave_row<8>), cus, 256, t, stride, pstride, units, rows, sink);
    RUN("V0 U2, 1024thr x1", (v_wave_row<2>), cus, 1024, t, stride, pstride, units, rows, sink);
    RUN("V0 U1, 1024thr x2", (v_wave_row<1>), cus * 2, 1024, t, stride, pstride, units, rows, sink);
    RUN("V1 wave/row, planes sequential, U4, 512thr x1", (v_wave_row_seq<4>), cus, 512, t, stride, pstride, units, rows, sink);
    RUN("V1 U8, 512thr x1", (v_wave_row_seq<8>), cus, 512, t, stride, pstride, units, rows, sink);
    RUN("V1 U4, 1024thr x1", (v_wave_row_seq<4>), cus, 1024, t, stride, pstride, units, rows, sink);
    RUN("V2 block/row, 512thr x1", v_block_row, cus, 512, t, stride, pstride, units, rows, sink);
    RUN("V2 block/row, 512thr x2", v_block_row, cus * 2, 512, t, stride, pstride, units, rows, sink);
    RUN("V2 block/row, 1024thr x1", v_block_row, cus, 1024, t, stride, pstride, units, rows, sink);
    RUN("V2 block/row, 256thr x4", v_block_row, cus * 4, 256, t, stride, pstride, units, rows, sink);
    RUN("V4 wave/row contiguous row ranges, U2, 512thr x1", (v_wave_row_blocked<2>), cus, 512, t, stride, pstride, units, rows, sink);
    report("V3 flat grid-stride, 512thr x1", time_ms([&] { hipLaunchKernelGGL(v_flat, dim3(cus), dim3(512), 0, 0, t, total_units, sink); }, 5), total_units * 16.0);
    report("V3 flat grid-stride, 1024thr x2", time_ms([&] { hipLaunchKernelGGL(v_flat, dim3(cus * 2), dim3(1024), 0, 0, t, total_units, sink); }, 5), total_units * 16.0);
    report("V3 flat grid-stride, 256thr x8", time_ms([&] { hipLaunchKernelGGL(v_flat, dim3(cus * 8), dim3(256), 0, 0, t, total_units, sink); }, 5), total_units * 16.0);
    if (units <= 80) {
        RUN("V5 sub-wave 16 lanes/row, 5 steps, 256thr x2", (v_subwave<16, 5>), cus * 2, 256, t, stride, pstride, units, rows, sink);
        RUN("V5 sub-wave 16 lanes/row, 5 steps, 256thr x4", (v_subwave<16, 5>), cus * 4, 256, t, stride, pstride, units, rows, sink);
        RUN("V5 sub-wave 16 lanes/row, 5 steps, 256thr x8", (v_subwave<16, 5>), cus * 8, 256, t, stride, pstride, units, rows, sink);
        RUN("V5 sub-wave 32 lanes/row, 3 steps, 256thr x4", (v_subwave<32, 3>), cus * 4, 256, t, stride, pstride, units, rows, sink);
        RUN("V5 sub-wave 32 lanes/row, 3 steps, 256thr x8", (v_subwave<32, 3>), cus * 8, 256, t, stride, pstride, units, rows, sink);
    }
    CK(hipFree(t));
    CK(hipFree(sink));
    return 0;
}
