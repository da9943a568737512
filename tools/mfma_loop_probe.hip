// mfma_loop_probe.hip -- the counting loop of k_boost_mfma_fp4<., 2, 2> (gw_boost_fp4.inc) taken apart: the same 32 MFMAs
// per pair of chunks on 8 accumulator tiles, with its ingredients switched on one at a time, to find which of them takes
// the matrix cores from the 36 nominal cycles per MFMA of a pure stream to the ~47 the kernel's steady state shows
// (tools/ab_boost.py --samples 120000: 6.7 POP/s with one or with two waves per SIMD).
//   bit 0  the MFMAs' operands ARE the expansion results (else: constant operands, the expansion goes to other registers)
//   bit 1  the sign-bit group's shifts (k = 3)
//   bit 2  operand rows refilled from memory inside the loop (a 2 MB table, L2-resident), else loaded once
//   bit 3  sched_group_barrier interleave (1 MFMA : 4 VALU)
//   bit 4  the expansion mask as a VGPR instead of an SGPR
//   bit 5  the refills addressed as uniform base (SGPR pair) + per-lane 32-bit offset instead of per-lane 64-bit pointers
//   bit 6  only the B row is refilled (2 of the 6 loads per 32 MFMAs), bit 7  only the A rows (4 of 6)
//   bit 8  the refills as dwordx2 loads (twice as many instructions, the same bytes)
//   bit 9  with bit 7: the B row comes from LDS (ds_read_b128 of a static buffer) instead of global memory -- what sharing
//          operand rows between the waves of a workgroup through LDS would turn a third of the refills into
//   bit 10 the B operands' expansion SHARED between the two waves of a workgroup that use the same B rows: a wave expands one
//          of the pair's two chunks (4 instead of 8 v_bitop3 per sample group), writes it to LDS (ds_write_b128), barrier, and
//          reads its partner's (ds_read_b128) -- not software-pipelined: the exchange sits in front of the group's MFMAs
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_loop_probe tools/mfma_loop_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <type_traits>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int MODE>
__global__ void __launch_bounds__(256, 2) k_loop(const u32x4 *tbl, float *out, int rounds, int sc_half, int sc_one, int sc_two,
                                                uint32_t chunk_mask) {
    constexpr bool FEED = MODE & 1, SHIFT = MODE & 2, LOADS = MODE & 4, SCHED = MODE & 8, VMASK = MODE & 16, SADDR = MODE & 32, ONLY_B = MODE & 64, ONLY_A = MODE & 128, X2 = MODE & 256, B_LDS = MODE & 512, B_SHARE = MODE & 1024;
    __shared__ u32x4 s_share[B_SHARE ? 2 * 2 * 2 * 64 : 1];  // [parity][wx][wy][lane]
    __shared__ u32x4 s_b[B_LDS ? 256 * 4 : 1];
    if (B_LDS) { for (int k = 0; k < 4; ++k) s_b[k * 256 + threadIdx.x] = tbl[k * 256 + threadIdx.x]; __syncthreads(); }
    constexpr int TA = 2, NV = 2, NP = 4, NT = 8, RING = 4;
    const uint32_t t = threadIdx.x, lane = t & 63u, fr = lane & 31u, fh = lane >> 5, wave = t >> 6;
    const u32x4 *ga[TA], *gb;
    for (int ta = 0; ta < TA; ++ta) ga[ta] = tbl + ((blockIdx.x * 7u + wave * 3u + ta) & 15u) * 128u * 64u + fr * 2u + fh;
    gb = tbl + ((blockIdx.x * 5u + wave + 9u) & 15u) * 128u * 64u + fr * 2u + fh;
    // SADDR: the row block is wave-uniform (readfirstlane), the lane adds 16 (2 fr + fh) bytes
    const uint32_t lane_off = (fr * 2u + fh) * 16u;
    uint64_t sa[TA], sb;
    for (int ta = 0; ta < TA; ++ta)
        sa[ta] = (uint64_t)tbl + (uint64_t)__builtin_amdgcn_readfirstlane((int)(((blockIdx.x * 7u + wave * 3u + ta) & 15u) * 128u * 64u)) * 16u;
    sb = (uint64_t)tbl + (uint64_t)__builtin_amdgcn_readfirstlane((int)(((blockIdx.x * 5u + wave + 9u) & 15u) * 128u * 64u)) * 16u;
    f32x16 acc[TA][NP];
    for (int ta = 0; ta < TA; ++ta)
        for (int c = 0; c < NP; ++c)
            for (int r = 0; r < 16; ++r) acc[ta][c][r] = 0.0f;
    u32x4 ua[RING][TA], ub[RING];
    for (int k = 0; k < RING; ++k) {
        for (int ta = 0; ta < TA; ++ta) ua[k][ta] = ga[ta][(uint64_t)k * 128u];
        ub[k] = gb[(uint64_t)k * 128u];
    }
    i32x8 ca = {0, 0, 0, 0, 0, 0, 0, 0}, cb = ca;  // constant operands of the non-FEED form
    for (int j = 0; j < 4; ++j) { ca[j] = (int)(ua[0][0][j] & 0x22222222u); cb[j] = (int)(ub[0][j] & 0x22222222u); }
    uint32_t sink = 0;
    auto cls = [](const u32x4 &v, int c, int w) -> uint32_t {
        const uint32_t p0 = v[w], p1 = v[2 + w];
        return c == 0 ? (p0 & ~p1) : (p0 & p1);
    };
    auto down3 = [](const u32x4 &v) -> u32x4 { return u32x4{v[0] >> 3, v[1] >> 3, v[2] >> 3, v[3] >> 3}; };
    auto chunk_pair = [&](const u32x4 (&va0)[TA], const u32x4 (&va1)[TA], const u32x4 &vb0, const u32x4 &vb1) {
        static_for<0, 4>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const int sc = k == 1 ? sc_one : (k == 2 ? sc_half : sc_two);
            uint32_t mask = 0x11111111u << (k < 3 ? k : 0);
            if (VMASK) mask ^= (lane >> 7);  // lane < 64: zero, but the compiler cannot know -> a VGPR
            u32x4 wa0[TA], wa1[TA];
            for (int ta = 0; ta < TA; ++ta) {
                wa0[ta] = (k == 3 && SHIFT) ? down3(va0[ta]) : va0[ta];
                wa1[ta] = (k == 3 && SHIFT) ? down3(va1[ta]) : va1[ta];
            }
            const u32x4 wb0 = (k == 3 && SHIFT) ? down3(vb0) : vb0, wb1 = (k == 3 && SHIFT) ? down3(vb1) : vb1;
            i32x8 xa[TA][NV], yb[NV];
            for (int c = 0; c < NV; ++c) {
                for (int ta = 0; ta < TA; ++ta) {
                    xa[ta][c] = i32x8{(int)(cls(wa0[ta], c, 0) & mask), (int)(cls(wa0[ta], c, 1) & mask),
                                      (int)(cls(wa1[ta], c, 0) & mask), (int)(cls(wa1[ta], c, 1) & mask), 0, 0, 0, 0};
                }
                if (!B_SHARE)
                    yb[c] = i32x8{(int)(cls(wb0, c, 0) & mask), (int)(cls(wb0, c, 1) & mask), (int)(cls(wb1, c, 0) & mask),
                                  (int)(cls(wb1, c, 1) & mask), 0, 0, 0, 0};
            }
            if (B_SHARE) {
                const uint32_t wy = wave >> 1, wx = wave & 1u;
                const u32x4 mine_raw = wy ? wb1 : wb0;   // (four v_cndmask per sample group here; a real kernel would branch once per wave)
                const u32x4 own = u32x4{cls(mine_raw, 0, 0) & mask, cls(mine_raw, 0, 1) & mask, cls(mine_raw, 1, 0) & mask, cls(mine_raw, 1, 1) & mask};
                u32x4 *slot = s_share + ((k & 1) * 4) * 64;
                slot[(wx * 2 + wy) * 64 + lane] = own;
                __syncthreads();
                const u32x4 oth = slot[(wx * 2 + (wy ^ 1u)) * 64 + lane];
                // (the order of the four dwords along K is the same for every wave's A operands: chunk of wy first)
                yb[0] = i32x8{(int)own[0], (int)own[1], (int)oth[0], (int)oth[1], 0, 0, 0, 0};
                yb[1] = i32x8{(int)own[2], (int)own[3], (int)oth[2], (int)oth[3], 0, 0, 0, 0};
            }
            if (!FEED) {  // the expansion results are kept alive through one more instruction each... no: folded into `sink` at the end of the step
                uint32_t s2 = 0;
                for (int c = 0; c < NV; ++c) {
                    for (int ta = 0; ta < TA; ++ta)
                        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(xa[ta][c][j]));
                    for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(yb[c][j]));
                }
                sink ^= s2;
            }
            for (int ta = 0; ta < TA; ++ta)
                for (int x = 0; x < NV; ++x)
                    for (int y = 0; y < NV; ++y)
                        acc[ta][NV * x + y] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                            FEED ? xa[ta][x] : ca, FEED ? yb[y] : cb, acc[ta][NV * x + y], 4, 4, 0, sc, 0, sc);
            if (SCHED) {
                for (int m = 0; m < NT; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
            }
        });
    };
    uint32_t c0 = 0;
    for (int it = 0; it < rounds; ++it, c0 += RING) {
        static_for<0, RING / 2>([&](auto kc) {
            constexpr int k = 2 * decltype(kc)::value;
            chunk_pair(ua[k], ua[k + 1], ub[k], ub[k + 1]);
            if (LOADS) {
                for (int d = 0; d < 2; ++d) {
                    const uint64_t nxt = (uint64_t)((c0 + k + d + RING) & chunk_mask) * 128u;
                    if (SADDR) {
                        for (int ta = 0; ta < TA; ++ta) ua[k + d][ta] = *(const u32x4 *)((const char *)(sa[ta] + nxt * 16u) + lane_off);
                        ub[k + d] = *(const u32x4 *)((const char *)(sb + nxt * 16u) + lane_off);
                    } else if (X2) {
                        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                        for (int ta = 0; ta < TA; ++ta) {
                            const u32x2 lo = ((const u32x2 *)(ga[ta] + nxt))[0], hi = ((const u32x2 *)(ga[ta] + nxt))[1];
                            ua[k + d][ta] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                        }
                        const u32x2 lo = ((const u32x2 *)(gb + nxt))[0], hi = ((const u32x2 *)(gb + nxt))[1];
                        ub[k + d] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                    } else {
                        if (!ONLY_B) for (int ta = 0; ta < TA; ++ta) ua[k + d][ta] = ga[ta][nxt];
                        if (!ONLY_A) ub[k + d] = gb[nxt];
                        if (B_LDS) ub[k + d] = s_b[((c0 + k + d) & 3u) * 256u + threadIdx.x];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    }
    float s = 0.0f;
    for (int ta = 0; ta < TA; ++ta)
        for (int c = 0; c < NP; ++c)
            for (int q = 0; q < 16; ++q) s += acc[ta][c][q];
    out[blockIdx.x * 256 + t] = s + (float)sink;
}

template <int MODE>
void run(const char *name, const u32x4 *dtbl, float *dout, int wgs_per_cu, int cus) {
    const int rounds = 400;  // x 64 MFMAs
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = cus * wgs_per_cu;
    hipLaunchKernelGGL((k_loop<MODE>), dim3(grid), dim3(256), 0, 0, dtbl, dout, 4, 126, 127, 128, 63u);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_loop<MODE>), dim3(grid), dim3(256), 0, 0, dtbl, dout, rounds, 126, 127, 128, 63u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)rounds * 64;
    const double ops = (double)grid * 4 * mfmas * 2.0 * 32 * 32 * 64;
    printf("%-76s %d wave(s)/SIMD  %8.3f ms  %5.1f %% of 10 POP/s   %.1f nominal cycles per MFMA per SIMD\n", name, wgs_per_cu, ms,
           ops / ms / 1e9 / 100.0, ms * 1e-3 * 2.4e9 / (mfmas * wgs_per_cu));
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    u32x4 *dtbl; float *dout;
    const size_t tbl_bytes = (size_t)16 * 128 * 64 * 16 + (size_t)64 * 128 * 16 * 2;
    hipMalloc(&dtbl, tbl_bytes * 2); hipMalloc(&dout, (size_t)cus * 2 * 256 * 4);
    hipMemset(dtbl, 0x5a, tbl_bytes * 2);
    printf("%d CUs, clock %.0f MHz\n", cus, p.clockRate / 1e3);
    for (int w = 1; w <= 2; ++w) {
        run<0>("constant operands, 96 expansion instructions per 32 MFMAs on the side", dtbl, dout, w, cus);
        run<8>("  + interleave 1 : 4", dtbl, dout, w, cus);
        run<1>("operands from the expansion", dtbl, dout, w, cus);
        run<9>("  + interleave 1 : 4", dtbl, dout, w, cus);
        run<11>("  + sign-bit shifts", dtbl, dout, w, cus);
        run<15>("  + operand rows refilled from L2 in the loop (the kernel's loop)", dtbl, dout, w, cus);
        run<31>("  + mask in a VGPR", dtbl, dout, w, cus);
        run<7>("the kernel's loop without the interleave pattern", dtbl, dout, w, cus);
        run<5>("  ... and without the sign-bit shifts", dtbl, dout, w, cus);
        run<7 + 64>("the kernel's loop without the pattern, 2 of the 6 refills", dtbl, dout, w, cus);
        run<7 + 128>("the kernel's loop without the pattern, 4 of the 6 refills", dtbl, dout, w, cus);
        run<7 + 256>("the kernel's loop without the pattern, refills as 12 dwordx2 loads", dtbl, dout, w, cus);
        run<7 + 128 + 512>("the kernel's loop without the pattern, 4 of the 6 refills + the B row from LDS", dtbl, dout, w, cus);
        run<7 + 1024>("the kernel's loop without the pattern, B expansion shared through LDS", dtbl, dout, w, cus);
        run<5 + 1024>("  ... and without the sign-bit shifts", dtbl, dout, w, cus);
    }
    return 0;
}
