// fp4_probe.hip -- does v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands count bit-vector dot products exactly?
// Rows of A and columns of B are 256-sample bit vectors: lane (r = l & 31, h = l >> 5) holds 128 samples of its row as
// four 32-bit words.  An operand dword takes ONE sample from each nibble of a word -- w & (0x11111111 << k) -- which leaves
// the FP4 values 0.5 (k = 0), 1 (k = 1), 2 (k = 2) in the nibbles; k = 3 is the sign bit, so that one is (w >> 3) & 0x11111111.
// All four dwords of an operand use the same k (one per word), so every product of that MFMA carries the same weight and
// the instruction's E8M0 block scales (byte 0 of a VGPR: 127 = 2^0) put it back to 1.  build: hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_probe(const uint32_t *a_words, const uint32_t *b_words, const int *scales, float *out, int mode) {
    const uint32_t l = threadIdx.x, r = l & 31u, h = l >> 5;
    uint32_t wa[4], wb[4];
    for (int j = 0; j < 4; ++j) { wa[j] = a_words[(r * 2 + h) * 4 + j]; wb[j] = b_words[(r * 2 + h) * 4 + j]; }
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    const int s_half = scales[0], s_one = scales[1], s_two = scales[2];  // E8M0 in byte 0: 2^-1, 2^0, 2^1 (run-time values)
    auto operand = [&](const uint32_t (&w)[4], int k) {
        i32x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < 4; ++j) v[j] = k < 3 ? (int)(w[j] & (0x11111111u << k)) : (int)((w[j] >> 3) & 0x11111111u);
        return v;
    };
    if (mode == 0) {  // plain: k = 1 only (value 1.0), unit scales: counts the samples 1, 5, 9, ... of every word
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(operand(wa, 1), operand(wb, 1), acc, 4, 4, 0, s_one, 0, s_one);
    } else {          // all four k, scales 2 / 1 / 0.5 / 2 on both sides: every sample counts once
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(operand(wa, 0), operand(wb, 0), acc, 4, 4, 0, s_two, 0, s_two);
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(operand(wa, 1), operand(wb, 1), acc, 4, 4, 0, s_one, 0, s_one);
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(operand(wa, 2), operand(wb, 2), acc, 4, 4, 0, s_half, 0, s_half);
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(operand(wa, 3), operand(wb, 3), acc, 4, 4, 0, s_two, 0, s_two);
    }
    // C layout of the 32 x 32 forms: col = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
    for (int q = 0; q < 16; ++q) out[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r] = acc[q];
}

int main() {
    std::vector<uint32_t> a(32 * 8), b(32 * 8);
    srand(7);
    for (auto &v : a) v = (uint32_t)rand() * 2654435761u ^ (uint32_t)rand();
    for (auto &v : b) v = (uint32_t)rand() * 2246822519u ^ (uint32_t)rand();
    const int scales[3] = {126, 127, 128};
    uint32_t *da, *db; int *ds; float *dout;
    hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&ds, 12); hipMalloc(&dout, 1024 * 4);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ds, scales, 12, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, ds, dout, mode);
        std::vector<float> out(1024);
        hipMemcpy(out.data(), dout, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        double maxv = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                int want = 0;
                for (int hh = 0; hh < 2; ++hh)
                    for (int w = 0; w < 4; ++w) {
                        uint32_t x = a[(i * 2 + hh) * 4 + w] & b[(j * 2 + hh) * 4 + w];
                        if (mode == 0) x &= 0x22222222u;
                        want += __builtin_popcount(x);
                    }
                const float got = out[i * 32 + j];
                if (got != (float)want) { if (bad < 5) printf("mode %d (%d,%d): got %g want %d\n", mode, i, j, got, want); ++bad; }
                if (got > maxv) maxv = got;
            }
        printf("mode %d: %d of 1024 cells differ (max value %g)\n", mode, bad, maxv);
        bad_total += bad;
    }
    return bad_total ? 1 : 0;
}
