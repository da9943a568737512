// gw_boost_mfma.hip -- the both-groups-at-once int8 MFMA pair kernel for the four corners (case groups of
// any size; see gw_boost.hip for the path)
#include "gw_boost_dev.h"

namespace gw {

// ---------------------------------------------------------------------------
// Matrix-core form of the same tile (tables without a missing call).
//
// The four counted corners of a pair are dot products of 0/1 class vectors over
// the samples of a group -- a binary GEMM, and unlike the per-SNP scan it re-uses
// every row M times, so it is compute- not HBM-bound.  v_bcnt_u32_b32 issues at
// half rate on gfx950 (tools/valu_rates.hip: 4.1 cycles; AND+BCNT 6.25 cycles per
// 64 x 32 sample pairs = 330 bit-MACs per cycle and SIMD); v_mfma_i32_32x32x32_i8
// does 1024 MACs per cycle and SIMD with exact int32 accumulation.  Price: every
// bit must become a byte.
//
// Wave = 32 x 32 SNP pairs: 4 corner products x 2 groups = 8 accumulator tiles
// (128 registers); workgroup = 2 x 2 waves that share nothing but the epilogue's
// margins.  No LDS and no barrier in the main loop: a first version that expanded
// each chunk once per workgroup into LDS was bound by LDS *stores* (ds_write_b128
// moves 79 B/clk/CU, the expanded operands of a 64 x 64 tile need 64 B/clk at the
// MFMA rate) and reached 48 % of the matrix peak.  Here every lane loads the 64
// sample bits it needs straight from the block-tiled table (k_retile_cc: lane
// (r, h) = row r, half h of the chunk, one 16-byte load, 1 KiB contiguous per wave)
// and expands them in registers.  The K order inside a chunk is ours to choose as
// long as A and B agree: lane half h owns samples 64h .. 64h+63, step s takes
// 16s .. 16s+15 of them.
//
// Expansion in ~1.3 VALU per 4 samples: the K order is free, so an operand dword
// takes one sample from each byte of the word -- a single v_and_b32 -- and carries
// a power-of-two weight that its partner compensates (expand_a / expand_b): every
// product is 64 and the accumulators hold 64 x count (exact; >> 6).
// ---------------------------------------------------------------------------
template <bool PRE>
__global__ void __launch_bounds__(256, 2) k_boost_mfma(BoostArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_mar[MT * 32];  // the A rows' margins (epilogue)

    const uint32_t a0 = a.a_begin + blockIdx.y * MT, b0 = a.b_begin + blockIdx.x * MT;
    const uint32_t a_end = a.a_begin + a.n_a, b_end = a.b_begin + a.n_b;
    if (a0 + 1u >= min(b0 + (uint32_t)MT, b_end)) return;  // nothing above the diagonal in this tile

    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t wy = wave >> 1, wx = wave & 1u;
    const uint32_t fr = lane & 31u, fh = lane >> 5;

    // this lane's A row and B row in the block-tiled table (rows past the block end repeat its last
    // row; their pairs are never reported)
    const uint32_t nchunk_c = a.wc / 4u, nchunk = nchunk_c + a.wt / 4u;  // planes are whole multiples of 8 chunks
    const uint32_t snp_a = min(a0 + wy * 32u + fr, a_end - 1u), snp_b = min(b0 + wx * 32u + fr, b_end - 1u);
    const u32x4 *ga = a.cct + ((uint64_t)(snp_a >> 6) * nchunk * 64u + (snp_a & 63u)) * 2u + fh;
    const u32x4 *gb = a.cct + ((uint64_t)(snp_b >> 6) * nchunk * 64u + (snp_b & 63u)) * 2u + fh;

    i32x16 acc[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][c][r] = 0;

    // the next RING chunks' bits wait in registers (the table of a large job sits in L2 / Infinity
    // Cache, 1-2 us away; a chunk takes ~0.25 us)
    constexpr int RING = 4;
    u32x4 ua[RING], ub[RING];  // {plane0 w0, plane0 w1, plane1 w0, plane1 w1} of the lane's 64 samples
#pragma unroll
    for (int k = 0; k < RING; ++k) {
        ua[k] = ga[(uint64_t)k * 128u];
        ub[k] = gb[(uint64_t)k * 128u];
    }
    const uint32_t last_chunk = nchunk - 1u;

    struct Frag { i32x4 xa, xb, ya, yb; };
    // one chunk = 4 steps of 16 samples per lane half: step s = half (s & 1) of word (s >> 1)
    auto chunk = [&](const u32x4 &va, const u32x4 &vb, i32x16 (&d)[4]) {
        uint32_t caa[2], cbb[2], baa[2], bbb[2], raa[2], rbb[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            caa[w] = va[w] & ~va[2 + w];  // A row: aa, bb
            cbb[w] = va[w] & va[2 + w];
            baa[w] = vb[w] & ~vb[2 + w];  // B row
            bbb[w] = vb[w] & vb[2 + w];
            raa[w] = __builtin_bswap32(__builtin_bitreverse32(baa[w])) >> 1;
            rbb[w] = __builtin_bswap32(__builtin_bitreverse32(bbb[w])) >> 1;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            Frag f;
            f.xa = expand_a(caa[s >> 1], s & 1);
            f.xb = expand_a(cbb[s >> 1], s & 1);
            f.ya = expand_b(baa[s >> 1], raa[s >> 1], s & 1);
            f.yb = expand_b(bbb[s >> 1], rbb[s >> 1], s & 1);
            d[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.xa, f.ya, d[0], 0, 0, 0);
            d[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.xa, f.yb, d[1], 0, 0, 0);
            d[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.xb, f.ya, d[2], 0, 0, 0);
            d[3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.xb, f.yb, d[3], 0, 0, 0);
            // issue order: an MFMA holds the matrix pipe for 32 cycles but the issue port for 8; the
            // ~6 VALU of expansion per MFMA go between them, not in front
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // 6 VALU
            }
        }
    };
    auto round = [&](uint32_t c0, i32x16 (&d)[4]) {
        static_for<0, RING>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            chunk(ua[k], ub[k], d);
            const uint64_t nxt = (uint64_t)min(c0 + k + RING, last_chunk) * 128u;  // past the end: re-read, unused
            ua[k] = ga[nxt];
            ub[k] = gb[nxt];
        });
    };
    for (uint32_t c0 = 0; c0 < nchunk_c; c0 += RING) round(c0, acc[0]);
    for (uint32_t c0 = nchunk_c; c0 < nchunk; c0 += RING) round(c0, acc[1]);

    // ---- per pair: corners -> tables -> bound / measure.  C layout of the 32x32 MFMA:
    // column (B index) = lane & 31, row (A index) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const uint32_t j = b0 + wx * 32u + fr, i_base = a0 + wy * 32u + 4u * fh;
    const bool j_ok = j < b_end;
    const SnpMargin *pm2 = a.mar + (j_ok ? j : b_end - 1u);
    uint32_t f2c[3], f2t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { f2c[k] = pm2->cases[k]; f2t[k] = pm2->ctrls[k]; }
    SnpMarginF mf2;
    if (PRE) mf2 = a.marf[j_ok ? j : b_end - 1u];
    const float nf = (float)a.n_individs, inv_n = 1.0f / (float)a.n_individs;

    // The 64 A rows' margins go through LDS: every lane needs 16 of them.
    // Per row 32 dwords: cases[3], ctrls[3], pad[2], then the 24 floats of SnpMarginF.
    for (uint32_t idx = t; idx < (uint32_t)MT * 32u; idx += 256u) {
        const uint32_t row = idx >> 5, k = idx & 31u;
        const uint32_t snp = min(a0 + row, a_end - 1u);
        uint32_t v = 0;
        if (k < 3u) v = a.mar[snp].cases[k];
        else if (k < 6u) v = a.mar[snp].ctrls[k - 3u];
        else if (k >= 8u && PRE) v = reinterpret_cast<const uint32_t *>(a.marf + snp)[k - 8u];
        s_mar[idx] = v;
    }
    __syncthreads();

    auto tables = [&](const uint32_t (&c4)[2][4], const uint32_t *f1c, const uint32_t *f1t, uint32_t (&ca)[9],
                      uint32_t (&co)[9]) {
        // table5.cpp:1088-1106 / 1125-1143: corners counted, the rest from the margins
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint32_t *f1 = g == 0 ? f1c : f1t;
            const uint32_t *f2 = g == 0 ? f2c : f2t;
            uint32_t *o = g == 0 ? ca : co;
            o[0] = c4[g][0]; o[2] = c4[g][1]; o[6] = c4[g][2]; o[8] = c4[g][3];
            o[1] = f1[0] - o[0] - o[2];
            o[7] = f1[2] - o[8] - o[6];
            o[3] = f2[0] - o[0] - o[6];
            o[5] = f2[2] - o[2] - o[8];
            o[4] = f2[1] - o[1] - o[7];
        }
    };

    uint32_t need = 0;  // bit r: pair of accumulator register r still wants its FP64 measure
    unsigned long long n_eval = 0;
    static_for<0, 16>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const uint32_t i = i_base + (r & 3) + 8 * (r >> 2);
        if (i < a_end && j_ok && i < j) {
            ++n_eval;
            if (PRE) {
                const uint32_t *row = s_mar + (i - a0) * 32u;
                uint32_t f1[8];
                *reinterpret_cast<u32x4 *>(f1) = *reinterpret_cast<const u32x4 *>(row);
                *reinterpret_cast<u32x4 *>(f1 + 4) = *reinterpret_cast<const u32x4 *>(row + 4);
                SnpMarginF mf1;
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    *reinterpret_cast<u32x4 *>(reinterpret_cast<uint32_t *>(&mf1) + 4 * q) =
                        *reinterpret_cast<const u32x4 *>(row + 8 + 4 * q);
                uint32_t c4[2][4], ca[9], co[9];
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int k = 0; k < 4; ++k) c4[g][k] = (uint32_t)acc[g][k][r] >> 6;  // every product was 64
                tables(c4, f1, f1 + 3, ca, co);
                float err;
                const float bound = boost_bound(ca, co, mf1, mf2, nf, inv_n, err);
                if (a.dense) a.dense[(size_t)(i - a.a_begin) * a.n_b + (j - a.b_begin)] = (double)(a.dense_err ? err : bound);
                else if (!bound_rules_out(bound, err, a.bound_threshold)) need |= 1u << r;
            } else {
                need |= 1u << r;
            }
        }
    });

    double vmax = -1.0e300, vmin = 1.0e300;
    while (need) {  // FP64 measure, one accumulator register at a time (per lane its own)
        const int r = __ffs((int)need) - 1;
        need &= need - 1u;
        uint32_t c4[2][4];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t v = 0;
#pragma unroll
                for (int q = 0; q < 16; ++q) v = r == q ? (uint32_t)acc[g][k][q] : v;
                c4[g][k] = v >> 6;
            }
        const uint32_t i = i_base + (r & 3) + 8 * (r >> 2);
        if constexpr (PRE) {  // (round 4) the pair leaves the kernel: k_boost_finish<3> runs the checks, the measure, the hit list
            boost_append_candidate<4>(a, i, j, c4);
            continue;
        }
        const SnpMargin m1 = a.mar[i], m2 = *pm2;
        uint32_t ca[9], co[9];
        tables(c4, m1.cases, m1.ctrls, ca, co);
        const double measure = boost_measure(ca, co, m1, m2, a.n_individs, a.frac, a.xlogx);
        vmax = fmax(vmax, measure);
        vmin = fmin(vmin, measure);
        if (measure > a.threshold) boost_append_hit(a, i, j, measure);  // :482-484
        if (a.dense) a.dense[(size_t)(i - a.a_begin) * a.n_b + (j - a.b_begin)] = measure;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n_eval += __shfl_xor(n_eval, off, 64);
        if (!PRE) {
            vmax = fmax(vmax, __shfl_xor(vmax, off, 64));
            vmin = fmin(vmin, __shfl_xor(vmin, off, 64));
        }
    }
    if (lane == 0 && n_eval) {
        atomicAdd(&a.tally[1], n_eval);
        if (!PRE) {
            atomicMax(&a.tally[2], ordered(vmax));
            atomicMin(&a.tally[3], ordered(vmin));
        }
    }
}

hipError_t launch_boost_mfma_both4(const BoostArgs &a, bool pre, int compute_units, hipStream_t stream) {
    const dim3 grid((a.n_b + MT - 1) / MT, (a.n_a + MT - 1) / MT);
    const bool screen = pre && !a.dense;   // candidates out, k_boost_finish behind the kernel
    hipError_t e = screen ? boost_screen_begin(a, stream) : hipSuccess;
    if (e != hipSuccess) return e;
    if (pre) hipLaunchKernelGGL(k_boost_mfma<true>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(k_boost_mfma<false>, grid, dim3(256), 0, stream, a);
    if ((e = hipGetLastError()) != hipSuccess || !screen) return e;
    // this kernel's bound is boost_bound (it counts four corners but evaluates all nine cells' logarithms): the finishing
    // kernel's second level for NV = 2 is boost_bound_split -- a different bound of the same measure, checked the same way
    return launch_boost_finish(a, false, compute_units, stream);
}

}  // namespace gw
