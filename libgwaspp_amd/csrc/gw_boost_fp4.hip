// gw_boost_fp4.hip -- the pair kernels that count the pair tables with v_mfma_scale_f32_32x32x64_f8f6f4 on FP4 (E2M1) operands:
//   k_boost_mfma_fp4<PRE, NV, TA, LV>   case groups < 65 536: four corners (NV = 2, TA = 2; tables without a missing call) or
//                                       nine cells (NV = 3, TA = 1), one group at a time, the case counts parked in LDS as uint16
//   k_boost_mfma_fp4_both<PRE, LV>      case groups of any size (cohorts < 2^24), four corners, both groups' accumulators in registers
// LV: the logarithm-free first-level bound ahead of the bound proper (gw_boost_dev.h boost_bound_quad): 0 none, 1 / 2.
//
// k_boost_mfma_seq<PRE, 2, 2> (gw_boost_seq.inc) is bound by in-order issue: an int8 MFMA covers 32 samples per lane and
// needs eight operand dwords made by one vector instruction each (4 samples per dword, one per byte) -- 25-27 of the
// 32 cycles the matrix core takes go to issuing the MFMA and its operand expansion (profiles/r03_c5_experiments.txt).
// FP4 halves both: an operand dword carries EIGHT samples, one per nibble, and one MFMA covers 64 per lane in the same
// 32 cycles.  The expansion is a single v_bitop3: dword = class of the two plane words & (0x11111111 << k) leaves the FP4
// values 0.5 (k = 0), 1 (k = 1), 2 (k = 2) in the nibbles; k = 3 would be the sign bit, so for that group the PLANES are
// shifted down by 3 once and treated like k = 0 (0.5).  No bit reversal on the B side: all four dwords of an operand use
// the SAME k (one dword from each of four words, i.e. two 64-sample units = two chunks of the block-tiled table per MFMA),
// so every product of an MFMA carries the same weight 4^(k-1), and the instruction's own E8M0 block scales (byte 0 of a
// VGPR per operand: 128 / 127 / 126 = 2 / 1 / 0.5) put it back to exactly 1.  Accumulators are FP32 and hold the counts
// exactly (< 2^24).  tools/fp4_probe.hip is the 32 x 32 x 256 check of all of this against popcounts;
// tools/mfma_rate_probe.hip measures what the matrix cores sustain for this instruction and how many vector instructions
// per MFMA are free (about five; the counting loop issues 4.3).
// Per pair of chunks (256 samples per row): 32 MFMAs (64 with int8) and 120 expansion instructions (264).
// The per-pair work: first-level bound -> boost_bound_split -> FP64 measure; hits and measures are bit-identical to the
// int8 and vector-ALU kernels'.
#include "gw_boost_dev.h"

namespace gw {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// NV = 2, TA = 2: the four corners on 64 x 32 pairs per wave (tables without a missing call);
// NV = 3, TA = 1: all nine cells on 32 x 32 pairs per wave (tables with missing calls), as k_boost_mfma_seq<., 3, 1>
// LV: the first-level bound -- 0 none, 1 one curvature constant (balanced cohorts), 2 two-piece quadratic
// BOTH: the case counts stay in registers beside the control counts instead of being parked in LDS as uint16 -- case
// groups of 65 536 and more (k_boost_mfma_fp4_both: NV = 2, TA = 1, the 128 accumulator registers split between the groups)
template <bool PRE, int NV, int TA, int LV, bool BOTH>
__device__ __forceinline__ void fp4_pair_tile(const BoostArgs &a) {
    constexpr bool L1 = LV != 0;
    constexpr int NP = NV * NV;            // products per tile pair
    constexpr int NT = TA * NP;            // accumulator tiles
    constexpr int WA = 32 * TA;            // A rows per wave
    __shared__ uint32_t s_case[BOTH ? 1 : NT * 8 * 256];  // [tile][register pair][thread]: two uint16 case counts
    // PRE: what the per-pair bound needs of the workgroup's 2 WA rows -- genotype counts of the two groups, P(status | g)
    // and K_i -- staged once: the bound reads 13 dwords per pair, and as loads from global memory (two rows per wave
    // instruction, ~10 instructions per pair at two waves per SIMD) their latency, not the arithmetic, set its pace
    constexpr int RS = NV == 3 ? 20 : 16;  // dwords per staged row (nine-cell tables: 1 / P(status | g) as well)
    __shared__ uint32_t s_row[PRE ? 2 * WA * RS : 1];

    // Tile order.  The grid is (column tiles padded to a multiple of 8) x (row tiles); workgroups are dispatched x-fastest and
    // dealt round-robin to the 8 XCDs, so an XCD sees the column tiles bx = xcd (mod 8) of row tile 0, then of row tile 1, ...
    // and every 160 KB column tile has left its 4 MB L2 long before the next row tile asks for it again.  The XCD-local
    // sequence number l is re-read as: GW_FP4_ROW_GROUP row tiles per column tile, then the next column tile -- the same
    // set of tiles per XCD, but a column tile now serves G consecutive workgroups.
    // (measured at c5, alternating processes on one box: G = 1 72.2, 2 70.8, 4 70.1, 6 69.7 - 69.8, 8 69.8 - 69.9, 12 71.5,
    // 16 74.4 ms: 6 row tiles of 128 rows, i.e. ~2 MB of operand rows per group)
#ifndef GW_FP4_ROW_GROUP
#define GW_FP4_ROW_GROUP (12 / TA)
#endif
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    if (GW_FP4_ROW_GROUP > 1 && (gridDim.x & 7u) == 0u) {
        constexpr uint32_t G = GW_FP4_ROW_GROUP;
        const uint32_t nc = gridDim.x >> 3, xcd = bx & 7u, l = by * nc + (bx >> 3);
        const uint32_t grp = l / (G * nc), rem = l - grp * (G * nc);
        const uint32_t g_here = min(G, gridDim.y - grp * G);  // the last group may be short
        const uint32_t vxl = rem / g_here;
        by = grp * G + (rem - vxl * g_here);
        bx = vxl * 8u + xcd;
    }
    const uint32_t a0 = a.a_begin + by * (2 * WA), b0 = a.b_begin + bx * MT;
    const uint32_t a_end = a.a_begin + a.n_a, b_end = a.b_begin + a.n_b;
    if (b0 >= b_end || a0 + 1u >= min(b0 + (uint32_t)MT, b_end)) return;  // grid padding; wholly on or below the diagonal

    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t wy = wave >> 1, wx = wave & 1u;
    const uint32_t fr = lane & 31u, fh = lane >> 5;

    if constexpr (PRE) {
        if (t < 2u * WA) {
            const uint32_t row = min(a0 + t, a_end - 1u);
            const SnpMargin *pm = a.mar + row;
            const SnpMarginF *pf = a.marf + row;
            uint32_t *o = s_row + t * RS;
            if constexpr (L1) {  // the first level works on floats: cases and everyone with genotype 0 / 2
                o[0] = __float_as_uint((float)pm->cases[0]);
                o[1] = __float_as_uint((float)pm->cases[2]);
                o[2] = __float_as_uint((float)(pm->cases[0] + pm->ctrls[0]));
                o[4] = __float_as_uint((float)(pm->cases[2] + pm->ctrls[2]));
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) { o[k] = pm->cases[k]; o[4 + k] = pm->ctrls[k]; }
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) o[8 + k] = __float_as_uint(pf->pa[k]);
            if constexpr (NV == 3) {
#pragma unroll
                for (int k = 0; k < 6; ++k) o[14 + k] = __float_as_uint(pf->qa[k]);
            }
            o[3] = __float_as_uint(pf->ki);
        }
        __syncthreads();
    }
    const uint32_t nchunk_c = a.wc / 4u, nchunk = nchunk_c + a.wt / 4u;  // both multiples of 8
    auto unit = [&](uint32_t snp) {
        return a.cct + ((uint64_t)(snp >> 6) * nchunk * 64u + (snp & 63u)) * 2u + fh;
    };
    const u32x4 *ga[TA];
#pragma unroll
    for (int ta = 0; ta < TA; ++ta) ga[ta] = unit(min(a0 + wy * WA + ta * 32u + fr, a_end - 1u));
    const u32x4 *gb = unit(min(b0 + wx * 32u + fr, b_end - 1u));

    f32x16 acc[TA][NP];  // [A tile][2 * class of A + class of B], one group at a time; exact counts
    auto clear = [&]() {
#pragma unroll
        for (int ta = 0; ta < TA; ++ta)
#pragma unroll
            for (int c = 0; c < NP; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ta][c][r] = 0.0f;
    };
    clear();

#ifndef GW_FP4_RING
#define GW_FP4_RING 4
#endif
    constexpr int RING = GW_FP4_RING;  // chunks in flight per operand row (MFMA groups of two chunks each)
    u32x4 ua[RING][TA], ub[RING];
#pragma unroll
    for (int k = 0; k < RING; ++k) {
#pragma unroll
        for (int ta = 0; ta < TA; ++ta) ua[k][ta] = ga[ta][(uint64_t)k * 128u];
        ub[k] = gb[(uint64_t)k * 128u];
    }
    const uint32_t last_chunk = nchunk - 1u;
    // E8M0 block scales, byte 0 of a VGPR each (run-time values: a compile-time constant would be taken for an f32)
    const int sc_half = a.fp4_scale[0], sc_one = a.fp4_scale[1], sc_two = a.fp4_scale[2];

    // class c of word w (0, 1) of a 64-sample unit {plane0 w0, plane0 w1, plane1 w0, plane1 w1}: NV = 2: aa, bb;
    // NV = 3: aa, ab, bb (plane0 = codes 1, 3; plane1 = codes 2, 3)
    auto cls = [](const u32x4 &v, int c, int w) -> uint32_t {
        const uint32_t p0 = v[w], p1 = v[2 + w];
        if (c == 0) return p0 & ~p1;
        if (c == NV - 1) return p0 & p1;
        return p1 & ~p0;
    };
    // operand: sample k of every nibble of four words (two units); k = 3 is the FP4 sign bit and goes through a shift
    auto expand = [](const uint32_t (&w)[4], int k) -> i32x8 {
        i32x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (int)(w[j] & (0x11111111u << (k < 3 ? k : 0)));  // k = 3: words already shifted down
        return v;
    };
    // k = 3 (the sign bit): the PLANES are shifted down once (two shifts per word) and the class is taken of the shifted
    // planes, so that class and mask fuse into one v_bitop3_b32 as they do for k < 3 -- shifting every class word
    // instead cost three instructions per operand dword where this costs two (24 of 144 per 32 MFMAs)
    auto down3 = [](const u32x4 &v) -> u32x4 { return u32x4{v[0] >> 3, v[1] >> 3, v[2] >> 3, v[3] >> 3}; };
    auto chunk_pair = [&](const u32x4 (&va0)[TA], const u32x4 (&va1)[TA], const u32x4 &vb0, const u32x4 &vb1) {
        static_for<0, 4>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const int sc = k == 1 ? sc_one : (k == 2 ? sc_half : sc_two);  // values 1 / 2 / 0.5 (k = 0, 3) -> products of 1
            uint32_t ca[TA][NV][4], cb[NV][4];
            u32x4 wa0[TA], wa1[TA];
#pragma unroll
            for (int ta = 0; ta < TA; ++ta) {
                wa0[ta] = k == 3 ? down3(va0[ta]) : va0[ta];
                wa1[ta] = k == 3 ? down3(va1[ta]) : va1[ta];
            }
            const u32x4 wb0 = k == 3 ? down3(vb0) : vb0, wb1 = k == 3 ? down3(vb1) : vb1;
#pragma unroll
            for (int c = 0; c < NV; ++c)
#pragma unroll
                for (int w = 0; w < 2; ++w) {
#pragma unroll
                    for (int ta = 0; ta < TA; ++ta) {
                        ca[ta][c][w] = cls(wa0[ta], c, w);
                        ca[ta][c][2 + w] = cls(wa1[ta], c, w);
                    }
                    cb[c][w] = cls(wb0, c, w);
                    cb[c][2 + w] = cls(wb1, c, w);
                }
            i32x8 xa[TA][NV], yb[NV];
#pragma unroll
            for (int c = 0; c < NV; ++c) {
#pragma unroll
                for (int ta = 0; ta < TA; ++ta) xa[ta][c] = expand(ca[ta][c], k);
                yb[c] = expand(cb[c], k);
            }
#pragma unroll
            for (int ta = 0; ta < TA; ++ta)
#pragma unroll
                for (int x = 0; x < NV; ++x)
#pragma unroll
                    for (int y = 0; y < NV; ++y)
                        acc[ta][NV * x + y] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                            xa[ta][x], yb[y], acc[ta][NV * x + y], 4, 4, 0, sc, 0, sc);  // cbsz = blgp = 4: FP4 E2M1
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA (32 cycles)
                __builtin_amdgcn_sched_group_barrier(0x002, NV == 3 ? 3 : 4, 0);  // its share of the VALU
            }
        });
    };
    auto round = [&](uint32_t c0) {  // RING = 4 chunks = two MFMA groups
        static_for<0, RING / 2>([&](auto kc) {
            constexpr int k = 2 * decltype(kc)::value;
            chunk_pair(ua[k], ua[k + 1], ub[k], ub[k + 1]);
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const uint64_t nxt = (uint64_t)min(c0 + k + d + RING, last_chunk) * 128u;  // past the end: re-read, unused
#pragma unroll
                for (int ta = 0; ta < TA; ++ta) ua[k + d][ta] = ga[ta][nxt];
                ub[k + d] = gb[nxt];
            }
#ifndef GW_FP4_LOADS_AT_END
            // The refill belongs HERE in the instruction stream, right behind the MFMA group that used the registers: left
            // alone, the scheduler sinks all of a round's loads below its last MFMA (the groups above only name MFMA and
            // VALU), and the first half of the next round then waits for loads issued a few instructions earlier
            // (measured in alternating processes on one box: 72.5 - 72.9 ms with the loads at the end, 71.4 - 71.5 so)
            __builtin_amdgcn_sched_barrier(0);  // nothing moves across: the loads stay between the two MFMA groups
#endif
        });
    };
    for (uint32_t c0 = 0; c0 < nchunk_c; c0 += RING) round(c0);
    f32x16 acc_case[BOTH ? TA : 1][BOTH ? NP : 1];
#pragma unroll
    for (int ta = 0; ta < TA; ++ta)
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            if constexpr (BOTH) {
                acc_case[ta][c] = acc[ta][c];
            } else {
#pragma unroll
                for (int r = 0; r < 16; r += 2)
                    s_case[((ta * NP + c) * 8 + (r >> 1)) * 256 + t] = (uint32_t)acc[ta][c][r] | ((uint32_t)acc[ta][c][r + 1] << 16);
            }
        }
    clear();
    for (uint32_t c0 = nchunk_c; c0 < nchunk; c0 += RING) round(c0);

    // ---- per pair (C layout as in k_boost_mfma); pair p = 16 ta + reg
    const uint32_t li_base = wy * WA + 4u * fh;  // the lane's first row within the workgroup's 2 WA rows
    const uint32_t j = b0 + wx * 32u + fr, i_base = a0 + li_base;
    const bool j_ok = j < b_end;
    // pair p = row i_base + off(p): valid <=> i < a_end, j < b_end, i < j <=> off(p) < limit -- one compare with a constant
    const uint32_t lim_abs = j_ok ? min(j, a_end) : 0u;
    const uint32_t limit = lim_abs > i_base ? lim_abs - i_base : 0u;
    const uint32_t *s_row_lane = s_row + li_base * RS;  // ... and its staged margins at a constant offset from here
    const SnpMargin *pm2 = a.mar + (j_ok ? j : b_end - 1u);
    uint32_t f2c[3], f2t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { f2c[k] = pm2->cases[k]; f2t[k] = pm2->ctrls[k]; }
    SnpMarginF mf2;
    if (PRE) mf2 = a.marf[j_ok ? j : b_end - 1u];
    float cx[3], csum[3];  // first level: the column's cases / everyone per genotype, as floats
#pragma unroll
    for (int k = 0; k < 3; ++k) { cx[k] = (float)f2c[k]; csum[k] = (float)(f2c[k] + f2t[k]); }
    const float nf = (float)a.n_individs, inv_n = 1.0f / (float)a.n_individs;
    // NV == 2 <=> the table has no missing call: the bound with the logarithms split (boost_bound_split)
    const float lg_n = a.lg_n;  // >= log2 n (rounded up on the host)
    auto bound_of = [&](const uint32_t (&ca)[9], const uint32_t (&co)[9], const SnpMarginF &f1, float &err) -> float {
        if constexpr (NV == 2) return boost_bound_split(ca, co, f1, mf2, nf, lg_n, err);
        else return boost_bound(ca, co, f1, mf2, nf, inv_n, err);
    };
    auto case_count = [&](int tile, int r) -> uint32_t {  // r may be a run-time value
        if constexpr (BOTH) {
            float x = 0.0f;
#pragma unroll
            for (int q = 0; q < 16 * NT; ++q) x = (tile * 16 + r) == q ? acc_case[(q >> 4) / NP][(q >> 4) % NP][q & 15] : x;
            return (uint32_t)x;
        } else {
            const uint32_t v = s_case[(tile * 8 + (r >> 1)) * 256 + t];
            return (r & 1) ? v >> 16 : v & 0xFFFFu;
        }
    };
    auto row_of = [&](int p) { return i_base + (uint32_t)((p >> 4) * 32 + (p & 3) + 8 * ((p & 15) >> 2)); };
    // the two 3 x 3 tables of a pair from its counted cells (cells[0] cases, [1] controls)
    auto tables = [&](const uint32_t (&cells)[2][NP], const uint32_t *f1c, const uint32_t *f1t, uint32_t (&ca)[9],
                      uint32_t (&co)[9]) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            uint32_t *o = g == 0 ? ca : co;
            if (NV == 3) {
#pragma unroll
                for (int c = 0; c < 9; ++c) o[c] = cells[g][c % NP];
            } else {  // table5.cpp:1088-1106 / 1125-1143: corners counted, the rest from the margins
                const uint32_t *f1 = g == 0 ? f1c : f1t;
                const uint32_t *f2 = g == 0 ? f2c : f2t;
                o[0] = cells[g][0]; o[2] = cells[g][1]; o[6] = cells[g][2 % NP]; o[8] = cells[g][3 % NP];
                o[1] = f1[0] - o[0] - o[2];
                o[7] = f1[2] - o[8] - o[6];
                o[3] = f2[0] - o[0] - o[6];
                o[5] = f2[2] - o[2] - o[8];
                o[4] = f2[1] - o[1] - o[7];
            }
        }
    };

    uint32_t need = 0;
    uint32_t n_valid = 0;  // at most 32 per lane
    // (the bound for two pairs at a time with packed FP32 multiply-adds -- 213 instead of 253 instructions per pair -- was
    // measured slower, 84.5 vs 82.8 ms in alternating processes on one box: profiles/r03_c5_experiments.txt)
    static_for<0, 16 * TA>([&](auto pc) {
        constexpr int p = decltype(pc)::value, ta = p >> 4, r = p & 15;
        constexpr uint32_t off = (uint32_t)((p >> 4) * 32 + (p & 3) + 8 * ((p & 15) >> 2));  // row_of(p) - i_base
        const uint32_t i = i_base + off;
        if (off < limit) {
            ++n_valid;
#ifdef GW_FP4_NO_BOUND   // measurement only (tools/build_variant_file.sh): the launch without any per-pair work
            if (a.n_individs < 0) {
#else
            if (PRE) {
#endif
                const uint32_t *rw = s_row_lane + off * RS;
                SnpMarginF f1;  // the fields the bounds of tables without / with missing calls read
                if constexpr (NV == 2) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) f1.pa[k] = __uint_as_float(rw[8 + k]);
                    f1.ki = __uint_as_float(rw[3]);
                } else {  // boost_bound reads pa and qa of the pair's first marker
#pragma unroll
                    for (int k = 0; k < 6; ++k) { f1.pa[k] = __uint_as_float(rw[8 + k]); f1.qa[k] = __uint_as_float(rw[14 + k]); }
                }
                float err, bound;
                if constexpr (L1) {  // first level, in floats throughout (the control counts are floats already)
                    float fx[9], fs[9];
                    static_assert(!L1 || NP == 4, "first level: four-corner tables");
                    constexpr int at[4] = {0, 2, 6, 8};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if constexpr (BOTH) fx[at[c]] = acc_case[ta][c][r];
                        else fx[at[c]] = (float)case_count(ta * NP + c, r);
                        fs[at[c]] = fx[at[c]] + acc[ta][c % NP][r];
                    }
                    const float rx0 = __uint_as_float(rw[0]), rx2 = __uint_as_float(rw[1]);
                    const float rs0 = __uint_as_float(rw[2]), rs2 = __uint_as_float(rw[4]);
                    fx[1] = rx0 - fx[0] - fx[2];   fs[1] = rs0 - fs[0] - fs[2];      // as `tables` below, exact in floats
                    fx[7] = rx2 - fx[8] - fx[6];   fs[7] = rs2 - fs[8] - fs[6];
                    fx[3] = cx[0] - fx[0] - fx[6]; fs[3] = csum[0] - fs[0] - fs[6];
                    fx[5] = cx[2] - fx[2] - fx[8]; fs[5] = csum[2] - fs[2] - fs[8];
                    fx[4] = cx[1] - fx[1] - fx[7]; fs[4] = csum[1] - fs[1] - fs[7];
                    bound = boost_bound_quad_f<LV == 2>(fx, fs, f1, mf2, nf, a.quad, err);
                } else {
                    uint32_t cells[2][NP], ca[9], co[9];
#pragma unroll
                    for (int c = 0; c < NP; ++c) {
                        cells[0][c] = case_count(ta * NP + c, r);
                        cells[1][c] = (uint32_t)acc[ta][c][r];
                    }
                    tables(cells, rw, rw + 4, ca, co);
                    bound = bound_of(ca, co, f1, err);
                }
                if (a.dense) a.dense[(size_t)(i - a.a_begin) * a.n_b + (j - a.b_begin)] = (double)(a.dense_err ? err : bound);
                else if (!bound_rules_out(bound, err, a.bound_threshold)) need |= 1u << p;
            } else {
#ifndef GW_FP4_NO_BOUND
                need |= 1u << p;
#endif
            }
        }
    });

    unsigned long long n_eval = n_valid;
    double vmax = -1.0e300, vmin = 1.0e300;
    while (need) {
        const int p = __ffs((int)need) - 1;
        need &= need - 1u;
        uint32_t cells[2][NP], ca[9], co[9];
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            uint32_t y = 0;
#pragma unroll
            for (int q = 0; q < 16 * TA; ++q) y = p == q ? (uint32_t)acc[q >> 4][c][q & 15] : y;
            cells[0][c] = case_count((p >> 4) * NP + c, p & 15);
            cells[1][c] = y;
        }
        const uint32_t i = row_of(p);
        const SnpMargin m1 = a.mar[i], m2 = *pm2;
        tables(cells, m1.cases, m1.ctrls, ca, co);
        float err2 = 0.0f, bound2 = 0.0f;
        if (PRE && !a.dense) {  // few pairs get here
            bound2 = bound_of(ca, co, a.marf[i], err2);
            // second level: what the logarithm-free bound let through meets the bound proper before any FP64
            if (L1 && bound_rules_out(bound2, err2, a.bound_threshold)) continue;
        }
        const double measure = boost_measure(ca, co, m1, m2, a.n_individs, a.frac, a.xlogx);
        if (!PRE) {
            vmax = fmax(vmax, measure);
            vmin = fmin(vmin, measure);
        } else if (!a.dense) {  // the bounds, checked against the measure
            check_bound(measure, bound2, err2, a.tally);
            if constexpr (L1) {
                float err1;
                const float bound1 = boost_bound_quad(ca, co, a.marf[i], mf2, nf, a.quad, err1);
                check_upper_bound(measure, bound1, err1, a.tally);
            }
        }
        if (measure > a.threshold) {  // :482-484
            const unsigned long long slot = atomicAdd(&a.tally[0], 1ull);
            if (slot < a.max_hits) {
                BoostHit h;
                h.snp_a = i;
                h.snp_b = j;
                h.measure = measure;
                a.hits[slot] = h;
            }
        }
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


template <bool PRE, int NV, int TA, int LV>
__global__ void __launch_bounds__(256, 2) k_boost_mfma_fp4(BoostArgs a) {
    fp4_pair_tile<PRE, NV, TA, LV, false>(a);
}
// case groups of any size (counts exact in the FP32 accumulators below 2^24): four corners, both groups' accumulators kept
template <bool PRE, int LV>
__global__ void __launch_bounds__(256, 2) k_boost_mfma_fp4_both(BoostArgs a) {
    fp4_pair_tile<PRE, 2, 1, LV, true>(a);
}

hipError_t launch_boost_mfma_fp4_both(const BoostArgs &a, bool pre, hipStream_t stream) {
    const dim3 grid(((a.n_b + MT - 1) / MT + 7u) & ~7u, (a.n_a + MT - 1) / MT);
    if (pre && a.quad.c_m > 0.0f && a.quad.c_d != 0.0f) hipLaunchKernelGGL((k_boost_mfma_fp4_both<true, 2>), grid, dim3(256), 0, stream, a);
    else if (pre && a.quad.c_m > 0.0f) hipLaunchKernelGGL((k_boost_mfma_fp4_both<true, 1>), grid, dim3(256), 0, stream, a);
    else if (pre) hipLaunchKernelGGL((k_boost_mfma_fp4_both<true, 0>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((k_boost_mfma_fp4_both<false, 0>), grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_boost_mfma_fp4(const BoostArgs &a, bool pre, bool nine, hipStream_t stream) {
    if (nine) {
        const dim3 grid(((a.n_b + MT - 1) / MT + 7u) & ~7u, (a.n_a + MT - 1) / MT);
        if (pre) hipLaunchKernelGGL((k_boost_mfma_fp4<true, 3, 1, 0>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_boost_mfma_fp4<false, 3, 1, 0>), grid, dim3(256), 0, stream, a);
    } else {
        const dim3 grid(((a.n_b + MT - 1) / MT + 7u) & ~7u, (a.n_a + 2 * MT - 1) / (2 * MT));
        if (pre && a.quad.c_m > 0.0f && a.quad.c_d != 0.0f) hipLaunchKernelGGL((k_boost_mfma_fp4<true, 2, 2, 2>), grid, dim3(256), 0, stream, a);
        else if (pre && a.quad.c_m > 0.0f) hipLaunchKernelGGL((k_boost_mfma_fp4<true, 2, 2, 1>), grid, dim3(256), 0, stream, a);
        else if (pre) hipLaunchKernelGGL((k_boost_mfma_fp4<true, 2, 2, 0>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_boost_mfma_fp4<false, 2, 2, 0>), grid, dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

}  // namespace gw
