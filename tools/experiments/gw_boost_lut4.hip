// gw_boost_lut4.hip -- k_boost_mfma_lut: the pair screen of computeBoost (algorithms/epistasis_func.cpp:393-486) for
// SMALL COHORTS without a missing call (n <= 16 383 selected individuals: the size of BASELINE config 5).
//
// Same counting as k_boost_mfma_seq<true, 2, 2> (gw_boost_seq.inc): four homozygote corners per pair and group as
// int8 MFMA dot products of 0/1 class vectors, one group at a time, 64 x 32 pairs per wave (margin-assisted tables,
// table5.cpp:1066-1143).  What differs is everything around it, because at n = 10 000 the counting phase is short and
// the per-pair evaluation decides the speed (profiles/r03_c5_pmc.json: the vector ALU issues 81 % of the time, the
// matrix cores are busy 67 %):
//   * the single-precision bound of the pre-filter is boost_bound_split's formula with the 27 logarithms LOOKED UP:
//     L[k] = k log2 k for k = 0 .. n sits in LDS (4 (n + 1) bytes), the cell counts are kept multiplied by 4 from the
//     accumulators on (acc = 64 count, acc >> 4 = 4 count), so a count is its own LDS byte address -- a cell costs
//     three ds_read_b32 and seven vector instructions instead of nineteen and three quarter-rate v_log_f32;
//   * the LDS that needs is the LDS the parked case counts used to take: they go to a 64 KB slice of global memory per
//     workgroup instead (each lane writes and reads back only its own dwords, coalesced; L2 traffic ~0.8 TB/s);
//   * which takes a buffer per RESIDENT workgroup, so the kernel is persistent: 2 workgroups per CU walk the tiles of
//     the triangle (static interleave), the table is staged once per workgroup, not once per tile.
// Hits and their FP64 measures are those of every other form: the bound only decides which pairs get the FP64 measure,
// and |bound - measure| <= err holds by boost_bound_split's derivation (a table entry is k log2 k rounded once, within
// u k log2 n -- tighter than the 2u log2 k + 4u of v_log_f32 that derivation allows).
#include "gw_boost_dev.h"

#include <algorithm>

namespace gw {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int LUT_TA = 2, LUT_NT = 8;               // 2 A tiles x 4 corners
constexpr uint32_t PARK_DWORDS = LUT_NT * 8u * 256u;  // [tile][register pair][thread]: two uint16 (4 x count) each
}  // namespace

__global__ void __launch_bounds__(256, 2) k_boost_mfma_lut(BoostArgs a) {
    __shared__ float s_lut[16384];                                   // k log2 k, k = 0 .. n_individs (<= 16 383); at LDS address 0
    __shared__ __attribute__((aligned(16))) uint32_t s_row[128 * 16];  // the tile's 128 first markers: what the bound needs of them
    __shared__ uint32_t s_late;
    for (uint32_t k = threadIdx.x; k <= (uint32_t)a.n_individs; k += 256u) s_lut[k] = a.klogk[k];
    // Start-up stagger.  A tile is a matrix-core phase (counting) followed by a vector-ALU / LDS phase (the per-pair
    // bound); the two workgroups of a CU start together and, being persistent and equally loaded, would stay in step
    // for the whole launch -- both counting, then both evaluating while the matrix cores idle.  The second workgroup
    // to arrive on a CU (a ticket per CU, keyed by the hardware's XCC / SE / SH / CU ids) waits once, about half a
    // tile; from then on one workgroup's evaluation runs under the other's counting.
    if (threadIdx.x == 0) {
        uint32_t late = 0;
        if (a.ticket && a.stagger_ticks) {
            uint32_t hw_id, xcc_id;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
            const uint32_t cu = ((xcc_id & 15u) << 8) | ((hw_id >> 8) & 0xFFu);  // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
            late = atomicAdd(a.ticket + cu, 1u) & 1u;
        }
        s_late = late;
    }
    __syncthreads();
    if (s_late) {
        const uint64_t t0 = wall_clock64();  // 100 MHz
        while (wall_clock64() - t0 < (uint64_t)a.stagger_ticks) __builtin_amdgcn_s_sleep(64);
    }

    const uint32_t a_end = a.a_begin + a.n_a, b_end = a.b_begin + a.n_b;
    const uint32_t X = (a.n_b + MT - 1u) / MT, Y = (a.n_a + 2u * MT - 1u) / (2u * MT);
    const uint32_t nchunk_c = a.wc / 4u, nchunk = nchunk_c + a.wt / 4u;
    const uint32_t last_chunk = nchunk - 1u;
    const float nf = (float)a.n_individs;

    // tiles of row y that reach above the diagonal: x >= x_min(y) (k_boost_mfma_seq's early exit, inverted)
    auto row_tiles = [&](uint32_t y, uint32_t &x_min) -> uint32_t {
        const uint32_t a0 = a.a_begin + y * (2u * MT);
        if (a0 + 1u >= b_end) { x_min = X; return 0u; }
        const long long d = (long long)a0 + 1 - (long long)a.b_begin;
        x_min = d <= 0 ? 0u : (uint32_t)(d >> 6);
        return x_min < X ? X - x_min : 0u;
    };

    uint32_t n_eval = 0;  // pairs this lane evaluated (a lane sees at most 32 per tile)
    uint32_t y = 0, off = blockIdx.x, x_min = 0;
    uint32_t in_row = row_tiles(0, x_min);
    for (;;) {
        while (y < Y && off >= in_row) {  // wave-uniform walk to this workgroup's next tile
            off -= in_row;
            ++y;
            if (y < Y) in_row = row_tiles(y, x_min);
        }
        if (y >= Y) break;
        const uint32_t a0 = a.a_begin + y * (2u * MT), b0 = a.b_begin + (x_min + off) * MT;
        off += gridDim.x;

        // Everything lane-dependent is derived anew per tile from a thread index the compiler cannot see through:
        // otherwise it hoists ~70 loop-invariant addresses out of this loop, spills them (the tile body needs every
        // register) and reloads them from scratch in the middle of the MFMA loop.
        uint32_t t = threadIdx.x;
        asm volatile("" : "+v"(t));
        const uint32_t lane = t & 63u, wave = t >> 6;
        const uint32_t wy = wave >> 1, wx = wave & 1u;
        const uint32_t fr = lane & 31u, fh = lane >> 5;
        // The workgroup's parking space as a buffer resource: buffer_load / store take the wave-uniform base, ONE lane
        // offset register (4 t) and the slot's offset as a scalar -- 64 distinct 64-bit lane addresses would not fit
        // the register file (the compiler spilled them).  0x00020000: raw dword buffer on gfx9 / CDNA.
        const __amdgpu_buffer_rsrc_t park =
            __builtin_amdgcn_make_buffer_rsrc(a.park + (size_t)blockIdx.x * PARK_DWORDS, 0, (int)(PARK_DWORDS * 4u), 0x00020000);
        const uint32_t t4 = t * 4u;

        // the 128 first markers of the tile, 16 dwords each: 4 x cases[3], 4 x ctrls[3], ki, -, pa[6], -, -
        __syncthreads();  // every wave is done with the previous tile's rows
        for (uint32_t idx = t; idx < 128u * 16u; idx += 256u) {
            const uint32_t row = idx >> 4, k = idx & 15u;
            const uint32_t snp = min(a0 + row, a_end - 1u);
            uint32_t v = 0;
            if (k < 3u) v = a.mar[snp].cases[k] << 2;
            else if (k < 6u) v = a.mar[snp].ctrls[k - 3u] << 2;
            else if (k == 6u) v = __float_as_uint(a.marf[snp].ki);
            else if (k >= 8u && k < 14u) v = __float_as_uint(a.marf[snp].pa[k - 8u]);
            s_row[idx] = v;
        }
        __syncthreads();

        auto unit = [&](uint32_t snp) {
            return a.cct + ((uint64_t)(snp >> 6) * nchunk * 64u + (snp & 63u)) * 2u + fh;
        };
        const u32x4 *ga[LUT_TA];
#pragma unroll
        for (int ta = 0; ta < LUT_TA; ++ta) ga[ta] = unit(min(a0 + wy * 64u + ta * 32u + fr, a_end - 1u));
        const u32x4 *gb = unit(min(b0 + wx * 32u + fr, b_end - 1u));

        f32x16 acc[LUT_TA][4];  // exact counts (FP4 operands, see gw_boost_fp4.hip)
        auto clear = [&]() {
#pragma unroll
            for (int ta = 0; ta < LUT_TA; ++ta)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ta][c][r] = 0.0f;
        };
        clear();
        constexpr int RING = 4;
        u32x4 ua[RING][LUT_TA], ub[RING];
#pragma unroll
        for (int k = 0; k < RING; ++k) {
#pragma unroll
            for (int ta = 0; ta < LUT_TA; ++ta) ua[k][ta] = ga[ta][(uint64_t)k * 128u];
            ub[k] = gb[(uint64_t)k * 128u];
        }
        const int sc_half = a.fp4_scale[0], sc_one = a.fp4_scale[1], sc_two = a.fp4_scale[2];
        auto cls = [](const u32x4 &v, int c, int w) -> uint32_t {  // aa / bb of word w
            const uint32_t p0 = v[w], p1 = v[2 + w];
            return c == 0 ? (p0 & ~p1) : (p0 & p1);
        };
        auto expand = [](const uint32_t (&w)[4], int k) -> i32x8 {  // gw_boost_fp4.hip: sample k of every nibble of four words
            i32x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = k < 3 ? (int)(w[j] & (0x11111111u << k)) : (int)((w[j] >> 3) & 0x11111111u);
            return v;
        };
        auto chunk_pair = [&](const u32x4 (&va0)[LUT_TA], const u32x4 (&va1)[LUT_TA], const u32x4 &vb0, const u32x4 &vb1) {
            uint32_t ca[LUT_TA][2][4], cb[2][4];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int w = 0; w < 2; ++w) {
#pragma unroll
                    for (int ta = 0; ta < LUT_TA; ++ta) {
                        ca[ta][c][w] = cls(va0[ta], c, w);
                        ca[ta][c][2 + w] = cls(va1[ta], c, w);
                    }
                    cb[c][w] = cls(vb0, c, w);
                    cb[c][2 + w] = cls(vb1, c, w);
                }
            static_for<0, 4>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const int sc = k == 1 ? sc_one : (k == 2 ? sc_half : sc_two);
                i32x8 xa[LUT_TA][2], yb[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
#pragma unroll
                    for (int ta = 0; ta < LUT_TA; ++ta) xa[ta][c] = expand(ca[ta][c], k);
                    yb[c] = expand(cb[c], k);
                }
#pragma unroll
                for (int ta = 0; ta < LUT_TA; ++ta)
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int yy = 0; yy < 2; ++yy)
                            acc[ta][2 * x + yy] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                                xa[ta][x], yb[yy], acc[ta][2 * x + yy], 4, 4, 0, sc, 0, sc);
#pragma unroll
                for (int m = 0; m < LUT_NT; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA (32 cycles)
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // its share of the VALU
                }
            });
        };
        auto round = [&](uint32_t c0) {
            static_for<0, RING / 2>([&](auto kc) {
                constexpr int k = 2 * decltype(kc)::value;
                chunk_pair(ua[k], ua[k + 1], ub[k], ub[k + 1]);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const uint64_t nxt = (uint64_t)min(c0 + k + d + RING, last_chunk) * 128u;  // past the end: re-read, unused
#pragma unroll
                    for (int ta = 0; ta < LUT_TA; ++ta) ua[k + d][ta] = ga[ta][nxt];
                    ub[k + d] = gb[nxt];
                }
            });
        };
        for (uint32_t c0 = 0; c0 < nchunk_c; c0 += RING) round(c0);
        // the case counts leave the registers: 4 x count of two accumulator registers per dword
#pragma unroll
        for (int ta = 0; ta < LUT_TA; ++ta)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 16; r += 2)
                    __builtin_amdgcn_raw_buffer_store_b32(
                        ((uint32_t)acc[ta][c][r] << 2) | (((uint32_t)acc[ta][c][r + 1] << 2) << 16), park, t4,
                        ((ta * 4 + c) * 8 + (r >> 1)) * 1024, 0);
        clear();
        for (uint32_t c0 = nchunk_c; c0 < nchunk; c0 += RING) round(c0);

        // ---- per pair (C layout of the 32 x 32 MFMA: register q of lane (fr, fh) = row 8 (q >> 2) + 4 fh + (q & 3), column fr)
        const uint32_t j = b0 + wx * 32u + fr, i_base = a0 + wy * 64u + 4u * fh;
        const bool j_ok = j < b_end;
        const uint32_t jj = j_ok ? j : b_end - 1u;
        const SnpMargin *pm2 = a.mar + jj;
        const SnpMarginF mf2 = a.marf[jj];
        uint32_t f2c4[3], f2t4[3];  // the second marker's genotype counts, x 4 like every count below
#pragma unroll
        for (int k = 0; k < 3; ++k) { f2c4[k] = pm2->cases[k] << 2; f2t4[k] = pm2->ctrls[k] << 2; }
        float rbq[6];               // rb / 4: float(4 s) * rbq = s * rb, exactly
#pragma unroll
        for (int k = 0; k < 6; ++k) rbq[k] = 0.25f * mf2.rb[k];
        auto row_of = [&](int p) { return i_base + (uint32_t)((p >> 4) * 32 + (p & 3) + 8 * ((p & 15) >> 2)); };
        auto lut = [&](uint32_t k4) -> float {  // k4 = 4 k = the byte offset of L[k]
            return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(s_lut) + k4);
        };
        // corners (x 4) -> the nine cells of a group's table (table5.cpp:1088-1106 / 1125-1143)
        auto table4 = [&](const uint32_t (&c4)[4], const uint32_t *f1, const uint32_t *f2, uint32_t (&o)[9]) {
            o[0] = c4[0]; o[2] = c4[1]; o[6] = c4[2]; o[8] = c4[3];
            o[1] = f1[0] - o[0] - o[2];
            o[7] = f1[2] - o[8] - o[6];
            o[3] = f2[0] - o[0] - o[6];
            o[5] = f2[2] - o[2] - o[8];
            o[4] = f2[1] - o[1] - o[7];
        };
        // boost_bound_split with the logarithms looked up (see there for the formula and the error bound).  Staged so
        // that the 27 table reads of a pair are in flight together: all addresses first, all reads, then the sums.
        struct RowF { float ki, pa[6]; };
        auto bound_lut = [&](const uint32_t (&ca)[9], const uint32_t (&co)[9], const RowF &f1, float &err) -> float {
            uint32_t s4[9];
            float lx[9], ly[9], ls[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) s4[c] = ca[c] + co[c];
#pragma unroll
            for (int c = 0; c < 9; ++c) { lx[c] = lut(ca[c]); ly[c] = lut(co[c]); ls[c] = lut(s4[c]); }
            float A = 0.0f, tao = 0.0f;
#pragma unroll
            for (int ga_ = 0; ga_ < 3; ++ga_) {
                float in_c = 0.0f, in_t = 0.0f;
#pragma unroll
                for (int gb_ = 0; gb_ < 3; ++gb_) {
                    const int c = 3 * ga_ + gb_;
                    A += (lx[c] + ly[c]) - ls[c];
                    const float fs4 = (float)s4[c];
                    in_c = __builtin_fmaf(fs4, rbq[gb_], in_c);
                    in_t = __builtin_fmaf(fs4, rbq[3 + gb_], in_t);
                }
                tao = __builtin_fmaf(f1.pa[ga_], in_c, tao);
                tao = __builtin_fmaf(f1.pa[3 + ga_], in_t, tao);
            }
            const float lt = __log2f(tao);
            const float r = 1.3862943611198906f * (((A + f1.ki) + mf2.kj) + nf * lt);  // 2 ln 2
            err = 5.9604645e-8f * (nf * (8.5f * a.lg_n + 69.0f + 7.2f * fabsf(lt)) + 6.0f * fabsf(r));
            return r;
        };
        auto row_data = [&](uint32_t i, uint32_t (&f1c4)[3], uint32_t (&f1t4)[3], RowF &f1) {
            const u32x4 *q = reinterpret_cast<const u32x4 *>(s_row + (i - a0) * 16u);  // the same address in a half-wave: broadcast
            const u32x4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            f1c4[0] = q0[0]; f1c4[1] = q0[1]; f1c4[2] = q0[2];
            f1t4[0] = q0[3]; f1t4[1] = q1[0]; f1t4[2] = q1[1];
            f1.ki = __uint_as_float(q1[2]);
            f1.pa[0] = __uint_as_float(q2[0]); f1.pa[1] = __uint_as_float(q2[1]); f1.pa[2] = __uint_as_float(q2[2]);
            f1.pa[3] = __uint_as_float(q2[3]); f1.pa[4] = __uint_as_float(q3[0]); f1.pa[5] = __uint_as_float(q3[1]);
        };

        uint32_t valid = 0;  // bit p: pair p of this lane is above the diagonal and inside the block
#pragma unroll
        for (int p = 0; p < 16 * LUT_TA; ++p) {
            const uint32_t i = row_of(p);
            valid |= (uint32_t)(i < a_end && j_ok && i < j) << p;
        }
        n_eval += (uint32_t)__popc(valid);
        uint32_t need = 0;
        // Two pairs (accumulator registers r, r + 1) share the four parked dwords of their case counts.  Those come
        // from L2 (~1 us away): the loads run three couples ahead of their use, outside every branch.
        constexpr int AHEAD = 3, NCOUPLE = 8 * LUT_TA;
        uint32_t pk[AHEAD + 1][4];
        auto fetch = [&](int q, uint32_t (&dst)[4]) {  // couple q = pairs 2q, 2q + 1: tile q >> 3, register pair q & 7
#pragma unroll
            for (int c = 0; c < 4; ++c)
                dst[c] = __builtin_amdgcn_raw_buffer_load_b32(park, t4, (((q >> 3) * 4 + c) * 8 + (q & 7)) * 1024, 0);
        };
        static_for<0, AHEAD>([&](auto qc) { fetch(decltype(qc)::value, pk[decltype(qc)::value]); });
        static_for<0, NCOUPLE>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q + AHEAD < NCOUPLE) fetch(q + AHEAD, pk[(q + AHEAD) % (AHEAD + 1)]);
            static_for<0, 2>([&](auto hc) {
                constexpr int p = 2 * q + decltype(hc)::value, ta = p >> 4, r = p & 15;
                if ((valid >> p) & 1u) {
                    const uint32_t i = row_of(p);
                    uint32_t c4[2][4], ca[9], co[9];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const uint32_t v = pk[q % (AHEAD + 1)][c];
                        c4[0][c] = (r & 1) ? v >> 16 : v & 0xFFFFu;
                        c4[1][c] = (uint32_t)acc[ta][c][r] << 2;
                    }
                    uint32_t f1c4[3], f1t4[3];
                    RowF f1;
                    row_data(i, f1c4, f1t4, f1);
                    table4(c4[0], f1c4, f2c4, ca);
                    table4(c4[1], f1t4, f2t4, co);
                    float err;
                    const float bound = bound_lut(ca, co, f1, err);
                    if (!bound_rules_out(bound, err, a.bound_threshold)) need |= 1u << p;
                }
            });
            // (evaluating both pairs of a couple unconditionally in one basic block, so that their dependent chains
            // interleave, was 128.6 vs 123.2 ms: the clamps a garbage pair needs cost more than the overlap gains)
        });

        while (need) {  // few pairs: the FP64 measure, and the bound again, checked against it
            const int p = __ffs((int)need) - 1;
            need &= need - 1u;
            uint32_t c4[2][4], ca4[9], co4[9], ca[9], co[9];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t yv = 0;
#pragma unroll
                for (int q = 0; q < 16 * LUT_TA; ++q) yv = p == q ? (uint32_t)acc[q >> 4][c][q & 15] : yv;
                const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(park, t4, (((p >> 4) * 4 + c) * 8 + ((p & 15) >> 1)) * 1024, 0);
                c4[0][c] = (p & 1) ? v >> 16 : v & 0xFFFFu;
                c4[1][c] = yv << 2;
            }
            const uint32_t i = row_of(p);
            const SnpMargin m1 = a.mar[i], m2 = *pm2;
            uint32_t f1c4[3], f1t4[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { f1c4[k] = m1.cases[k] << 2; f1t4[k] = m1.ctrls[k] << 2; }
            table4(c4[0], f1c4, f2c4, ca4);
            table4(c4[1], f1t4, f2t4, co4);
#pragma unroll
            for (int c = 0; c < 9; ++c) { ca[c] = ca4[c] >> 2; co[c] = co4[c] >> 2; }
            const double measure = boost_measure(ca, co, m1, m2, a.n_individs, a.frac, a.xlogx);
            float err;
            RowF f1;
            row_data(i, f1c4, f1t4, f1);
            const float bound = bound_lut(ca4, co4, f1, err);
            check_bound(measure, bound, err, a.tally);
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
        }
    }
    unsigned long long total = n_eval;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
    if ((threadIdx.x & 63u) == 0 && total) atomicAdd(&a.tally[1], total);
}

// k log2 k for k = 0 .. n (computed in FP64, rounded once)
__global__ void k_boost_klogk(uint32_t n_entries, float *klogk) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_entries) return;
    klogk[k] = k ? (float)((double)k * log2((double)k)) : 0.0f;
}

hipError_t launch_boost_klogk(uint32_t n_entries, float *d_klogk, hipStream_t stream) {
    hipLaunchKernelGGL(k_boost_klogk, dim3((n_entries + 255u) / 256u), dim3(256), 0, stream, n_entries, d_klogk);
    return hipGetLastError();
}

size_t boost_lut_park_bytes(int compute_units) { return (size_t)compute_units * 2u * PARK_DWORDS * sizeof(uint32_t); }

hipError_t launch_boost_mfma_lut4(const BoostArgs &a, int compute_units, hipStream_t stream) {
    const uint32_t X = (a.n_b + MT - 1u) / MT, Y = (a.n_a + 2u * MT - 1u) / (2u * MT);
    const unsigned long long tiles = (unsigned long long)X * Y;
    const uint32_t grid = (uint32_t)std::min<unsigned long long>(tiles, (unsigned long long)compute_units * 2u);
    hipLaunchKernelGGL(k_boost_mfma_lut, dim3(grid), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace gw
