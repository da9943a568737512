/*
 * gw_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See gw_oracle.h for scope, usage rules and pinning status.
 *
 * Reference paths are relative to /root/reference/src/libgwaspp/.
 */
#include "gw_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* geometry                                                                  */
/* ------------------------------------------------------------------------ */

static uint32_t round_up4(uint32_t v) { return (v + 3u) & ~3u; }

/* compressed_genotype_table5.cpp:44-72 */
void gwo_geometry_stream(uint32_t n, gwo_geom *g) {
    g->n_samples = n;
    g->data_blocks = round_up4(n / 16u + 1u); /* always one pad block, then x4 */
    g->blocks_per_row = 2u * g->data_blocks + 1u;
    g->words = g->data_blocks / 4u;
}

/* compressed_genotype_table3.cpp:44-67 */
void gwo_geometry_block(uint32_t n, gwo_geom *g) {
    g->n_samples = n;
    g->data_blocks = round_up4(n / 8u + 1u);
    g->blocks_per_row = g->data_blocks + 1u;
    g->words = g->data_blocks / 4u;
}

/* ------------------------------------------------------------------------ */
/* PopCount: 64 Ki byte table, common_genotype_func.h:56-90                  */
/* ------------------------------------------------------------------------ */

static uint8_t g_pop16[0x10000];
static pthread_once_t g_pop_once = PTHREAD_ONCE_INIT;

static void build_pop16(void) {
    /* same table as common_genotype_func.cpp:157-171 (ones16 over 0..65535),
     * filled by the recurrence pop(i) = pop(i>>1) + (i&1). */
    g_pop16[0] = 0;
    for (uint32_t i = 1; i < 0x10000u; ++i) g_pop16[i] = (uint8_t)(g_pop16[i >> 1] + (i & 1u));
}

int gwo_popcount16(uint16_t v) {
    pthread_once(&g_pop_once, build_pop16);
    return g_pop16[v];
}

static inline int pop64_lut(uint64_t v) {
    return g_pop16[(uint16_t)v] + g_pop16[(uint16_t)(v >> 16)] + g_pop16[(uint16_t)(v >> 32)] +
           g_pop16[(uint16_t)(v >> 48)];
}

int gwo_popcount64(uint64_t v) {
    pthread_once(&g_pop_once, build_pop16);
    return pop64_lut(v);
}

/* rows start one uint16 into the buffer, so 64-bit words are 2-byte aligned
 * only (the reference reinterpret_casts; memcpy is the defined-behaviour
 * spelling of the same load). */
static inline uint64_t ld64(const uint16_t *p) {
    uint64_t v;
    memcpy(&v, p, sizeof v);
    return v;
}

/* ------------------------------------------------------------------------ */
/* ingest                                                                    */
/* ------------------------------------------------------------------------ */

/* transformations[] / lookup[][] of table5.cpp:96-118: alphabet "ACGT"
 * (genotype/genotype.h:80-82) -> 0..3, anything else -> 4 (= unknown). */
static inline unsigned allele_index(char c) {
    switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return 4;
    }
}

static inline uint16_t encode_pair(char c1, char c2) {
    unsigned a = allele_index(c1), b = allele_index(c2);
    if (a > 3 || b > 3) return 0xFFFF;
    return (uint16_t)(4u * a + b);
}

/* table5.cpp:1220-1223 */
static inline int enc_is_hom(uint16_t e) { return e == 0 || e == 5 || e == 10 || e == 15; }

/* HeaderStateMachine, common_genotype.h:257-304.  order = bits 15..12 of the
 * header.  Returns the 2-bit code assigned to the new genotype, 0 on the
 * reference's assert paths. */
typedef struct {
    uint16_t order; /* 0x0000 .. 0x7000 */
    uint16_t head;
} hdr_state;

static int header_step(hdr_state *s, uint16_t enc) {
    const int hom = enc_is_hom(enc);
    uint16_t clear, shift;
    int code;
    switch (s->order) {
    case 0x0000:
        clear = 0x0000;
        if (hom) { s->order = 0x1000; shift = 8; code = 1; }
        else     { s->order = 0x2000; shift = 4; code = 2; }
        break;
    case 0x1000:
        clear = 0x0F00;
        if (hom) { s->order = 0x3000; shift = 0; code = 3; }
        else     { s->order = 0x4000; shift = 4; code = 2; }
        break;
    case 0x2000:
        if (!hom) return 0; /* second heterozygote spelling: assert */
        clear = 0x00F0; s->order = 0x4000; shift = 8; code = 1;
        break;
    case 0x3000:
        if (hom) return 0; /* third homozygote: assert */
        clear = 0x0F0F; s->order = 0x7000; shift = 4; code = 2;
        break;
    case 0x4000:
        if (!hom) return 0;
        clear = 0x0FF0; s->order = 0x7000; shift = 0; code = 3;
        break;
    default:
        return 0; /* a fourth distinct genotype: assert( geno_code < 0x7000 ) */
    }
    s->head = (uint16_t)((s->head & clear) | s->order | (uint16_t)(enc << shift));
    return code;
}

/* shared text walk for both layouts */
static int pack_row(const char *b, const char *e, const gwo_geom *g, uint16_t *row, int block_layout) {
    memset(row, 0, (size_t)g->blocks_per_row * sizeof(uint16_t));
    if (b >= e) return 0;
    uint16_t seen[16];
    for (int i = 0; i < 16; ++i) seen[i] = 0xFFFF;
    hdr_state st = {0, 0};
    uint16_t *p0 = row + 1;
    uint16_t *p1 = row + 1 + g->data_blocks; /* stream only (table5.cpp:63,302) */
    uint32_t col = 0;
    const char *p = b;
    do {
        if (col >= g->n_samples) return -1; /* more genotypes than columns */
        char c1 = *p++;
        char c2 = (p < e) ? *p++ : '\0';
        uint16_t enc = encode_pair(c1, c2);
        if (enc != 0xFFFF) {
            int code = seen[enc];
            if (code == 0xFFFF) {
                code = header_step(&st, enc);
                if (code == 0) return -1;
                seen[enc] = (uint16_t)code;
            }
            if (block_layout) {
                /* table3.cpp:189-190: sample c at bits 2(c&7) of block c>>3 */
                p0[col >> 3] |= (uint16_t)(code << ((col & 7u) * 2u));
            } else {
                /* table5.cpp:333-344: plane0 for codes 1,3; plane1 for 2,3 */
                uint16_t bit = (uint16_t)(1u << (col & 15u));
                if (code & 1) p0[col >> 4] |= bit;
                if (code & 2) p1[col >> 4] |= bit;
            }
        }
        ++col;
    } while (++p < e); /* ++p skips the delimiter, as table5.cpp:352 */
    row[0] = st.head;
    return 0;
}

int gwo_pack_row_stream(const char *b, const char *e, const gwo_geom *g, uint16_t *row) {
    return pack_row(b, e, g, row, 0);
}
int gwo_pack_row_block(const char *b, const char *e, const gwo_geom *g, uint16_t *row) {
    return pack_row(b, e, g, row, 1);
}

/* ParseHeader, common_genotype.h:306-334 */
void gwo_parse_header(uint16_t h, uint16_t out[4]) {
    uint16_t order = h & 0xF000;
    uint16_t aa = (h >> 8) & 0xF, ab = (h >> 4) & 0xF, bb = h & 0xF;
    out[0] = 0; /* the macro never writes xx; ResetHeaderTable (common_genotype.h:58-60,
                 * 0xFFFF000000000000 on a little-endian union) leaves it 0 */
    out[1] = out[2] = out[3] = 0xFFFF;
    switch (order) {
    case 0x7000: out[1] = aa; out[2] = ab; out[3] = bb; break;
    case 0x4000: out[1] = aa; out[2] = ab; break;
    case 0x3000: out[1] = aa; out[3] = bb; break;
    case 0x2000: out[2] = ab; break;
    case 0x1000: out[1] = aa; break;
    default: break;
    }
}

/* table5.cpp:383-417 */
int gwo_stream_code_at(const uint16_t *row, const gwo_geom *g, uint32_t c) {
    uint16_t bit = (uint16_t)(1u << (c & 15u));
    int lo = (row[1 + (c >> 4)] & bit) != 0;
    int hi = (row[1 + g->data_blocks + (c >> 4)] & bit) != 0;
    return lo | (hi << 1);
}

int gwo_block_code_at(const uint16_t *row, const gwo_geom *g, uint32_t c) {
    (void)g;
    return (row[1 + (c >> 3)] >> ((c & 7u) * 2u)) & 3;
}

/* ------------------------------------------------------------------------ */
/* CaseControlSet masks, case_control_set.cpp:77-113 (setCases) / :115-150    */
/* ------------------------------------------------------------------------ */

int gwo_ccs_set(const int32_t *idx, size_t n_idx, uint32_t n, uint16_t *block_mask,
                uint16_t *stream_mask) {
    gwo_geom gb, gs;
    gwo_geometry_block(n, &gb);  /* block_count  = (n>>3)+1 -> x4, :37,46-48 */
    gwo_geometry_stream(n, &gs); /* stream_count = (n>>4)+1 -> x4, :38,50-52 */
    if (block_mask) memset(block_mask, 0, (size_t)gb.data_blocks * 2u);
    if (stream_mask) memset(stream_mask, 0, (size_t)gs.data_blocks * 2u);
    /* membership bitmap so that duplicates count once, like std::set<int> */
    uint8_t *member = (uint8_t *)calloc((size_t)n / 8u + 1u, 1);
    int count = 0;
    for (size_t k = 0; k < n_idx; ++k) {
        int32_t i = idx[k];
        if (i < 0 || (uint32_t)i >= n) { free(member); return -1; } /* assert(*it <= max_index) */
        if (member[i >> 3] & (1u << (i & 7))) continue;
        member[i >> 3] |= (uint8_t)(1u << (i & 7));
        ++count;
        if (block_mask) block_mask[i >> 3] |= (uint16_t)(3u << ((i & 7) * 2));
        if (stream_mask) stream_mask[i >> 4] |= (uint16_t)(1u << (i & 15));
    }
    free(member);
    return count;
}

/* ------------------------------------------------------------------------ */
/* stream layout distributions                                               */
/* ------------------------------------------------------------------------ */

/* table5.cpp:577-607 */
void gwo_stream_dist(const uint16_t *row, const gwo_geom *g, gwo_freq *out) {
    pthread_once(&g_pop_once, build_pop16);
    const uint16_t *p0 = row + 1, *p1 = row + 1 + g->data_blocks;
    uint32_t aa = 0, ab = 0, bb = 0, xx = 0;
    for (uint32_t w = 0; w < g->words; ++w) {
        uint64_t a = ld64(p0 + 4 * w), b = ld64(p1 + 4 * w);
        bb += (uint32_t)pop64_lut(a & b);
        ab += (uint32_t)pop64_lut(b);
        aa += (uint32_t)pop64_lut(a);
        xx += (uint32_t)pop64_lut(~(a | b)); /* pad bits included: quirk Q3 */
    }
    out->aa = aa - bb;
    out->ab = ab - bb;
    out->bb = bb;
    out->xx = xx;
}

/* table5.cpp:609-657 -- the north-star word loop */
void gwo_stream_cc_dist(const uint16_t *row, const gwo_geom *g, const uint16_t *cs,
                        const uint16_t *ct, uint32_t case_count, uint32_t ctrl_count,
                        gwo_freq *cases, gwo_freq *ctrls) {
    pthread_once(&g_pop_once, build_pop16);
    const uint16_t *p0 = row + 1, *p1 = row + 1 + g->data_blocks;
    uint32_t caa = 0, cab = 0, cbb = 0, taa = 0, tab = 0, tbb = 0;
    for (uint32_t w = 0; w < g->words; ++w) {
        uint64_t d0 = ld64(p0 + 4 * w), d1 = ld64(p1 + 4 * w);
        uint64_t m = ld64(cs + 4 * w);
        uint64_t a = d0 & m, b = d1 & m;
        cbb += (uint32_t)pop64_lut(a & b);
        cab += (uint32_t)pop64_lut(b);
        caa += (uint32_t)pop64_lut(a);
        m = ld64(ct + 4 * w);
        a = d0 & m;
        b = d1 & m;
        tbb += (uint32_t)pop64_lut(a & b);
        tab += (uint32_t)pop64_lut(b);
        taa += (uint32_t)pop64_lut(a);
    }
    ctrls->aa = taa - tbb;
    ctrls->ab = tab - tbb;
    ctrls->bb = tbb;
    ctrls->xx = ctrl_count - ctrls->aa - ctrls->ab - ctrls->bb; /* :647-649 */
    cases->aa = caa - cbb;
    cases->ab = cab - cbb;
    cases->bb = cbb;
    cases->xx = case_count - cases->aa - cases->ab - cases->bb; /* :651-653 */
}

/* table5.cpp:449-468 */
void gwo_compact_geometry(uint32_t case_count, uint32_t ctrl_count, uint32_t *case_blocks,
                          uint32_t *ctrl_blocks) {
    *case_blocks = round_up4(case_count / 16u + 1u);
    *ctrl_blocks = round_up4(ctrl_count / 16u + 1u);
}

/* table5.cpp:492-575, one row.  Output row layout (:495-499):
 * [case plane0 | case plane1 | ctrl plane0 | ctrl plane1]. */
void gwo_stream_compact_row(const uint16_t *row, const gwo_geom *g, const uint16_t *cs,
                            const uint16_t *ct, uint32_t case_blocks, uint32_t ctrl_blocks,
                            uint16_t *out) {
    memset(out, 0, (size_t)2u * (case_blocks + ctrl_blocks) * sizeof(uint16_t));
    const uint16_t *p0 = row + 1, *p1 = row + 1 + g->data_blocks;
    uint16_t *c0 = out, *c1 = out + case_blocks;
    uint16_t *t0 = out + 2u * case_blocks, *t1 = t0 + ctrl_blocks;
    uint64_t cw0 = 0, cw1 = 0, tw0 = 0, tw1 = 0;
    uint64_t cbit = 1, tbit = 1;
    uint32_t cpos = 0, tpos = 0; /* 64-bit word index in the outputs */
    for (uint32_t w = 0; w < g->words; ++w) {
        uint64_t a = ld64(p0 + 4 * w), b = ld64(p1 + 4 * w);
        uint64_t mc = ld64(cs + 4 * w), mt = ld64(ct + 4 * w);
        for (uint64_t bit = 1; bit != 0; bit <<= 1) {
            if (cbit == 0) {
                memcpy(c0 + 4 * cpos, &cw0, 8);
                memcpy(c1 + 4 * cpos, &cw1, 8);
                ++cpos; cw0 = cw1 = 0; cbit = 1;
            }
            if (tbit == 0) {
                memcpy(t0 + 4 * tpos, &tw0, 8);
                memcpy(t1 + 4 * tpos, &tw1, 8);
                ++tpos; tw0 = tw1 = 0; tbit = 1;
            }
            if (mc & bit) { /* a case wins over a control if both are set (:531-548) */
                if (a & bit) cw0 |= cbit;
                if (b & bit) cw1 |= cbit;
                cbit <<= 1;
            } else if (mt & bit) {
                if (a & bit) tw0 |= tbit;
                if (b & bit) tw1 |= tbit;
                tbit <<= 1;
            }
        }
    }
    if (cbit != 1) { memcpy(c0 + 4 * cpos, &cw0, 8); memcpy(c1 + 4 * cpos, &cw1, 8); }
    if (tbit != 1) { memcpy(t0 + 4 * tpos, &tw0, 8); memcpy(t1 + 4 * tpos, &tw1, 8); }
}

/* table5.cpp:659-701 */
void gwo_stream_cc_dist_compacted(const uint16_t *crow, uint32_t case_blocks, uint32_t ctrl_blocks,
                                  uint32_t case_count, uint32_t ctrl_count, gwo_freq *cases,
                                  gwo_freq *ctrls) {
    pthread_once(&g_pop_once, build_pop16);
    const uint16_t *c0 = crow, *c1 = crow + case_blocks;
    const uint16_t *t0 = crow + 2u * case_blocks, *t1 = t0 + ctrl_blocks;
    uint32_t aa = 0, ab = 0, bb = 0;
    for (uint32_t i = 0; i < case_blocks; i += 4) {
        uint64_t a = ld64(c0 + i), b = ld64(c1 + i);
        bb += (uint32_t)pop64_lut(a & b);
        ab += (uint32_t)pop64_lut(b);
        aa += (uint32_t)pop64_lut(a);
    }
    cases->aa = aa - bb; cases->ab = ab - bb; cases->bb = bb;
    cases->xx = case_count - cases->aa - cases->ab - cases->bb;
    aa = ab = bb = 0;
    for (uint32_t i = 0; i < ctrl_blocks; i += 4) {
        uint64_t a = ld64(t0 + i), b = ld64(t1 + i);
        bb += (uint32_t)pop64_lut(a & b);
        ab += (uint32_t)pop64_lut(b);
        aa += (uint32_t)pop64_lut(a);
    }
    ctrls->aa = aa - bb; ctrls->ab = ab - bb; ctrls->bb = bb;
    ctrls->xx = ctrl_count - ctrls->aa - ctrls->ab - ctrls->bb;
}

/* ------------------------------------------------------------------------ */
/* block layout distributions                                                */
/* ------------------------------------------------------------------------ */

#define HI_BITS 0xAAAAAAAAAAAAAAAAull
#define LO_BITS 0x5555555555555555ull

/* DecodeBitStrings2BitBlock, compressed_genotype_table3.h:195-202 */
static inline void decode_block(uint64_t v, uint64_t *aa, uint64_t *ab, uint64_t *bb) {
    uint64_t hi = (v & HI_BITS) >> 1, lo = v & LO_BITS;
    *bb = hi & lo;
    *aa = *bb ^ lo;
    *ab = *bb ^ hi;
}

/* IncrementFrequencyValue2BitBlock, table3.h:204-211: fold the upper half's
 * even bits onto the lower half's odd bits, count 32 bits. */
static inline uint32_t fold_count(uint64_t x) {
    x |= (x & 0x5555555500000000ull) >> 31;
    uint32_t lo = (uint32_t)x;
    return (uint32_t)(g_pop16[(uint16_t)lo] + g_pop16[(uint16_t)(lo >> 16)]);
}

/* table3.cpp:556-594 */
void gwo_block_dist(const uint16_t *row, const gwo_geom *g, gwo_freq *out) {
    pthread_once(&g_pop_once, build_pop16);
    uint32_t aa = 0, ab = 0, bb = 0;
    for (uint32_t w = 0; w < g->words; ++w) {
        uint64_t a, h, b;
        decode_block(ld64(row + 1 + 4 * w), &a, &h, &b);
        aa += fold_count(a);
        ab += fold_count(h);
        bb += fold_count(b);
    }
    out->aa = aa; out->ab = ab; out->bb = bb;
    out->xx = 0; /* left at zero by the reference: quirk Q3 */
}

/* table3.cpp:596-640 */
void gwo_block_cc_dist(const uint16_t *row, const gwo_geom *g, const uint16_t *cs,
                       const uint16_t *ct, gwo_freq *cases, gwo_freq *ctrls) {
    pthread_once(&g_pop_once, build_pop16);
    uint32_t caa = 0, cab = 0, cbb = 0, taa = 0, tab = 0, tbb = 0;
    for (uint32_t w = 0; w < g->words; ++w) {
        uint64_t v = ld64(row + 1 + 4 * w);
        uint64_t a, h, b;
        decode_block(v & ld64(cs + 4 * w), &a, &h, &b);
        caa += fold_count(a); cab += fold_count(h); cbb += fold_count(b);
        decode_block(v & ld64(ct + 4 * w), &a, &h, &b);
        taa += fold_count(a); tab += fold_count(h); tbb += fold_count(b);
    }
    cases->aa = caa; cases->ab = cab; cases->bb = cbb; cases->xx = 0;
    ctrls->aa = taa; ctrls->ab = tab; ctrls->bb = tbb; ctrls->xx = 0;
}

/* ------------------------------------------------------------------------ */
/* pairwise contingency, stream layout                                       */
/* ------------------------------------------------------------------------ */

/* DecodeBitStreams2BitStream with xx, compressed_genotype_table5.h:110-115 */
static inline void decode_stream(uint64_t p0, uint64_t p1, uint64_t cls[4]) {
    cls[3] = ~(p0 | p1);
    cls[2] = p0 & p1;
    cls[0] = p0 ^ cls[2];
    cls[1] = p1 ^ cls[2];
}

static inline void add_pair_word(const uint64_t a[4], const uint64_t b[4], uint32_t out[16]) {
    /* table5.cpp:775-797: the reference guards the xx terms with if(b_xx)/
     * if(a_xx); the guarded terms are zero when the guard is false, so the
     * unguarded sum is the same number. */
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[4 * i + j] += (uint32_t)pop64_lut(a[i] & b[j]);
}

/* table5.cpp:749-800 */
void gwo_stream_contingency(const uint16_t *ra, const uint16_t *rb, const gwo_geom *g,
                            uint32_t out[16]) {
    pthread_once(&g_pop_once, build_pop16);
    memset(out, 0, 16 * sizeof(uint32_t));
    for (uint32_t w = 0; w < g->words; ++w) {
        uint64_t a[4], b[4];
        decode_stream(ld64(ra + 1 + 4 * w), ld64(ra + 1 + g->data_blocks + 4 * w), a);
        decode_stream(ld64(rb + 1 + 4 * w), ld64(rb + 1 + g->data_blocks + 4 * w), b);
        add_pair_word(a, b, out);
    }
}

/* table5.cpp:806-895: the mask is ANDed into the planes BEFORE decoding, so
 * every non-member (and every pad bit) decodes as xx in both markers and
 * lands in xx_xx (quirk Q7). */
void gwo_stream_cc_contingency(const uint16_t *ra, const uint16_t *rb, const gwo_geom *g,
                               const uint16_t *cs, const uint16_t *ct, uint32_t out_case[16],
                               uint32_t out_ctrl[16]) {
    pthread_once(&g_pop_once, build_pop16);
    memset(out_case, 0, 16 * sizeof(uint32_t));
    memset(out_ctrl, 0, 16 * sizeof(uint32_t));
    for (uint32_t w = 0; w < g->words; ++w) {
        uint64_t a0 = ld64(ra + 1 + 4 * w), a1 = ld64(ra + 1 + g->data_blocks + 4 * w);
        uint64_t b0 = ld64(rb + 1 + 4 * w), b1 = ld64(rb + 1 + g->data_blocks + 4 * w);
        uint64_t a[4], b[4];
        uint64_t m = ld64(cs + 4 * w);
        decode_stream(a0 & m, a1 & m, a);
        decode_stream(b0 & m, b1 & m, b);
        add_pair_word(a, b, out_case);
        m = ld64(ct + 4 * w);
        decode_stream(a0 & m, a1 & m, a);
        decode_stream(b0 & m, b1 & m, b);
        add_pair_word(a, b, out_ctrl);
    }
}

/* table3.cpp:733-849; AddToContingency2BitBlock, compressed_genotype_table3.h:110-115 */
void gwo_block_cc_contingency(const uint16_t *ra, const uint16_t *rb, const gwo_geom *g, const uint16_t *cs,
                              const uint16_t *ct, uint32_t out_case[16], uint32_t out_ctrl[16]) {
    pthread_once(&g_pop_once, build_pop16);
    memset(out_case, 0, 16 * sizeof(uint32_t));
    memset(out_ctrl, 0, 16 * sizeof(uint32_t));
    for (uint32_t w = 0; w < g->words; ++w) {
        const uint64_t va = ld64(ra + 1 + 4 * w), vb = ld64(rb + 1 + 4 * w);
        for (int grp = 0; grp < 2; ++grp) {
            const uint64_t m = ld64((grp ? ct : cs) + 4 * w);
            if (!m) continue; /* :757, :779 */
            uint32_t *out = grp ? out_ctrl : out_case;
            uint64_t a[3], b[3];
            decode_block(va & m, &a[0], &a[1], &a[2]);
            decode_block(vb & m, &b[0], &b[1], &b[2]);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) out[4 * i + j] += fold_count(a[i] & b[j]);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* arithmetic on the counts                                                  */
/* ------------------------------------------------------------------------ */

/* algorithms/maf_func.h:46-54.  (2aa+ab) is divided by tot, not 2*tot (Q5). */
void gwo_minor_allele_frequency(const gwo_freq *ft, double *tot, double *maf) {
    double t = (double)ft->aa;
    double m = 2.0 * t;
    t += (double)ft->ab;
    m += (double)ft->ab;
    t += (double)ft->bb;
    m /= t;
    if (m < 0.5) m = 1.0 - m;
    *tot = t;
    *maf = m;
}

/* common_genotype_func.cpp:173-219 */
int gwo_marginal_information(const gwo_freq *cases, const gwo_freq *ctrls, uint32_t n_individs,
                             gwo_marginal *m) {
    memset(m, 0, sizeof *m);
    m->cases = *cases;
    m->controls = *ctrls;
    const uint32_t ca[4] = {cases->aa, cases->ab, cases->bb, cases->xx};
    const uint32_t co[4] = {ctrls->aa, ctrls->ab, ctrls->bb, ctrls->xx};
    uint32_t n_cases = ca[0] + ca[1] + ca[2] + ca[3];
    uint32_t n_ctrls = co[0] + co[1] + co[2] + co[3];
    if (n_cases + n_ctrls != n_individs) return -1;
    uint32_t mar[4];
    const double n = (double)n_individs;
    for (int i = 0; i < 4; ++i) {
        mar[i] = ca[i] + co[i];
        if (mar[i] > 0) {
            double t = (double)mar[i] / n;
            m->marginal_entropy += -(t)*log(t);
        }
        if (ca[i] > 0) {
            double t = (double)ca[i] / n;
            m->marginal_entropy_y += -(t)*log(t);
            m->pbc[i] = (double)ca[i] / (double)n_cases;
            m->pca[i] = (double)ca[i] / (double)mar[i];
        }
        if (co[i] > 0) {
            double t = (double)co[i] / n;
            m->marginal_entropy_y += -(t)*log(t);
            m->pbc[4 + i] = (double)co[i] / (double)n_ctrls;
            m->pca[4 + i] = (double)co[i] / (double)mar[i];
        }
    }
    m->margins.aa = mar[0]; m->margins.ab = mar[1]; m->margins.bb = mar[2]; m->margins.xx = mar[3];
    return 0;
}

/* HWE chi-square, df 1: p = (2aa+ab)/(2n); E = (n p^2, 2 n p q, n q^2);
 * sum (O-E)^2/E over cells with E > 0, in the order aa, ab, bb. */
static double hwe_chi2(double aa, double ab, double bb) {
    double n = aa + ab + bb;
    if (n <= 0.0) return 0.0;
    double p = (2.0 * aa + ab) / (2.0 * n);
    double q = 1.0 - p;
    double e0 = n * p * p, e1 = 2.0 * n * p * q, e2 = n * q * q;
    double x = 0.0, d;
    if (e0 > 0.0) { d = aa - e0; x += d * d / e0; }
    if (e1 > 0.0) { d = ab - e1; x += d * d / e1; }
    if (e2 > 0.0) { d = bb - e2; x += d * d / e2; }
    return x;
}

void gwo_assoc_stats(const gwo_freq *ca, const gwo_freq *co, double out[GWO_N_STATS]) {
    double tot;
    const double caa = ca->aa, cab = ca->ab, cbb = ca->bb;
    const double taa = co->aa, tab = co->ab, tbb = co->bb;
    const double r1 = caa + cab + cbb, r2 = taa + tab + tbb, n = r1 + r2;

    /* [0],[1]: reference MAF; a group with no calls gives 0 instead of 0/0 */
    if (r1 > 0.0) gwo_minor_allele_frequency(ca, &tot, &out[0]); else out[0] = 0.0;
    if (r2 > 0.0) gwo_minor_allele_frequency(co, &tot, &out[1]); else out[1] = 0.0;

    /* [2]: conventional MAF over everyone */
    if (n > 0.0) {
        double p = (2.0 * (caa + taa) + (cab + tab)) / (2.0 * n);
        out[2] = (p > 0.5) ? 1.0 - p : p;
    } else out[2] = 0.0;

    out[3] = hwe_chi2(taa, tab, tbb);
    out[4] = hwe_chi2(caa + taa, cab + tab, cbb + tbb);

    /* [5]: genotypic 2x3 Pearson, cells in the order case aa,ab,bb, ctrl aa,ab,bb */
    {
        const double o[2][3] = {{caa, cab, cbb}, {taa, tab, tbb}};
        const double r[2] = {r1, r2};
        const double c[3] = {caa + taa, cab + tab, cbb + tbb};
        double x = 0.0;
        if (n > 0.0) {
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 3; ++j) {
                    double e = r[i] * c[j] / n;
                    if (e > 0.0) { double d = o[i][j] - e; x += d * d / e; }
                }
        }
        out[5] = x;
    }
    /* [6]: allelic 2x2 Pearson: n (ad-bc)^2 / ((a+b)(c+d)(a+c)(b+d)) */
    {
        double a = 2.0 * caa + cab, b = 2.0 * cbb + cab;
        double c = 2.0 * taa + tab, d = 2.0 * tbb + tab;
        double den = (a + b) * (c + d) * (a + c) * (b + d);
        double det = a * d - b * c;
        out[6] = (den > 0.0) ? (a + b + c + d) * det * det / den : 0.0;
    }
    /* [7]: call rate */
    {
        double members = r1 + (double)ca->xx + r2 + (double)co->xx;
        out[7] = (members > 0.0) ? n / members : 0.0;
    }
}

/* ------------------------------------------------------------------------ */
/* whole-table scans                                                         */
/* ------------------------------------------------------------------------ */

static inline void put_record(uint32_t *counts, double *stats, uint32_t r, const gwo_freq *ca,
                              const gwo_freq *co) {
    uint32_t *c = counts + (size_t)8 * r;
    c[0] = ca->aa; c[1] = ca->ab; c[2] = ca->bb; c[3] = ca->xx;
    c[4] = co->aa; c[5] = co->ab; c[6] = co->bb; c[7] = co->xx;
    if (stats) gwo_assoc_stats(ca, co, stats + (size_t)GWO_N_STATS * r);
}

/* inline_cc_maf, algorithms/maf_func.cpp:292-316 over table5.cpp:609-657 */
void gwo_scan_stream_cc(const uint16_t *table, uint32_t n_snps, const gwo_geom *g,
                        const uint16_t *cs, const uint16_t *ct, uint32_t case_count,
                        uint32_t ctrl_count, uint32_t *counts, double *stats) {
    for (uint32_t r = 0; r < n_snps; ++r) {
        gwo_freq ca, co;
        gwo_stream_cc_dist(table + (size_t)r * g->blocks_per_row, g, cs, ct, case_count, ctrl_count,
                           &ca, &co);
        put_record(counts, stats, r, &ca, &co);
    }
}

void gwo_scan_block_cc(const uint16_t *table, uint32_t n_snps, const gwo_geom *g,
                       const uint16_t *cs, const uint16_t *ct, uint32_t case_count,
                       uint32_t ctrl_count, uint32_t *counts, double *stats) {
    for (uint32_t r = 0; r < n_snps; ++r) {
        gwo_freq ca, co;
        gwo_block_cc_dist(table + (size_t)r * g->blocks_per_row, g, cs, ct, &ca, &co);
        ca.xx = case_count - ca.aa - ca.ab - ca.bb;
        co.xx = ctrl_count - co.aa - co.ab - co.bb;
        put_record(counts, stats, r, &ca, &co);
    }
}

typedef struct {
    const uint16_t *table, *cs, *ct;
    const gwo_geom *g;
    uint32_t r0, r1, case_count, ctrl_count;
    uint32_t *counts;
    double *stats;
} mt_job;

static void *mt_worker(void *arg) {
    mt_job *j = (mt_job *)arg;
    const gwo_geom *g = j->g;
    for (uint32_t r = j->r0; r < j->r1; ++r) {
        const uint16_t *row = j->table + (size_t)r * g->blocks_per_row;
        const uint16_t *p0 = row + 1, *p1 = row + 1 + g->data_blocks;
        uint32_t caa = 0, cab = 0, cbb = 0, taa = 0, tab = 0, tbb = 0;
        for (uint32_t w = 0; w < g->words; ++w) {
            uint64_t d0 = ld64(p0 + 4 * w), d1 = ld64(p1 + 4 * w);
            uint64_t mc = ld64(j->cs + 4 * w), mt = ld64(j->ct + 4 * w);
            uint64_t a = d0 & mc, b = d1 & mc;
            caa += (uint32_t)__builtin_popcountll(a);
            cab += (uint32_t)__builtin_popcountll(b);
            cbb += (uint32_t)__builtin_popcountll(a & b);
            a = d0 & mt; b = d1 & mt;
            taa += (uint32_t)__builtin_popcountll(a);
            tab += (uint32_t)__builtin_popcountll(b);
            tbb += (uint32_t)__builtin_popcountll(a & b);
        }
        gwo_freq ca = {caa - cbb, cab - cbb, cbb, 0}, co = {taa - tbb, tab - tbb, tbb, 0};
        ca.xx = j->case_count - ca.aa - ca.ab - ca.bb;
        co.xx = j->ctrl_count - co.aa - co.ab - co.bb;
        put_record(j->counts, j->stats, r, &ca, &co);
    }
    return NULL;
}

void gwo_scan_stream_cc_mt(const uint16_t *table, uint32_t n_snps, const gwo_geom *g,
                           const uint16_t *cs, const uint16_t *ct, uint32_t case_count,
                           uint32_t ctrl_count, uint32_t *counts, double *stats, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if ((uint32_t)n_threads > n_snps && n_snps > 0) n_threads = (int)n_snps;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    mt_job *jobs = (mt_job *)malloc(sizeof(mt_job) * (size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) {
        uint32_t r0 = (uint32_t)(((uint64_t)n_snps * (uint64_t)t) / (uint64_t)n_threads);
        uint32_t r1 = (uint32_t)(((uint64_t)n_snps * (uint64_t)(t + 1)) / (uint64_t)n_threads);
        jobs[t] = (mt_job){table, cs, ct, g, r0, r1, case_count, ctrl_count, counts, stats};
        pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
    }
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}

/* ------------------------------------------------------------------------ */
/* synthetic input                                                           */
/* ------------------------------------------------------------------------ */

/* affy6 column of data/maf_spectrum.tab (bins 0..50 = MAF 0.00 .. 0.50) */
static const uint32_t k_affy6[51] = {
    23486, 18503, 16090, 16751, 15610, 33044, 17674, 16029, 16403, 15372, 29748, 16045, 14741,
    14874, 13997, 27101, 15027, 13233, 14199, 12800, 25738, 13828, 13188, 13368, 12272, 24398,
    12968, 12653, 12657, 11557, 23319, 12626, 11183, 12830, 11411, 22332, 11922, 10943, 12225,
    10944, 21966, 11463, 10873, 11675, 11194, 21596, 11478, 10541, 11511, 10658, 6014};

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

#define SIM_SNP_MUL 0x632BE59BD9B4E019ull
#define SIM_WORD_MUL 0xD1342543DE82EF95ull
#define SIM_DRAW_MUL 0xA0761D6478BD642Full
#define SIM_PHENO_TAG 0x00CA5EC0117201ull

static inline uint64_t sim_snp_key(uint64_t seed, uint64_t snp) {
    return splitmix64(seed + SIM_SNP_MUL * (snp + 1));
}

/* simulate_data.cpp:176-192: pick a MAF bin with probability proportional to
 * the spectrum column, then p = bin/100 + U{0..999}/100000. */
uint32_t gwo_sim_snp_p_e5(uint64_t seed, uint64_t snp) {
    uint64_t key = sim_snp_key(seed, snp);
    uint64_t total = 0;
    for (int i = 0; i < 51; ++i) total += k_affy6[i];
    uint64_t u = splitmix64(key ^ 1u) % total;
    uint32_t bin = 0;
    while (u >= k_affy6[bin]) { u -= k_affy6[bin]; ++bin; }
    uint32_t r = (uint32_t)(splitmix64(key ^ 2u) % 1000u);
    return bin * 1000u + r;
}

/* 64 Bernoulli(thr/65536) bits from 16 counter-based random words: bit i is
 * set iff the 16-bit number whose k-th bit is ~R_k[i] is below thr (compare
 * from the least significant bit up). */
static inline uint64_t bernoulli64(uint64_t key, uint32_t base, uint32_t thr) {
    uint64_t lt = 0;
    for (uint32_t k = 0; k < 16; ++k) {
        uint64_t r = splitmix64(key + SIM_DRAW_MUL * (uint64_t)(base + k + 1));
        lt = ((thr >> k) & 1u) ? (lt | r) : (lt & r);
    }
    return lt;
}

void gwo_sim_row_stream(uint64_t seed, uint64_t snp, const gwo_geom *g, uint32_t missing_q16,
                        uint16_t *row) {
    memset(row, 0, (size_t)g->blocks_per_row * 2u);
    uint64_t key = sim_snp_key(seed, snp);
    uint32_t p_e5 = gwo_sim_snp_p_e5(seed, snp);
    uint32_t thr = (uint32_t)(((uint64_t)p_e5 * 65536ull) / 100000ull);
    uint16_t *p0 = row + 1, *p1 = row + 1 + g->data_blocks;
    for (uint32_t w = 0; w < g->words; ++w) {
        uint64_t wkey = key + SIM_WORD_MUL * (uint64_t)(w + 1);
        uint64_t m1 = bernoulli64(wkey, 0, thr);  /* allele 1 is the minor allele */
        uint64_t m2 = bernoulli64(wkey, 16, thr); /* allele 2 is the minor allele */
        uint64_t a = ~(m1 ^ m2);                  /* homozygous: codes 1 and 3 */
        uint64_t b = m1 | m2;                     /* carries the minor allele: 2, 3 */
        if (missing_q16) {
            uint64_t miss = bernoulli64(wkey, 32, missing_q16);
            a &= ~miss;
            b &= ~miss;
        }
        uint32_t first = w * 64u;
        uint64_t valid;
        if (first >= g->n_samples) valid = 0;
        else if (g->n_samples - first >= 64u) valid = ~0ull;
        else valid = (1ull << (g->n_samples - first)) - 1ull;
        a &= valid;
        b &= valid;
        memcpy(p0 + 4 * w, &a, 8);
        memcpy(p1 + 4 * w, &b, 8);
    }
    /* header as the packer would write it for alleles A (major) / C (minor)
     * with all three genotypes present: order 7, aa=AA(0), ab=AC(1), bb=CC(5) */
    row[0] = (uint16_t)(0x7000 | (0u << 8) | (1u << 4) | 5u);
}

/* The generator with the reference simulator's exact allele count (data/simulate_data.cpp:191-205: exactly
 * s = floor(p * 2N) 'B' alleles among the 2N alleles of a SNP, shuffled; consecutive alleles pair up into genotypes).
 * A uniformly random arrangement = the s smallest of 2N independent random keys: allele `pos` gets a 32-bit key (one
 * splitmix64 per sample for its two alleles) and the minor alleles are the s positions with the smallest (key, pos).
 * Same rule as the device's k_simulate_exact, stated with a sort. */
#define SIM_ALLELE_MUL 0x9FB21C651E98DF25ull
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
int gwo_sim_row_stream_exact(uint64_t seed, uint64_t snp, const gwo_geom *g, uint32_t missing_q16, uint16_t *row) {
    const uint32_t n = g->n_samples;
    memset(row, 0, (size_t)g->blocks_per_row * 2u);
    uint64_t key = sim_snp_key(seed, snp);
    uint32_t p_e5 = gwo_sim_snp_p_e5(seed, snp);
    uint64_t s = ((uint64_t)p_e5 * 2ull * n) / 100000ull;   /* floor(p * 2N), simulate_data.cpp:193 */
    uint64_t *e = (uint64_t *)malloc((size_t)2u * n * sizeof(uint64_t));
    if (!e) return -1;
    for (uint32_t c = 0; c < n; ++c) {
        uint64_t h = splitmix64(key + SIM_ALLELE_MUL * (uint64_t)(c + 1u));
        e[2u * c] = ((h >> 32) << 32) | (2u * c);
        e[2u * c + 1u] = ((h & 0xFFFFFFFFull) << 32) | (2u * c + 1u);
    }
    uint64_t thr = 0;
    if (s >= 1) {
        uint64_t *sorted = (uint64_t *)malloc((size_t)2u * n * sizeof(uint64_t));
        if (!sorted) { free(e); return -1; }
        memcpy(sorted, e, (size_t)2u * n * sizeof(uint64_t));
        qsort(sorted, (size_t)2u * n, sizeof(uint64_t), cmp_u64);
        thr = sorted[s - 1];
        free(sorted);
    }
    uint16_t *p0 = row + 1, *p1 = row + 1 + g->data_blocks;
    for (uint32_t w = 0; w < g->words; ++w) {
        uint32_t first = w * 64u;
        uint64_t valid;
        if (first >= n) valid = 0;
        else if (n - first >= 64u) valid = ~0ull;
        else valid = (1ull << (n - first)) - 1ull;
        uint64_t m1 = 0, m2 = 0;
        for (uint32_t bit = 0; bit < 64u && first + bit < n; ++bit) {
            uint32_t c = first + bit;
            m1 |= (uint64_t)(s >= 1 && e[2u * c] <= thr) << bit;
            m2 |= (uint64_t)(s >= 1 && e[2u * c + 1u] <= thr) << bit;
        }
        uint64_t a = ~(m1 ^ m2), b = m1 | m2;
        if (missing_q16) {
            uint64_t wkey = key + SIM_WORD_MUL * (uint64_t)(w + 1);
            uint64_t miss = bernoulli64(wkey, 32, missing_q16);
            a &= ~miss;
            b &= ~miss;
        }
        a &= valid;
        b &= valid;
        memcpy(p0 + 4 * w, &a, 8);
        memcpy(p1 + 4 * w, &b, 8);
    }
    free(e);
    row[0] = (uint16_t)(0x7000 | (0u << 8) | (1u << 4) | 5u);
    return 0;
}

void gwo_sim_phenotype_stream(uint64_t seed, const gwo_geom *g, uint16_t *cs, uint16_t *ct,
                              uint32_t *case_count, uint32_t *ctrl_count) {
    uint32_t nc = 0, nt = 0;
    for (uint32_t w = 0; w < g->words; ++w) {
        uint64_t coin = splitmix64((seed ^ SIM_PHENO_TAG) + SIM_WORD_MUL * (uint64_t)(w + 1));
        uint32_t first = w * 64u;
        uint64_t valid;
        if (first >= g->n_samples) valid = 0;
        else if (g->n_samples - first >= 64u) valid = ~0ull;
        else valid = (1ull << (g->n_samples - first)) - 1ull;
        uint64_t c = coin & valid, t = ~coin & valid;
        nc += (uint32_t)__builtin_popcountll(c);
        nt += (uint32_t)__builtin_popcountll(t);
        memcpy(cs + 4 * w, &c, 8);
        memcpy(ct + 4 * w, &t, 8);
    }
    *case_count = nc;
    *ctrl_count = nt;
}

/* ------------------------------------------------------------------------ */
/* BOOST / KSA pre-screen, algorithms/epistasis_func.cpp:404-470             */
/* ------------------------------------------------------------------------ */

double gwo_boost_measure(const uint32_t ca_tab[16], const uint32_t co_tab[16], const gwo_marginal *m1,
                         const gwo_marginal *m2, uint32_t n_individs) {
    const uint32_t mar2[4] = {m2->margins.aa, m2->margins.ab, m2->margins.bb, m2->margins.xx};
    const int n = (int)n_individs; /* an int in the reference (:352-354) */
    double tao = 0.0, im = 0.0;
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) { /* the xx column/row of the 4x4 tables is skipped (:436-445) */
            const uint32_t ca = ca_tab[4 * a + b], co = co_tab[4 * a + b];
            const double pab = (double)(ca + co) / (double)mar2[b];
            const double t2 = (pab) * (m2->pbc[b]) * (m1->pca[a]);
            const double t3 = (pab) * (m2->pbc[4 + b]) * (m1->pca[4 + a]);
            tao += t2 + t3;
            if (ca > 0) {
                const double t1 = (double)ca / n;
                im += t1 * log(t1);
                if (t2 > 0) im += -t1 * log(t2);
            }
            if (co > 0) {
                const double t1 = (double)co / n;
                im += t1 * log(t1);
                if (t3 > 0) im += -t1 * log(t3);
            }
        }
    }
    return (im + log(tao)) * n * 2.0;
}

void gwo_boost_all_pairs(const uint16_t *table, uint32_t n_snps, const gwo_geom *g, const uint16_t *cs,
                         const uint16_t *ct, uint32_t case_count, uint32_t ctrl_count, double *measures) {
    const uint32_t n_individs = case_count + ctrl_count;
    gwo_marginal *mar = (gwo_marginal *)calloc(n_snps, sizeof(gwo_marginal));
    for (uint32_t i = 0; i < n_snps; ++i) { /* computeMargins, epistasis_func.cpp:706-721 */
        gwo_freq ca, co;
        gwo_stream_cc_dist(table + (size_t)i * g->blocks_per_row, g, cs, ct, case_count, ctrl_count, &ca, &co);
        gwo_marginal_information(&ca, &co, n_individs, &mar[i]);
    }
    const uint32_t pad = g->words * 64u;
    size_t k = 0;
    for (uint32_t i = 0; i < n_snps; ++i)
        for (uint32_t j = i + 1; j < n_snps; ++j) {
            uint32_t a[16], b[16];
            gwo_stream_cc_contingency(table + (size_t)i * g->blocks_per_row, table + (size_t)j * g->blocks_per_row,
                                      g, cs, ct, a, b);
            /* margin-assisted getter: exact member tables (table5.cpp:1031,1062) */
            a[15] -= pad - case_count;
            b[15] -= pad - ctrl_count;
            measures[k++] = gwo_boost_measure(a, b, &mar[i], &mar[j], n_individs);
        }
    free(mar);
}

/* computeGTest, algorithms/epistasis_func.cpp:508-704, for one pair */
void gwo_gtest(const uint32_t ca_tab[16], const uint32_t co_tab[16], const gwo_marginal *m1,
               const gwo_marginal *m2, uint32_t n_individs, double *measure, double *zval) {
    const uint32_t f1c[3] = {m1->cases.aa, m1->cases.ab, m1->cases.bb};
    const uint32_t f1t[3] = {m1->controls.aa, m1->controls.ab, m1->controls.bb};
    const uint32_t f2c[3] = {m2->cases.aa, m2->cases.ab, m2->cases.bb};
    const uint32_t f2t[3] = {m2->controls.aa, m2->controls.ab, m2->controls.bb};
    double mu_ca[9], mu_co[9], old_ca[9], old_co[9];
    for (int i = 0; i < 9; ++i) { mu_ca[i] = mu_co[i] = 1.0; old_ca[i] = old_co[i] = 0.0; }
    double err = 18.0 * fabs(mu_ca[0] - old_ca[0]); /* :552-556 */
    while (err > 0.001) {
        memcpy(old_ca, mu_ca, sizeof mu_ca);
        memcpy(old_co, mu_co, sizeof mu_co);
        double ik[6] = {0, 0, 0, 0, 0, 0}, jk[6] = {0, 0, 0, 0, 0, 0};
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) { /* :571-598 */
                const int i = 3 * a + b;
                const uint32_t tot = ca_tab[4 * a + b] + co_tab[4 * a + b];
                const double mij = mu_ca[i] + mu_co[i];
                if (mij > 0) {
                    mu_ca[i] = mu_ca[i] * (tot) / mij;
                    mu_co[i] = mu_co[i] * (tot) / mij;
                } else {
                    mu_ca[i] = 0;
                    mu_co[i] = 0;
                }
                ik[a] += mu_ca[i]; ik[3 + a] += mu_co[i];
                jk[b] += mu_ca[i]; jk[3 + b] += mu_co[i];
            }
        err = 0.0;
        for (int a = 0; a < 3; ++a) { /* :611-645 */
            const double t1 = ik[a] > 0 ? f1c[a] / ik[a] : 0.0;
            const double t2 = ik[3 + a] > 0 ? f1t[a] / ik[3 + a] : 0.0;
            for (int b = 0; b < 3; ++b) {
                const int i = 3 * a + b;
                const double t3 = jk[b] > 0 ? f2c[b] / jk[b] : 0.0;
                const double t4 = jk[3 + b] > 0 ? f2t[b] / jk[3 + b] : 0.0;
                mu_ca[i] = mu_ca[i] * (t1)*t3;
                mu_co[i] = mu_co[i] * (t2)*t4;
                err += fabs(mu_ca[i] - old_ca[i]);
                err += fabs(mu_co[i] - old_co[i]);
            }
        }
    }
    double tao = 0.0, im = 0.0; /* :648-683 */
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            const int i = 3 * a + b;
            const uint32_t ca = ca_tab[4 * a + b], co = co_tab[4 * a + b];
            double t1, t2;
            if (ca > 0) { t1 = (double)ca / n_individs; im += t1 * log(t1); } else t1 = 0.0;
            if (mu_ca[i] > 0) { t2 = mu_ca[i] / n_individs; im += -t1 * log(t2); tao += t2; }
            if (co > 0) { t1 = (double)co / n_individs; im += t1 * log(t1); } else t1 = 0.0;
            if (mu_co[i] > 0) { t2 = mu_co[i] / n_individs; im += -t1 * log(t2); tao += t2; }
        }
    *measure = (im + log(tao)) * n_individs * 2.0;
    /* allelic joint distribution and the odds-ratio z value, :686-701 (uint arithmetic) */
    uint32_t aj[8];
    const uint32_t *t = ca_tab;
    for (int g = 0; g < 2; ++g, t = co_tab) {
        aj[4 * g + 0] = (t[0] << 2) + (t[1] << 1) + (t[4] << 1) + t[5];
        aj[4 * g + 1] = (t[2] << 2) + (t[1] << 1) + (t[6] << 1) + t[5];
        aj[4 * g + 2] = (t[8] << 2) + (t[9] << 1) + (t[4] << 1) + t[5];
        aj[4 * g + 3] = (t[10] << 2) + (t[9] << 1) + (t[6] << 1) + t[5];
    }
    const double or_aff = log((double)(aj[0] * aj[3]) / (double)(aj[1] * aj[2]));
    const double v_aff = 1 / (double)aj[0] + 1 / (double)aj[1] + 1 / (double)aj[2] + 1 / (double)aj[3];
    const double or_unf = log((double)(aj[4] * aj[7]) / (double)(aj[5] * aj[6]));
    const double v_unf = 1 / (double)aj[4] + 1 / (double)aj[5] + 1 / (double)aj[6] + 1 / (double)aj[7];
    *zval = (or_aff - or_unf) / sqrt(v_aff + v_unf);
}

/* ------------------------------------------------------------------------ */
/* pairwise LLR, src/test/pairwise.c:50-133                                  */
/* ------------------------------------------------------------------------ */

double gwo_pairwise_llr(const uint32_t cs[9], const uint32_t ct[9]) {
    double cn[3][3];
    double cs1[3] = {0, 0, 0}, cs2[3] = {0, 0, 0}, ct1[3] = {0, 0, 0}, ct2[3] = {0, 0, 0};
    double c1[3] = {0, 0, 0}, c2[3] = {0, 0, 0};
    double ns = 0, nt = 0, n = 0;
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) {
            double s = cs[3 * a + b], t = ct[3 * a + b];
            cn[a][b] = s + t;
            cs1[a] += s; cs2[b] += s;
            ct1[a] += t; ct2[b] += t;
            c1[a] += cn[a][b]; c2[b] += cn[a][b];
        }
        ns += cs1[a]; nt += ct1[a]; n += c1[a];
    }
    double ll = 0.0, tao = 0.0;
    for (int a = 0; a < 3; ++a) {
        double psa = cs1[a] / c1[a], pta = 1.0 - psa;
        for (int b = 0; b < 3; ++b) {
            double s = cs[3 * a + b], t = ct[3 * a + b];
            double pab = cn[a][b] / c2[b];
            double pbs = cs2[b] / ns, pbt = ct2[b] / nt;
            if (s > 0) ll += s * log(s / n);
            if (t > 0) ll += t * log(t / n);
            double ps = pab * pbs * psa, pt = pab * pbt * pta;
            tao += ps + pt;
            if (ps > 0) ll -= s * log(ps);
            if (pt > 0) ll -= t * log(pt);
        }
    }
    ll += n * log(tao);
    return 2.0 * ll;
}
