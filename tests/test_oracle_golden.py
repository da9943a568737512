"""Pins the CPU oracle (oracle/gw_oracle.c) to the reference's own golden
vectors: the per-SNP genotype counts that scripts/perl/genotype_set_builder.pl
writes next to the random .tped/.tfam it generates.
CPU only."""
import numpy as np
import pytest

from golden_util import (expected_in_table_space, golden_sets, read_dist, read_tfam_status,
                         read_tped)

PLAIN = golden_sets("plain")
CC = golden_sets("cc")


def _n_of(name):
    return int(name.split("_n")[1])


def test_fixture_inventory():
    assert len(PLAIN) >= 20 and len(CC) >= 20


def test_geometry_matches_survey_table(oracle):
    # SURVEY.md section 8 geometry table (derived from table3.cpp:44-67, table5.cpp:44-72)
    for n, words_a, words_b, stride in [(10_000, 313, 157, None), (100_000, 3126, 1563, 25_010),
                                        (500_000, 15_626, 7813, 125_010)]:
        ga, gb = oracle.geometry_block(n), oracle.geometry_stream(n)
        assert ga.words == words_a and gb.words == words_b
        if stride:
            assert ga.blocks_per_row * 2 == stride and gb.blocks_per_row * 2 == stride
    assert oracle.geometry_block(10_000).blocks_per_row * 2 == 2506
    assert oracle.geometry_stream(10_000).blocks_per_row * 2 == 2514


def test_lut_popcount(oracle):
    rng = np.random.default_rng(1)
    for v in [0, 1, 0xFFFF, 0xFFFFFFFFFFFFFFFF, 0x8000000000000000] + list(
            rng.integers(0, 2**63, size=200, dtype=np.uint64)):
        assert oracle.lib().gwo_popcount64(int(v)) == bin(int(v)).count("1")


@pytest.mark.parametrize("name", PLAIN)
def test_plain_distribution(oracle, name):
    """getGenotypeDistribution, both 2-bit layouts, vs *.expected.dist."""
    n = _n_of(name)
    lines = read_tped(name)
    exp = read_dist(name, "dist")
    assert len(lines) == len(exp)
    texts = [oracle.tped_genotype_text(ln) for ln in lines]
    gs, gb = oracle.geometry_stream(n), oracle.geometry_block(n)
    ts = oracle.pack_rows(texts, n, "stream")
    tb = oracle.pack_rows(texts, n, "block")
    all_idx = np.arange(n, dtype=np.int32)
    bm_all, sm_all, cnt = oracle.ccs_masks(all_idx, n)
    bm_none, sm_none, _ = oracle.ccs_masks(np.zeros(0, np.int32), n)
    assert cnt == n
    for r, ln in enumerate(lines):
        aa, ab, bb, xx = expected_in_table_space(exp[r], ln)
        assert aa + ab + bb + xx == n
        # stream layout: xx also counts the pad bits of the last words (quirk Q3)
        pad = gs.words * 64 - n
        assert oracle.stream_dist(ts[r], gs) == (aa, ab, bb, xx + pad)
        # block layout: xx left at 0 (quirk Q3)
        assert oracle.block_dist(tb[r], gb) == (aa, ab, bb, 0)
        # everyone a case: the cc path returns the exact missing count
        ca, co = oracle.stream_cc_dist(ts[r], gs, sm_all, sm_none, n, 0)
        assert ca == (aa, ab, bb, xx) and co == (0, 0, 0, 0)
        ca, co = oracle.block_cc_dist(tb[r], gb, bm_all, bm_none)
        assert ca == (aa, ab, bb, 0) and co == (0, 0, 0, 0)


@pytest.mark.parametrize("name", CC)
def test_case_control_distribution(oracle, name):
    """getCaseControlGenotypeDistribution: inline masks (both layouts) and the
    selectCaseControl + compacted variant, vs *.expected.{case,control}.dist."""
    n = _n_of(name)
    lines = read_tped(name)
    exp_case, exp_ctrl = read_dist(name, "case.dist"), read_dist(name, "control.dist")
    cases, ctrls = read_tfam_status(name)
    assert len(cases) + len(ctrls) == n
    texts = [oracle.tped_genotype_text(ln) for ln in lines]
    gs, gb = oracle.geometry_stream(n), oracle.geometry_block(n)
    ts = oracle.pack_rows(texts, n, "stream")
    tb = oracle.pack_rows(texts, n, "block")
    bm_c, sm_c, nc = oracle.ccs_masks(cases, n)
    bm_t, sm_t, nt = oracle.ccs_masks(ctrls, n)
    assert (nc, nt) == (len(cases), len(ctrls))
    for r, ln in enumerate(lines):
        ec = expected_in_table_space(exp_case[r], ln)
        et = expected_in_table_space(exp_ctrl[r], ln)
        assert sum(ec) == nc and sum(et) == nt
        assert oracle.stream_cc_dist(ts[r], gs, sm_c, sm_t, nc, nt) == (ec, et)
        assert oracle.stream_cc_dist_selected(ts[r], gs, sm_c, sm_t, nc, nt) == (ec, et)
        ca, co = oracle.block_cc_dist(tb[r], gb, bm_c, bm_t)
        assert ca[:3] == ec[:3] and co[:3] == et[:3] and ca[3] == 0 and co[3] == 0
    # whole-table drivers return the same records
    counts, stats = oracle.scan_stream_cc(ts, gs, sm_c, sm_t, nc, nt)
    counts_b, _ = oracle.scan_block_cc(tb, gb, bm_c, bm_t, nc, nt)
    counts_mt, stats_mt = oracle.scan_stream_cc(ts, gs, sm_c, sm_t, nc, nt, threads=3)
    for r, ln in enumerate(lines):
        ec = expected_in_table_space(exp_case[r], ln)
        et = expected_in_table_space(exp_ctrl[r], ln)
        assert tuple(counts[r]) == ec + et
    assert np.array_equal(counts, counts_b) and np.array_equal(counts, counts_mt)
    assert np.array_equal(stats, stats_mt)


@pytest.mark.parametrize("name", [s for s in PLAIN if _n_of(s) <= 257])
def test_call_round_trip(oracle, name):
    """--valid-geno analogue (validation_tests/validate_func.cpp:31-55): every
    call decodes back to what was written, in both layouts."""
    n = _n_of(name)
    lines = read_tped(name)
    texts = [oracle.tped_genotype_text(ln) for ln in lines]
    gs, gb = oracle.geometry_stream(n), oracle.geometry_block(n)
    ts = oracle.pack_rows(texts, n, "stream")
    tb = oracle.pack_rows(texts, n, "block")
    import ctypes as C
    hdr = (C.c_uint16 * 4)()
    sym = "ACGT"
    for r, ln in enumerate(lines):
        al = ln.split()[4:]
        oracle.lib().gwo_parse_header(int(ts[r, 0]), hdr)
        assert ts[r, 0] == tb[r, 0]
        for c in range(n):
            a, b = al[2 * c], al[2 * c + 1]
            cs = oracle.lib().gwo_stream_code_at(ts[r].ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(gs), c)
            cb = oracle.lib().gwo_block_code_at(tb[r].ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(gb), c)
            assert cs == cb
            if a == "0":
                assert cs == 0
            else:
                enc = hdr[cs]
                assert sym[enc >> 2] == a and sym[enc & 3] == b


def test_pack_codes_helper_equals_text_packer(oracle):
    rng = np.random.default_rng(5)
    for n in (1, 15, 16, 17, 63, 64, 65, 130, 1000):
        codes = rng.integers(0, 4, size=(5, n), dtype=np.uint8)
        codes[:, 0] = 1  # first homozygote is 'AA' -> code space == allele space
        if n > 2:
            codes[:, 1] = 2
            codes[:, 2] = 3
        texts = [oracle.codes_to_text(codes[r]) for r in range(5)]
        for layout in ("stream", "block"):
            a = oracle.pack_rows(texts, n, layout)
            b = oracle.pack_codes(codes, layout)
            if n > 2:
                assert np.array_equal(a, b)
            else:
                assert np.array_equal(a[:, 1:], b[:, 1:])


def test_packer_rejects_what_the_reference_asserts_on(oracle):
    n = 4
    # two heterozygote spellings in one row (common_genotype.h:282-294)
    with pytest.raises(ValueError):
        oracle.pack_rows([b"AC\tCA\tAA\tCC\t"], n)
    # a fourth distinct genotype
    with pytest.raises(ValueError):
        oracle.pack_rows([b"AA\tAC\tCC\tGG\t"], n)
    # unknown symbols are missing calls, not errors (quirk Q2)
    t = oracle.pack_rows([b"AB\t00\tNN\tAA\t"], n)
    g = oracle.geometry_stream(n)
    assert oracle.stream_dist(t[0], g)[:3] == (1, 0, 0)


@pytest.mark.parametrize("name", [s for s in CC if _n_of(s) <= 1000])
def test_pair_tables_agree_across_compression_levels(oracle, name):
    """What scripts/perl/contin_output_comparer.pl checks on two --epi-debug dumps: the
    case/control pair tables of the 2-bit block table (table3.cpp:733-849) and of the 2-bit stream
    table (table5.cpp:806-895) hold the same nine joint-genotype cells; and the cells add up to the
    single-SNP counts the golden files give."""
    n = _n_of(name)
    lines = read_tped(name)
    exp_case = read_dist(name, "case.dist")
    cases, ctrls = read_tfam_status(name)
    texts = [oracle.tped_genotype_text(ln) for ln in lines]
    gs, gb = oracle.geometry_stream(n), oracle.geometry_block(n)
    ts, tb = oracle.pack_rows(texts, n, "stream"), oracle.pack_rows(texts, n, "block")
    bm_c, sm_c, nc = oracle.ccs_masks(cases, n)
    bm_t, sm_t, nt = oracle.ccs_masks(ctrls, n)
    m = len(lines)
    for i in range(m):
        for j in range(i + 1, m):
            sc, st = oracle.stream_cc_contingency(ts[i], ts[j], gs, sm_c, sm_t)
            bc, bt = oracle.block_cc_contingency(tb[i], tb[j], gb, bm_c, bm_t)
            nine = [4 * a + b for a in range(3) for b in range(3)]
            assert np.array_equal(sc[nine], bc[nine]) and np.array_equal(st[nine], bt[nine])
            assert not bc[[3, 7, 11, 12, 13, 14, 15]].any()        # CGT3 leaves the xx margins at zero
            # row sums of the case table = marker i's case counts (no missing calls in these files)
            ei = expected_in_table_space(exp_case[i], lines[i])
            assert tuple(int(sc[4 * a:4 * a + 3].sum()) for a in range(3)) == ei[:3]
            # the stream table sends every non-member and pad bit to xx_xx (quirk Q7)
            assert int(sc.sum()) == gs.words * 64 and int(sc[15]) == gs.words * 64 - nc
