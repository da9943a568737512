"""Parity of the HIP path (through the C-ABI, libgwaspp_amd/libgwscan.so) with
the CPU oracle and with the reference's golden files.  Integer counts are
bit-exact; FP64 statistics within rel 1e-10 (BASELINE.json north_star).
Needs a real MI355X: pytest -m gpu."""
import os

import numpy as np
import pytest

import libgwaspp_amd as gw
from libgwaspp_amd import capi
from golden_util import (expected_in_table_space, golden_sets, read_dist, read_tfam_status,
                         read_tped)

pytestmark = pytest.mark.gpu

RTOL = 1e-10   # north_star: double-precision chi-square within 1e-10 of the CPU path
ATOL = 1e-12


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if gw.device_count() == 0:
        pytest.fail("no HIP device visible: the gpu tests must run on the GPU box "
                    "(the product has no CPU fallback)")


def _assert_stats(got, want):
    assert got.shape == want.shape
    assert np.all(np.isfinite(got))
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)


def _random_codes(rng, m, n, p_missing=0.02):
    """codes in table space; row r has its own allele frequency."""
    q = rng.uniform(0.02, 0.5, size=(m, 1))
    a = rng.random((m, n)) < q
    b = rng.random((m, n)) < q
    codes = np.where(a & b, 3, np.where(a | b, 2, 1)).astype(np.uint8)
    codes[rng.random((m, n)) < p_missing] = 0
    return codes


def _random_masks(oracle, rng, n, p_case=0.45, p_ctrl=0.45):
    u = rng.random(n)
    cases = np.flatnonzero(u < p_case).astype(np.int32)
    ctrls = np.flatnonzero((u >= p_case) & (u < p_case + p_ctrl)).astype(np.int32)
    _, cs, nc = oracle.ccs_masks(cases, n)
    _, ct, nt = oracle.ccs_masks(ctrls, n)
    return cs, ct, nc, nt


# --------------------------------------------------------------------------
# the reference's golden files, end to end through the drop-in boundary
# --------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_sets("cc"))
def test_golden_case_control(oracle, name):
    n = int(name.split("_n")[1])
    lines = read_tped(name)
    exp_case, exp_ctrl = read_dist(name, "case.dist"), read_dist(name, "control.dist")
    cases, ctrls = read_tfam_status(name)
    cs, nc = gw.build_stream_mask(cases, n)
    ct, nt = gw.build_stream_mask(ctrls, n)
    with gw.GwTable(n, len(lines)) as t:
        for r, ln in enumerate(lines):
            t.add_genotype_row(r, oracle.tped_genotype_text(ln))
        t.set_case_control(cs, ct, nc, nt)
        counts, stats = t.scan_case_control()
        for r, ln in enumerate(lines):
            want = expected_in_table_space(exp_case[r], ln) + expected_in_table_space(exp_ctrl[r], ln)
            assert tuple(int(v) for v in counts[r]) == want
            _assert_stats(stats[r], oracle.assoc_stats(want[:4], want[4:]))


@pytest.mark.parametrize("name", golden_sets("plain"))
def test_golden_distribution(oracle, name):
    n = int(name.split("_n")[1])
    lines = read_tped(name)
    exp = read_dist(name, "dist")
    with gw.GwTable(n, len(lines)) as t:
        for r, ln in enumerate(lines):
            t.add_genotype_row(r, oracle.tped_genotype_text(ln))
        dist = t.scan_distribution()
        hdr = t.headers(0, len(lines))
        ref_rows = oracle.pack_rows([oracle.tped_genotype_text(ln) for ln in lines], n)
        assert np.array_equal(hdr, ref_rows[:, 0])
        for r, ln in enumerate(lines):
            assert tuple(int(v) for v in dist[r]) == expected_in_table_space(exp[r], ln)
        # everyone a case: the cc scan returns the same counts (25 % missing here)
        allm, cnt = gw.build_stream_mask(np.arange(n, dtype=np.int32), n)
        none, _ = gw.build_stream_mask([], n)
        t.set_case_control(allm, none, cnt, 0)
        counts, _ = t.scan_case_control()
        assert np.array_equal(counts[:, :4], dist) and not counts[:, 4:].any()


# --------------------------------------------------------------------------
# seeded random inputs vs the oracle, every kernel shape
# --------------------------------------------------------------------------
SIZES = [(1, 5), (63, 40), (64, 40), (65, 40), (127, 33), (128, 33), (129, 33), (1000, 300),
         (4097, 257), (8192, 64), (10_000, 500), (100_003, 40)]


@pytest.mark.parametrize("n,m", SIZES)
def test_random_vs_oracle_both_layouts(oracle, n, m):
    rng = np.random.default_rng(1000 + n)
    codes = _random_codes(rng, m, n)
    cs, ct, nc, nt = _random_masks(oracle, rng, n)
    gs = oracle.geometry_stream(n)
    ts = oracle.pack_codes(codes, "stream")
    tb = oracle.pack_codes(codes, "block")
    want_counts, want_stats = oracle.scan_stream_cc(ts, gs, cs, ct, nc, nt)
    bm_c, _, _ = oracle.ccs_masks(np.flatnonzero(np.unpackbits(cs.view(np.uint8), bitorder="little")[:n]), n)
    bm_t, _, _ = oracle.ccs_masks(np.flatnonzero(np.unpackbits(ct.view(np.uint8), bitorder="little")[:n]), n)
    # the reference's two layouts agree with each other
    cb, _ = oracle.scan_block_cc(tb, oracle.geometry_block(n), bm_c, bm_t, nc, nt)
    assert np.array_equal(cb, want_counts)
    for layout, table in ((gw.LAYOUT_STREAM, ts), (gw.LAYOUT_BLOCK, tb)):
        with gw.GwTable(n, m) as t:
            t.upload_rows(0, table, layout)
            t.set_case_control(cs, ct, nc, nt)
            counts, stats = t.scan_case_control()
            assert np.array_equal(counts, want_counts)
            _assert_stats(stats, want_stats)
            # counts-only form
            c2, s2 = t.scan_case_control(stats=False)
            assert s2 is None and np.array_equal(c2, want_counts)
            # sub-range
            if m > 7:
                c3, s3 = t.scan_case_control(3, m - 7)
                assert np.array_equal(c3, want_counts[3:m - 4]) and np.array_equal(s3, stats[3:m - 4])
            # plain distribution, exact xx
            dist = t.scan_distribution()
            for r in (0, m // 2, m - 1):
                aa, ab, bb, xx = oracle.stream_dist(ts[r], gs)
                assert tuple(int(v) for v in dist[r]) == (aa, ab, bb, n - aa - ab - bb)
            # round trip in both reference packings
            assert np.array_equal(t.download_rows(0, m, gw.LAYOUT_STREAM), ts)
            assert np.array_equal(t.download_rows(0, m, gw.LAYOUT_BLOCK), tb)


@pytest.mark.parametrize("n,m", [(1000, 700), (20_000, 1200), (70_001, 300)])
def test_every_kernel_shape_gives_identical_results(oracle, n, m):
    rng = np.random.default_rng(77 + n)
    codes = _random_codes(rng, m, n)
    cs, ct, nc, nt = _random_masks(oracle, rng, n)
    ts = oracle.pack_codes(codes, "stream")
    want_counts, want_stats = oracle.scan_stream_cc(ts, oracle.geometry_stream(n), cs, ct, nc, nt)
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, ts)
        t.set_case_control(cs, ct, nc, nt)
        for unroll in (1, 2, 4, 8):
            for lds in (0, 1):
                for nt_loads in (0, 1):
                    for threads, per_cu, pipe in ((256, 4, 1), (512, 2, 0), (1024, 1, 1), (0, 0, 0), (0, 0, 1)):
                        t.set_option(capi.OPT_PIPELINE, pipe)
                        t.set_option(capi.OPT_UNROLL, unroll)
                        t.set_option(capi.OPT_MASK_IN_LDS, lds)
                        t.set_option(capi.OPT_NONTEMPORAL, nt_loads)
                        t.set_option(capi.OPT_BLOCK_THREADS, threads)
                        t.set_option(capi.OPT_BLOCKS_PER_CU, per_cu)
                        counts, stats = t.scan_case_control()
                        assert np.array_equal(counts, want_counts), (unroll, lds, nt_loads, threads, pipe)
                        _assert_stats(stats, want_stats)


def test_mask_edge_cases(oracle):
    n, m = 3001, 50
    rng = np.random.default_rng(9)
    codes = _random_codes(rng, m, n, p_missing=0.1)
    ts = oracle.pack_codes(codes, "stream")
    gs = oracle.geometry_stream(n)
    everyone = np.arange(n, dtype=np.int32)
    cases = everyone[::3]
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, ts)
        with pytest.raises(gw.GwError) as e:   # scan before the masks are installed
            t.scan_case_control()
        assert e.value.code == -4
        combos = {
            "empty": (np.zeros(0, np.int32), np.zeros(0, np.int32)),
            "all cases": (everyone, np.zeros(0, np.int32)),
            "all controls": (np.zeros(0, np.int32), everyone),
            "neither for most": (cases[:5], everyone[1:2]),
            # a sample in both sets: the inline path applies the two masks independently
            "overlap": (everyone[: n // 2 + 100], everyone[n // 2 - 100:]),
        }
        for label, (ca, co) in combos.items():
            _, cs, nc = oracle.ccs_masks(ca, n)
            _, ct, nt = oracle.ccs_masks(co, n)
            t.set_case_control(cs, ct, nc, nt)
            counts, stats = t.scan_case_control()
            wc, ws = oracle.scan_stream_cc(ts, gs, cs, ct, nc, nt)
            assert np.array_equal(counts, wc), label
            _assert_stats(stats, ws)
            g_cs, g_ct, g_nc, g_nt = t.get_case_control()
            assert np.array_equal(g_cs, cs) and np.array_equal(g_ct, ct) and (g_nc, g_nt) == (nc, nt)
        # setAllAsCases-style mask with the pad bits set too (case_control_set.cpp:152-169)
        cs = np.full(gs.data_blocks, 0xFFFF, np.uint16)
        ct = np.zeros(gs.data_blocks, np.uint16)
        t.set_case_control(cs, ct, n, 0)
        counts, _ = t.scan_case_control()
        assert np.array_equal(counts[:, :4], t.scan_distribution())


def test_monomorphic_and_all_missing_rows(oracle):
    n = 777
    codes = np.zeros((4, n), np.uint8)
    codes[1] = 1
    codes[2] = 2
    codes[3] = 3
    ts = oracle.pack_codes(codes, "stream")
    rng = np.random.default_rng(4)
    cs, ct, nc, nt = _random_masks(oracle, rng, n)
    with gw.GwTable(n, 4) as t:
        t.upload_rows(0, ts)
        t.set_case_control(cs, ct, nc, nt)
        counts, stats = t.scan_case_control()
        wc, ws = oracle.scan_stream_cc(ts, oracle.geometry_stream(n), cs, ct, nc, nt)
        assert np.array_equal(counts, wc)
        _assert_stats(stats, ws)
        assert tuple(counts[0]) == (0, 0, 0, nc, 0, 0, 0, nt)


def test_argument_errors():
    with gw.GwTable(100, 10) as t:
        with pytest.raises(gw.GwError) as e:
            t.scan_distribution(5, 6)
        assert e.value.code == -5
        with pytest.raises(gw.GwError) as e:
            t.upload_rows(9, np.zeros((2, 2 * gw.stream_blocks(100) + 1), np.uint16))
        assert e.value.code == -5
        with pytest.raises(gw.GwError) as e:
            t.set_case_control(np.zeros(3, np.uint16), np.zeros(3, np.uint16), 0, 0)
        assert e.value.code == -1
        with pytest.raises(gw.GwError) as e:
            t.add_genotype_row(0, b"AC\tCA\t")
        assert e.value.code == -6
        assert t.scan_distribution(10, 0).shape == (0, 4)
    with gw.GwTable(100, 0) as t:   # empty table
        assert t.scan_distribution().shape == (0, 4)


# --------------------------------------------------------------------------
# pairwise contingency (callers of the scan)
# --------------------------------------------------------------------------
@pytest.mark.parametrize("n", [70, 128, 1000, 4097])
def test_pair_contingency(oracle, n):
    m = 12
    rng = np.random.default_rng(n)
    codes = _random_codes(rng, m, n, p_missing=0.05)
    ts = oracle.pack_codes(codes, "stream")
    gs = oracle.geometry_stream(n)
    cs, ct, nc, nt = _random_masks(oracle, rng, n)
    pairs = np.array([(i, j) for i in range(m) for j in range(i + 1, m)], dtype=np.uint32)
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, ts)
        got = t.pair_contingency(pairs)
        t.set_case_control(cs, ct, nc, nt)
        got_c, got_t = t.pair_case_control_contingency(pairs)
        for k, (i, j) in enumerate(pairs):
            assert np.array_equal(got[k], oracle.stream_contingency(ts[i], ts[j], gs))
            wc, wt = oracle.stream_cc_contingency(ts[i], ts[j], gs, cs, ct)
            assert np.array_equal(got_c[k], wc) and np.array_equal(got_t[k], wt)
        with pytest.raises(gw.GwError):
            t.pair_contingency(np.array([[0, m]], np.uint32))


# --------------------------------------------------------------------------
# the synthetic-input generator: device == oracle restatement
# --------------------------------------------------------------------------
@pytest.mark.parametrize("n,missing", [(1, 0), (64, 0), (1000, 0), (10_000, 655), (33_333, 0)])
def test_simulator_equals_oracle(oracle, n, missing):
    seed, m, first = 0x5EED, 37, 1_000_003
    with gw.GwTable(n, m) as t:
        t.simulate(seed, first, missing)
        t.simulate_phenotype(seed)
        rows = t.download_rows(0, m)
        want, g = oracle.sim_table_stream(seed, first, m, n, missing)
        assert np.array_equal(rows, want)
        cs, ct, nc, nt = t.get_case_control()
        ocs, oct_, onc, ont = oracle.sim_phenotype_stream(seed, n)
        assert np.array_equal(cs, ocs) and np.array_equal(ct, oct_) and (nc, nt) == (onc, ont)
        counts, stats = t.scan_case_control()
        wc, ws = oracle.scan_stream_cc(want, g, ocs, oct_, onc, ont)
        assert np.array_equal(counts, wc)
        _assert_stats(stats, ws)


@pytest.mark.parametrize("n,m", [(1, 3), (64, 20), (129, 40), (5000, 300), (70_001, 200)])
def test_complementary_mask_kernel(oracle, n, m):
    """When the control mask is exactly the complement of the case mask the scan uses one mask
    and controls = all - cases (k_scan_cc_compl); same records as the two-mask kernel."""
    rng = np.random.default_rng(n)
    codes = _random_codes(rng, m, n, p_missing=0.05)
    ts = oracle.pack_codes(codes, "stream")
    gs = oracle.geometry_stream(n)
    is_case = rng.random(n) < 0.4
    _, cs, nc = oracle.ccs_masks(np.flatnonzero(is_case), n)
    _, ct, nt = oracle.ccs_masks(np.flatnonzero(~is_case), n)
    want_counts, want_stats = oracle.scan_stream_cc(ts, gs, cs, ct, nc, nt)
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, ts)
        t.set_case_control(cs, ct, nc, nt)
        assert t.get_option(capi.OPT_COMPLEMENT) == 0   # auto: both masks fit the LDS here
        t.set_option(capi.OPT_COMPLEMENT, 2)
        assert t.get_option(capi.OPT_COMPLEMENT) == 1
        for unroll in (0, 1, 2, 4, 8):
            t.set_option(capi.OPT_UNROLL, unroll)
            counts, stats = t.scan_case_control()
            assert np.array_equal(counts, want_counts)
            _assert_stats(stats, want_stats)
        t.set_option(capi.OPT_COMPLEMENT, 0)
        assert t.get_option(capi.OPT_COMPLEMENT) == 0
        c2, s2 = t.scan_case_control()
        assert np.array_equal(c2, want_counts) and np.array_equal(s2, stats)
        t.set_option(capi.OPT_COMPLEMENT, 2)
        # one sample in neither group: no longer complementary -> two-mask kernel, same API
        if n > 2:
            ct2 = ct.copy()
            drop = int(np.flatnonzero(~is_case)[0])
            ct2[drop >> 4] &= np.uint16(~(1 << (drop & 15)) & 0xFFFF)
            t.set_case_control(cs, ct2, nc, nt - 1)
            assert t.get_option(capi.OPT_COMPLEMENT) == 0
            wc, ws = oracle.scan_stream_cc(ts, gs, cs, ct2, nc, nt - 1)
            counts, stats = t.scan_case_control()
            assert np.array_equal(counts, wc)
            _assert_stats(stats, ws)


def test_batch_text_ingest(oracle):
    """gw_table_add_genotype_rows: many rows packed by host threads == row-at-a-time ingest == oracle."""
    n, m = 777, 300
    rng = np.random.default_rng(3)
    codes = _random_codes(rng, m, n, p_missing=0.1)
    codes[:, 0], codes[:, 1], codes[:, 2] = 1, 2, 3
    texts = [oracle.codes_to_text(codes[r]) for r in range(m)]
    want = oracle.pack_rows(texts, n)
    with gw.GwTable(n, m + 5) as t:
        t.add_genotype_rows(2, texts, n_threads=5)
        assert np.array_equal(t.download_rows(2, m), want)
        assert not t.download_rows(0, 2).any() and not t.download_rows(m + 2, 3).any()
        # mixing with the row-at-a-time form keeps the order of writes
        t.add_genotype_row(2, texts[7])
        t.add_genotype_rows(3, texts[:2])
        got = t.download_rows(2, 3)
        assert np.array_equal(got[0], want[7]) and np.array_equal(got[1:], want[:2])
        # a row the reference asserts on: error names the SNP, earlier content stays
        bad = list(texts[:4]) + [b"AC\tCA\t"]
        with pytest.raises(gw.GwError) as e:
            t.add_genotype_rows(10, bad)
        assert e.value.code == -6 and "SNP 14" in str(e.value)
        with pytest.raises(gw.GwError) as e:
            t.add_genotype_rows(m + 3, texts[:5])   # runs past the table
        assert e.value.code == -5


@pytest.mark.parametrize("layout,lname", [(gw.LAYOUT_STREAM, "stream"), (gw.LAYOUT_BLOCK, "block")])
def test_pitched_row_buffers(oracle, layout, lname):
    """rows `row_stride_bytes` apart with slack between them (a caller's own allocation), both ways."""
    n, m = 1234, 77
    rng = np.random.default_rng(8)
    table = oracle.pack_codes(_random_codes(rng, m, n), lname)
    wide = np.full((m, table.shape[1] + 5), 0xDEAD, dtype=np.uint16)
    wide[:, :table.shape[1]] = table
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, wide, layout)
        assert np.array_equal(t.download_rows(0, m, layout), table)
        back = t.download_rows(0, m, layout, pad_blocks=3)
        assert np.array_equal(back[:, :table.shape[1]], table) and not back[:, table.shape[1]:].any()
        with pytest.raises(gw.GwError):   # a stride shorter than a row
            t.upload_rows(0, table[:, :-1].copy(), layout)


@pytest.mark.parametrize("seed", range(24))
def test_randomized_configurations(oracle, seed):
    """Random table sizes, packings, masks (complementary, overlapping, sparse), missing rates,
    kernel shapes and SNP sub-ranges against the oracle."""
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([rng.integers(1, 200), rng.integers(200, 5000), rng.integers(5000, 40_000)]))
    m = int(rng.integers(1, 150))
    codes = _random_codes(rng, m, n, p_missing=float(rng.choice([0.0, 0.01, 0.3])))
    lname = str(rng.choice(["stream", "block"]))
    table = oracle.pack_codes(codes, lname)
    ts = table if lname == "stream" else oracle.pack_codes(codes, "stream")
    kind = int(rng.integers(0, 4))
    u = rng.random(n)
    if kind == 0:      # every sample a case or a control
        cases, ctrls = np.flatnonzero(u < 0.5), np.flatnonzero(u >= 0.5)
    elif kind == 1:    # some in neither
        cases, ctrls = np.flatnonzero(u < 0.3), np.flatnonzero(u > 0.6)
    elif kind == 2:    # overlap
        cases, ctrls = np.flatnonzero(u < 0.6), np.flatnonzero(u > 0.4)
    else:              # sparse
        cases, ctrls = np.flatnonzero(u < 0.02), np.flatnonzero(u > 0.97)
    _, cs, nc = oracle.ccs_masks(cases, n)
    _, ct, nt = oracle.ccs_masks(ctrls, n)
    gs = oracle.geometry_stream(n)
    want_c, want_s = oracle.scan_stream_cc(ts, gs, cs, ct, nc, nt)
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, table, gw.LAYOUT_STREAM if lname == "stream" else gw.LAYOUT_BLOCK)
        t.set_case_control(cs, ct, nc, nt)
        t.set_option(capi.OPT_UNROLL, int(rng.choice([0, 1, 2, 4, 8])))
        t.set_option(capi.OPT_BLOCK_THREADS, int(rng.choice([0, 256, 512, 1024])))
        t.set_option(capi.OPT_BLOCKS_PER_CU, int(rng.integers(0, 5)))
        t.set_option(capi.OPT_MASK_IN_LDS, int(rng.choice([-1, 0, 1])))
        t.set_option(capi.OPT_NONTEMPORAL, int(rng.integers(0, 2)))
        t.set_option(capi.OPT_PIPELINE, int(rng.choice([-1, 0, 1])))
        t.set_option(capi.OPT_COMPLEMENT, int(rng.integers(0, 3)))
        b = int(rng.integers(0, m))
        k = int(rng.integers(0, m - b + 1))
        counts, stats = t.scan_case_control(b, k)
        assert np.array_equal(counts, want_c[b:b + k])
        _assert_stats(stats, want_s[b:b + k])
        dist = t.scan_distribution(b, k)
        for r in range(b, b + k, max(1, k // 5)):
            aa, ab, bb, _ = oracle.stream_dist(ts[r], gs)
            assert tuple(int(v) for v in dist[r - b]) == (aa, ab, bb, n - aa - ab - bb)


def test_allocation_failure_is_reported_not_fatal():
    """A table that cannot fit in HBM: GW_E_NOMEM with a message, and the device stays usable."""
    info = capi.device_info(0)
    n = 500_000
    too_many = int(info.hbm_bytes / (n / 4.0) * 1.5)
    with pytest.raises(gw.GwError) as e:
        gw.GwTable(n, too_many)
    assert e.value.code == -3, str(e.value)
    with gw.GwTable(1000, 10) as t:      # still works afterwards
        assert t.scan_distribution().shape == (10, 4)


@pytest.mark.parametrize("n,m", [(70, 9), (1000, 777), (8191, 130), (8193, 131), (16400, 257), (24576, 66), (24700, 67),
                                 (32767, 50), (32768, 50), (40000, 33), (57345, 21), (65535, 20), (65536, 20)])
def test_short_row_kernel(oracle, n, m):
    """Rows of at most 256 units (N <= 32 767) go through k_scan_cc_sub by default -- 16 or 32 lanes per
    row, double-buffered batches, masks in registers, packed reductions: same counts and statistics as the
    oracle and as the row-at-a-time kernel, for every lane-group size and step count, ragged row counts
    and sub-ranges."""
    rng = np.random.default_rng(n * 7 + m)
    codes = _random_codes(rng, m, n, p_missing=0.03)
    cs, ct, nc, nt = _random_masks(oracle, rng, n)
    ts = oracle.pack_codes(codes, "stream")
    want_counts, want_stats = oracle.scan_stream_cc(ts, oracle.geometry_stream(n), cs, ct, nc, nt)
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, ts)
        t.set_case_control(cs, ct, nc, nt)
        assert t.get_option(capi.OPT_SHORT_ROWS) == (1 if n <= 32767 else 0)
        counts, stats = t.scan_case_control()
        assert np.array_equal(counts, want_counts)
        _assert_stats(stats, want_stats)
        if m > 9:
            c2, s2 = t.scan_case_control(5, m - 9)
            assert np.array_equal(c2, want_counts[5:m - 4]) and np.array_equal(s2, stats[5:m - 4])
        c1, _ = t.scan_case_control(m - 1, 1)
        assert np.array_equal(c1, want_counts[m - 1:])
        # the plain distribution takes the same traversal (k_scan_dist_sub): exact counts incl. missing
        gs = oracle.geometry_stream(n)
        want_dist = np.array([(lambda d: (d[0], d[1], d[2], n - d[0] - d[1] - d[2]))(oracle.stream_dist(ts[r], gs))
                              for r in range(m)], dtype=np.uint32)
        dist = t.scan_distribution()
        assert np.array_equal(dist, want_dist)
        if m > 9:
            assert np.array_equal(t.scan_distribution(5, m - 9), want_dist[5:m - 4])
        t.set_option(capi.OPT_SHORT_ROWS, 0)
        assert t.get_option(capi.OPT_SHORT_ROWS) == 0
        c3, s3 = t.scan_case_control()
        assert np.array_equal(c3, counts) and np.array_equal(s3, stats)
        assert np.array_equal(t.scan_distribution(), want_dist)
        # everyone a case / nobody a control: counts reach the 16-bit boundary of the packed reduction
        everyone = np.arange(n, dtype=np.int32)
        _, cs_all, nc_all = oracle.ccs_masks(everyone, n)
        _, ct_none, nt_none = oracle.ccs_masks(np.zeros(0, np.int32), n)
        t.set_option(capi.OPT_SHORT_ROWS, 1)
        t.set_case_control(cs_all, ct_none, nc_all, nt_none)
        want_c, want_s = oracle.scan_stream_cc(ts, oracle.geometry_stream(n), cs_all, ct_none, nc_all, nt_none)
        c4, s4 = t.scan_case_control()
        assert np.array_equal(c4, want_c)
        _assert_stats(s4, want_s)


@pytest.mark.parametrize("n,m,seg", [(70_001, 130, 64), (70_001, 130, 300), (70_001, 77, 546), (20_000, 200, 100),
                                     (9_000, 50, 1), (140_000, 40, 512)])
def test_column_segmented_scan(oracle, n, m, seg):
    """GW_OPT_COLUMN_SEGMENT: the samples are walked in segments (masks of a segment in LDS), six partial
    sums per SNP accumulate, a last kernel finishes the rows -- the form very large cohorts (masks beyond
    the LDS) take automatically.  Forced here on small tables: same records as the one-pass scan and the
    oracle, for segments that do and do not divide the row, ranges, and kernel shapes."""
    rng = np.random.default_rng(n + 3 * m + seg)
    codes = _random_codes(rng, m, n, p_missing=0.02)
    cs, ct, nc, nt = _random_masks(oracle, rng, n)
    ts = oracle.pack_codes(codes, "stream")
    want_counts, want_stats = oracle.scan_stream_cc(ts, oracle.geometry_stream(n), cs, ct, nc, nt)
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, ts)
        t.set_case_control(cs, ct, nc, nt)
        assert t.get_option(capi.OPT_COLUMN_SEGMENT) == 0   # auto: these tables scan in one pass
        ref_counts, ref_stats = t.scan_case_control()
        t.set_option(capi.OPT_COLUMN_SEGMENT, seg)
        assert t.get_option(capi.OPT_COLUMN_SEGMENT) == seg
        counts, stats = t.scan_case_control()
        assert np.array_equal(counts, want_counts) and np.array_equal(counts, ref_counts)
        _assert_stats(stats, want_stats)
        assert np.array_equal(stats, ref_stats)
        c2, s2 = t.scan_case_control(3, m - 5)
        assert np.array_equal(c2, counts[3:m - 2]) and np.array_equal(s2, stats[3:m - 2])
        for unroll, threads, pipe in ((2, 256, 1), (4, 512, 0), (1, 1024, 0)):
            t.set_option(capi.OPT_UNROLL, unroll)
            t.set_option(capi.OPT_BLOCK_THREADS, threads)
            t.set_option(capi.OPT_PIPELINE, pipe)
            c3, s3 = t.scan_case_control()
            assert np.array_equal(c3, counts) and np.array_equal(s3, stats), (unroll, threads, pipe)


# --------------------------------------------------------------------------
# masks made by the REAL reference CaseControlSet: tests/golden/ref_ccs/masks.npz holds the bytes behind
# stream_case_begin() / stream_control_begin() of the reference object (written in the build container by
# tests/golden/ref_ccs/make_ref_ccs.py from the reference's own case_control_set.cpp compiled in place); only that
# data travels to the GPU box
# --------------------------------------------------------------------------
@pytest.mark.parametrize("n,m", [(70, 20), (1000, 64), (4097, 33), (10_000, 200)])
def test_scan_with_masks_from_the_real_reference_object(oracle, n, m):
    fixture = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_ccs", "masks.npz")
    with np.load(fixture, allow_pickle=False) as z:
        key = f"scan{n}x{m}"
        cases, ctrls = z[key + "_cases"], z[key + "_ctrls"]
        sc, st, cnt = z[key + "_stream_case"], z[key + "_stream_ctrl"], z[key + "_counts"]
    rng = np.random.default_rng(3 * n + m)
    codes = _random_codes(rng, m, n, p_missing=0.03)
    ts = oracle.pack_codes(codes, "stream")
    nc, nt = int(cnt[0]), int(cnt[1])
    assert nc == cases.size and nt == ctrls.size
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, ts)
        t.set_case_control(sc, st, nc, nt)
        counts, stats = t.scan_case_control()
    # expected counts straight from the code array (not from the oracle's masks)
    for r in range(m):
        want = tuple(int((codes[r][grp] == c).sum()) for grp in (cases, ctrls) for c in (1, 2, 3, 0))
        assert tuple(int(v) for v in counts[r]) == want
        _assert_stats(stats[r], oracle.assoc_stats(want[:4], want[4:]))



# --------------------------------------------------------------------------
# GW_OPT_DEFER_RECORDS: the same records whether they are written as rows finish or in one burst at the end of the
# launch, also when a call is cut into several launches (each piece's parking area is a few KB of LDS per wave) and
# when the last 64-row group of a wave is partial
# --------------------------------------------------------------------------
@pytest.mark.parametrize("n,m,missing", [(2_000, 3_400_003, 655), (9_999, 300_000, 0), (20_000, 70_001, 655),
                                         (70_000, 40_000, 655), (100_000, 9_000, 0), (1_000, 37, 655)])
def test_deferred_records_equal_interleaved_records(oracle, n, m, missing):
    with gw.GwTable(n, m) as t:
        t.simulate(0xD1FE, 5, missing)
        t.simulate_phenotype(0xD1FE)
        t.set_option(capi.OPT_DEFER_RECORDS, 0)
        c0, s0 = t.scan_case_control()
        k0 = t.last_scan_kernel()
        t.set_option(capi.OPT_DEFER_RECORDS, 1)
        assert t.get_option(capi.OPT_DEFER_RECORDS) == 1
        c1, s1 = t.scan_case_control()
        k1 = t.last_scan_kernel()
        assert k0.endswith("false>") and k1.endswith("true>") and k0[:-6] == k1[:-5], (k0, k1)
        assert np.array_equal(c0, c1) and np.array_equal(s0, s1)
        lo, cnt = m // 7, m - m // 7 - 3                       # a sub-range is a slice, counts-only too
        c2, s2 = t.scan_case_control(lo, cnt)
        assert np.array_equal(c2, c0[lo:lo + cnt]) and np.array_equal(s2, s0[lo:lo + cnt])
        # and the oracle on a few rows spread over the pieces
        rows_at = sorted({0, 1, m // 3, m // 2, m - 2, m - 1})
        cs, ct, nc, nt = t.get_case_control()
        g = oracle.geometry_stream(n)
        for r in rows_at:
            row = t.download_rows(r, 1)
            want_c, want_s = oracle.scan_stream_cc(row, g, cs, ct, nc, nt)
            assert np.array_equal(c1[r], want_c[0])
            _assert_stats(s1[r], want_s[0])


@pytest.mark.parametrize("per_cu", [1, 2])
def test_deferred_records_with_few_workgroups_per_cu(per_cu):
    """GW_OPT_BLOCKS_PER_CU 1 / 2 with deferred records on the sub-wave kernels: the parking area stays within the 64 KB of
    dynamic LDS a workgroup may ask for (the launch used to fail there), the records are the interleaved ones."""
    n, m = 9_000, 600_000
    with gw.GwTable(n, m) as t:
        t.simulate(0xD1FE, 5, 0)
        t.simulate_phenotype(0xD1FE)
        t.set_option(capi.OPT_DEFER_RECORDS, 0)
        c0, s0 = t.scan_case_control()
        t.set_option(capi.OPT_BLOCKS_PER_CU, per_cu)
        t.set_option(capi.OPT_DEFER_RECORDS, 1)
        c1, s1 = t.scan_case_control()
        assert t.last_scan_kernel().startswith("gw::k_scan_cc_sub<16") and t.last_scan_kernel().endswith("true>")
        assert np.array_equal(c0, c1) and np.array_equal(s0, s1)


def test_deferred_records_auto_policy():
    """auto: the 16-lane kernels when one call's records exceed the 256 MB Infinity Cache (3 M SNPs x 96 B)."""
    with gw.GwTable(2_000, 3_000_000) as t:
        t.simulate(1, 0, 0)
        t.simulate_phenotype(1)
        assert t.get_option(capi.OPT_DEFER_RECORDS) == 1
        c, s = t.scan_case_control()
        assert t.last_scan_kernel() == "gw::k_scan_cc_sub<16, 1, true, true>"
        c_small, s_small = t.scan_case_control(1000, 50_000)          # a small call: interleaved form
        assert t.last_scan_kernel() == "gw::k_scan_cc_sub<16, 1, true, false>"
        assert np.array_equal(c_small, c[1000:51_000]) and np.array_equal(s_small, s[1000:51_000])
    with gw.GwTable(2_000, 100_000) as t:
        assert t.get_option(capi.OPT_DEFER_RECORDS) == 0


# --------------------------------------------------------------------------
# the exact-count generator (gw_table_simulate_exact): exactly floor(p * 2N) minor alleles per SNP in a random arrangement,
# as data/simulate_data.cpp:191-205 produces by filling and shuffling; device rows == the oracle's restatement bit for bit
# --------------------------------------------------------------------------
@pytest.mark.parametrize("n,m,missing", [(1, 9, 0), (63, 20, 0), (64, 20, 655), (1000, 64, 0), (4097, 33, 655), (10_000, 40, 0),
                                         (70_001, 6, 0)])
def test_exact_count_generator_equals_the_oracle(oracle, n, m, missing):
    seed, begin = 0xE7AC7, 123_456
    with gw.GwTable(n, m) as t:
        t.simulate(seed, begin, missing, exact=True)
        rows = t.download_rows(0, m)
        t.simulate(seed, begin, missing)                 # the binomial generator is another table
        other = t.download_rows(0, m)
    want, g = oracle.sim_table_stream(seed, begin, m, n, missing, exact=True)
    assert np.array_equal(rows, want)
    assert n < 64 or not np.array_equal(rows, other)
    if not missing:                                      # the defining property, straight from the planes
        nb = g.data_blocks
        for r in range(m):
            p0 = np.unpackbits(rows[r, 1:1 + nb].view(np.uint8), bitorder="little")[:n].astype(np.int64)
            p1 = np.unpackbits(rows[r, 1 + nb:1 + 2 * nb].view(np.uint8), bitorder="little")[:n].astype(np.int64)
            minor_alleles = int((p1 & (1 - p0)).sum() + 2 * (p0 & p1).sum())      # code 2: one, code 3: two
            assert minor_alleles == oracle.sim_snp_p_e5(seed, begin + r) * 2 * n // 100000
            assert int((p0 | p1).sum()) == n             # no missing call


def test_exact_count_generator_at_full_row_length(oracle):
    """500 000 samples (config 3's rows): sampled rows of a table generated by global index, against the oracle."""
    n, m, seed = 500_000, 64, 0x5EED
    with gw.GwTable(n, m) as t:
        t.simulate(seed, 999_936, 0, exact=True)         # the last 64 SNPs of the 1 M-SNP job
        rows = t.download_rows(0, m)
    for r in (0, 17, 63):
        want, _ = oracle.sim_table_stream(seed, 999_936 + r, 1, n, 0, exact=True)
        assert np.array_equal(rows[r], want[0])


def test_new_table_reads_as_unknown_everywhere_without_a_create_time_memset(oracle):
    """gw_table_create no longer zero-fills (a table that is generated in place never pays the 20 ms for 125 GB); the fill
    happens at the first use that does not overwrite the whole table.  Every reader must still see 'unknown' everywhere
    (table5.cpp:82-83), whatever comes first: a scan, a download, a partial upload, a pair table."""
    n, m = 1000, 50
    rng = np.random.default_rng(4)
    row7 = oracle.pack_codes(_random_codes(rng, 1, n, p_missing=0.1), "stream")
    with gw.GwTable(n, m) as t:                       # first use: a distribution scan
        assert np.array_equal(t.scan_distribution(), np.tile(np.array([0, 0, 0, n], np.uint32), (m, 1)))
    with gw.GwTable(n, m) as t:                       # first use: a partial upload
        t.upload_rows(7, row7)
        rows = t.download_rows(0, m)
        assert np.array_equal(rows[7, 1:], row7[0, 1:]) and not rows[:7, 1:].any() and not rows[8:, 1:].any()
    with gw.GwTable(n, m) as t:                       # first use: a download
        assert not t.download_rows(0, m)[:, 1:].any()
    with gw.GwTable(n, m) as t:                       # first use: a pair table (every sample missing at both markers)
        tab = t.pair_contingency([(0, 1), (3, 49)])
        assert np.array_equal(tab[:, :15], np.zeros((2, 15), np.uint32)) and np.all(tab[:, 15] == oracle.geometry_stream(n).words * 64)
    with gw.GwTable(n, m) as t:                       # generated in place, then read: nothing of the old contents, no fill
        t.simulate(9, 0, 0, exact=True)
        d = t.scan_distribution()
        assert np.all(d[:, 3] == 0) and np.all(d[:, :3].sum(axis=1) == n)
