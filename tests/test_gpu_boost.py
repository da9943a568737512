"""Pairwise path on the GPU vs the oracle: selectCaseControl's compacted rows
(table5.cpp:443-575), the BOOST / KSA interaction measure of every SNP pair
(algorithms/epistasis_func.cpp:349-506) and its threshold list.  The measure
itself is pinned to the reference's pairwise.c known-answer tables in
tests/test_oracle_arith.py.  Needs a GPU."""
import ctypes as C

import numpy as np
import pytest

import libgwaspp_amd as gw

pytestmark = pytest.mark.gpu


def _codes(rng, m, n, p_missing):
    q = rng.uniform(0.05, 0.5, size=(m, 1))
    a = rng.random((m, n)) < q
    b = rng.random((m, n)) < q
    codes = np.where(a & b, 3, np.where(a | b, 2, 1)).astype(np.uint8)
    if p_missing:
        codes[rng.random((m, n)) < p_missing] = 0
    return codes


def _masks(oracle, rng, n, neither=0.1):
    u = rng.random(n)
    cases = np.flatnonzero(u < 0.5 - neither / 2).astype(np.int32)
    ctrls = np.flatnonzero(u >= 0.5 + neither / 2).astype(np.int32)
    _, cs, nc = oracle.ccs_masks(cases, n)
    _, ct, nt = oracle.ccs_masks(ctrls, n)
    return cs, ct, nc, nt


def _pair_matrix(flat, m):
    out = np.full((m, m), np.nan)
    k = 0
    for i in range(m):
        for j in range(i + 1, m):
            out[i, j] = flat[k]
            k += 1
    return out


@pytest.mark.parametrize("n,m,p_missing", [(70, 9, 0.0), (1000, 40, 0.0), (1000, 40, 0.03), (4097, 70, 0.0),
                                           (4097, 33, 0.01), (129, 65, 0.05), (20_000, 48, 0.0)])
def test_boost_measures_vs_oracle(oracle, n, m, p_missing):
    if gw.device_count() == 0:
        pytest.fail("no HIP device visible")
    rng = np.random.default_rng(n * 31 + m)
    ts = oracle.pack_codes(_codes(rng, m, n, p_missing), "stream")
    gs = oracle.geometry_stream(n)
    cs, ct, nc, nt = _masks(oracle, rng, n)
    want = _pair_matrix(oracle.boost_all_pairs(ts, gs, cs, ct, nc, nt), m)
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, ts)
        t.set_case_control(cs, ct, nc, nt)
        got = t.boost_measures(0, m, 0, m)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want)
        np.testing.assert_allclose(got[ok], want[ok], rtol=1e-10, atol=1e-10)
        # block form: any rectangle of the pair space gives the same numbers
        a0, na, b0, nb = 1, m // 2, m // 3, m - m // 3
        blk = t.boost_measures(a0, na, b0, nb)
        sub = want[a0:a0 + na, b0:b0 + nb]
        ok = ~np.isnan(sub)
        assert np.array_equal(np.isnan(blk), ~ok)
        np.testing.assert_allclose(blk[ok], sub[ok], rtol=1e-10, atol=1e-10)
        # threshold list
        finite = want[~np.isnan(want)]
        thr = float(np.median(finite))
        hits, n_hits, n_pairs, vmax, vmin = t.boost_screen(0, m, 0, m, threshold=thr)
        assert n_pairs == m * (m - 1) // 2
        assert n_hits == int((finite > thr).sum()) == len(hits)
        for h in hits:
            assert h["snp_a"] < h["snp_b"]
            assert h["measure"] == got[h["snp_a"], h["snp_b"]] and h["measure"] > thr
        assert vmax == np.nanmax(got) and vmin == np.nanmin(got)
        # a capped list reports the full count
        few, n_hits2, *_ = t.boost_screen(0, m, 0, m, threshold=thr, max_hits=3)
        assert n_hits2 == n_hits and len(few) == min(3, n_hits)
        # the int8-MFMA kernels (4 products per pair without missing calls, 9 with) are the default: the
        # vector-ALU kernel must give the same counts, hence bit-identical measures
        assert t.get_option(gw.capi.OPT_BOOST_MATRIX_CORES) == 1
        for mode in (0, 2):  # 0: vector ALU; 2: the both-groups MFMA kernels (what large case groups get)
            t.set_option(gw.capi.OPT_BOOST_MATRIX_CORES, mode)
            again = t.boost_measures(0, m, 0, m)
            ok_all = ~np.isnan(got)
            assert np.array_equal(np.isnan(again), np.isnan(got)) and np.array_equal(again[ok_all], got[ok_all])
            blk2 = t.boost_measures(a0, na, b0, nb)
            assert np.array_equal(blk2[~np.isnan(blk2)], blk[~np.isnan(blk)])
            h2, n2, p2, *_ = t.boost_screen(0, m, 0, m, threshold=thr, extremes=False)
            assert (n2, p2) == (n_hits, n_pairs)
            assert np.array_equal(np.sort(h2, order=["snp_a", "snp_b"]), np.sort(hits, order=["snp_a", "snp_b"]))


@pytest.mark.parametrize("n", [33, 70, 1000, 4097, 300_001])
def test_select_case_control_compaction(oracle, n):
    """gw_table_download_selected == the reference's m_cases_controls rows (compressed_genotype_table5.cpp:443-575), from one
    word (n = 33: two) to rows whose compact form needs more than the default 48 KB of dynamic LDS (n = 300 001: 75 KB)."""
    m = 7
    rng = np.random.default_rng(n)
    ts = oracle.pack_codes(_codes(rng, m, n, 0.04), "stream")
    gs = oracle.geometry_stream(n)
    cs, ct, nc, nt = _masks(oracle, rng, n)
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, ts)
        t.set_case_control(cs, ct, nc, nt)
        rows, cb, tb = t.download_selected(0, m)
        ocb, otb = C.c_uint32(), C.c_uint32()
        oracle.lib().gwo_compact_geometry(nc, nt, C.byref(ocb), C.byref(otb))
        assert (cb, tb) == (ocb.value, otb.value)
        for r in range(m):
            want = np.zeros(2 * (cb + tb), np.uint16)
            p16 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint16))
            oracle.lib().gwo_stream_compact_row(p16(ts[r]), C.byref(gs), p16(cs), p16(ct), cb, tb, p16(want))
            assert np.array_equal(rows[r], want)


def test_boost_on_simulated_table_matches_oracle_and_finds_planted_pair(oracle):
    """End to end on the synthetic generator, plus one planted interaction that
    must come out of the threshold list."""
    n, m, seed = 6000, 90, 0x5EED
    rng = np.random.default_rng(5)
    table, gs = oracle.sim_table_stream(seed, 0, m, n)
    cs, ct, nc, nt = oracle.sim_phenotype_stream(seed, n)
    # plant: make SNP 7 x SNP 11 jointly predictive of case status
    is_case = np.unpackbits(cs.view(np.uint8), bitorder="little")[:n].astype(bool)
    codes = np.array([[oracle.lib().gwo_stream_code_at(table[r].ctypes.data_as(C.POINTER(C.c_uint16)),
                                                       C.byref(gs), c) for c in range(n)] for r in (7, 11)],
                     dtype=np.uint8)
    a = rng.integers(1, 4, size=n).astype(np.uint8)
    b = rng.integers(1, 4, size=n).astype(np.uint8)
    flip = is_case & (rng.random(n) < 0.6)
    b[flip] = a[flip]                       # cases: the two genotypes agree far more often
    codes[0], codes[1] = a, b
    planted = oracle.pack_codes(codes, "stream")
    table[7], table[11] = planted[0], planted[1]
    want = _pair_matrix(oracle.boost_all_pairs(table, gs, cs, ct, nc, nt), m)
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, table)
        t.set_case_control(cs, ct, nc, nt)
        hits, n_hits, n_pairs, vmax, _ = t.boost_screen(0, m, 0, m, threshold=30.0)   # the reference's threshold
        assert n_pairs == m * (m - 1) // 2
        assert (7, 11) in {(int(h["snp_a"]), int(h["snp_b"])) for h in hits}
        assert n_hits == int((want[~np.isnan(want)] > 30.0).sum())
        assert vmax == pytest.approx(np.nanmax(want), rel=1e-10)


def _prefilter_table(rng, n, m, p_missing, low_maf):
    """Random table with a few strongly interacting pairs so that the measures span 0 .. thousands."""
    q = rng.uniform(0.01, 0.12, size=(m, 1)) if low_maf else rng.uniform(0.05, 0.5, size=(m, 1))
    a = rng.random((m, n)) < q
    b = rng.random((m, n)) < q
    codes = np.where(a & b, 3, np.where(a | b, 2, 1)).astype(np.uint8)
    if p_missing:
        codes[rng.random((m, n)) < p_missing] = 0
    return codes


@pytest.mark.parametrize("n,m,p_missing,low_maf", [(600, 48, 0.0, False), (600, 48, 0.05, True), (6000, 64, 0.0, True),
                                                   (6000, 40, 0.02, False), (40_000, 40, 0.0, False),
                                                   (40_000, 36, 0.0, True), (250_000, 24, 0.0, False),
                                                   (250_000, 20, 0.01, True)])
def test_boost_prefilter_gives_the_same_hit_list(oracle, n, m, p_missing, low_maf):
    """gw_boost_screen without extremes rules pairs out by a single-precision bound of the measure plus the
    bound's own running error bound (GW_OPT_BOOST_PREFILTER, derivation in gw_boost_dev.h): the hit list and
    its FP64 measures must not change, also for thresholds that sit exactly on a pair's measure; every pair's
    bound must lie within its error bound of the FP64 measure (the guarantee the screen relies on, and checks
    itself at run time: GW_OPT_BOOST_PREFILTER_TRIPS stays 0); and the FP64 repeat that a violation would
    trigger gives the same list."""
    rng = np.random.default_rng(n + 7 * m + int(low_maf))
    codes = _prefilter_table(rng, n, m, p_missing, low_maf)
    u = rng.random(n)
    is_case = u < 0.45
    # plant interactions of different strength: cases copy SNP i's genotype into SNP j
    for (i, j, frac) in ((1, 5, 0.5), (2, 9, 0.15), (3, 11, 0.05), (4, 13, 0.02)):
        flip = is_case & (rng.random(n) < frac)
        codes[j, flip] = codes[i, flip]
    ts = oracle.pack_codes(codes, "stream")
    cases = np.flatnonzero(is_case).astype(np.int32)
    ctrls = np.flatnonzero(u >= 0.55).astype(np.int32)
    _, cs, nc = oracle.ccs_masks(cases, n)
    _, ct, nt = oracle.ccs_masks(ctrls, n)
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, ts)
        t.set_case_control(cs, ct, nc, nt)
        exact = t.boost_measures(0, m, 0, m)
        finite = np.sort(exact[np.isfinite(exact)])
        n_sel = nc + nt

        # the bound and its error bound (diagnostic forms)
        t.set_option(gw.capi.OPT_BOOST_PREFILTER, 2)
        bound = t.boost_measures(0, m, 0, m)
        t.set_option(gw.capi.OPT_BOOST_PREFILTER, 3)
        err_bound = t.boost_measures(0, m, 0, m)
        t.set_option(gw.capi.OPT_BOOST_PREFILTER, 1)
        both = np.isfinite(exact) & np.isfinite(bound) & np.isfinite(err_bound)
        assert both.sum() >= 0.9 * np.isfinite(exact).sum()
        err = np.abs(bound[both] - exact[both])
        assert np.all(err <= err_bound[both]), (err.max(), (err / err_bound[both]).max())
        # ... and for ordinary pairs it is far tighter than the empirical slack (2n * 1e-4 + 0.05) it replaced
        assert np.median(err_bound[both]) < 0.1 * (2.0 * n_sel * 1e-4 + 0.05)
        print(f"n={n}: max |bound - measure| / err_bound = {(err / err_bound[both]).max():.3f}, "
              f"median err_bound = {np.median(err_bound[both]):.4f}")

        thresholds = [30.0, float(np.median(finite)), float(finite[int(0.9 * len(finite))]), float(finite[-1]),
                      float(finite[-2]), float(finite[len(finite) // 3]), -1.0e9, 1.0e9]
        thresholds += [float(v) for v in rng.choice(finite, size=6)]           # exactly on a measure
        thresholds += [float(np.nextafter(v, -np.inf)) for v in rng.choice(finite, size=3)]
        for thr in thresholds:
            ref_hits, ref_n, ref_pairs, _, _ = t.boost_screen(0, m, 0, m, threshold=thr)           # all FP64
            hits, n_hits, n_pairs, vmax, vmin = t.boost_screen(0, m, 0, m, threshold=thr, extremes=False)
            assert vmax is None and vmin is None
            assert (n_hits, n_pairs) == (ref_n, ref_pairs) == (int((finite > thr).sum()), m * (m - 1) // 2)
            assert np.array_equal(np.sort(hits, order=["snp_a", "snp_b"]), np.sort(ref_hits, order=["snp_a", "snp_b"]))
        assert t.get_option(gw.capi.OPT_BOOST_PREFILTER_TRIPS) == 0     # no pair ever left its error bound
        # the repeat in FP64 that a violation triggers (forced here): same list, and it is counted
        thr = float(finite[int(0.9 * len(finite))])
        want_hits, want_n, *_ = t.boost_screen(0, m, 0, m, threshold=thr)
        t.set_option(gw.capi.OPT_BOOST_PREFILTER_TRIPS, 1)
        hits1, n1, p1, *_ = t.boost_screen(0, m, 0, m, threshold=thr, extremes=False)
        t.set_option(gw.capi.OPT_BOOST_PREFILTER_TRIPS, 0)
        assert t.get_option(gw.capi.OPT_BOOST_PREFILTER_TRIPS) == 1
        assert (n1, p1) == (want_n, m * (m - 1) // 2)
        assert np.array_equal(np.sort(hits1, order=["snp_a", "snp_b"]), np.sort(want_hits, order=["snp_a", "snp_b"]))
        # switched off: still the same
        t.set_option(gw.capi.OPT_BOOST_PREFILTER, 0)
        hits0, n0, *_ = t.boost_screen(0, m, 0, m, threshold=30.0, extremes=False)
        assert n0 == int((finite > 30.0).sum())


def _quad_constants(n_cases, n):
    """gw_boost.hip quad_bound_constants, restated: expansion point, the two curvature constants (as the c_m -+ c_d the
    device uses), offset"""
    f32 = np.float32
    p = float(f32(n_cases / n))
    h0 = -(p * np.log2(p) + (1 - p) * np.log2(1 - p))
    h1 = np.log2((1 - p) / p)
    gp = 1.0 / (2.0 * np.log(2.0) * p * (1 - p))
    c_lo = max((h0 - h1 * p) / p ** 2, gp) * (1 + 1e-6)
    c_hi = max((h0 + h1 * (1 - p)) / (1 - p) ** 2, gp) * (1 + 1e-6)
    k0 = float(np.nextafter(f32(-n * h0 - h1 * (n_cases - p * n)), f32(np.inf)))
    if abs(c_hi - c_lo) <= 0.02 * max(c_hi, c_lo):     # balanced cohort: one constant for both sides (LV = 1)
        c = float(np.nextafter(f32(max(c_lo, c_hi)), f32(np.inf)))
        return p, c, c, k0, 1
    c_d = float(f32(0.5 * (c_hi - c_lo)))
    c_m = float(np.nextafter(f32(max(c_lo + c_d, c_hi - c_d)), f32(np.inf)))
    return p, c_m - c_d, c_m + c_d, k0, 2


@pytest.mark.parametrize("n,m,case_share,low_maf", [(600, 48, 0.5, False), (6000, 64, 0.45, True), (6000, 48, 0.15, False),
                                                    (40_000, 40, 0.5, False), (40_000, 36, 0.85, True),
                                                    (250_000, 24, 0.25, False), (250_000, 24, 0.3, False)])
def test_first_level_bound_is_an_upper_bound_within_its_error_model(oracle, n, m, case_share, low_maf):
    """GW_OPT_BOOST_FIRST_LEVEL (k_boost_mfma_fp4<true, 2, 2, 1 or 2>): the logarithm-free bound of gw_boost_dev.h
    boost_bound_quad.  For every pair: measure <= bound1 + err1 (what the screen relies on); the device's value lies
    within err1 of the same formula evaluated in FP64 from counts taken from the genotype codes (the error model);
    the quadratic lies below the binary entropy for this cohort's constants; and the hit lists with the first level
    on, off and without any pre-filter are the same, also for thresholds that sit on a pair's measure.  Balanced and
    lopsided cohorts (one curvature constant where the two sides differ by under 2 %, the two-piece form elsewhere)."""
    rng = np.random.default_rng(5 * n + m + int(100 * case_share))
    codes = _prefilter_table(rng, n, m, 0.0, low_maf)
    u = rng.random(n)
    is_case = u < case_share * 0.9
    is_ctrl = (u >= case_share * 0.9) & (u < 0.9)
    for (i, j, frac) in ((1, 5, 0.5), (2, 9, 0.15), (3, 11, 0.05), (4, 13, 0.02)):
        flip = is_case & (rng.random(n) < frac)
        codes[j, flip] = codes[i, flip]
    ts = oracle.pack_codes(codes, "stream")
    cases, ctrls = np.flatnonzero(is_case).astype(np.int32), np.flatnonzero(is_ctrl).astype(np.int32)
    _, cs, nc = oracle.ccs_masks(cases, n)
    _, ct, nt = oracle.ccs_masks(ctrls, n)
    n_sel = nc + nt
    p, c_lo, c_hi, k0, level = _quad_constants(nc, n_sel)
    # case groups of 65 536 and more: the kernel that keeps both groups' accumulators in registers
    # (the diagnostic form writes a value per pair: the DENSE instantiations of gw_boost_fp4_full.hip)
    dn = lambda dense: "true" if dense else "false"
    kname = ((lambda lv, dense: f"gw::k_boost_mfma_fp4_both<true, {lv}, {dn(dense)}>") if nc >= 65536 else
             (lambda lv, dense: f"gw::k_boost_mfma_fp4<true, 2, 2, {lv}, {dn(dense)}>"))
    c = max(c_lo, c_hi)
    z = np.linspace(0.0, 1.0, 200_001)
    ent = lambda v: -(np.where(v > 0, v * np.log2(np.maximum(v, 1e-300)), 0.0) + np.where(v < 1, (1 - v) * np.log2(np.maximum(1 - v, 1e-300)), 0.0))
    assert np.all(ent(np.array([p]))[0] + np.log2((1 - p) / p) * (z - p) - np.where(z < p, c_lo, c_hi) * (z - p) ** 2 <= ent(z) + 1e-12)
    # the cells of every pair, from the codes: X[a][b] (cases), S[a][b] (both groups), m x m each
    ind = lambda grp, k: (codes[:, grp] == k).astype(np.float32)
    X = np.array([[ind(is_case, a) @ ind(is_case, b).T for b in (1, 2, 3)] for a in (1, 2, 3)], dtype=np.float64)
    Y = np.array([[ind(is_ctrl, a) @ ind(is_ctrl, b).T for b in (1, 2, 3)] for a in (1, 2, 3)], dtype=np.float64)
    S = X + Y
    xlgx = lambda v: np.where(v > 0, v * np.log2(np.maximum(v, 1.0)), 0.0)
    A = (xlgx(X) + xlgx(Y) - xlgx(S)).sum(axis=(0, 1))
    D = X - p * S
    A1 = k0 + (np.where(D < 0, c_lo, c_hi) * np.where(S > 0, D ** 2 / np.maximum(S, 1.0), 0.0)).sum(axis=(0, 1))
    assert np.all(A <= A1 + 1e-6)                       # the inequality itself, in FP64
    # round 4: n lg tao enters as n lg e (tao - 1) -- tao (epistasis_func.cpp:424-470) from the same cells, in FP64
    Ra_c, Ra = X.sum(axis=1), S.sum(axis=1)             # 3, m, m: SNP i's class counts (cases / everyone), same for every j
    Cb_c, Cb = X.sum(axis=0), S.sum(axis=0)
    tao = np.zeros((m, m))
    for Xg, Ra_g, Cb_g, n_g in ((X, Ra_c, Cb_c, float(nc)), (Y, Ra - Ra_c, Cb - Cb_c, float(nt))):
        for ka in range(3):
            for kb in range(3):
                with np.errstate(divide="ignore", invalid="ignore"):
                    tao += np.nan_to_num(S[ka, kb] * Ra_g[ka] * Cb_g[kb] / (Ra[ka] * Cb[kb] * n_g))
    lin_slack = 2.0 * n_sel * (tao - 1.0 - np.log(tao))   # >= 0: what the linearised logarithm adds to the bound
    assert np.all(lin_slack[np.triu_indices(m, 1)] >= -1e-9)
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, ts)
        t.set_case_control(cs, ct, nc, nt)
        exact = t.boost_measures(0, m, 0, m)
        t.set_option(gw.capi.OPT_BOOST_PREFILTER, 4)
        b1 = t.boost_measures(0, m, 0, m)
        assert t.last_boost_kernel() == kname(level, True)
        t.set_option(gw.capi.OPT_BOOST_PREFILTER, 5)
        e1 = t.boost_measures(0, m, 0, m)
        t.set_option(gw.capi.OPT_BOOST_PREFILTER, 1)
        iu = np.triu_indices(m, 1)
        ex, bb, ee = exact[iu], b1[iu], e1[iu]
        want = ex + 2.0 * np.log(2.0) * (A1 - A)[iu] + lin_slack[iu]    # the same formula in FP64
        ok = np.isfinite(ex) & np.isfinite(bb) & np.isfinite(ee)
        assert ok.sum() >= 0.9 * np.isfinite(ex).sum()
        dev = np.abs(bb[ok] - want[ok])
        worst = int(np.argmax(dev / ee[ok]))
        print(f"worst pair: device {bb[ok][worst]:.4f} formula {want[ok][worst]:.4f} err1 {ee[ok][worst]:.4f} measure {ex[ok][worst]:.4f}")
        assert np.all(ex[ok] <= bb[ok] + ee[ok])
        assert np.all(dev <= ee[ok]), (dev.max(), (dev / ee[ok]).max())
        reach = np.mean(bb[ok] + ee[ok] >= 30.0)
        print(f"n={n_sel} p={p:.3f} c={c_lo:.3f}/{c_hi:.3f}: max |bound1 - formula| / err1 = {(dev / ee[ok]).max():.3f}, median err1 = {np.median(ee[ok]):.4f}, "
              f"median slack = {np.median(bb[ok] - ex[ok]):.2f}, first level reaches 30 for {reach:.3f} of the pairs")
        finite = np.sort(ex[np.isfinite(ex)])
        thresholds = [30.0, 25.0, float(np.median(finite)), float(finite[-1]), float(finite[-2]), -1.0e9, 1.0e9]
        thresholds += [float(v) for v in rng.choice(finite, size=3)]
        high = finite[finite >= 25.0]                      # where the first level is in use: exactly on / just below a measure
        thresholds += [float(v) for v in rng.choice(high, size=min(4, len(high)), replace=False)]
        thresholds += [float(np.nextafter(v, -np.inf)) for v in rng.choice(high, size=min(2, len(high)), replace=False)]
        assert len(high) >= 1
        for thr in thresholds:
            ref_hits, ref_n, ref_pairs, _, _ = t.boost_screen(0, m, 0, m, threshold=thr)           # all FP64
            for first in (1, 0):
                t.set_option(gw.capi.OPT_BOOST_FIRST_LEVEL, first)
                hits, n_hits, n_pairs, *_ = t.boost_screen(0, m, 0, m, threshold=thr, extremes=False)
                # (the first level is only used where it is selective: thresholds of 25 and more)
                assert t.last_boost_kernel() == kname(level if first and thr >= 25.0 else 0, False)
                assert (n_hits, n_pairs) == (ref_n, ref_pairs) == (int((finite > thr).sum()), m * (m - 1) // 2)
                assert np.array_equal(np.sort(hits, order=["snp_a", "snp_b"]), np.sort(ref_hits, order=["snp_a", "snp_b"]))
        t.set_option(gw.capi.OPT_BOOST_FIRST_LEVEL, 1)
        assert t.get_option(gw.capi.OPT_BOOST_PREFILTER_TRIPS) == 0     # neither level's run-time check fired


@pytest.mark.parametrize("p_missing", [0.0, 0.02])
def test_boost_blocks_of_any_geometry_agree_across_kernels(oracle, p_missing):
    """Rectangles of the pair space that start and end anywhere (not on the 32/64/128-SNP tile grids of
    the kernels), on a table several tiles wide: every kernel family (vector ALU, one-group-at-a-time
    MFMA, both-groups MFMA) must return the measures of the full matrix, and the screen its hits."""
    n, m = 2500, 203
    rng = np.random.default_rng(11 + int(p_missing * 100))
    ts = oracle.pack_codes(_codes(rng, m, n, p_missing), "stream")
    cs, ct, nc, nt = _masks(oracle, rng, n)
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, ts)
        t.set_case_control(cs, ct, nc, nt)
        t.set_option(gw.capi.OPT_BOOST_MATRIX_CORES, 0)
        full = t.boost_measures(0, m, 0, m)  # vector-ALU kernel (checked against the oracle above)
        finite = full[np.isfinite(full)]
        thr = float(np.quantile(finite, 0.97))
        rects = [(0, m, 0, m), (0, 1, 0, m), (m - 1, 1, 0, m), (5, 70, 3, 190), (63, 66, 64, 65), (64, 64, 128, 75),
                 (127, 3, 0, m), (1, 130, 131, 72), (190, 13, 100, 103)]
        for _ in range(6):
            a0 = int(rng.integers(0, m)); na = int(rng.integers(1, m - a0 + 1))
            b0 = int(rng.integers(0, m)); nb = int(rng.integers(1, m - b0 + 1))
            rects.append((a0, na, b0, nb))
        for mode in (1, 2, 0):
            t.set_option(gw.capi.OPT_BOOST_MATRIX_CORES, mode)
            for (a0, na, b0, nb) in rects:
                blk = t.boost_measures(a0, na, b0, nb)
                want = full[a0:a0 + na, b0:b0 + nb].copy()
                want[np.arange(a0, a0 + na)[:, None] >= np.arange(b0, b0 + nb)[None, :]] = np.nan
                assert np.array_equal(np.isnan(blk), np.isnan(want)), (mode, a0, na, b0, nb)
                ok = ~np.isnan(want)
                assert np.array_equal(blk[ok], want[ok]), (mode, a0, na, b0, nb)
                hits, n_hits, n_pairs, _, _ = t.boost_screen(a0, na, b0, nb, threshold=thr, extremes=False)
                assert n_pairs == int(ok.sum())
                assert n_hits == int((want[ok] > thr).sum()) == len(hits)
                for h in hits:
                    assert h["measure"] == full[h["snp_a"], h["snp_b"]]


def test_new_masks_on_the_same_table_reuse_the_selection_buffers(oracle):
    """selectCaseControl again with other masks (another phenotype of the same cohort): the compacted planes are
    re-used while the padded group sizes stay the same and re-made when they change; measures follow the masks."""
    n, m = 3000, 40
    rng = np.random.default_rng(17)
    ts = oracle.pack_codes(_codes(rng, m, n, 0.01), "stream")
    gs = oracle.geometry_stream(n)
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, ts)
        for neither in (0.1, 0.12, 0.7, 0.1):      # same padded geometry twice, then much smaller groups, then back
            cs, ct, nc, nt = _masks(oracle, rng, n, neither=neither)
            t.set_case_control(cs, ct, nc, nt)
            got = t.boost_measures(0, m, 0, m)
            want = _pair_matrix(oracle.boost_all_pairs(ts, gs, cs, ct, nc, nt), m)
            assert np.array_equal(np.isnan(got), np.isnan(want))
            ok = ~np.isnan(want)
            np.testing.assert_allclose(got[ok], want[ok], rtol=1e-10, atol=1e-10)
            rec = t.scan_margins()
            for r in (0, m - 1):
                ca, co = oracle.stream_cc_dist(ts[r], gs, cs, ct, nc, nt)
                assert tuple(int(v) for v in rec[r]["cases"]) == tuple(ca) and tuple(int(v) for v in rec[r]["controls"]) == tuple(co)


@pytest.mark.parametrize("n,m", [(600, 48), (3000, 500), (9000, 1400)])
def test_block_calls_on_fp4_equal_the_int8_kernels(n, m):
    """Block and whole-triangle calls of the pre-filtered screen: the FP4 kernels (candidates out, k_boost_finish behind
    them) against the int8 one-group-at-a-time kernel (everything in one kernel) -- same hits, same FP64 measures, same
    pair tally, with the first level (thresholds from 25), without it, and with a threshold every pair passes (every
    pair becomes a candidate); the pre-filter's error model holds.  The retired options are refused."""
    with gw.GwTable(n, m) as t:
        t.simulate(11, 3, 0)
        t.simulate_phenotype(11)
        calls = [(0, m, 0, m), (0, m // 3, 0, m), (m // 3, m - m // 3, m // 3, m - m // 3),
                 (5, min(130, m - 5), min(70, m // 2), m - min(70, m // 2))]
        for thr in (30.0, 6.0, -1.0e9):
            for (a, na, b, nb) in calls:
                t.set_option(gw.capi.OPT_BOOST_FP4, 0)
                want, wh, wp, _, _ = t.boost_screen(a, na, b, nb, threshold=thr, extremes=False)
                assert t.last_boost_kernel().startswith("gw::k_boost_mfma_seq<true, 2, 2")
                t.set_option(gw.capi.OPT_BOOST_FP4, 1)
                got4, gh4, gp4, _, _ = t.boost_screen(a, na, b, nb, threshold=thr, extremes=False)
                # (first level from a threshold of 25: one curvature constant or the two-piece form, by the cohort's balance)
                assert t.last_boost_kernel() in ([f"gw::k_boost_mfma_fp4<true, 2, 2, {v}, false>" for v in (1, 2)] if thr >= 25.0 else
                                                 ["gw::k_boost_mfma_fp4<true, 2, 2, 0, false>"])
                assert (gh4, gp4) == (wh, wp)
                assert np.array_equal(np.sort(got4, order=["snp_a", "snp_b"]), np.sort(want, order=["snp_a", "snp_b"]))
        assert t.get_option(gw.capi.OPT_BOOST_PREFILTER_TRIPS) == 0
        for opt in (gw.capi.OPT_BOOST_LUT, gw.capi.OPT_BOOST_STAGGER_US):     # experiments of round 3, no longer in the library
            with pytest.raises(gw.GwError):
                t.set_option(opt, 1)
            t.set_option(opt, 0)
            assert t.get_option(opt) == 0
    with gw.GwTable(2000, 100) as t:    # a table with missing calls: nine cells
        t.simulate(11, 3, 655)
        t.simulate_phenotype(11)
        t.boost_screen(0, 100, 0, 100, extremes=False)      # threshold 30: the nine-cell first level
        assert t.last_boost_kernel() in ("gw::k_boost_mfma_fp4<true, 3, 1, 1, false>", "gw::k_boost_mfma_fp4<true, 3, 1, 2, false>")
        t.boost_screen(0, 100, 0, 100, threshold=8.0, extremes=False)
        assert t.last_boost_kernel() == "gw::k_boost_mfma_fp4<true, 3, 1, 0, false>"


@pytest.mark.parametrize("missing", [0, 655])
def test_screen_in_row_blocks_equals_one_launch(missing):
    """GW_OPT_BOOST_ROWS_PER_LAUNCH: the pre-filtered screen of a block pair row block by row block through one candidate
    list (what a call over all pairs of ~130 000 SNPs and more does by itself, because the list's cap would leave it less
    than a record per 512 pairs) -- same hits, same tallies, host form and device form, whole triangle and an off-diagonal block."""
    n, m = 2500, 700
    with gw.GwTable(n, m) as t:
        t.simulate(23, 0, missing)
        t.simulate_phenotype(23)
        thr = 12.0 if missing == 0 else -1.0e9      # (with missing calls the measures of unlinked pairs are negative: every pair)
        for (a, na, b, nb) in ((0, m, 0, m), (37, 500, 300, 400)):
            t.set_option(gw.capi.OPT_BOOST_ROWS_PER_LAUNCH, 0)
            want, wh, wp, _, _ = t.boost_screen(a, na, b, nb, threshold=thr, max_hits=1 << 18, extremes=False)
            assert wh > 20
            for rows in (1, 128, 300):          # rounded up to 128, 128, 384
                t.set_option(gw.capi.OPT_BOOST_ROWS_PER_LAUNCH, rows)
                got, gh, gp, _, _ = t.boost_screen(a, na, b, nb, threshold=thr, max_hits=1 << 18, extremes=False)
                assert (gh, gp) == (wh, wp)
                assert np.array_equal(np.sort(got, order=["snp_a", "snp_b"]), np.sort(want, order=["snp_a", "snp_b"]))
        # the device form, list in HBM
        cap = 1 << 18
        lst = gw.capi.DeviceBuffer(gw.capi.BOOST_LIST_HEADER_BYTES + 16 * cap)
        lists = []
        st = t.stream()
        for rows in (0, 128):
            t.set_option(gw.capi.OPT_BOOST_ROWS_PER_LAUNCH, rows)
            t.boost_list_init(lst.ptr, st)
            t.boost_screen_device(0, m, 0, m, lst.ptr, cap, thr, st)
            hits, n_hits, n_pairs, aux0 = gw.parse_boost_list(lst.download(st), cap)
            lists.append((np.sort(hits, order=["snp_a", "snp_b"]), n_hits, n_pairs, aux0))
        assert lists[0][1:] == lists[1][1:] and np.array_equal(lists[0][0], lists[1][0]) and lists[0][3] == 0
        t.set_option(gw.capi.OPT_BOOST_ROWS_PER_LAUNCH, 0)
        assert t.get_option(gw.capi.OPT_BOOST_PREFILTER_TRIPS) == 0
        with pytest.raises(gw.GwError):
            t.set_option(gw.capi.OPT_BOOST_ROWS_PER_LAUNCH, -1)


def test_candidate_list_overflow_repeats_the_block_in_fp64():
    """GW_OPT_BOOST_CANDIDATES: a candidate list too small for what the in-kernel bound lets through is reported through
    the tally (like a violated error model) and the block is repeated with every pair in FP64 -- same hits, one trip."""
    n, m = 3000, 400
    with gw.GwTable(n, m) as t:
        t.simulate(17, 0, 0)
        t.simulate_phenotype(17)
        want, wh, wp, _, _ = t.boost_screen(0, m, 0, m, threshold=13.0, extremes=False)
        assert 100 < wh < 4096 and t.get_option(gw.capi.OPT_BOOST_PREFILTER_TRIPS) == 0
        t.set_option(gw.capi.OPT_BOOST_CANDIDATES, 64)
        got, gh, gp, _, _ = t.boost_screen(0, m, 0, m, threshold=13.0, extremes=False)
        assert t.get_option(gw.capi.OPT_BOOST_PREFILTER_TRIPS) == 1
        assert t.last_boost_kernel() == "gw::k_boost_mfma_fp4<false, 2, 2, 0, false>"
        assert (gh, gp) == (wh, wp)
        assert np.array_equal(np.sort(got, order=["snp_a", "snp_b"]), np.sort(want, order=["snp_a", "snp_b"]))
        # the device-list form reports it in the header (slot 2), where a sharded job's ranks all see it
        lst = gw.capi.DeviceBuffer(gw.capi.BOOST_LIST_HEADER_BYTES + 16 * 4096)
        s = t.stream()
        t.boost_list_init(lst.ptr, s)
        t.boost_screen_device(0, m, 0, m, lst.ptr, 4096, 13.0, s)
        _, n_hits, n_pairs, aux0 = gw.parse_boost_list(lst.download(s), 4096)
        assert aux0 > 0 and n_pairs == wp
        t.set_option(gw.capi.OPT_BOOST_CANDIDATES, 0)
        t.boost_list_init(lst.ptr, s)
        t.boost_screen_device(0, m, 0, m, lst.ptr, 4096, 13.0, s)
        hits, n_hits, n_pairs, aux0 = gw.parse_boost_list(lst.download(s), 4096)
        assert aux0 == 0 and (n_hits, n_pairs) == (wh, wp)
        assert np.array_equal(np.sort(hits[:n_hits], order=["snp_a", "snp_b"]), np.sort(want, order=["snp_a", "snp_b"]))


@pytest.mark.parametrize("n,m,force", [(3000, 200, True), (9000, 131, True), (140_000, 90, False), (200_000, 70, False)])
def test_fp4_both_groups_kernel_equals_the_int8_and_vector_kernels(n, m, force):
    """k_boost_mfma_fp4_both: case groups of 65 536 and more (or GW_OPT_BOOST_MATRIX_CORES = 2) on tables without a missing
    call keep both groups' FP32 accumulators in registers instead of parking uint16 case counts in LDS.  Every pair's FP64
    measure, the extremes and the hit lists with and without the pre-filter must be those of the int8 both-groups kernel
    (k_boost_mfma) and of the vector-ALU kernel, bit for bit."""
    with gw.GwTable(n, m) as t:
        t.simulate(29, 5, 0, exact=True)
        t.simulate_phenotype(29)
        res = {}
        for name, (fp4, mfma) in {"fp4": (1, 2 if force else 1), "int8": (0, 2 if force else 1), "valu": (0, 0)}.items():
            t.set_option(gw.capi.OPT_BOOST_FP4, fp4)
            t.set_option(gw.capi.OPT_BOOST_MATRIX_CORES, mfma)
            dense = t.boost_measures(0, m, 0, m)
            k_dense = t.last_boost_kernel()
            hits, nh, npairs, vmax, vmin = t.boost_screen(0, m, 0, m, threshold=8.0)
            pre_hits, pre_nh, pre_pairs, *_ = t.boost_screen(0, m, 0, m, threshold=8.0, extremes=False)
            k_pre = t.last_boost_kernel()
            hi_hits, hi_nh, *_ = t.boost_screen(3, m - 3, 1, m - 1, threshold=26.0, extremes=False)   # first level in use
            res[name] = (dense, np.sort(hits, order=["snp_a", "snp_b"]), nh, npairs, vmax, vmin,
                         np.sort(pre_hits, order=["snp_a", "snp_b"]), pre_nh, pre_pairs, np.sort(hi_hits, order=["snp_a", "snp_b"]), hi_nh,
                         k_dense, k_pre, t.last_boost_kernel())
        assert res["fp4"][11] == "gw::k_boost_mfma_fp4_both<false, 0, true>" and res["fp4"][12] == "gw::k_boost_mfma_fp4_both<true, 0, false>"
        assert res["fp4"][13] in ("gw::k_boost_mfma_fp4_both<true, 1, false>", "gw::k_boost_mfma_fp4_both<true, 2, false>")
        assert res["int8"][11] == "gw::k_boost_mfma<false>" and res["int8"][12] == "gw::k_boost_mfma<true>"
        for other in ("int8", "valu"):
            assert np.array_equal(res["fp4"][0], res[other][0], equal_nan=True)
            for k in (1, 6, 9):
                assert np.array_equal(res["fp4"][k], res[other][k])
            assert res["fp4"][2:6] == res[other][2:6] and res["fp4"][7:9] == res[other][7:9] and res["fp4"][10] == res[other][10]
        assert np.array_equal(res["fp4"][1], res["fp4"][6])          # pre-filter on / off: the same list
        assert t.get_option(gw.capi.OPT_BOOST_PREFILTER_TRIPS) == 0


@pytest.mark.parametrize("missing", [0, 655])
@pytest.mark.parametrize("n,m", [(300, 70), (2000, 300), (20_000, 257), (70_000, 130)])
def test_fp4_counting_equals_int8_counting(n, m, missing):
    """k_boost_mfma_fp4 (v_mfma_scale_f32_32x32x64_f8f6f4 on one-sample-per-nibble operands, E8M0 scales 2 / 1 / 0.5) against
    the int8 kernel and the vector-ALU kernel: every pair's FP64 measure (the dense form runs every pair through the
    counted tables), extremes, hit lists -- bit-identical, for cohorts on both sides of every padding boundary."""
    with gw.GwTable(n, m) as t:
        t.simulate(23, 9, missing, exact=True)        # with missing calls: the nine-cell kernels
        t.simulate_phenotype(23)
        res = {}
        for name, (fp4, mfma) in {"fp4": (1, 1), "int8": (0, 1), "valu": (0, 0)}.items():
            t.set_option(gw.capi.OPT_BOOST_FP4, fp4)
            t.set_option(gw.capi.OPT_BOOST_MATRIX_CORES, mfma)
            dense = t.boost_measures(0, m, 0, m)
            hits, nh, npairs, vmax, vmin = t.boost_screen(0, m, 0, m, threshold=8.0)
            res[name] = (dense, np.sort(hits, order=["snp_a", "snp_b"]), nh, npairs, vmax, vmin, t.last_boost_kernel())
        shape = "3, 1" if missing else "2, 2"
        assert res["fp4"][6] == f"gw::k_boost_mfma_fp4<false, {shape}, 0, false>" and res["int8"][6].startswith(f"gw::k_boost_mfma_seq<false, {shape}")
        for other in ("int8", "valu"):
            assert np.array_equal(res["fp4"][0], res[other][0], equal_nan=True)
            assert np.array_equal(res["fp4"][1], res[other][1]) and res["fp4"][2:6] == res[other][2:6]


def test_gtest_on_the_device_equals_the_oracle(oracle):
    """gw_boost_gtest: BOOST stage 2 (computeGTest, epistasis_func.cpp:508-704) on the device against the oracle's
    restatement, pair by pair: >= 1000 pairs of a table with missing calls, planted interactions, rare variants (tables
    with empty cells, rows and columns) -- the IPF-fitted measure to 1e-9 relative (the device's log differs from libm's
    in the last place), z likewise (NaN / inf where the reference divides by an empty allele class, the same on both
    sides); every fit settles.  Then the device-list form behind a screen: {measure, z} of every hit in HBM."""
    n, m = 3000, 96
    rng = np.random.default_rng(42)
    codes = _prefilter_table(rng, n, m, 0.004, True)
    status = rng.random(n) < 0.45
    neither = rng.random(n) < 0.05
    for (i, j, frac) in ((1, 5, 0.6), (2, 9, 0.3), (3, 11, 0.1), (40, 77, 0.5)):
        flip = status & (rng.random(n) < frac)
        codes[j, flip] = codes[i, flip]
    codes[20, :] = np.where(rng.random(n) < 0.002, 3, 1)          # nearly monomorphic rows: empty cells all over the table
    codes[21, :] = np.where(rng.random(n) < 0.001, 2, 1)
    codes[22, :] = 1                                               # monomorphic: two empty rows / columns
    ts = oracle.pack_codes(codes, "stream")
    gs = oracle.geometry_stream(n)
    cases = np.flatnonzero(status & ~neither).astype(np.int32)
    ctrls = np.flatnonzero(~status & ~neither).astype(np.int32)
    _, cs, nc = oracle.ccs_masks(cases, n)
    _, ct, nt = oracle.ccs_masks(ctrls, n)
    n_sel = nc + nt
    mar = []
    for r in range(m):
        ca, co = oracle.stream_cc_dist(ts[r], gs, cs, ct, nc, nt)
        mar.append(oracle.marginal_information(ca, co, n_sel))
    pairs = [(i, j) for i in range(m) for j in range(i + 1, m)]
    sel = rng.choice(len(pairs), size=1400, replace=False)
    must = [pairs.index(p) for p in ((1, 5), (2, 9), (3, 11), (40, 77), (20, 21), (20, 22), (21, 22), (5, 22), (0, 20))]
    ij = np.array([pairs[k] for k in sorted(set(sel.tolist()) | set(must))], dtype=np.uint32)
    assert ij.shape[0] >= 1000
    pad = gs.words * 64
    want = np.zeros((ij.shape[0], 2))
    for k, (i, j) in enumerate(ij):
        c16, t16 = oracle.stream_cc_contingency(ts[i], ts[j], gs, cs, ct)
        c16[15] -= pad - nc
        t16[15] -= pad - nt
        want[k] = oracle.gtest(c16, t16, mar[i], mar[j], n_sel)
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, ts)
        t.set_case_control(cs, ct, nc, nt)
        measure, z, unsettled = t.boost_gtest(ij)
        assert unsettled == 0
        for got, w, what in ((measure, want[:, 0], "measure"), (z, want[:, 1], "z")):
            both_nan = np.isnan(got) & np.isnan(w)
            same_inf = np.isinf(got) & np.isinf(w) & (np.sign(got) == np.sign(w))
            fin = np.isfinite(got) & np.isfinite(w)
            assert np.all(both_nan | same_inf | fin), what
            assert fin.sum() >= 900, what
            assert np.allclose(got[fin], w[fin], rtol=1e-9, atol=1e-9), (what, np.max(np.abs(got[fin] - w[fin])))
        print(f"gtest: {ij.shape[0]} pairs, {int(np.isfinite(z).sum())} finite z, max |measure - oracle| "
              f"{np.nanmax(np.abs(measure - want[:, 0])):.2e}")
        # stage 1 + stage 2 in HBM: the screen's list, then {measure, z} per hit, nothing on the host in between
        cap = 8192
        lst = gw.capi.DeviceBuffer(gw.capi.BOOST_LIST_HEADER_BYTES + 16 * cap)
        out = gw.capi.DeviceBuffer(16 * cap)
        status_buf = gw.capi.DeviceBuffer(8)
        s = t.stream()
        t.boost_list_init(lst.ptr, s)
        status_buf.memset(0, s)
        t.boost_screen_device(0, m, 0, m, lst.ptr, cap, -1.0e9, s)  # every pair with a measure (missing calls drag it below zero: mean -44)
        t.boost_gtest_device(lst.ptr, cap, out.ptr, status_buf.ptr, s)
        hits, n_hits, _, aux0 = gw.parse_boost_list(lst.download(s), cap)
        assert aux0 == 0 and 1000 <= n_hits <= cap
        mz = out.download(s).view(np.float64).reshape(cap, 2)[:n_hits]
        assert int(status_buf.download(s).view(np.uint64)[0]) == 0
        m2, z2, _ = t.boost_gtest(hits)                      # the host form on the same records
        assert np.array_equal(mz[:, 0], m2, equal_nan=True) and np.array_equal(mz[:, 1], z2, equal_nan=True)
        assert {(1, 5), (40, 77)} <= {(int(h["snp_a"]), int(h["snp_b"])) for h in hits}


def test_list_header_follows_the_form_the_screen_takes():
    """gw_boost_list_init and gw_boost_screen_device take the same decision about the pre-filter (one predicate in
    gw_capi.cpp): with it the header's slots 2 / 3 are counters that stay 0, without it (GW_OPT_BOOST_PREFILTER 0: every
    pair in FP64) they are the ordered maximum / minimum of the measures -- the extremes gw_boost_screen reports."""
    n, m = 4000, 300
    cap = 1 << 14

    def ordered_to_double(u):
        u = int(u)
        u = (u & 0x7FFFFFFFFFFFFFFF) if (u & 0x8000000000000000) else (~u & 0xFFFFFFFFFFFFFFFF)
        return np.array([u], dtype=np.uint64).view(np.float64)[0]

    with gw.GwTable(n, m) as t:
        t.simulate(23, 0, 0, exact=True)
        t.simulate_phenotype(23)
        hits, nh, npairs, vmax, vmin = t.boost_screen(0, m, 0, m, threshold=12.0)
        lst = gw.capi.DeviceBuffer(gw.capi.BOOST_LIST_HEADER_BYTES + 16 * cap)
        s = t.stream()
        for pre in (1, 0):
            t.set_option(gw.capi.OPT_BOOST_PREFILTER, pre)
            t.boost_list_init(lst.ptr, s)
            head0 = lst.download(s)[:32].view(np.uint64).copy()
            t.boost_screen_device(0, m, 0, m, lst.ptr, cap, 12.0, s)
            raw = lst.download(s)
            got, n_hits, n_pairs, aux0 = gw.parse_boost_list(raw, cap)
            head = raw[:32].view(np.uint64)
            assert (n_hits, n_pairs) == (nh, npairs)
            assert np.array_equal(np.sort(got, order=["snp_a", "snp_b"]), np.sort(hits, order=["snp_a", "snp_b"]))
            if pre:
                assert t.last_boost_kernel().startswith("gw::k_boost_mfma_fp4<true")
                assert list(head0) == [0, 0, 0, 0] and int(head[2]) == 0 and int(head[3]) == 0
            else:
                assert t.last_boost_kernel().startswith("gw::k_boost_mfma_fp4<false")
                assert list(head0) == [0, 0, 0, 0xFFFFFFFFFFFFFFFF]
                assert ordered_to_double(head[2]) == vmax and ordered_to_double(head[3]) == vmin
        t.set_option(gw.capi.OPT_BOOST_PREFILTER, 1)


def test_gtest_and_candidates_edge_cases():
    """gw_boost_gtest: no pairs, a pair outside the table, no masks; the candidate list's overflow repeat on the nine-cell
    kernel (tables with missing calls) and on the both-groups kernel, same hits as with room."""
    with gw.GwTable(1000, 40) as t:
        t.simulate(3, 0, 0)
        with pytest.raises(gw.GwError):                      # masks first (gw_table_set_case_control)
            t.boost_gtest(np.array([[0, 1]], dtype=np.uint32))
        t.simulate_phenotype(3)
        m0, z0, u0 = t.boost_gtest(np.zeros((0, 2), dtype=np.uint32))
        assert len(m0) == 0 and len(z0) == 0 and u0 == 0
        with pytest.raises(gw.GwError):
            t.boost_gtest(np.array([[0, 40]], dtype=np.uint32))
        m1, z1, _ = t.boost_gtest(np.array([[0, 1], [1, 0], [5, 5]], dtype=np.uint32))   # any order, also i == j: just tables
        assert np.isfinite(m1[0]) and m1[0] == m1[0]
    for missing, mode, prefix in ((655, 1, "gw::k_boost_mfma_fp4<true, 3, 1, 0, false>"), (0, 2, "gw::k_boost_mfma_fp4_both<true, 0, false>")):
        with gw.GwTable(3000, 300) as t:
            t.simulate(31, 0, missing)
            t.simulate_phenotype(31)
            t.set_option(gw.capi.OPT_BOOST_MATRIX_CORES, mode)
            exact = t.boost_measures(0, 300, 0, 300)
            fin = np.sort(exact[np.isfinite(exact)])
            thr = float(fin[-400])                           # ~400 pairs pass
            assert thr < 25.0                                 # (below 25 the screen runs without the first level)
            want, wh, wp, _, _ = t.boost_screen(0, 300, 0, 300, threshold=thr, extremes=False)
            assert t.last_boost_kernel() == prefix and t.get_option(gw.capi.OPT_BOOST_PREFILTER_TRIPS) == 0
            t.set_option(gw.capi.OPT_BOOST_CANDIDATES, 50)
            got, gh, gp, _, _ = t.boost_screen(0, 300, 0, 300, threshold=thr, extremes=False)
            assert t.get_option(gw.capi.OPT_BOOST_PREFILTER_TRIPS) == 1
            assert (gh, gp) == (wh, wp) and wh >= 399
            assert np.array_equal(np.sort(got, order=["snp_a", "snp_b"]), np.sort(want, order=["snp_a", "snp_b"]))


@pytest.mark.parametrize("n,m,case_share,p_missing,neither", [(800, 40, 0.5, 0.01, 0.0), (6000, 56, 0.45, 0.01, 0.05),
                                                              (6000, 40, 0.2, 0.03, 0.0), (40_000, 36, 0.6, 0.005, 0.1)])
def test_first_level_bound_for_tables_with_missing_calls(oracle, n, m, case_share, p_missing, neither):
    """The nine-cell form of the first level (gw_boost_dev.h boost_bound_quad9_f, k_boost_mfma_fp4<true, 3, 1, 1 or 2>): for every
    pair measure <= bound1 + err1; the device's value lies within err1 of the same formula in FP64 from the pair's counted cells
    (the quadratic under the entropy, the twelve row / column-sum products, n lg e (tao - 1)); and the hit lists with the first
    level on, off and without any pre-filter are the same for thresholds on and just below pair measures."""
    rng = np.random.default_rng(17 * n + m)
    codes = _prefilter_table(rng, n, m, p_missing, False)
    u = rng.random(n)
    is_case = u < case_share * (1.0 - neither)
    is_ctrl = (u >= case_share * (1.0 - neither)) & (u < 1.0 - neither)
    for (i, j, frac) in ((1, 5, 0.95), (2, 9, 0.8), (3, 11, 0.6), (4, 13, 0.4)):     # strong ones: missing calls drag the measure down
        flip = is_case & (rng.random(n) < frac)
        codes[j, flip] = codes[i, flip]
    ts = oracle.pack_codes(codes, "stream")
    _, cs, nc = oracle.ccs_masks(np.flatnonzero(is_case).astype(np.int32), n)
    _, ct, nt = oracle.ccs_masks(np.flatnonzero(is_ctrl).astype(np.int32), n)
    n_sel = nc + nt
    p, c_lo, c_hi, _, level = _quad_constants(nc, n_sel)
    h0 = -(p * np.log2(p) + (1 - p) * np.log2(1 - p))
    h1 = np.log2((1 - p) / p)
    ind = lambda grp, k: (codes[:, grp] == k).astype(np.float32)
    X = np.array([[ind(is_case, a) @ ind(is_case, b).T for b in (1, 2, 3)] for a in (1, 2, 3)], dtype=np.float64)
    Y = np.array([[ind(is_ctrl, a) @ ind(is_ctrl, b).T for b in (1, 2, 3)] for a in (1, 2, 3)], dtype=np.float64)
    S = X + Y
    xlgx = lambda v: np.where(v > 0, v * np.log2(np.maximum(v, 1.0)), 0.0)
    A = (xlgx(X) + xlgx(Y) - xlgx(S)).sum(axis=(0, 1))
    D = X - p * S
    quad = (np.where(D < 0, c_lo, c_hi) * np.where(S > 0, D ** 2 / np.maximum(S, 1.0), 0.0)).sum(axis=(0, 1))
    A1 = quad - h0 * S.sum(axis=(0, 1)) - h1 * (X.sum(axis=(0, 1)) - p * S.sum(axis=(0, 1)))
    assert np.all(A <= A1 + 1e-6)                       # the inequality itself, in FP64
    # the SNPs' own counts (over every selected individual, also those missing at the other SNP) and tao
    cnt = lambda grp: np.array([(codes[:, grp] == k).sum(axis=1) for k in (1, 2, 3)], dtype=np.float64)   # 3, m
    xc, yc = cnt(is_case), cnt(is_ctrl)
    mg = xc + yc
    tao = np.zeros((m, m))
    for G, n_g in ((xc, float(nc)), (yc, float(nt))):
        for ka in range(3):
            for kb in range(3):
                with np.errstate(divide="ignore", invalid="ignore"):
                    tao += np.nan_to_num(S[ka, kb] * G[ka][:, None] * G[kb][None, :] / (mg[ka][:, None] * mg[kb][None, :] * n_g))
    lin_slack = 2.0 * n_sel * (tao - 1.0 - np.log(tao))
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, ts)
        t.set_case_control(cs, ct, nc, nt)
        exact = t.boost_measures(0, m, 0, m)
        t.set_option(gw.capi.OPT_BOOST_PREFILTER, 4)
        b1 = t.boost_measures(0, m, 0, m)
        assert t.last_boost_kernel() == f"gw::k_boost_mfma_fp4<true, 3, 1, {level}, true>"
        t.set_option(gw.capi.OPT_BOOST_PREFILTER, 5)
        e1 = t.boost_measures(0, m, 0, m)
        t.set_option(gw.capi.OPT_BOOST_PREFILTER, 1)
        iu = np.triu_indices(m, 1)
        ex, bb, ee = exact[iu], b1[iu], e1[iu]
        want = ex + 2.0 * np.log(2.0) * (A1 - A)[iu] + lin_slack[iu]      # the same formula in FP64
        ok = np.isfinite(ex) & np.isfinite(bb) & np.isfinite(ee) & np.isfinite(want)
        assert ok.sum() >= 0.9 * np.isfinite(ex).sum()
        dev = np.abs(bb[ok] - want[ok])
        assert np.all(ex[ok] <= bb[ok] + ee[ok])
        assert np.all(dev <= ee[ok]), (dev.max(), (dev / ee[ok]).max())
        print(f"n={n_sel} p={p:.3f} missing {p_missing}: max |bound1 - formula| / err1 = {(dev / ee[ok]).max():.3f}, median err1 = {np.median(ee[ok]):.4f}, "
              f"median slack = {np.median(bb[ok] - ex[ok]):.2f}, first level reaches 30 for {np.mean(bb[ok] + ee[ok] >= 30.0):.4f} of the pairs")
        finite = np.sort(ex[np.isfinite(ex)])
        high = finite[finite >= 25.0]
        assert len(high) >= 1
        thresholds = [30.0, 25.0, float(finite[-1]), float(finite[-2]), 1.0e9] + [float(v) for v in high[:3]]
        thresholds += [float(np.nextafter(v, -np.inf)) for v in high[:2]]
        for thr in thresholds:
            ref_hits, ref_n, ref_pairs, _, _ = t.boost_screen(0, m, 0, m, threshold=thr)           # all FP64
            for first in (1, 0):
                t.set_option(gw.capi.OPT_BOOST_FIRST_LEVEL, first)
                hits, n_hits, n_pairs, *_ = t.boost_screen(0, m, 0, m, threshold=thr, extremes=False)
                assert t.last_boost_kernel() == f"gw::k_boost_mfma_fp4<true, 3, 1, {level if first else 0}, false>"
                assert (n_hits, n_pairs) == (ref_n, ref_pairs)
                assert np.array_equal(np.sort(hits, order=["snp_a", "snp_b"]), np.sort(ref_hits, order=["snp_a", "snp_b"]))
        t.set_option(gw.capi.OPT_BOOST_FIRST_LEVEL, 1)
        assert t.get_option(gw.capi.OPT_BOOST_PREFILTER_TRIPS) == 0     # neither level's run-time check fired
