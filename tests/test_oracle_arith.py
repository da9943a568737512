"""Arithmetic on the counts: MinorAlleleFrequency, computeMarginalInformation,
the build-defined chi-square / HWE statistics and the pairwise LLR KATs.
CPU only."""
import math

import numpy as np
import pytest


def test_minor_allele_frequency_quirk(oracle):
    # algorithms/maf_func.h:46-54: (2aa+ab)/tot, flipped below 0.5 (quirk Q5)
    tot, maf = oracle.minor_allele_frequency((10, 20, 70, 5))
    assert tot == 100.0 and maf == pytest.approx(1.0 - 0.4)
    tot, maf = oracle.minor_allele_frequency((70, 20, 10, 0))
    assert maf == pytest.approx(1.6)  # not a frequency: the reference divides by tot, not 2*tot
    tot, maf = oracle.minor_allele_frequency((25, 0, 75, 0))
    assert maf == 0.5


def test_marginal_information(oracle):
    cases, ctrls = (30, 50, 20, 3), (40, 45, 10, 2)
    n = sum(cases) + sum(ctrls)
    m = oracle.marginal_information(cases, ctrls, n)
    mar = [a + b for a, b in zip(cases, ctrls)]
    assert (m.margins.aa, m.margins.ab, m.margins.bb, m.margins.xx) == tuple(mar)
    h = -sum(v / n * math.log(v / n) for v in mar if v)
    hy = -sum(v / n * math.log(v / n) for v in cases + ctrls if v)
    assert m.marginal_entropy == pytest.approx(h, rel=1e-14)
    assert m.marginal_entropy_y == pytest.approx(hy, rel=1e-14)
    for i in range(4):
        assert m.pbc[i] == cases[i] / sum(cases)
        assert m.pbc[4 + i] == ctrls[i] / sum(ctrls)
        assert m.pca[i] == cases[i] / mar[i]
        assert m.pca[4 + i] == ctrls[i] / mar[i]
    with pytest.raises(ValueError):
        oracle.marginal_information(cases, ctrls, n + 1)  # common_genotype_func.cpp:180 assert


def test_assoc_stats_against_scipy(oracle):
    """The chi-square / HWE definitions are the build's own (the reference has
    none: SURVEY.md 8a row 13) -- cross-checked here against scipy's Pearson
    chi-square so that the definition itself is the textbook one."""
    from scipy.stats import chi2_contingency
    rng = np.random.default_rng(3)
    for _ in range(50):
        ca = tuple(int(v) for v in rng.integers(1, 5000, size=3)) + (int(rng.integers(0, 50)),)
        co = tuple(int(v) for v in rng.integers(1, 5000, size=3)) + (int(rng.integers(0, 50)),)
        s = oracle.assoc_stats(ca, co)
        geno = chi2_contingency(np.array([ca[:3], co[:3]]), correction=False)[0]
        alle = chi2_contingency(np.array([[2 * ca[0] + ca[1], 2 * ca[2] + ca[1]],
                                          [2 * co[0] + co[1], 2 * co[2] + co[1]]]), correction=False)[0]
        assert s[5] == pytest.approx(geno, rel=1e-11)
        assert s[6] == pytest.approx(alle, rel=1e-11)
        for k, (aa, ab, bb) in ((3, co[:3]), (4, tuple(a + b for a, b in zip(ca[:3], co[:3])))):
            nn = aa + ab + bb
            p = (2 * aa + ab) / (2 * nn)
            e = (nn * p * p, 2 * nn * p * (1 - p), nn * (1 - p) ** 2)
            assert s[k] == pytest.approx(sum((o - x) ** 2 / x for o, x in zip((aa, ab, bb), e)), rel=1e-11)
        assert s[0] == oracle.minor_allele_frequency(ca)[1]
        assert s[1] == oracle.minor_allele_frequency(co)[1]
        p = (2 * (ca[0] + co[0]) + ca[1] + co[1]) / (2 * (sum(ca[:3]) + sum(co[:3])))
        assert s[2] == pytest.approx(min(p, 1 - p), rel=1e-15)
        assert s[7] == pytest.approx((sum(ca[:3]) + sum(co[:3])) / (sum(ca) + sum(co)), rel=1e-15)


def test_assoc_stats_degenerate(oracle):
    z = oracle.assoc_stats((0, 0, 0, 0), (0, 0, 0, 0))
    assert np.all(z == 0) and np.all(np.isfinite(z))
    s = oracle.assoc_stats((10, 0, 0, 0), (7, 0, 0, 1))  # monomorphic
    assert np.all(np.isfinite(s)) and s[5] == 0 and s[6] == 0 and s[3] == 0
    s = oracle.assoc_stats((0, 0, 0, 5), (3, 4, 5, 0))  # no called cases
    assert np.all(np.isfinite(s)) and s[0] == 0 and s[5] == 0


# src/test/pairwise.c:20-37: the five literal case/control 3x3 tables.  The
# file stores no expected output; the values below are the ones SURVEY.md
# section 4 records for them ("[probed]", produced during the survey from the
# reference's own pairwise_epi_test).
PAIRWISE_KAT = [
    ("boost0", [[5, 24, 14], [57, 138, 96], [151, 315, 187]],
     [[4, 16, 10], [71, 176, 86], [120, 330, 200]], 5.6412445164852567),
    ("boost1", [[227, 225, 29], [191, 189, 58], [24, 33, 11]],
     [[229, 219, 67], [195, 199, 26], [33, 42, 3]], 33.381542422601164),
    ("boost2", [[32, 103, 93], [49, 250, 203], [52, 94, 111]],
     [[33, 121, 96], [76, 203, 227], [20, 124, 113]], 30.715608133058666),
    ("biforce1", [[2, 10, 12], [31, 130, 87], [75, 295, 345]],
     [[3, 14, 3], [16, 90, 143], [84, 353, 307]], 39.162377641819205),
    ("biforce2", [[21, 214, 289], [7, 119, 260], [0, 0, 77]],
     [[12, 123, 542], [3, 69, 223], [4, 14, 23]], 80.980275431820317),
]


@pytest.mark.parametrize("name,cs,ct,llr", PAIRWISE_KAT)
def test_pairwise_llr_kat(oracle, name, cs, ct, llr):
    assert oracle.pairwise_llr(cs, ct) == pytest.approx(llr, rel=1e-12)


def test_simulator_shape(oracle):
    """Restated simulate_data.cpp: p = bin/100 + U{0..999}/1e5 with the bin
    drawn from the affy6 spectrum; genotypes in HWE; coin-flip phenotype."""
    seed = 0x5EED
    ps = np.array([oracle.sim_snp_p_e5(seed, s) for s in range(4000)]) / 1e5
    assert ps.min() >= 0 and ps.max() < 0.51
    assert 0.18 < ps.mean() < 0.30  # affy6 spectrum mean is ~0.236
    n = 20_000
    table, g = oracle.sim_table_stream(seed, 100, 8, n)
    cs, ct, nc, nt = oracle.sim_phenotype_stream(seed, n)
    assert nc + nt == n and abs(nc - n / 2) < 4 * (n ** 0.5)
    for r in range(8):
        aa, ab, bb, xx = oracle.stream_dist(table[r], g)
        assert aa + ab + bb == n  # no missing by default
        p = oracle.sim_snp_p_e5(seed, 100 + r) / 1e5
        q = (ab + 2 * bb) / (2 * n)
        assert abs(q - p) < 5 * (p * (1 - p) / (2 * n)) ** 0.5 + 2e-5
    t2, _ = oracle.sim_table_stream(seed, 100, 2, n, missing_q16=655)  # ~1 %
    aa, ab, bb, xx = oracle.stream_dist(t2[0], g)
    miss = n - (aa + ab + bb)
    assert 100 < miss < 320


@pytest.mark.parametrize("name,cs,ct,llr", PAIRWISE_KAT)
def test_boost_measure_equals_pairwise_llr(oracle, name, cs, ct, llr):
    """computeBoost's interaction measure (epistasis_func.cpp:424-470) is the
    same statistic as src/test/pairwise.c's 2*log-likelihood ratio, so the five
    literal tables of pairwise.c:20-37 pin it too."""
    cs, ct = np.array(cs, dtype=np.uint32), np.array(ct, dtype=np.uint32)
    case16, ctrl16 = np.zeros((4, 4), np.uint32), np.zeros((4, 4), np.uint32)
    case16[:3, :3], ctrl16[:3, :3] = cs, ct
    n = int(cs.sum() + ct.sum())
    m1 = oracle.marginal_information(tuple(int(v) for v in cs.sum(axis=1)) + (0,),
                                     tuple(int(v) for v in ct.sum(axis=1)) + (0,), n)
    m2 = oracle.marginal_information(tuple(int(v) for v in cs.sum(axis=0)) + (0,),
                                     tuple(int(v) for v in ct.sum(axis=0)) + (0,), n)
    assert oracle.boost_measure(case16, ctrl16, m1, m2, n) == pytest.approx(llr, rel=1e-10)


def test_boost_measure_equals_its_information_identity(oracle):
    """An independent derivation of computeBoost's measure (epistasis_func.cpp:424-470) for tables without a missing
    call, in numpy from the genotype codes: with lg = log2, x / y the case / control count of a cell and s = x + y,
        measure = 2 ln 2 [ sum_cells (x lg x + y lg y - s lg s) + n H(status | SNP i) + n H(status | SNP j)
                           - n H(status) + n lg tao ],
        tao = sum_{a, b, c} p(a, b) p(a, c) p(b, c) / (p(a) p(b) p(c))          (c = status),
    i.e. n times [H(C|A) + H(C|B) - H(C|A,B) - H(C)] in bits plus the Kirkwood normaliser -- the identity the device's
    single-precision bounds are built on (gw_boost_dev.h boost_bound_split / boost_bound_quad).  All pairs of 160
    simulated SNPs (exact-count rows, affy6 spectrum) x 3000 samples against the oracle's restatement of the reference
    loop.  Parity unpinned like the measure itself: two derivations agreeing, not a reference-held output."""
    n, m, seed = 3000, 160, 0xB005
    table, g = oracle.sim_table_stream(seed, 0, m, n, 0, exact=True)
    cs, ct, nc, nt = oracle.sim_phenotype_stream(seed, n)
    nb = g.data_blocks
    bits = lambda blocks: np.unpackbits(np.ascontiguousarray(blocks).astype("<u2").view(np.uint8), bitorder="little").reshape(blocks.shape[0], -1)[:, :n]
    codes = bits(table[:, 1:1 + nb]) + 2 * bits(table[:, 1 + nb:1 + 2 * nb])
    case, ctrl = bits(cs[None, :])[0].astype(bool), bits(ct[None, :])[0].astype(bool)
    assert case.sum() == nc and ctrl.sum() == nt and not np.any(codes == 0)
    ind = lambda grp, k: (codes[:, grp] == k).astype(np.float64)
    X = np.array([[ind(case, a) @ ind(case, b).T for b in (1, 2, 3)] for a in (1, 2, 3)])
    Y = np.array([[ind(ctrl, a) @ ind(ctrl, b).T for b in (1, 2, 3)] for a in (1, 2, 3)])
    S, nn = X + Y, float(nc + nt)
    xlgx = lambda v: np.where(v > 0, v * np.log2(np.maximum(v, 1.0)), 0.0)
    A = (xlgx(X) + xlgx(Y) - xlgx(S)).sum(axis=(0, 1))                      # -n H(C | A, B)
    cond = lambda mx, my: -(xlgx(mx) + xlgx(my) - xlgx(mx + my)).sum(axis=0)   # n H(C | one SNP)
    Ki, Kj = cond(X.sum(axis=1), Y.sum(axis=1)), cond(X.sum(axis=0), Y.sum(axis=0))
    hC = -(nc * np.log2(nc / nn) + nt * np.log2(nt / nn))                  # n H(C)
    pab = S / nn
    pa, pb = pab.sum(axis=1), pab.sum(axis=0)
    tao = np.zeros_like(A)
    for cnts, tot in ((X, nc), (Y, nt)):
        pac, pbc = cnts.sum(axis=1) / nn, cnts.sum(axis=0) / nn
        for a in range(3):
            for b in range(3):
                with np.errstate(divide="ignore", invalid="ignore"):
                    tao += np.nan_to_num(pab[a, b] * pac[a] * pbc[b] / (pa[a] * pb[b] * (tot / nn)))
    mine = 2.0 * np.log(2.0) * (A + Ki + Kj - hC + nn * np.log2(tao))[np.triu_indices(m, 1)]
    ref = oracle.boost_all_pairs(table, g, cs, ct, nc, nt)
    ok = np.isfinite(ref) & np.isfinite(mine)
    assert ok.sum() > 0.9 * ref.size
    assert np.max(np.abs(mine[ok] - ref[ok])) < 1e-8
