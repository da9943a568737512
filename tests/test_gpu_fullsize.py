"""Parity at BASELINE.json's full sizes through size-independent properties
(the oracle cannot scan 125 GB in seconds): conservation of the group counts,
case + control = unconditional distribution, mask-swap symmetry, identical
results from differently shaped kernels, and the oracle on sampled rows of the
very table that was scanned.  Needs a GPU with the table's HBM footprint free."""
import numpy as np
import pytest

import libgwaspp_amd as gw
from libgwaspp_amd import capi

pytestmark = pytest.mark.gpu

SEED = 0x5EED
CONFIGS = {
    "c1": (10_000, 50_000),        # BASELINE.json configs[0]
    "c2": (100_000, 500_000),      # configs[1], 12.5 GB
    "c3": (500_000, 1_000_000),    # configs[2], 125 GB
    # not BASELINE configs: typical cohort sizes, through the sub-wave kernels (16 and 32 lanes per row)
    "n6k": (6_000, 1_000_000),
    "n20k": (20_000, 1_000_000),
}


@pytest.mark.parametrize("cfg", ["c1", "c2", "c3", "n6k", "n20k"])
def test_full_size_properties(oracle, cfg):
    if gw.device_count() == 0:
        pytest.fail("no HIP device visible")
    n, m = CONFIGS[cfg]
    info = capi.device_info(0)
    need = (n / 4.0) * m * 1.01 + (2 << 30)
    if info.hbm_free_bytes < need:
        pytest.skip(f"{cfg} needs {need / 1e9:.0f} GB of free HBM")
    first = 7_000_000  # global SNP index of row 0 (a shard in the middle of a bigger job)
    with gw.GwTable(n, m) as t:
        t.simulate(SEED, first, 0, exact=True)   # the exact-count generator bench.py uses (simulate_data.cpp:191-205)
        t.simulate_phenotype(SEED)
        cs, ct, nc, nt = t.get_case_control()
        assert nc + nt == n
        counts, stats = t.scan_case_control()
        dist = t.scan_distribution()

        # conservation: every member of a group lands in exactly one class
        assert np.all(counts[:, :4].sum(axis=1, dtype=np.uint64) == nc)
        assert np.all(counts[:, 4:].sum(axis=1, dtype=np.uint64) == nt)
        # the simulator makes no missing calls, and everyone is a case or a control
        assert not counts[:, 3].any() and not counts[:, 7].any() and not dist[:, 3].any()
        assert np.array_equal(counts[:, :3] + counts[:, 4:7], dist[:, :3])
        assert np.all(np.isfinite(stats))
        assert np.all(stats[:, 7] == 1.0)

        # mask-swap symmetry
        t.set_case_control(ct, cs, nt, nc)
        swapped, sstats = t.scan_case_control()
        assert np.array_equal(swapped[:, :4], counts[:, 4:]) and np.array_equal(swapped[:, 4:], counts[:, :4])
        # the two association tests are symmetric in the groups (up to summation order)
        np.testing.assert_allclose(sstats[:, 5:7], stats[:, 5:7], rtol=1e-11, atol=1e-13)
        t.set_case_control(cs, ct, nc, nt)

        # differently shaped kernels: masks through L2 instead of LDS, other unroll / geometry
        for opts in ({capi.OPT_MASK_IN_LDS: 0, capi.OPT_UNROLL: 2, capi.OPT_BLOCK_THREADS: 256},
                     {capi.OPT_MASK_IN_LDS: 1, capi.OPT_UNROLL: 8, capi.OPT_NONTEMPORAL: 0}):
            for k, v in opts.items():
                t.set_option(k, v)
            c2, s2 = t.scan_case_control()
            assert np.array_equal(c2, counts) and np.array_equal(s2, stats)

        # the oracle on sampled rows of this very table (and its generator by global index)
        g = oracle.geometry_stream(n)
        rng = np.random.default_rng(1)
        sample = np.unique(np.concatenate([[0, 1, m // 2, m - 2, m - 1], rng.integers(0, m, size=27)]))
        for r in sample:
            row = t.download_rows(int(r), 1)
            want, _ = oracle.sim_table_stream(SEED, first + int(r), 1, n, exact=True)
            assert np.array_equal(row, want)
            ca, co = oracle.stream_cc_dist(row[0], g, cs, ct, nc, nt)
            assert tuple(int(v) for v in counts[r]) == ca + co
            np.testing.assert_allclose(stats[r], oracle.assoc_stats(ca, co), rtol=1e-10, atol=1e-12)

        # the data are in Hardy-Weinberg proportions by construction: chi2(1) behaviour
        poly = stats[:, 2] > 0.05
        assert 0.8 < stats[poly, 4].mean() < 1.2
        assert 0.35 < np.median(stats[poly, 4]) < 0.56
        # and phenotype is independent of genotype: genotypic test ~ chi2(2)
        assert 1.8 < stats[poly, 5].mean() < 2.2


def test_million_sample_rows_use_the_single_mask_kernel(oracle):
    """N = 1 M: two masks (250 kB) exceed the 160 KiB LDS, one fits -> with complementary masks the
    scan stages the case mask only; otherwise it walks the samples in column segments whose two masks
    fit (or, with that switched off, reads the masks through L2).  Same records every way, checked
    against the oracle on sampled rows."""
    if gw.device_count() == 0:
        pytest.fail("no HIP device visible")
    n, m = 1_000_000, 3000
    with gw.GwTable(n, m) as t:
        t.simulate(SEED, 0, 300)
        t.simulate_phenotype(SEED)
        assert t.get_option(capi.OPT_COMPLEMENT) == 1 and t.get_option(capi.OPT_MASK_IN_LDS) == 1
        cs, ct, nc, nt = t.get_case_control()
        counts, stats = t.scan_case_control()
        t.set_option(capi.OPT_COMPLEMENT, 0)
        # two masks do not fit: the scan walks the samples in segments of 4096 units (both masks in LDS) ...
        assert t.get_option(capi.OPT_MASK_IN_LDS) == 0 and t.get_option(capi.OPT_COLUMN_SEGMENT) == 4096
        c2, s2 = t.scan_case_control()
        assert np.array_equal(counts, c2) and np.array_equal(stats, s2)
        # ... or, with segmentation off, in one pass with the masks read through L2
        t.set_option(capi.OPT_COLUMN_SEGMENT, -1)
        assert t.get_option(capi.OPT_COLUMN_SEGMENT) == 0
        c3, s3 = t.scan_case_control()
        assert np.array_equal(counts, c3) and np.array_equal(stats, s3)
        t.set_option(capi.OPT_COLUMN_SEGMENT, 0)
        g = oracle.geometry_stream(n)
        for r in (0, 1, m // 2, m - 1):
            row = t.download_rows(r, 1)
            ca, co = oracle.stream_cc_dist(row[0], g, cs, ct, nc, nt)
            assert tuple(int(v) for v in counts[r]) == ca + co
            np.testing.assert_allclose(stats[r], oracle.assoc_stats(ca, co), rtol=1e-10, atol=1e-12)


def test_config5_all_pairs_of_100k_snps(oracle):
    """BASELINE.json configs[4] at full size on one GPU: the BOOST / KSA pre-screen of computeBoost
    (algorithms/epistasis_func.cpp:393-486 over the margin-assisted tables of table5.cpp:989-1150) over all
    4 999 950 000 pairs of 100 000 SNPs x 10 000 samples, three ways -- int8-MFMA kernels with the
    single-precision pre-filter (the default), the same with every pair's FP64 measure, and the vector-ALU
    kernels -- which must list the same pairs with the same measures; the oracle's measure on eight
    sampled 32 x 32 blocks of the pair space and on 50 of the listed pairs."""
    if gw.device_count() == 0:
        pytest.fail("no HIP device visible")
    n, m, thr = 10_000, 100_000, 30.0
    order = ["snp_a", "snp_b"]
    with gw.GwTable(n, m) as t:
        t.simulate(SEED, 0, 0, exact=True)       # bench.py's input (22 096 pairs over 30)
        t.simulate_phenotype(SEED)
        cs, ct, nc, nt = t.get_case_control()
        gs = oracle.geometry_stream(n)
        lists = {}
        for name, mfma, pre in (("mfma+prefilter", 1, 1), ("mfma", 1, 0), ("valu+prefilter", 0, 1)):
            t.set_option(capi.OPT_BOOST_MATRIX_CORES, mfma)
            t.set_option(capi.OPT_BOOST_PREFILTER, pre)
            hits, n_hits, n_pairs, vmax, vmin = t.boost_screen(0, m, 0, m, threshold=thr, max_hits=1 << 18,
                                                               extremes=False)
            assert n_pairs == m * (m - 1) // 2 == 4_999_950_000          # 64-bit pair tally
            assert n_hits == len(hits) and 1000 < n_hits < (1 << 18)
            assert np.all(hits["snp_a"] < hits["snp_b"]) and np.all(hits["measure"] > thr)
            lists[name] = np.sort(hits, order=order)
        ref = lists["mfma+prefilter"]
        for name in ("mfma", "valu+prefilter"):
            assert np.array_equal(lists[name], ref), name                # same pairs, bit-identical measures
        assert len(np.unique(ref[order])) == len(ref)                    # no pair listed twice
        t.set_option(capi.OPT_BOOST_MATRIX_CORES, 1)
        t.set_option(capi.OPT_BOOST_PREFILTER, 1)

        # the oracle on sampled blocks: 8 x (32 x 32) = 8192 pairs, anywhere in the triangle
        rng = np.random.default_rng(5)
        listed = {(int(h["snp_a"]), int(h["snp_b"])): float(h["measure"]) for h in ref}
        checked = 0
        for _ in range(8):
            a0 = int(rng.integers(0, m - 64))
            b0 = int(rng.integers(a0 + 32, m - 32))
            got = t.boost_measures(a0, 32, b0, 32)
            rows = np.concatenate([t.download_rows(a0, 32), t.download_rows(b0, 32)])
            flat = oracle.boost_all_pairs(rows, gs, cs, ct, nc, nt)
            k = 0
            for i in range(64):
                for j in range(i + 1, 64):
                    if i < 32 <= j:
                        w, g = flat[k], got[i, j - 32]
                        assert (np.isnan(w) and np.isnan(g)) or abs(g - w) <= 1e-10 * abs(w) + 1e-8
                        if not np.isnan(w):
                            # the screen's verdict on this pair is the oracle's
                            assert ((a0 + i, b0 + j - 32) in listed) == (g > thr)
                        checked += 1
                    k += 1
        assert checked == 8 * 32 * 32

        # ... and on 50 of the listed pairs, from the pair's own rows
        pad = gs.words * 64
        for h in ref[rng.choice(len(ref), size=50, replace=False)]:
            ra, rb = t.download_rows(int(h["snp_a"]), 1)[0], t.download_rows(int(h["snp_b"]), 1)[0]
            mar = []
            for row in (ra, rb):
                ca, co = oracle.stream_cc_dist(row, gs, cs, ct, nc, nt)
                mar.append(oracle.marginal_information(ca, co, nc + nt))
            c16, t16 = oracle.stream_cc_contingency(ra, rb, gs, cs, ct)
            c16[15] -= pad - nc          # exact tables (table5.cpp:1031,1062): members missing at both markers
            t16[15] -= pad - nt
            want = oracle.boost_measure(c16, t16, mar[0], mar[1], nc + nt)
            assert want > thr and abs(h["measure"] - want) <= 1e-10 * abs(want) + 1e-8
