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
        t.simulate(SEED, first, 0)
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
            want, _ = oracle.sim_table_stream(SEED, first + int(r), 1, n)
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
