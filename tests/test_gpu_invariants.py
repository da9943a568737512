"""Size-independent properties of the case/control scan, checked on every kernel family (sub-wave rows, row at a time,
continuous stream, single mask, column segments): the counts depend on WHO is in a group, not on where the samples
sit or how the scan is cut up.  No oracle involved -- the GPU is compared with itself under transformations whose
effect is known exactly.  Needs a real MI355X: pytest -m gpu."""
import numpy as np
import pytest

import libgwaspp_amd as gw
from libgwaspp_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if gw.device_count() == 0:
        pytest.fail("no HIP device visible: the gpu tests must run on the GPU box")


def _pack(codes):
    """codes [m, n] in {0..3} -> reference stream packing (header 0x7015), vectorised (same as oracle.pack_codes,
    restated here so that this file needs no oracle)."""
    m, n = codes.shape
    nb = ((n // 16 + 1) + 3) & ~3
    c = np.zeros((m, nb * 16), dtype=np.uint8)
    c[:, :n] = codes
    w = (1 << np.arange(16, dtype=np.uint32))
    lo = ((c & 1).reshape(m, nb, 16) * w).sum(axis=2).astype(np.uint16)
    hi = (((c >> 1) & 1).reshape(m, nb, 16) * w).sum(axis=2).astype(np.uint16)
    rows = np.zeros((m, 1 + 2 * nb), dtype=np.uint16)
    rows[:, 0] = 0x7015
    rows[:, 1:1 + nb] = lo
    rows[:, 1 + nb:] = hi
    return rows


def _mask(idx, n):
    m, cnt = gw.build_stream_mask(np.asarray(idx, dtype=np.int32), n)
    return m, cnt


SHAPES = [  # (n_samples, n_snps, options, kernel of the two-mask scan, kernel of the complementary scan)
    (3_000, 700, {}, "gw::k_scan_cc_sub<16, ", "gw::k_scan_cc_sub<16, "),
    (24_000, 300, {}, "gw::k_scan_cc_sub<32, ", "gw::k_scan_cc_sub<32, "),
    (24_000, 300, {capi.OPT_SHORT_ROWS: 0}, "gw::k_scan_cc<", "gw::k_scan_cc<"),
    (90_000, 150, {}, "gw::k_scan_cc<1, true, true, false>", "gw::k_scan_cc<1, true, true, false>"),
    # records deferred to the end of the launch (GW_OPT_DEFER_RECORDS): sub-wave kernels and the row-at-a-time kernel
    (3_000, 700, {capi.OPT_DEFER_RECORDS: 1}, "gw::k_scan_cc_sub<16, 2, true, true>", "gw::k_scan_cc_sub<16, 2, true, true>"),
    (24_000, 300, {capi.OPT_DEFER_RECORDS: 1}, "gw::k_scan_cc_sub<32, 6, true, true>", "gw::k_scan_cc_sub<32, 6, true, true>"),
    (90_000, 150, {capi.OPT_DEFER_RECORDS: 1}, "gw::k_scan_cc<1, true, true, true>", "gw::k_scan_cc<1, true, true, true>"),
    (90_000, 150, {capi.OPT_PIPELINE: 1}, "gw::k_scan_cc_pipe<", "gw::k_scan_cc_pipe<"),
    (90_000, 150, {capi.OPT_COMPLEMENT: 2}, "gw::k_scan_cc<", "gw::k_scan_cc_compl<"),   # single mask where the phenotype allows
    (90_000, 150, {capi.OPT_COLUMN_SEGMENT: 128}, "gw::k_scan_cc", "gw::k_scan_cc"),     # column segments
    (90_000, 150, {capi.OPT_MASK_IN_LDS: 0}, "gw::k_scan_cc<4, false, true, false>", "gw::k_scan_cc<4, false, true, false>"),
]


@pytest.mark.parametrize("n,m,opts,kernel,kernel_compl", SHAPES)
def test_scan_invariants(n, m, opts, kernel, kernel_compl):
    rng = np.random.default_rng(n + m + len(opts))
    q = rng.uniform(0.03, 0.5, size=(m, 1))
    a, b = rng.random((m, n)) < q, rng.random((m, n)) < q
    codes = np.where(a & b, 3, np.where(a | b, 2, 1)).astype(np.uint8)
    codes[rng.random((m, n)) < 0.02] = 0
    u = rng.random(n)
    A = np.flatnonzero(u < 0.3)                  # three disjoint groups + a rest
    B = np.flatnonzero((u >= 0.3) & (u < 0.55))
    C = np.flatnonzero((u >= 0.55) & (u < 0.9))

    def scan(table, cases, ctrls, begin=0, count=None):
        cm, nc = _mask(cases, n)
        tm, nt = _mask(ctrls, n)
        table.set_case_control(cm, tm, nc, nt)
        return table.scan_case_control(begin, count)

    with gw.GwTable(n, m) as t:
        for k, v in opts.items():
            t.set_option(k, v)
        t.upload_rows(0, _pack(codes))
        cAB, sAB = scan(t, A, B)
        assert t.last_scan_kernel().startswith(kernel), t.last_scan_kernel()
        if capi.OPT_COLUMN_SEGMENT in opts:
            assert t.get_option(capi.OPT_COLUMN_SEGMENT) == 128
        # the counts are what they say: class sizes inside each group, straight from the code array (a few rows)
        for r in (0, m // 2, m - 1):
            want = [int((codes[r, g] == c).sum()) for g in (A, B) for c in (1, 2, 3, 0)]
            assert [int(v) for v in cAB[r]] == want
        # 1. swapping the groups swaps the two halves of the record
        cBA, _ = scan(t, B, A)
        assert np.array_equal(cBA[:, :4], cAB[:, 4:]) and np.array_equal(cBA[:, 4:], cAB[:, :4])
        # 2. additivity over disjoint groups: cases(A u C) = cases(A) + cases(C)
        cACB, _ = scan(t, np.concatenate([A, C]), B)
        cCB, _ = scan(t, C, B)
        assert np.array_equal(cACB[:, :4], cAB[:, :4] + cCB[:, :4])
        assert np.array_equal(cACB[:, 4:], cAB[:, 4:]) and np.array_equal(cCB[:, 4:], cAB[:, 4:])
        # 3. everybody a case or a control (complementary masks): the two groups add up to the plain distribution
        everyone = np.arange(n)
        rest = np.setdiff1d(everyone, A)
        cAr, _ = scan(t, A, rest)
        assert t.last_scan_kernel().startswith(kernel_compl), t.last_scan_kernel()
        dist = t.scan_distribution()
        assert np.array_equal(cAr[:, :4] + cAr[:, 4:], dist)
        # 4. a sub-range of rows is a slice of the full scan, records bit-identical
        cAB2, sAB2 = scan(t, A, B)
        assert np.array_equal(cAB2, cAB) and np.array_equal(sAB2, sAB)            # idempotent
        lo, cnt = m // 3, m - m // 3 - 1
        cp, sp = scan(t, A, B, lo, cnt)
        assert np.array_equal(cp, cAB[lo:lo + cnt]) and np.array_equal(sp, sAB[lo:lo + cnt])
        # 5. the statistics are a function of the counts alone: rows with equal counts have equal statistics
        _, first, inv = np.unique(cAB, axis=0, return_index=True, return_inverse=True)
        assert np.array_equal(sAB, sAB[first][inv.reshape(-1)])

    # 6. where the samples sit does not matter: permute the columns of the table and the group members alike
    perm = rng.permutation(n)
    inv_perm = np.empty(n, dtype=np.int64)
    inv_perm[perm] = np.arange(n)
    with gw.GwTable(n, m) as t2:
        for k, v in opts.items():
            t2.set_option(k, v)
        t2.upload_rows(0, _pack(codes[:, perm]))       # new column j holds old sample perm[j]
        cm, nc = _mask(inv_perm[A], n)
        tm, nt = _mask(inv_perm[B], n)
        t2.set_case_control(cm, tm, nc, nt)
        cP, sP = t2.scan_case_control()
        assert np.array_equal(cP, cAB) and np.array_equal(sP, sAB)
