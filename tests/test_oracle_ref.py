"""Oracle masks vs the REAL reference CaseControlSet (oracle/_ref/libref_ccs.so,
compiled by oracle/Makefile from /root/reference sources in place).  Runs
wherever oracle/_ref exists (the build container; the prebuilt .so also travels
to the GPU box).  CPU only."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    return oracle


@pytest.mark.parametrize("n", [1, 7, 8, 15, 16, 17, 63, 64, 65, 70, 127, 128, 129, 1000, 4097, 10_000])
def test_masks_equal_reference(ref, n):
    rng = np.random.default_rng(n)
    status = rng.integers(0, 3, size=n)  # 0 control, 1 case, 2 neither
    cases = np.flatnonzero(status == 1).astype(np.int32)
    ctrls = np.flatnonzero(status == 0).astype(np.int32)
    bc, bt, sc, st, counts = ref.ref_ccs(n, cases, ctrls)
    obc, osc, nc = ref.ccs_masks(cases, n)
    obt, ost, nt = ref.ccs_masks(ctrls, n)
    assert (nc, nt, nc + nt) == tuple(int(v) for v in counts)
    assert np.array_equal(bc, obc) and np.array_equal(bt, obt)
    assert np.array_equal(sc, osc) and np.array_equal(st, ost)


def test_duplicates_count_once(ref):
    n = 100
    cases = np.array([3, 3, 5, 99, 5], dtype=np.int32)
    _, _, cnt = ref.ccs_masks(cases, n)
    bc, bt, sc, st, counts = ref.ref_ccs(n, cases, np.zeros(0, np.int32))
    assert cnt == 3 == int(counts[0])
