"""Oracle masks vs the REAL reference CaseControlSet.

tests/golden/ref_ccs/masks.npz holds the four masks and the counts the reference object produced for 23 column sets
(written once in the build container by tests/golden/ref_ccs/make_ref_ccs.py, which compiles the reference's own
case_control_set.cpp + index_set.cpp in place and calls setCases / setControls).  These tests run everywhere from
that data; where oracle/_ref is built (the container) the file is also re-checked against the live object.  CPU only."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "ref_ccs", "masks.npz")


@pytest.fixture(scope="module")
def masks():
    with np.load(FIXTURE, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def _keys(masks):
    return sorted(k[:-2] for k in masks if k.endswith("_n"))


def test_fixture_is_complete(masks):
    keys = _keys(masks)
    assert len(keys) == 23
    for k in keys:
        for part in ("cases", "ctrls", "block_case", "block_ctrl", "stream_case", "stream_ctrl", "counts"):
            assert f"{k}_{part}" in masks


def test_oracle_masks_equal_the_reference_objects(oracle, masks):
    for k in _keys(masks):
        n = int(masks[k + "_n"][0])
        cases, ctrls = masks[k + "_cases"], masks[k + "_ctrls"]
        obc, osc, nc = oracle.ccs_masks(cases, n)
        obt, ost, nt = oracle.ccs_masks(ctrls, n)
        assert (nc, nt, nc + nt) == tuple(int(v) for v in masks[k + "_counts"]), k
        assert np.array_equal(masks[k + "_block_case"], obc) and np.array_equal(masks[k + "_block_ctrl"], obt), k
        assert np.array_equal(masks[k + "_stream_case"], osc) and np.array_equal(masks[k + "_stream_ctrl"], ost), k


def test_duplicates_count_once(masks):
    assert int(masks["dup100_counts"][0]) == 3          # {3, 3, 5, 99, 5}
    assert int(np.unpackbits(masks["dup100_stream_case"].view(np.uint8)).sum()) == 3


def test_product_mask_builder_equals_the_reference_objects(masks):
    """gw_build_stream_mask (the C-ABI's host-side helper, no device needed) against the same data."""
    import libgwaspp_amd as gw
    for k in _keys(masks):
        n = int(masks[k + "_n"][0])
        for grp, idx in (("case", masks[k + "_cases"]), ("ctrl", masks[k + "_ctrls"])):
            got, cnt = gw.build_stream_mask(idx, n)
            assert np.array_equal(got, masks[f"{k}_stream_{grp}"]), (k, grp)
            assert cnt == int(masks[k + "_counts"][0 if grp == "case" else 1]), (k, grp)


def test_fixture_equals_the_live_reference_object(oracle, masks):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference on this machine); the committed fixture stands alone")
    for k in _keys(masks):
        n = int(masks[k + "_n"][0])
        bc, bt, sc, st, counts = oracle.ref_ccs(n, masks[k + "_cases"], masks[k + "_ctrls"])
        assert np.array_equal(bc, masks[k + "_block_case"]) and np.array_equal(bt, masks[k + "_block_ctrl"]), k
        assert np.array_equal(sc, masks[k + "_stream_case"]) and np.array_equal(st, masks[k + "_stream_ctrl"]), k
        assert np.array_equal(counts, masks[k + "_counts"]), k
