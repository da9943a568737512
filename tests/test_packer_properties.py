"""Property tests (hypothesis) of the product's host packer against the oracle: any byte string is
either refused by both or packed identically by both, in both reference packings.  CPU only."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import libgwaspp_amd as gw

SYMS = "ACGT0N"


@st.composite
def genotype_rows(draw):
    n = draw(st.integers(1, 70))
    samples = draw(st.one_of(st.just(n), st.integers(0, n + 2)))
    alphabet = draw(st.sampled_from(["AC", "ACGT0N", "CT0", "AG", "ACGT"]))
    text = "".join(draw(st.sampled_from(alphabet)) + draw(st.sampled_from(alphabet)) + "\t" for _ in range(samples))
    return n, text.encode()


@settings(max_examples=400, deadline=None)
@given(genotype_rows())
def test_packer_agrees_with_oracle_on_arbitrary_rows(oracle, case):
    n, text = case
    for layout, lname in ((gw.LAYOUT_STREAM, "stream"), (gw.LAYOUT_BLOCK, "block")):
        try:
            want = oracle.pack_rows([text], n, lname)[0]
        except ValueError:
            want = None
        if want is None:
            with pytest.raises(gw.GwError):
                gw.pack_genotype_row(text, n, layout)
        else:
            assert np.array_equal(gw.pack_genotype_row(text, n, layout), want)


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 3000), st.lists(st.integers(-2, 3001), max_size=60))
def test_mask_builder_agrees_with_oracle(oracle, n, idx):
    try:
        _, want, cnt = oracle.ccs_masks(idx, n)
    except ValueError:
        with pytest.raises(gw.GwError):
            gw.build_stream_mask(idx, n)
        return
    got, gcnt = gw.build_stream_mask(idx, n)
    assert gcnt == cnt and np.array_equal(got, want)
