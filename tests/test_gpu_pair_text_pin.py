"""The GPU pair path against counts taken straight from the golden .tped/.tfam text
(tests/text_pairs.py; the oracle is not the reference here): k_pair's plain and case/control 4 x 4
tables (gw_pair_contingency, gw_pair_case_control_contingency) cell for cell on every golden set, and the
all-pairs kernels' counts (int8-MFMA four-corner form on the sets without missing calls, nine-cell form on
the sets with 25 % missing calls, the both-groups forms, and the vector-ALU form) through the interaction
measure evaluated on the text-derived tables.  Needs a real MI355X: pytest -m gpu."""
import numpy as np
import pytest

import libgwaspp_amd as gw
from libgwaspp_amd import capi
from golden_util import golden_sets, read_tfam_status, read_tped
from text_pairs import joint_counts, row_classes, stream_table

pytestmark = pytest.mark.gpu

ALL = golden_sets("cc") + golden_sets("plain")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if gw.device_count() == 0:
        pytest.fail("no HIP device visible: the gpu tests must run on the GPU box")


def _load(oracle, t, lines, n):
    for r, ln in enumerate(lines):
        t.add_genotype_row(r, oracle.tped_genotype_text(ln))


@pytest.mark.parametrize("name", ALL)
def test_k_pair_tables_equal_text_counts(oracle, name):
    n = int(name.split("_n")[1])
    lines = read_tped(name)
    m = len(lines)
    cls = [row_classes(ln) for ln in lines]
    cases, ctrls = read_tfam_status(name)
    cs, nc = gw.build_stream_mask(cases, n)
    ct, nt = gw.build_stream_mask(ctrls, n)
    pairs = np.array([(i, j) for i in range(m) for j in range(i + 1, m)], dtype=np.uint32)
    with gw.GwTable(n, m) as t:
        _load(oracle, t, lines, n)
        plain = t.pair_contingency(pairs)
        t.set_case_control(cs, ct, nc, nt)
        got_c, got_t = t.pair_case_control_contingency(pairs)
    for k, (i, j) in enumerate(pairs):
        assert np.array_equal(plain[k], stream_table(cls[i], cls[j], n))
        assert np.array_equal(got_c[k], stream_table(cls[i], cls[j], n, cases))
        assert np.array_equal(got_t[k], stream_table(cls[i], cls[j], n, ctrls))


@pytest.mark.parametrize("name", [s for s in ALL if int(s.split("_n")[1]) >= 63])
def test_all_pairs_kernels_count_what_the_text_holds(oracle, name):
    """gw_boost_measures (every kernel family) == the interaction measure (epistasis_func.cpp:424-470)
    of the text-derived case/control tables and margins.  A single count off by one moves the measure by
    ~1e-3 or more at these sizes; the comparison is to 1e-9."""
    n = int(name.split("_n")[1])
    lines = read_tped(name)
    m = len(lines)
    cls = [row_classes(ln) for ln in lines]
    cases, ctrls = read_tfam_status(name)
    if len(cases) == 0 or len(ctrls) == 0:
        # the 'plain' sets (25 % missing calls) put everybody in one group: split them by sample index so that
        # the nine-cell kernels (tables WITH missing calls) are pinned to the text too
        cases = np.arange(0, n, 2, dtype=np.int32)
        ctrls = np.arange(1, n, 2, dtype=np.int32)
    cs, nc = gw.build_stream_mask(cases, n)
    ct, nt = gw.build_stream_mask(ctrls, n)
    mar = []
    for c in cls:
        ca = tuple(int(v) for v in np.bincount(c[cases], minlength=4))
        co = tuple(int(v) for v in np.bincount(c[ctrls], minlength=4))
        mar.append(oracle.marginal_information(ca, co, nc + nt))
    want = np.full((m, m), np.nan)
    for i in range(m):
        for j in range(i + 1, m):
            want[i, j] = oracle.boost_measure(joint_counts(cls[i], cls[j], cases), joint_counts(cls[i], cls[j], ctrls),
                                              mar[i], mar[j], nc + nt)
    with gw.GwTable(n, m) as t:
        _load(oracle, t, lines, n)
        t.set_case_control(cs, ct, nc, nt)
        for mode in (1, 0, 2):   # int8 MFMA (one group at a time), vector ALU, int8 MFMA (both groups)
            t.set_option(capi.OPT_BOOST_MATRIX_CORES, mode)
            got = t.boost_measures(0, m, 0, m)
            # a class nobody has gives 0/0 in the reference's formula: NaN on both sides
            assert np.array_equal(np.isnan(got), np.isnan(want)), mode
            ok = ~np.isnan(want)
            np.testing.assert_allclose(got[ok], want[ok], rtol=1e-9, atol=1e-9)
        rec = t.scan_margins()
    for r in range(m):   # the device margins are the text's single-SNP counts
        assert tuple(int(v) for v in rec[r]["cases"]) == mar[r].cases.tuple()
        assert tuple(int(v) for v in rec[r]["controls"]) == mar[r].controls.tuple()


@pytest.mark.parametrize("name", golden_sets("plain"))
def test_gpu_masked_scan_with_missing_calls_equals_text_counts(oracle, name):
    """The north-star kernel on the reference generator's rows WITH missing calls (25 %), groups laid over them
    (tests/test_pair_text_pin.py split_groups): per-group counts straight from the .tped text, a sample in neither group."""
    from test_pair_text_pin import split_groups
    n = int(name.split("_n")[1])
    lines = read_tped(name)
    cases, ctrls, _ = split_groups(name, n)
    ts = oracle.pack_rows([oracle.tped_genotype_text(ln) for ln in lines], n, "stream")
    _, sm_c, nc = oracle.ccs_masks(cases, n)
    _, sm_t, nt = oracle.ccs_masks(ctrls, n)
    with gw.GwTable(n, len(lines)) as t:
        t.upload_rows(0, ts)
        t.set_case_control(sm_c, sm_t, nc, nt)
        counts, _ = t.scan_case_control()
    for r, ln in enumerate(lines):
        cls = row_classes(ln)
        want = [int(v) for g in (cases, ctrls) for v in np.bincount(cls[g], minlength=4)[:4]]
        assert [int(v) for v in counts[r]] == want
