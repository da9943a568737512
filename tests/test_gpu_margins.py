"""Device-computed marginal_information (gw_scan_margins) against the oracle's
restatement of computeMarginalInformation (genotype/common_genotype_func.cpp:173-219), field by field:
counts bit-exact, entropies and probabilities within 1e-10.  Also through the drop-in table's
three-argument getter (table5.cpp:703-747) as computeMargins calls it (epistasis_func.cpp:706-721).
Needs a real MI355X: pytest -m gpu."""
import os
import subprocess

import numpy as np
import pytest

import libgwaspp_amd as gw
from libgwaspp_amd import capi
from golden_util import PERL_DIR, golden_sets, read_tfam_status, read_tped

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "libgwaspp_amd", "bin", "gwas_hip")
RTOL, ATOL = 1e-10, 1e-14


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if gw.device_count() == 0:
        pytest.fail("no HIP device visible: the gpu tests must run on the GPU box")


def _codes(rng, m, n, p_missing):
    q = rng.uniform(0.02, 0.5, size=(m, 1))
    a, b = rng.random((m, n)) < q, rng.random((m, n)) < q
    codes = np.where(a & b, 3, np.where(a | b, 2, 1)).astype(np.uint8)
    codes[rng.random((m, n)) < p_missing] = 0
    return codes


def _check_record(rec, want, written_only=True):
    """rec: one MARGINAL_DTYPE record from the device; want: oracle.Marginal."""
    assert tuple(int(v) for v in rec["margins"]) == want.margins.tuple()
    assert tuple(int(v) for v in rec["cases"]) == want.cases.tuple()
    assert tuple(int(v) for v in rec["controls"]) == want.controls.tuple()
    np.testing.assert_allclose(rec["marginal_entropy"], want.marginal_entropy, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(rec["marginal_entropy_y"], want.marginal_entropy_y, rtol=RTOL, atol=ATOL)
    ca, co = want.cases.tuple(), want.controls.tuple()
    for g in range(4):
        # quirk Q6: the reference writes dPbc / dPca only where the count is non-zero; the device record
        # holds zero elsewhere
        for off, cnt in ((0, ca[g]), (4, co[g])):
            if cnt > 0:
                np.testing.assert_allclose(rec["pbc"][off + g], want.pbc[off + g], rtol=RTOL, atol=ATOL)
                np.testing.assert_allclose(rec["pca"][off + g], want.pca[off + g], rtol=RTOL, atol=ATOL)
            else:
                assert rec["pbc"][off + g] == 0.0 and rec["pca"][off + g] == 0.0


@pytest.mark.parametrize("n,m,p_missing", [(70, 33, 0.1), (1000, 257, 0.02), (10_000, 600, 0.0), (40_000, 130, 0.01)])
def test_margins_vs_oracle(oracle, n, m, p_missing):
    rng = np.random.default_rng(n + m)
    codes = _codes(rng, m, n, p_missing)
    codes[0, :] = 1                     # a monomorphic row: empty classes everywhere
    codes[1, :] = 0                     # a row of missing calls only
    rows = oracle.pack_codes(codes, "stream")
    g = oracle.geometry_stream(n)
    u = rng.random(n)
    cases = np.flatnonzero(u < 0.4).astype(np.int32)
    ctrls = np.flatnonzero((u >= 0.4) & (u < 0.9)).astype(np.int32)   # 10 % in neither group
    _, cs, nc = oracle.ccs_masks(cases, n)
    _, ct, nt = oracle.ccs_masks(ctrls, n)
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, rows)
        t.set_case_control(cs, ct, nc, nt)
        got = t.scan_margins()
        part = t.scan_margins(5, 11)                        # a sub-range
        counts, _ = t.scan_case_control()
        # once the table is "selected" (the pair path ran) the records come from the compaction's margins
        t.boost_measures(0, min(m, 8), 0, min(m, 8))
        again = t.scan_margins()
    assert got.dtype.itemsize == 192
    assert np.array_equal(got[5:16].view(np.uint8), part.view(np.uint8))
    assert np.array_equal(got.view(np.uint8), again.view(np.uint8))
    for r in range(m):
        ca, co = oracle.stream_cc_dist(rows[r], g, cs, ct, nc, nt)
        assert tuple(int(v) for v in counts[r]) == tuple(ca) + tuple(co)
        _check_record(got[r], oracle.marginal_information(ca, co, nc + nt))


def test_margins_sample_in_both_masks_is_a_case(oracle):
    """selectCaseControl keeps a sample that is in both sets as a case only (table5.cpp:531-548); the
    margins are those of the selected table."""
    n, m = 500, 40
    rng = np.random.default_rng(11)
    codes = _codes(rng, m, n, 0.03)
    rows = oracle.pack_codes(codes, "stream")
    g = oracle.geometry_stream(n)
    cases = np.arange(0, 260, dtype=np.int32)
    ctrls = np.arange(200, 500, dtype=np.int32)             # 200..259 in both
    _, cs, nc = oracle.ccs_masks(cases, n)
    _, ct, nt = oracle.ccs_masks(ctrls, n)
    only = np.arange(260, 500, dtype=np.int32)
    _, ct_only, nt_only = oracle.ccs_masks(only, n)
    with gw.GwTable(n, m) as t:
        t.upload_rows(0, rows)
        t.set_case_control(cs, ct, nc, nt)
        got = t.scan_margins()
    for r in range(m):
        ca, co = oracle.stream_cc_dist(rows[r], g, cs, ct_only, nc, nt_only)
        _check_record(got[r], oracle.marginal_information(ca, co, nc + nt_only))


def test_margins_need_masks_and_range():
    with gw.GwTable(100, 10) as t:
        with pytest.raises(gw.GwError) as e:
            t.scan_margins()
        assert e.value.code == -4                            # GW_E_STATE
        t.simulate(1, 0, 0)
        t.simulate_phenotype(1)
        with pytest.raises(gw.GwError) as e:
            t.scan_margins(5, 6)
        assert e.value.code == -5                            # GW_E_RANGE
        assert len(t.scan_margins(10, 0)) == 0


@pytest.mark.parametrize("name", golden_sets("cc")[:3])
@pytest.mark.parametrize("window", [5, 1 << 18])
def test_three_argument_getter_through_the_drop_in_table(oracle, name, window):
    """gwas_hip --margins = computeMargins' loop over GenoTable::getCaseControlGenotypeDistribution(i, ccgd, m)
    on the golden .tped/.tfam: every field against the oracle."""
    if not os.path.exists(CLI):
        subprocess.run(["make", "-C", os.path.join(ROOT, "libgwaspp_amd", "host"), "-s"], check=True)
    n = int(name.split("_n")[1])
    lines = read_tped(name)
    cases, ctrls = read_tfam_status(name)
    _, cs, nc = oracle.ccs_masks(cases, n)
    _, ct, nt = oracle.ccs_masks(ctrls, n)
    g = oracle.geometry_stream(n)
    out = subprocess.run([CLI, "-g", os.path.join(PERL_DIR, name + ".tped"), "-p", os.path.join(PERL_DIR, name + ".tfam"),
                          "--scan-window", str(window), "--margins"], check=True, capture_output=True, text=True).stdout
    recs = [ln.split("\t") for ln in out.splitlines() if ln.count("\t") == 30]
    assert len(recs) == len(lines)
    for r, ln in enumerate(lines):
        row = oracle.pack_rows([oracle.tped_genotype_text(ln)], n, "stream")[0]
        ca, co = oracle.stream_cc_dist(row, g, cs, ct, nc, nt)
        want = oracle.marginal_information(ca, co, nc + nt)
        f = recs[r]
        assert int(f[0]) == r
        assert tuple(int(v) for v in f[1:5]) == want.margins.tuple()
        assert tuple(int(v) for v in f[5:9]) == tuple(ca)
        assert tuple(int(v) for v in f[9:13]) == tuple(co)
        rec = np.zeros(1, dtype=capi.MARGINAL_DTYPE)[0]
        rec["margins"], rec["cases"], rec["controls"] = want.margins.tuple(), tuple(ca), tuple(co)
        rec["marginal_entropy"], rec["marginal_entropy_y"] = float(f[13]), float(f[14])
        rec["pbc"] = [float(v) for v in f[15:23]]
        rec["pca"] = [float(v) for v in f[23:31]]
        _check_record(rec, want)
