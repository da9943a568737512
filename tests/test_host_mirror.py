"""The C++ mirror of the reference's table API (libgwaspp_amd/host): its
GPU-free parts against the oracle and the real reference CaseControlSet.
CPU only."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "libgwaspp_amd", "host")
BUILD = os.path.join(ROOT, "tests", "cpp", "build")
PROBE = os.path.join(BUILD, "host_cpu_probe")


@pytest.fixture(scope="module")
def probe():
    os.makedirs(BUILD, exist_ok=True)
    # the probe links the product's host library (its GPU-free classes and functions are what is probed)
    pkg = os.path.join(ROOT, "libgwaspp_amd")
    subprocess.run(["make", "-C", os.path.join(pkg, "csrc"), "-s"], check=True)
    subprocess.run(["make", "-C", HOST, "-s"], check=True)
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(HOST, "include"), "-I" + os.path.join(ROOT, "include"),
                    "-o", PROBE, os.path.join(ROOT, "tests", "cpp", "host_cpu_probe.cpp"),
                    "-L" + pkg, "-lgwaspp_hip", "-lgwscan", "-Wl,-rpath," + pkg], check=True)

    def run(*args):
        r = subprocess.run([PROBE] + [str(a) for a in args], check=True, capture_output=True, text=True)
        return r.stdout.splitlines()
    return run


def _hex_words(line):
    return np.array([int(t, 16) for t in line.split()[1:]], dtype=np.uint16)


@pytest.mark.parametrize("n", [1, 8, 15, 16, 17, 64, 70, 129, 1000, 4097])
def test_case_control_set_mirror(oracle, probe, n):
    rng = np.random.default_rng(n)
    status = rng.integers(0, 3, size=n)
    cases, ctrls = np.flatnonzero(status == 1), np.flatnonzero(status == 0)
    out = probe("ccs", n, ",".join(map(str, cases)) or "-", ",".join(map(str, ctrls)) or "-")
    got = {ln.split()[0]: ln for ln in out}
    obc, osc, nc = oracle.ccs_masks(cases, n)
    obt, ost, nt = oracle.ccs_masks(ctrls, n)
    assert np.array_equal(_hex_words(got["block_case"]), obc)
    assert np.array_equal(_hex_words(got["block_ctrl"]), obt)
    assert np.array_equal(_hex_words(got["stream_case"]), osc)
    assert np.array_equal(_hex_words(got["stream_ctrl"]), ost)
    assert got["counts"].split()[1:] == [str(nc), str(nt), str(nc + nt)]
    if oracle.have_ref():   # and against the real reference object
        bc, bt, sc, st, counts = oracle.ref_ccs(n, cases, ctrls)
        assert np.array_equal(_hex_words(got["block_case"]), bc)
        assert np.array_equal(_hex_words(got["stream_ctrl"]), st)
    # setAllAsCases: stream mask all ones incl. pad bits, block mask trimmed at the last sample
    assert np.all(_hex_words(got["all_stream_case"]) == 0xFFFF)
    allb = _hex_words(got["all_block_case"])
    assert np.all(allb[:-1] == 0xFFFF)


def test_marginal_information_mirror(oracle, probe):
    cases, ctrls = (30, 50, 20, 3), (40, 0, 10, 2)
    vals = [float(v) for v in probe("marginal", *cases, *ctrls)[0].split()]
    m = oracle.marginal_information(cases, ctrls, sum(cases) + sum(ctrls))
    want = [m.margins.aa, m.margins.ab, m.margins.bb, m.margins.xx, m.marginal_entropy,
            m.marginal_entropy_y] + list(m.pbc) + list(m.pca)
    assert vals == [float(w) for w in want]


def test_minor_allele_frequency_mirror(oracle, probe):
    for ft in [(10, 20, 70, 5), (70, 20, 10, 0), (25, 0, 75, 0), (1, 1, 1, 1)]:
        tot, maf = (float(v) for v in probe("maf", *ft)[0].split())
        assert (tot, maf) == oracle.minor_allele_frequency(ft)


def test_boost_measure_and_gtest_mirror(oracle, probe):
    """computeBoost's measure and computeGTest (epistasis_func.cpp:424-470, 508-704) in the C++
    mirror vs the oracle's restatement, incl. the pairwise.c known-answer tables."""
    from test_oracle_arith import PAIRWISE_KAT
    rng = np.random.default_rng(11)
    tables = [(np.array(cs), np.array(ct), llr) for _, cs, ct, llr in PAIRWISE_KAT]
    for _ in range(20):
        tables.append((rng.integers(1, 400, size=(3, 3)), rng.integers(1, 400, size=(3, 3)), None))
    tables.append((np.array([[5, 0, 3], [0, 0, 0], [2, 9, 1]]), np.array([[1, 0, 0], [0, 0, 0], [7, 2, 2]]), None))
    for cs, ct, llr in tables:
        got = [float(v) for v in probe("gtest", *cs.reshape(-1), *ct.reshape(-1))[0].split()]
        c16, t16 = np.zeros((4, 4), np.uint32), np.zeros((4, 4), np.uint32)
        c16[:3, :3], t16[:3, :3] = cs, ct
        n = int(cs.sum() + ct.sum())
        m1 = oracle.marginal_information(tuple(int(v) for v in cs.sum(axis=1)) + (0,),
                                         tuple(int(v) for v in ct.sum(axis=1)) + (0,), n)
        m2 = oracle.marginal_information(tuple(int(v) for v in cs.sum(axis=0)) + (0,),
                                         tuple(int(v) for v in ct.sum(axis=0)) + (0,), n)
        want = (oracle.boost_measure(c16, t16, m1, m2, n),) + oracle.gtest(c16, t16, m1, m2, n)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=0, equal_nan=True)
        if llr is not None:
            assert got[0] == pytest.approx(llr, rel=1e-10)



def test_pairwise_epi_test_mirror(oracle, probe):
    """EpistasisPerformance's per-pair test in the C++ mirror == the oracle's restatement of pairwise.c:50-133, and
    its chi-square(4) tail == scipy's."""
    from scipy.stats import chi2
    rng = np.random.default_rng(3)
    tables = [(rng.integers(0, 60, size=9), rng.integers(0, 60, size=9)) for _ in range(12)]
    tables.append((np.array([10, 0, 3, 0, 0, 0, 4, 0, 9]), np.array([8, 0, 2, 0, 0, 0, 5, 0, 7])))   # empty classes
    for cs, ct in tables:
        ll, p = (float(v) for v in probe("llr", *cs, *ct)[0].split())
        want = oracle.pairwise_llr(cs.reshape(3, 3), ct.reshape(3, 3))
        if np.isnan(want):
            assert np.isnan(ll)
            continue
        assert ll == pytest.approx(want, rel=1e-13, abs=1e-10)
        assert p == pytest.approx(chi2.sf(max(ll, 0.0), 4), rel=1e-12)
    # the five literal tables of src/test/pairwise.c:20-37 (LLR and p-values as SURVEY.md section 4 records them)
    from test_oracle_arith import PAIRWISE_KAT
    pvals = {"boost0": 0.22759012753331526, "boost1": 9.97784906723507e-07, "boost2": 3.4987703837442206e-06,
             "biforce1": 6.448641043097075e-08, "biforce2": 1.0796960060009832e-16}
    for name, cs, ct, llr in PAIRWISE_KAT:
        ll, p = (float(v) for v in probe("llr", *np.array(cs).reshape(-1), *np.array(ct).reshape(-1))[0].split())
        assert ll == pytest.approx(llr, rel=1e-12)
        assert p == pytest.approx(pvals[name], rel=1e-9)
