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
    srcs = [os.path.join(ROOT, "tests", "cpp", "host_cpu_probe.cpp"),
            os.path.join(HOST, "src", "case_control_set.cpp"),
            os.path.join(HOST, "src", "common_genotype_func.cpp")]
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(HOST, "include"), "-o", PROBE] + srcs,
                   check=True)

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
