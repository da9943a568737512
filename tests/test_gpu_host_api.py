"""The reference's own test flow against the drop-in table: the CLI harness
(libgwaspp_amd/bin/gwas_hip, after src/test/gwas_basic.cpp) reads the golden
.tped/.tfam pairs through GeneticData -> GenoTable::addGenotypeRow and prints
per-SNP counts, which are diffed with genotype_set_builder.pl's expected files
(SURVEY.md section 4).  Needs a GPU."""
import os
import subprocess

import numpy as np
import pytest

import libgwaspp_amd as gw
from golden_util import (PERL_DIR, expected_in_table_space, golden_sets, read_dist, read_tfam_status,
                         read_tped)

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "libgwaspp_amd", "bin", "gwas_hip")


@pytest.fixture(scope="module", autouse=True)
def _cli():
    if gw.device_count() == 0:
        pytest.fail("no HIP device visible")
    if not os.path.exists(CLI):
        subprocess.run(["make", "-C", os.path.join(ROOT, "libgwaspp_amd", "host"), "-s"], check=True)


def _run(name, mode, out=None, extra=()):
    cmd = [CLI, "-g", os.path.join(PERL_DIR, name + ".tped"), "-p", os.path.join(PERL_DIR, name + ".tfam")]
    if out:
        cmd += ["-o", out]
    r = subprocess.run(cmd + list(extra) + [mode], check=True, capture_output=True, text=True)
    return r.stdout


def _table_lines(text):
    # the table constructors log to stdout like the reference's: keep number rows only
    rows = []
    for ln in text.splitlines():
        t = ln.split()
        if t and all(x.lstrip("-").isdigit() for x in t):
            rows.append([int(x) for x in t])
    return rows


@pytest.mark.parametrize("name", golden_sets("plain"))
def test_inline_maf_print_equals_expected_dist(name):
    lines = read_tped(name)
    exp = read_dist(name, "dist")
    rows = _table_lines(_run(name, "--test-inline-maf"))
    assert len(rows) == len(lines)
    for r, ln in enumerate(lines):
        aa, ab, bb, xx = expected_in_table_space(exp[r], ln)
        assert rows[r] == [xx, aa, ab, bb]     # "xx aa ab bb", maf_func.cpp:332


@pytest.mark.parametrize("name", golden_sets("cc"))
def test_inline_cc_print_equals_expected_dists(name, tmp_path):
    lines = read_tped(name)
    exp_c, exp_t = read_dist(name, "case.dist"), read_dist(name, "control.dist")
    out = str(tmp_path / "cc")
    _run(name, "--test-inline-cc-maf", out)
    got_c = _table_lines(open(out + ".case").read())
    got_t = _table_lines(open(out + ".control").read())
    assert len(got_c) == len(got_t) == len(lines)
    for r, ln in enumerate(lines):
        aa, ab, bb, xx = expected_in_table_space(exp_c[r], ln)
        assert got_c[r] == [xx, aa, ab, bb]
        aa, ab, bb, xx = expected_in_table_space(exp_t[r], ln)
        assert got_t[r] == [xx, aa, ab, bb]


@pytest.mark.parametrize("name", ["plain_m4_n1", "plain_m4_n17", "plain_m4_n129", "plain_m40_n70"])
def test_valid_geno_round_trip(name):
    """--valid-geno (validate_func.cpp:31-55): every call decodes back to the text."""
    lines = read_tped(name)
    out = [ln.split("\t") for ln in _run(name, "--valid-geno").splitlines() if ln.startswith("rs")]
    assert len(out) == len(lines)
    for r, ln in enumerate(lines):
        t = ln.split()
        want = [(a + b) if a != "0" else "00" for a, b in zip(t[4::2], t[5::2])]
        assert out[r][0] == t[1] and out[r][1:] == want


@pytest.mark.parametrize("name", ["cc_m6_n70", "cc_m4_n129", "cc_m4_n16"])
def test_epi_debug_three_getters(oracle, name):
    """The three getCaseControlContingencyTable overloads, with the xx_xx
    conventions of table5.cpp:806-895 (inline), :896-987 (compacted) and
    :989-1150 (margin-assisted)."""
    n = int(name.split("_n")[1])
    lines = read_tped(name)
    cases, ctrls = read_tfam_status(name)
    ts = oracle.pack_rows([oracle.tped_genotype_text(ln) for ln in lines], n)
    gs = oracle.geometry_stream(n)
    _, cs, nc = oracle.ccs_masks(cases, n)
    _, ct, nt = oracle.ccs_masks(ctrls, n)
    rows = [ln.split("\t") for ln in _run(name, "--epi-debug").splitlines() if "\tcase\t" in ln or "\tctrl\t" in ln]
    k = 0
    m = len(lines)
    import ctypes as C
    cb, tb = C.c_uint32(), C.c_uint32()
    oracle.lib().gwo_compact_geometry(nc, nt, C.byref(cb), C.byref(tb))
    for i in range(m):
        for j in range(i + 1, m):
            wc, wt = oracle.stream_cc_contingency(ts[i], ts[j], gs, cs, ct)
            for variant in range(3):
                for which, want, members, words in (("case", wc, nc, cb.value // 4), ("ctrl", wt, nt, tb.value // 4)):
                    got = rows[k]
                    k += 1
                    assert got[:3] == [str(i), str(j), which]
                    vals = np.array([int(v) for v in got[3:]], dtype=np.int64)
                    w = want.astype(np.int64).copy()
                    if variant == 1:      # compacted rows: pad bits of the compacted planes
                        w[15] += words * 64 - gs.words * 64
                    elif variant == 2:    # margin-assisted: exact
                        w[15] -= gs.words * 64 - members
                    assert np.array_equal(vals, w), (i, j, variant, which)
    assert k == len(rows)


def test_assoc_statistics(oracle):
    name = "cc_m24_n1000"
    n = 1000
    lines = read_tped(name)
    cases, ctrls = read_tfam_status(name)
    ts = oracle.pack_rows([oracle.tped_genotype_text(ln) for ln in lines], n)
    _, cs, nc = oracle.ccs_masks(cases, n)
    _, ct, nt = oracle.ccs_masks(ctrls, n)
    wc, ws = oracle.scan_stream_cc(ts, oracle.geometry_stream(n), cs, ct, nc, nt)
    rows = [ln.split("\t") for ln in _run(name, "--assoc").splitlines() if ln.count("\t") == 15]
    assert len(rows) == len(lines)
    for r, t in enumerate(rows):
        assert [int(v) for v in t[:8]] == [int(v) for v in wc[r]]
        np.testing.assert_allclose([float(v) for v in t[8:]], ws[r], rtol=1e-10, atol=1e-12)


def test_compute_boost_report(oracle, tmp_path):
    """computeBoost through the harness (--test-boost-epi): KSA pre-screen of all pairs on the
    GPU, exact G test on the hits, the reference's report lines -- against the oracle's
    restatement of the same pipeline (epistasis_func.cpp:349-506, 508-704)."""
    import ctypes as C
    n, m = 3000, 40
    rng = np.random.default_rng(21)
    status = rng.integers(0, 2, size=n)
    q = rng.uniform(0.1, 0.5, size=(m, 1))
    a = (rng.random((m, n)) < q).astype(np.uint8) + (rng.random((m, n)) < q).astype(np.uint8)  # minor-allele count
    # two planted interactions: in cases the second SNP copies the first
    for i, j in ((3, 17), (20, 31)):
        copy = (status == 1) & (rng.random(n) < 0.7)
        a[j, copy] = a[i, copy]
    a[:, 0], a[:, 1], a[:, 2] = 0, 1, 2     # every row starts AA, AC, CC: code space == allele space
    sym = np.array(["A A", "A C", "C C"])
    with open(tmp_path / "d.tped", "w") as f:
        for r in range(m):
            f.write(f"0 rs{r} 0 {r} " + " ".join(sym[a[r]]) + "\n")
    with open(tmp_path / "d.tfam", "w") as f:
        for s in range(n):
            f.write(f"Fam_{s} Ind_{s} Pat_{s} Mat_{s} x {status[s]}\n")
    cmd = [CLI, "-g", str(tmp_path / "d.tped"), "-p", str(tmp_path / "d.tfam"), "--test-boost-epi"]
    text = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    rows = [ln.split("\t") for ln in text.splitlines() if ln.count("\t") == 6]
    located = [ln for ln in text.splitlines() if ln.startswith("Located ")]

    # the oracle's pipeline
    ts = oracle.pack_codes(a + 1, "stream")
    gs = oracle.geometry_stream(n)
    _, cs, nc = oracle.ccs_masks(np.flatnonzero(status == 1), n)
    _, ct, nt = oracle.ccs_masks(np.flatnonzero(status == 0), n)
    flat = oracle.boost_all_pairs(ts, gs, cs, ct, nc, nt)
    pad = gs.words * 64
    mar = []
    for r in range(m):
        ca, co = oracle.stream_cc_dist(ts[r], gs, cs, ct, nc, nt)
        mar.append(oracle.marginal_information(ca, co, n))
    want, k, passed = [], 0, 0
    for i in range(m):
        for j in range(i + 1, m):
            if flat[k] > 30.0:
                passed += 1
                c16, t16 = oracle.stream_cc_contingency(ts[i], ts[j], gs, cs, ct)
                c16[15] -= pad - nc
                t16[15] -= pad - nt
                measure, z = oracle.gtest(c16, t16, mar[i], mar[j], n)
                if measure > 30.0:
                    want.append((i, j, measure, z))
            k += 1
    assert located == [f"Located {passed} potential interactions"]
    assert len(rows) == len(want) >= 2
    assert {(3, 17), (20, 31)} <= {(w[0], w[1]) for w in want}
    for idx, (got, w) in enumerate(zip(rows, want)):
        assert [int(got[0]), int(got[1]), int(got[2])] == [idx, w[0], w[1]]
        assert float(got[3]) == 0.0 and float(got[4]) == 0.0
        assert float(got[5]) == pytest.approx(w[2], abs=2e-6)
        assert float(got[6]) == pytest.approx(w[3], abs=2e-6)


def test_plain_c_example_runs(tmp_path):
    """examples/scan_demo.c end to end on the GPU (C99 caller of the C-ABI)."""
    exe = str(tmp_path / "scan_demo")
    pkg = os.path.join(ROOT, "libgwaspp_amd")
    subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "scan_demo.c"),
                    "-L" + pkg, "-lgwscan", "-Wl,-rpath," + pkg, "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    lines = [ln for ln in out.splitlines() if ln.startswith("SNP ")]
    assert len(lines) == 3
    # SNP 0: samples 0,2,4,6 are cases: AA CC 00 AA -> aa 2, bb 1, xx 1; controls AC AA AC CC -> aa 1, ab 2, bb 1
    assert "SNP 0  cases aa/ab/bb/xx 2 0 1 1  controls 1 2 1 0" in lines[0]
    assert "SNP 2  cases aa/ab/bb/xx 4 0 0 0  controls 4 0 0 0" in lines[2]


def test_contingency_performance_harness():
    """--contin-perform: every pair's 4x4 table (batched on the device); each table sums to the
    table's 64-bit-word capacity (pad bits land in xx_xx, table5.cpp:749-800)."""
    name = "plain_m40_n70"
    out = _run(name, "--contin-perform")
    line = [ln for ln in out.splitlines() if ln.startswith("Contingencies 40:")][0]
    words = (((70 // 16 + 1) + 3) & ~3) // 4
    assert f"(sum of all cells {40 * 39 // 2 * words * 64})" in line


@pytest.mark.parametrize("window", [1, 3, 7, 64])
def test_getters_served_from_prefetched_windows(window, tmp_path):
    """The per-row getters are served from device scans of `window` rows; the window after the one being
    walked is scanned ahead on a worker thread.  Walking 40- and 24-row tables through 1 .. 40 windows, with and
    without the worker, must print what one big window prints."""
    name_cc = "cc_m24_n1000"
    name_plain = "plain_m40_n70"
    ref_plain = _table_lines(_run(name_plain, "--test-inline-maf"))
    out = str(tmp_path / "ref")
    _run(name_cc, "--test-inline-cc-maf", out)
    ref_cc = (_table_lines(open(out + ".case").read()), _table_lines(open(out + ".control").read()))
    ref_assoc = _run(name_cc, "--assoc")
    for extra in (["--scan-window", str(window)], ["--scan-window", str(window), "--no-prefetch"]):
        assert _table_lines(_run(name_plain, "--test-inline-maf", extra=extra)) == ref_plain
        o = str(tmp_path / ("w%d_%d" % (window, len(extra))))
        _run(name_cc, "--test-inline-cc-maf", o, extra=extra)
        assert (_table_lines(open(o + ".case").read()), _table_lines(open(o + ".control").read())) == ref_cc
        assert _run(name_cc, "--assoc", extra=extra) == ref_assoc


def test_case_control_contingency_performance_harness():
    """--contin-cc-perform (ContingencyCCPerformance, epistasis_func.cpp:137-203): selectCaseControl, computeMargins,
    then the margin-assisted case/control tables of every pair, batched on the device.  Every exact table sums to its
    group, so the checksum is pairs x (cases + controls)."""
    for name in ("cc_m24_n1000", "plain_m40_n70"):          # without and with missing calls
        cases, ctrls = read_tfam_status(name)
        m = len(read_tped(name))
        out = _run(name, "--contin-cc-perform")
        line = [ln for ln in out.splitlines() if ln.startswith(f"Case/Control Contingencies {m}:")][0]
        pairs = m * (m - 1) // 2
        assert f"({pairs} pairs, sum of all cells {pairs * (len(cases) + len(ctrls))})" in line


def test_epistasis_performance_harness(oracle):
    """--epi-perform (EpistasisPerformance, epistasis_func.cpp:307-347): the harness's own groups (even columns < 400
    cases, odd ones controls), the inline-mask case/control tables of every pair and the two-locus LLR test on each.
    The sum of the LLRs against the oracle's pairwise.c restatement on tables counted from the .tped text."""
    from text_pairs import joint_counts, row_classes
    for name in ("cc_m24_n1000", "cc_m6_n70", "plain_m40_n70"):
        n = int(name.split("_n")[1])
        lines = read_tped(name)
        m = len(lines)
        cls = [row_classes(ln) for ln in lines]
        cases = np.array([i << 1 for i in range(200) if (i << 1) < n], dtype=np.int64)
        ctrls = np.array([(i << 1) + 1 for i in range(200) if (i << 1) + 1 < n], dtype=np.int64)
        total, undefined = 0.0, 0
        for i in range(m):
            for j in range(i + 1, m):
                nine = [4 * a + b for a in range(3) for b in range(3)]
                cs = joint_counts(cls[i], cls[j], cases)[nine].reshape(3, 3)
                ct = joint_counts(cls[i], cls[j], ctrls)[nine].reshape(3, 3)
                ll = oracle.pairwise_llr(cs, ct)
                if np.isnan(ll):
                    undefined += 1
                else:
                    total += ll
        out = _run(name, "--epi-perform")
        line = [ln for ln in out.splitlines() if ln.startswith(f"Case/Control Contingencies {m}:")][0]
        got = line[line.index("(") + 1:line.index(")")].split(", ")
        assert got[0] == f"{m * (m - 1) // 2} pairs" and got[1] == f"{undefined} undefined"
        assert float(got[2].split()[-1]) == pytest.approx(total, rel=1e-8)


# ---- the drop-in table on tables generated in HBM (gwas_hip --simulate) ------------------------------------------
def _simulate(args, mode):
    return subprocess.run([CLI, "--simulate"] + [str(a) for a in args] + [mode], capture_output=True, text=True, timeout=600)


def test_inline_cc_maf_walk_over_a_simulated_table_equals_the_batch_scan():
    """gwas_hip --simulate N M --time-inline-cc-maf: the reference's harness loop (maf_func.cpp:292-316) walking the per-row
    getter of HipGenotypeTable over every row, window prefetch on and off, counts equal to one batch scan; one JSON line."""
    import json
    r = _simulate([10_000, 30_000, "--scan-window", "4096", "--missing-q16", "655"], "--time-inline-cc-maf")
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["walk_equals_batch_scan"] is True and d["n_snps"] == 30_000 and d["n_samples"] == 10_000
    assert d["snps_per_s_prefetch"] > 0 and d["snps_per_s_no_prefetch"] > 0 and d["kernel_snps_per_s"] > d["snps_per_s_prefetch"]
    assert d["kernel"].startswith("gw::k_scan_cc_sub<16, 5, true")


@pytest.mark.parametrize("extra", [[], ["--mask-check-every-call"]])
def test_case_control_set_edited_in_place_reaches_the_device(extra):
    """The reference's CaseControlSet has no change counter (case_control_set.h:48-91): an edit that keeps the object and
    both counts must still be noticed -- by the next window of rows and by selectCaseControl (every byte compared), or by
    the very next getter call with setMaskCheckEveryCall."""
    r = _simulate([3000, 500, "--scan-window", "64"] + extra, "--mask-edit-selftest")
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith(("PASS", "FAIL"))]
    assert lines and all(ln.startswith("PASS") for ln in lines), lines
    assert len(lines) == (2 if extra else 4)
