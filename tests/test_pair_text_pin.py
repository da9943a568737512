"""Pins the ORACLE's pair tables to counts taken straight from the golden .tped/.tfam text
(tests/text_pairs.py), so that rows a9 / a10 of SURVEY.md 8 no longer rest on the oracle checking
itself: plain pair tables (table5.cpp:749-800) on the 'plain' sets (25 % missing calls), case/control
pair tables of both packings (table5.cpp:806-895, table3.cpp:733-849) on every set, and the
interaction measure's inputs.  CPU only."""
import numpy as np
import pytest

from golden_util import golden_sets, read_tfam_status, read_tped
from text_pairs import joint_counts, reference_words, row_classes, stream_table

ALL = golden_sets("cc") + golden_sets("plain")
NINE = [4 * a + b for a in range(3) for b in range(3)]


def _n_of(name):
    return int(name.split("_n")[1])


def _pairs(m, limit=400):
    p = [(i, j) for i in range(m) for j in range(i + 1, m)]
    return p if len(p) <= limit else p[:: len(p) // limit + 1]


def test_text_classes_reproduce_the_generators_single_snp_counts():
    """the text reader itself against the reference generator's expected files"""
    from golden_util import expected_in_table_space, read_dist
    for name in golden_sets("plain"):
        lines, exp = read_tped(name), read_dist(name, "dist")
        for r, ln in enumerate(lines):
            c = np.bincount(row_classes(ln), minlength=4)
            assert (int(c[0]), int(c[1]), int(c[2]), int(c[3])) == expected_in_table_space(exp[r], ln)
    for name in golden_sets("cc"):
        lines = read_tped(name)
        cases, ctrls = read_tfam_status(name)
        for grp, suffix in ((cases, "case.dist"), (ctrls, "control.dist")):
            exp = read_dist(name, suffix)
            for r, ln in enumerate(lines):
                c = np.bincount(row_classes(ln)[grp], minlength=4)
                assert (int(c[0]), int(c[1]), int(c[2]), int(c[3])) == expected_in_table_space(exp[r], ln)


@pytest.mark.parametrize("name", ALL)
def test_oracle_pair_tables_equal_text_counts(oracle, name):
    n = _n_of(name)
    lines = read_tped(name)
    cls = [row_classes(ln) for ln in lines]
    assert all(len(c) == n for c in cls)
    cases, ctrls = read_tfam_status(name)
    texts = [oracle.tped_genotype_text(ln) for ln in lines]
    gs, gb = oracle.geometry_stream(n), oracle.geometry_block(n)
    assert gs.words == reference_words(n)
    ts, tb = oracle.pack_rows(texts, n, "stream"), oracle.pack_rows(texts, n, "block")
    bm_c, sm_c, nc = oracle.ccs_masks(cases, n)
    bm_t, sm_t, nt = oracle.ccs_masks(ctrls, n)
    for i, j in _pairs(len(lines)):
        # getContingencyTable(i, j): all 16 cells
        assert np.array_equal(oracle.stream_contingency(ts[i], ts[j], gs), stream_table(cls[i], cls[j], n))
        # getCaseControlContingencyTable(i, j, ccs): all 16 cells per group
        sc, st = oracle.stream_cc_contingency(ts[i], ts[j], gs, sm_c, sm_t)
        assert np.array_equal(sc, stream_table(cls[i], cls[j], n, cases))
        assert np.array_equal(st, stream_table(cls[i], cls[j], n, ctrls))
        # the block table's case/control tables: nine cells, xx margins left at zero (table3.cpp:733-849)
        bc, bt = oracle.block_cc_contingency(tb[i], tb[j], gb, bm_c, bm_t)
        assert np.array_equal(bc[NINE], joint_counts(cls[i], cls[j], cases)[NINE])
        assert np.array_equal(bt[NINE], joint_counts(cls[i], cls[j], ctrls)[NINE])


def split_groups(name, n):
    """A seeded three-way split (case / control / neither) laid over a 'plain' golden set -- the reference generator's
    --case-control mode writes no missing call, its plain mode no groups; the masked scan WITH missing calls is pinned
    by putting groups on the plain sets' rows (25 % missing calls) and counting classes per group from the text."""
    rng = np.random.default_rng(sum(map(ord, name)))   # a seed that does not depend on the interpreter's hash salt
    status = rng.integers(0, 3, size=n)
    return np.flatnonzero(status == 1), np.flatnonzero(status == 0), np.flatnonzero(status == 2)


@pytest.mark.parametrize("name", golden_sets("plain"))
def test_oracle_masked_scan_with_missing_calls_equals_text_counts(oracle, name):
    """getCaseControlGenotypeDistribution (table5.cpp:609-657, table3.cpp:596-640) on rows WITH missing calls: per group
    aa / ab / bb from the .tped text, xx = group - aa - ab - bb (table5.cpp:649,653); and the three groups' counts add
    up, class for class, to the generator's own expected file for the row."""
    from golden_util import expected_in_table_space, read_dist
    n = _n_of(name)
    lines, exp = read_tped(name), read_dist(name, "dist")
    cases, ctrls, rest = split_groups(name, n)
    gs, gb = oracle.geometry_stream(n), oracle.geometry_block(n)
    texts = [oracle.tped_genotype_text(ln) for ln in lines]
    ts, tb = oracle.pack_rows(texts, n, "stream"), oracle.pack_rows(texts, n, "block")
    bm_c, sm_c, nc = oracle.ccs_masks(cases, n)
    bm_t, sm_t, nt = oracle.ccs_masks(ctrls, n)
    for r, ln in enumerate(lines):
        cls = row_classes(ln)
        per = [np.bincount(cls[g], minlength=4) for g in (cases, ctrls, rest)]
        total = per[0] + per[1] + per[2]
        assert (int(total[0]), int(total[1]), int(total[2]), int(total[3])) == expected_in_table_space(exp[r], ln)
        ca, co = oracle.stream_cc_dist(ts[r], gs, sm_c, sm_t, nc, nt)
        assert tuple(ca) == tuple(int(v) for v in per[0]) and tuple(co) == tuple(int(v) for v in per[1])
        ba, bo = oracle.block_cc_dist(tb[r], gb, bm_c, bm_t)       # the block table leaves xx at 0 (quirk Q3)
        assert tuple(ba)[:3] == tuple(int(v) for v in per[0][:3]) and tuple(bo)[:3] == tuple(int(v) for v in per[1][:3])
