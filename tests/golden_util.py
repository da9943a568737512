"""Readers for the reference-generated fixtures under tests/golden/perl
(written by the reference's scripts/perl/genotype_set_builder.pl; see
tests/golden/make_golden.sh)."""
import glob
import os

import numpy as np

PERL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "perl")


def golden_sets(kind):
    """names of the 'plain' or 'cc' sets, smallest first."""
    names = [os.path.basename(p)[:-5] for p in glob.glob(os.path.join(PERL_DIR, f"{kind}_*.tped"))]
    return sorted(names, key=lambda s: (int(s.split("_n")[1]), s))


def read_tped(name):
    with open(os.path.join(PERL_DIR, name + ".tped")) as f:
        return [ln for ln in f.read().splitlines() if ln.strip()]


def read_dist(name, suffix):
    """[[xx, aa, ab, bb], ...] with aa = count of 'A A', bb = count of 'C C'."""
    with open(os.path.join(PERL_DIR, f"{name}.expected.{suffix}")) as f:
        return np.array([[int(t) for t in ln.split()] for ln in f.read().splitlines() if ln.strip()],
                        dtype=np.int64)


def read_tfam_status(name):
    """column 6 of the .tfam: '1' = case, '0' = control
    (individual/tfam_annotation_file.cpp:68-77)."""
    cases, ctrls = [], []
    with open(os.path.join(PERL_DIR, name + ".tfam")) as f:
        for i, ln in enumerate(l for l in f.read().splitlines() if l.strip()):
            s = ln.split()[5][0]
            if s == "1":
                cases.append(i)
            elif s == "0":
                ctrls.append(i)
    return np.array(cases, dtype=np.int32), np.array(ctrls, dtype=np.int32)


def first_homozygote_is_cc(tped_line):
    """The reference labels code 1 ('aa') by the first homozygote met scanning
    the row left to right (quirk Q1, common_genotype.h:257-304); the generator's
    expected files label by allele (aa = 'A A').  True -> swap aa and bb."""
    al = tped_line.split()[4:]
    for i in range(0, len(al), 2):
        if al[i] == al[i + 1] and al[i] in "AC":
            return al[i] == "C"
    return False


def expected_in_table_space(expected_row, tped_line):
    """[xx, aa, ab, bb] by allele -> (aa, ab, bb, xx) in first-seen code space."""
    xx, aa, ab, bb = (int(v) for v in expected_row)
    if first_homozygote_is_cc(tped_line):
        aa, bb = bb, aa
    return (aa, ab, bb, xx)
