"""Pair tables counted straight from the .tped / .tfam TEXT, without the oracle and without any packed
row: the reference's golden-by-construction idea (scripts/perl/genotype_set_builder.pl:119-207 counts
genotypes while it writes them) extended from single SNPs to SNP pairs.  This is what pins the pair path
 independently of oracle/gw_oracle.c.

Conventions restated from the reference (not from the oracle):
  * class of a call: 0 = first homozygote met scanning the row left to right ("aa", code 1, quirk Q1:
    genotype/common_genotype.h:257-304), 1 = heterozygote, 2 = the other homozygote, 3 = missing
    (anything outside "ACGT" / "1234", quirk Q2: individual/tped_genotype_file.cpp:161-177);
  * CONTIN_TABLE_T is row-major 4 x 4, first marker down, second across
    (genotype/common_genotype.h:175-192): cell = 4 * class_A + class_B;
  * the stream table's xx_xx also holds every bit that is not a member of the group: non-members and
    the pad bits of the last 64-bit word (table5.cpp:749-800, 806-895; quirk Q7).
"""
import numpy as np

_DIGIT = {"1": "A", "2": "C", "3": "G", "4": "T"}


def row_classes(tped_line: str) -> np.ndarray:
    """uint8[n_samples] of classes 0..3 for one .tped line."""
    toks = tped_line.split()[4:]
    assert len(toks) % 2 == 0
    first_hom = None
    out = np.empty(len(toks) // 2, dtype=np.uint8)
    for s in range(len(toks) // 2):
        a = _DIGIT.get(toks[2 * s][0], toks[2 * s][0])
        b = _DIGIT.get(toks[2 * s + 1][0], toks[2 * s + 1][0])
        if a not in "ACGT" or b not in "ACGT":
            out[s] = 3
        elif a != b:
            out[s] = 1
        else:
            if first_hom is None:
                first_hom = a
            out[s] = 0 if a == first_hom else 2
    return out


def joint_counts(cls_a: np.ndarray, cls_b: np.ndarray, members=None) -> np.ndarray:
    """int64[16]: joint class counts of two SNPs over `members` (index array; None = everybody)."""
    if members is not None:
        cls_a, cls_b = cls_a[members], cls_b[members]
    return np.bincount(4 * cls_a.astype(np.int64) + cls_b.astype(np.int64), minlength=16)[:16]


def reference_words(n_samples: int) -> int:
    """64-bit words per bit-plane of the stream table: N/16+1 blocks rounded up to a multiple of 4
    (table5.cpp:52-60), four blocks per word."""
    return (((n_samples // 16 + 1) + 3) & ~3) // 4


def stream_table(cls_a, cls_b, n_samples: int, members=None) -> np.ndarray:
    """What CompressedGenotypeTable5::getContingencyTable / the inline-mask
    getCaseControlContingencyTable report: the joint counts with xx_xx absorbing pad bits and non-members."""
    t = joint_counts(cls_a, cls_b, members)
    group = n_samples if members is None else len(members)
    t[15] += reference_words(n_samples) * 64 - group
    return t
