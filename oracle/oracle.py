"""ctypes view of the CPU oracle (oracle/liboracle.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  Nothing under libgwaspp_amd/ does.  See oracle/gw_oracle.h for
what each function restates (reference file:line) and its pinning status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libref_ccs.so")

N_STATS = 8


class Geom(C.Structure):
    _fields_ = [("n_samples", C.c_uint32), ("data_blocks", C.c_uint32),
                ("blocks_per_row", C.c_uint32), ("words", C.c_uint32)]


class Freq(C.Structure):
    _fields_ = [("aa", C.c_uint32), ("ab", C.c_uint32), ("bb", C.c_uint32), ("xx", C.c_uint32)]

    def tuple(self):
        return (self.aa, self.ab, self.bb, self.xx)


class Marginal(C.Structure):
    _fields_ = [("margins", Freq), ("cases", Freq), ("controls", Freq),
                ("marginal_entropy", C.c_double), ("marginal_entropy_y", C.c_double),
                ("pbc", C.c_double * 8), ("pca", C.c_double * 8)]


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "gw_oracle.c"))):
        subprocess.run(["make", "-C", _HERE, "-s", "all"], check=True)
    elif os.path.isdir("/root/reference") and not os.path.exists(_REF_PATH):
        subprocess.run(["make", "-C", _HERE, "-s", "ref"], check=True)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u16p, u32p, f64p = C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), C.POINTER(C.c_double)
        gp, fp = C.POINTER(Geom), C.POINTER(Freq)
        L.gwo_geometry_stream.argtypes = [C.c_uint32, gp]
        L.gwo_geometry_block.argtypes = [C.c_uint32, gp]
        L.gwo_popcount64.argtypes = [C.c_uint64]
        L.gwo_popcount64.restype = C.c_int
        for f in (L.gwo_pack_row_stream, L.gwo_pack_row_block):
            f.argtypes = [C.c_char_p, C.c_char_p, gp, u16p]
            f.restype = C.c_int
        L.gwo_parse_header.argtypes = [C.c_uint16, u16p]
        for f in (L.gwo_stream_code_at, L.gwo_block_code_at):
            f.argtypes = [u16p, gp, C.c_uint32]
            f.restype = C.c_int
        L.gwo_ccs_set.argtypes = [C.POINTER(C.c_int32), C.c_size_t, C.c_uint32, u16p, u16p]
        L.gwo_ccs_set.restype = C.c_int
        L.gwo_stream_dist.argtypes = [u16p, gp, fp]
        L.gwo_stream_cc_dist.argtypes = [u16p, gp, u16p, u16p, C.c_uint32, C.c_uint32, fp, fp]
        L.gwo_compact_geometry.argtypes = [C.c_uint32, C.c_uint32, u32p, u32p]
        L.gwo_stream_compact_row.argtypes = [u16p, gp, u16p, u16p, C.c_uint32, C.c_uint32, u16p]
        L.gwo_stream_cc_dist_compacted.argtypes = [u16p, C.c_uint32, C.c_uint32, C.c_uint32,
                                                   C.c_uint32, fp, fp]
        L.gwo_block_dist.argtypes = [u16p, gp, fp]
        L.gwo_block_cc_dist.argtypes = [u16p, gp, u16p, u16p, fp, fp]
        L.gwo_stream_contingency.argtypes = [u16p, u16p, gp, u32p]
        L.gwo_stream_cc_contingency.argtypes = [u16p, u16p, gp, u16p, u16p, u32p, u32p]
        L.gwo_block_cc_contingency.argtypes = [u16p, u16p, gp, u16p, u16p, u32p, u32p]
        L.gwo_minor_allele_frequency.argtypes = [fp, f64p, f64p]
        L.gwo_marginal_information.argtypes = [fp, fp, C.c_uint32, C.POINTER(Marginal)]
        L.gwo_marginal_information.restype = C.c_int
        L.gwo_assoc_stats.argtypes = [fp, fp, f64p]
        L.gwo_scan_stream_cc.argtypes = [u16p, C.c_uint32, gp, u16p, u16p, C.c_uint32, C.c_uint32,
                                         u32p, f64p]
        L.gwo_scan_block_cc.argtypes = L.gwo_scan_stream_cc.argtypes
        L.gwo_scan_stream_cc_mt.argtypes = L.gwo_scan_stream_cc.argtypes + [C.c_int]
        L.gwo_sim_row_stream.argtypes = [C.c_uint64, C.c_uint64, gp, C.c_uint32, u16p]
        L.gwo_sim_row_stream_exact.argtypes = [C.c_uint64, C.c_uint64, gp, C.c_uint32, u16p]
        L.gwo_sim_phenotype_stream.argtypes = [C.c_uint64, gp, u16p, u16p, u32p, u32p]
        L.gwo_sim_snp_p_e5.argtypes = [C.c_uint64, C.c_uint64]
        L.gwo_sim_snp_p_e5.restype = C.c_uint32
        L.gwo_boost_measure.argtypes = [u32p, u32p, C.POINTER(Marginal), C.POINTER(Marginal), C.c_uint32]
        L.gwo_boost_measure.restype = C.c_double
        L.gwo_boost_all_pairs.argtypes = [u16p, C.c_uint32, gp, u16p, u16p, C.c_uint32, C.c_uint32, f64p]
        L.gwo_gtest.argtypes = [u32p, u32p, C.POINTER(Marginal), C.POINTER(Marginal), C.c_uint32, f64p, f64p]
        L.gwo_pairwise_llr.argtypes = [u32p, u32p]
        L.gwo_pairwise_llr.restype = C.c_double
        _lib = L
    return _lib


def _p16(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint16))


def _p32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _pf64(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


# ------------------------------------------------------------------ geometry
def geometry_stream(n: int) -> Geom:
    g = Geom()
    lib().gwo_geometry_stream(n, C.byref(g))
    return g


def geometry_block(n: int) -> Geom:
    g = Geom()
    lib().gwo_geometry_block(n, C.byref(g))
    return g


# -------------------------------------------------------------------- ingest
def tped_genotype_text(line: str, delim: str = "\t") -> bytes:
    """What TpedGenotypeFile::parseNextGenotypeRecord hands to addGenotypeRow
    (individual/tped_genotype_file.cpp:127-190): skip four columns, drop the
    blank between the two alleles, map 1..4 -> ACGT, put `delim` after each
    pair."""
    toks = line.strip().split()
    alleles = toks[4:]
    assert len(alleles) % 2 == 0
    digit = {"1": "A", "2": "C", "3": "G", "4": "T"}
    out = []
    for i in range(0, len(alleles), 2):
        a, b = alleles[i][0], alleles[i + 1][0]
        out.append(digit.get(a, a) + digit.get(b, b) + delim)
    return "".join(out).encode()


def pack_rows(texts, n: int, layout: str = "stream") -> np.ndarray:
    """Packs genotype text rows into a reference-layout table
    uint16[len(texts), blocks_per_row]."""
    g = geometry_stream(n) if layout == "stream" else geometry_block(n)
    f = lib().gwo_pack_row_stream if layout == "stream" else lib().gwo_pack_row_block
    table = np.zeros((len(texts), g.blocks_per_row), dtype=np.uint16)
    for r, t in enumerate(texts):
        buf = C.create_string_buffer(t, len(t))
        base = C.addressof(buf)
        rc = f(C.cast(base, C.c_char_p), C.cast(base + len(t), C.c_char_p), C.byref(g), _p16(table[r]))
        if rc != 0:
            raise ValueError(f"row {r}: the reference would assert on this genotype row")
    return table


def codes_to_text(codes: np.ndarray, delim: str = "\t") -> bytes:
    """codes (0 missing, 1 'AA', 2 'AC', 3 'CC' BY ALLELE) -> genotype text."""
    lut = np.array(["00", "AA", "AC", "CC"])
    return (delim.join(lut[codes]) + delim).encode()


def pack_codes(codes: np.ndarray, layout: str = "stream") -> np.ndarray:
    """Packs a [M, N] array of 2-bit codes (already in the table's own code
    space: 1 = first homozygote) straight into the reference memory layout,
    header 0x7015.  Vectorised helper for parity inputs; its output is checked
    against gwo_pack_row_* in tests/test_oracle_golden.py."""
    codes = np.asarray(codes, dtype=np.uint8)
    m, n = codes.shape
    g = geometry_stream(n) if layout == "stream" else geometry_block(n)
    table = np.zeros((m, g.blocks_per_row), dtype=np.uint16)
    table[:, 0] = 0x7015
    if layout == "stream":
        nb = g.data_blocks
        pad = nb * 16 - n
        c = np.pad(codes, ((0, 0), (0, pad)))
        lo = (c & 1).reshape(m, nb, 16).astype(np.uint32)
        hi = ((c >> 1) & 1).reshape(m, nb, 16).astype(np.uint32)
        w = (1 << np.arange(16, dtype=np.uint32))
        table[:, 1:1 + nb] = (lo * w).sum(axis=2).astype(np.uint16)
        table[:, 1 + nb:1 + 2 * nb] = (hi * w).sum(axis=2).astype(np.uint16)
    else:
        nb = g.data_blocks
        pad = nb * 8 - n
        c = np.pad(codes, ((0, 0), (0, pad))).reshape(m, nb, 8).astype(np.uint32)
        w = (1 << (2 * np.arange(8, dtype=np.uint32)))
        table[:, 1:1 + nb] = (c * w).sum(axis=2).astype(np.uint16)
    return table


# --------------------------------------------------------------------- masks
def ccs_masks(idx, n: int):
    """(block_mask uint16[], stream_mask uint16[], count) for a column set."""
    idx = np.ascontiguousarray(np.asarray(idx, dtype=np.int32))
    gb, gs = geometry_block(n), geometry_stream(n)
    bm = np.zeros(gb.data_blocks, dtype=np.uint16)
    sm = np.zeros(gs.data_blocks, dtype=np.uint16)
    cnt = lib().gwo_ccs_set(idx.ctypes.data_as(C.POINTER(C.c_int32)), idx.size, n, _p16(bm), _p16(sm))
    if cnt < 0:
        raise ValueError("case/control index out of range")
    return bm, sm, cnt


# ------------------------------------------------------------- distributions
def stream_dist(row, g):
    f = Freq()
    lib().gwo_stream_dist(_p16(row), C.byref(g), C.byref(f))
    return f.tuple()


def block_dist(row, g):
    f = Freq()
    lib().gwo_block_dist(_p16(row), C.byref(g), C.byref(f))
    return f.tuple()


def stream_cc_dist(row, g, case_stream, ctrl_stream, case_count, ctrl_count):
    a, b = Freq(), Freq()
    lib().gwo_stream_cc_dist(_p16(row), C.byref(g), _p16(case_stream), _p16(ctrl_stream),
                             case_count, ctrl_count, C.byref(a), C.byref(b))
    return a.tuple(), b.tuple()


def block_cc_dist(row, g, case_block, ctrl_block):
    a, b = Freq(), Freq()
    lib().gwo_block_cc_dist(_p16(row), C.byref(g), _p16(case_block), _p16(ctrl_block),
                            C.byref(a), C.byref(b))
    return a.tuple(), b.tuple()


def stream_cc_dist_selected(row, g, case_stream, ctrl_stream, case_count, ctrl_count):
    """selectCaseControl + mask-free per-row variant (table5.cpp:443-575, 659-701)."""
    cb, tb = C.c_uint32(), C.c_uint32()
    lib().gwo_compact_geometry(case_count, ctrl_count, C.byref(cb), C.byref(tb))
    crow = np.zeros(2 * (cb.value + tb.value), dtype=np.uint16)
    lib().gwo_stream_compact_row(_p16(row), C.byref(g), _p16(case_stream), _p16(ctrl_stream),
                                 cb.value, tb.value, _p16(crow))
    a, b = Freq(), Freq()
    lib().gwo_stream_cc_dist_compacted(_p16(crow), cb.value, tb.value, case_count, ctrl_count,
                                       C.byref(a), C.byref(b))
    return a.tuple(), b.tuple()


def stream_contingency(row_a, row_b, g):
    out = np.zeros(16, dtype=np.uint32)
    lib().gwo_stream_contingency(_p16(row_a), _p16(row_b), C.byref(g), _p32(out))
    return out


def stream_cc_contingency(row_a, row_b, g, case_stream, ctrl_stream):
    a, b = np.zeros(16, dtype=np.uint32), np.zeros(16, dtype=np.uint32)
    lib().gwo_stream_cc_contingency(_p16(row_a), _p16(row_b), C.byref(g), _p16(case_stream),
                                    _p16(ctrl_stream), _p32(a), _p32(b))
    return a, b


def block_cc_contingency(row_a, row_b, g, case_block, ctrl_block):
    a, b = np.zeros(16, dtype=np.uint32), np.zeros(16, dtype=np.uint32)
    lib().gwo_block_cc_contingency(_p16(row_a), _p16(row_b), C.byref(g), _p16(case_block), _p16(ctrl_block),
                                   _p32(a), _p32(b))
    return a, b


# ---------------------------------------------------------------- arithmetic
def minor_allele_frequency(ft):
    f = Freq(*ft)
    tot, maf = C.c_double(), C.c_double()
    lib().gwo_minor_allele_frequency(C.byref(f), C.byref(tot), C.byref(maf))
    return tot.value, maf.value


def marginal_information(cases, ctrls, n_individs):
    m = Marginal()
    rc = lib().gwo_marginal_information(C.byref(Freq(*cases)), C.byref(Freq(*ctrls)), n_individs,
                                        C.byref(m))
    if rc != 0:
        raise ValueError("nCases + nControls != nIndivids (reference assert)")
    return m


def assoc_stats(cases, ctrls) -> np.ndarray:
    out = np.zeros(N_STATS, dtype=np.float64)
    lib().gwo_assoc_stats(C.byref(Freq(*cases)), C.byref(Freq(*ctrls)), _pf64(out))
    return out


def boost_measure(case16, ctrl16, m1: "Marginal", m2: "Marginal", n_individs: int) -> float:
    a = np.ascontiguousarray(np.asarray(case16, dtype=np.uint32).reshape(16))
    b = np.ascontiguousarray(np.asarray(ctrl16, dtype=np.uint32).reshape(16))
    return lib().gwo_boost_measure(_p32(a), _p32(b), C.byref(m1), C.byref(m2), n_individs)


def boost_all_pairs(table, g, case_stream, ctrl_stream, case_count, ctrl_count) -> np.ndarray:
    table = np.ascontiguousarray(table, dtype=np.uint16)
    m = table.shape[0]
    out = np.zeros(m * (m - 1) // 2, dtype=np.float64)
    lib().gwo_boost_all_pairs(_p16(table), m, C.byref(g), _p16(case_stream), _p16(ctrl_stream),
                              case_count, ctrl_count, _pf64(out))
    return out


def gtest(case16, ctrl16, m1: "Marginal", m2: "Marginal", n_individs: int):
    """-> (exact measure after IPF, allelic odds-ratio z)"""
    a = np.ascontiguousarray(np.asarray(case16, dtype=np.uint32).reshape(16))
    b = np.ascontiguousarray(np.asarray(ctrl16, dtype=np.uint32).reshape(16))
    me, z = C.c_double(), C.c_double()
    lib().gwo_gtest(_p32(a), _p32(b), C.byref(m1), C.byref(m2), n_individs, C.byref(me), C.byref(z))
    return me.value, z.value


def pairwise_llr(case_tab, ctrl_tab) -> float:
    a = np.ascontiguousarray(np.asarray(case_tab, dtype=np.uint32).reshape(9))
    b = np.ascontiguousarray(np.asarray(ctrl_tab, dtype=np.uint32).reshape(9))
    return lib().gwo_pairwise_llr(_p32(a), _p32(b))


# --------------------------------------------------------------------- scans
def scan_stream_cc(table, g, case_stream, ctrl_stream, case_count, ctrl_count, stats=True,
                   threads: int = 0):
    """counts uint32[M,8] (+ stats float64[M,8]) over a reference-layout table.
    threads=0: the faithful single-thread LUT loop; >0: the best-effort
    hardware-popcount pthread variant."""
    table = np.ascontiguousarray(table, dtype=np.uint16)
    m = table.shape[0]
    assert table.shape[1] == g.blocks_per_row
    counts = np.zeros((m, 8), dtype=np.uint32)
    st = np.zeros((m, N_STATS), dtype=np.float64) if stats else None
    sp = _pf64(st) if stats else None
    if threads > 0:
        lib().gwo_scan_stream_cc_mt(_p16(table), m, C.byref(g), _p16(case_stream), _p16(ctrl_stream),
                                    case_count, ctrl_count, _p32(counts), sp, threads)
    else:
        lib().gwo_scan_stream_cc(_p16(table), m, C.byref(g), _p16(case_stream), _p16(ctrl_stream),
                                 case_count, ctrl_count, _p32(counts), sp)
    return counts, st


def scan_block_cc(table, g, case_block, ctrl_block, case_count, ctrl_count, stats=True):
    table = np.ascontiguousarray(table, dtype=np.uint16)
    m = table.shape[0]
    assert table.shape[1] == g.blocks_per_row
    counts = np.zeros((m, 8), dtype=np.uint32)
    st = np.zeros((m, N_STATS), dtype=np.float64) if stats else None
    lib().gwo_scan_block_cc(_p16(table), m, C.byref(g), _p16(case_block), _p16(ctrl_block),
                            case_count, ctrl_count, _p32(counts), _pf64(st) if stats else None)
    return counts, st


# ----------------------------------------------------------------- simulator
def sim_table_stream(seed: int, snp_begin: int, n_snps: int, n: int, missing_q16: int = 0, exact: bool = False):
    """exact=True: exactly floor(p * 2N) minor alleles per SNP (simulate_data.cpp:191-205) instead of binomial sampling"""
    g = geometry_stream(n)
    table = np.zeros((n_snps, g.blocks_per_row), dtype=np.uint16)
    for r in range(n_snps):
        if exact:
            if lib().gwo_sim_row_stream_exact(seed, snp_begin + r, C.byref(g), missing_q16, _p16(table[r])) != 0:
                raise MemoryError("gwo_sim_row_stream_exact")
        else:
            lib().gwo_sim_row_stream(seed, snp_begin + r, C.byref(g), missing_q16, _p16(table[r]))
    return table, g


def sim_phenotype_stream(seed: int, n: int):
    g = geometry_stream(n)
    cs = np.zeros(g.data_blocks, dtype=np.uint16)
    ct = np.zeros(g.data_blocks, dtype=np.uint16)
    nc, nt = C.c_uint32(), C.c_uint32()
    lib().gwo_sim_phenotype_stream(seed, C.byref(g), _p16(cs), _p16(ct), C.byref(nc), C.byref(nt))
    return cs, ct, nc.value, nt.value


def sim_snp_p_e5(seed: int, snp: int) -> int:
    return lib().gwo_sim_snp_p_e5(seed, snp)


# ------------------------------------------------ real reference (oracle/_ref)
def have_ref() -> bool:
    return os.path.exists(_REF_PATH)


def ref_ccs(n: int, cases, ctrls):
    """Masks produced by the REAL reference CaseControlSet (oracle/_ref)."""
    R = C.CDLL(_REF_PATH)
    i32p, u16p, u32p = C.POINTER(C.c_int32), C.POINTER(C.c_uint16), C.POINTER(C.c_uint32)
    R.ref_ccs_build.argtypes = [C.c_int, i32p, C.c_int, i32p, C.c_int, u16p, u16p, u16p, u16p, u32p]
    cases = np.ascontiguousarray(cases, dtype=np.int32)
    ctrls = np.ascontiguousarray(ctrls, dtype=np.int32)
    gb, gs = geometry_block(n), geometry_stream(n)
    bc = np.zeros(gb.data_blocks, np.uint16)
    bt = np.zeros(gb.data_blocks, np.uint16)
    sc = np.zeros(gs.data_blocks, np.uint16)
    st = np.zeros(gs.data_blocks, np.uint16)
    counts = np.zeros(3, np.uint32)
    R.ref_ccs_build(n, cases.ctypes.data_as(i32p), cases.size, ctrls.ctypes.data_as(i32p),
                    ctrls.size, _p16(bc), _p16(bt), _p16(sc), _p16(st), _p32(counts))
    return bc, bt, sc, st, counts
