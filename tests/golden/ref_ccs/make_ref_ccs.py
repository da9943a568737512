#!/usr/bin/env python3
"""Writes tests/golden/ref_ccs/masks.npz: phenotype masks produced by the REAL reference CaseControlSet.

Container only (needs /root/reference).  `make -C oracle ref` compiles the reference's own, unmodified
analyzable/case_control_set.cpp + util/index_set/index_set.cpp in place into oracle/_ref/libref_ccs.so (driver:
oracle/ref_ccs_shim.cpp); this script calls setCases / setControls (case_control_set.cpp:77-150) on that object for
a list of cohort sizes and column sets and stores, per case,
    <key>_cases, <key>_ctrls                      int32 column indices handed in (may hold duplicates)
    <key>_block_case, <key>_block_ctrl            uint16, case_begin() / control_begin()   (0b11 per member)
    <key>_stream_case, <key>_stream_ctrl          uint16, stream_case_begin() / stream_control_begin()
    <key>_counts                                  uint32 getCaseCount, getControlCount, getTotalCount
The file is data (inputs and the reference's outputs); nothing of the reference itself travels with it.  The column
sets come from numpy's seeded generator, so the file is reproducible bit for bit.
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)

SIZES = [1, 7, 8, 15, 16, 17, 63, 64, 65, 70, 127, 128, 129, 1000, 4097, 10_000]
SCAN_SHAPES = [(70, 20), (1000, 64), (4097, 33), (10_000, 200)]   # tests/test_gpu_parity.py


def status_sets(seed, n):
    rng = np.random.default_rng(seed)
    status = rng.integers(0, 3, size=n)  # 0 control, 1 case, 2 neither
    return np.flatnonzero(status == 1).astype(np.int32), np.flatnonzero(status == 0).astype(np.int32)


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("the reference tree is not on this machine")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "all"], check=True)
    from oracle import oracle
    assert oracle.have_ref()
    out = {}

    def add(key, n, cases, ctrls):
        bc, bt, sc, st, counts = oracle.ref_ccs(n, cases, ctrls)
        out[key + "_n"] = np.array([n], np.int64)
        out[key + "_cases"], out[key + "_ctrls"] = cases, ctrls
        out[key + "_block_case"], out[key + "_block_ctrl"] = bc, bt
        out[key + "_stream_case"], out[key + "_stream_ctrl"] = sc, st
        out[key + "_counts"] = counts

    for n in SIZES:
        add(f"size{n}", n, *status_sets(n, n))
    for n, m in SCAN_SHAPES:
        add(f"scan{n}x{m}", n, *status_sets(3 * n + m, n))
    add("dup100", 100, np.array([3, 3, 5, 99, 5], np.int32), np.zeros(0, np.int32))
    add("all_cases300", 300, np.arange(300, dtype=np.int32), np.zeros(0, np.int32))
    add("overlap64", 64, np.arange(0, 40, dtype=np.int32), np.arange(30, 64, dtype=np.int32))
    path = os.path.join(HERE, "masks.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out) // 8} mask sets, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
