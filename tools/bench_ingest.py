#!/usr/bin/env python3
"""Ingest rates of the data formats either side of the scan:
genotype text -> packed rows (host threads) -> HBM, and packed reference rows -> HBM."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import libgwaspp_amd as gw  # noqa: E402

n, m = 100_000, 2048
rng = np.random.default_rng(0)
lut = np.array(["00", "AA", "AC", "CC"])
texts = []
for r in range(64):                       # 64 distinct rows, re-used
    codes = rng.choice(4, size=n, p=[0.01, 0.49, 0.40, 0.10])
    codes[:3] = [1, 2, 3]
    texts.append(("\t".join(lut[codes]) + "\t").encode())
rows = [texts[r % 64] for r in range(m)]
out = {"n_samples": n, "rows": m, "genotypes": n * m}
try:
    cores = len(os.sched_getaffinity(0))
except AttributeError:
    cores = os.cpu_count() or 1
cores = min(cores, 16)
with gw.GwTable(n, m) as t:
    for threads in (1, cores):
        t0 = time.perf_counter()
        t.add_genotype_rows(0, rows, n_threads=threads)
        dt = time.perf_counter() - t0
        out[f"text_ingest_{threads}_threads_Mgenotypes_per_s"] = n * m / dt / 1e6
    stream = t.download_rows(0, m, gw.LAYOUT_STREAM)
    block = t.download_rows(0, m, gw.LAYOUT_BLOCK)
big_s, big_b = np.tile(stream, (16, 1)), np.tile(block, (16, 1))
with gw.GwTable(n, big_s.shape[0]) as t:
    for name, arr, lay in (("stream", big_s, gw.LAYOUT_STREAM), ("block", big_b, gw.LAYOUT_BLOCK)):
        t.upload_rows(0, arr[:64], lay)
        t0 = time.perf_counter()
        t.upload_rows(0, arr, lay)
        dt = time.perf_counter() - t0
        out[f"upload_{name}_rows_GBps"] = arr.nbytes / dt / 1e9
        t0 = time.perf_counter()
        back = t.download_rows(0, arr.shape[0], lay)
        out[f"download_{name}_rows_GBps"] = arr.nbytes / (time.perf_counter() - t0) / 1e9
        assert np.array_equal(back, arr)
out["note"] = ("text = 'AC<tab>' triplets as TpedGenotypeFile hands them to addGenotypeRow; reference measured at "
               "56-108 M genotypes/s on one core (BASELINE.md section 2); packed rows from pageable numpy memory")
print(json.dumps(out))
