#!/usr/bin/env python3
"""Pairs/s of the BOOST / KSA pre-screen (BASELINE.json configs[4]-shaped:
all pairs of M SNPs, N samples) on one GPU.
    python tools/bench_boost.py --snps 100000 --samples 10000
(the CPU baseline and the parity check of this path live in `bench.py --workload c5`)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libgwaspp_amd as gw  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--snps", type=int, default=100_000)
ap.add_argument("--samples", type=int, default=10_000)
ap.add_argument("--missing-q16", type=int, default=0)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--valu", action="store_true", help="AND + popcount kernel instead of the int8 MFMA one")
ap.add_argument("--mode", type=int, default=1, help="GW_OPT_BOOST_MATRIX_CORES: 1 default, 2 both-groups kernels")
args = ap.parse_args()

m, n = args.snps, args.samples
t = gw.GwTable(n, m)
t.simulate(0x5EED, 0, args.missing_q16)
t.simulate_phenotype(0x5EED)
if args.valu:
    t.set_option(gw.capi.OPT_BOOST_MATRIX_CORES, 0)
elif args.mode != 1:
    t.set_option(gw.capi.OPT_BOOST_MATRIX_CORES, args.mode)
t0 = time.perf_counter()
t.boost_screen(0, min(m, 64), 0, min(m, 64))  # builds the compacted rows + margins
prep = time.perf_counter() - t0
best = None
for _ in range(args.reps):  # every pair's FP64 measure (extremes requested)
    t0 = time.perf_counter()
    hits, n_hits, n_pairs, vmax, vmin = t.boost_screen(0, m, 0, m, threshold=30.0, max_hits=1 << 20)
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
all_fp64 = best
best = None
for _ in range(args.reps):  # hits only: single-precision bound first (the default screen)
    t0 = time.perf_counter()
    hits2, n_hits2, n_pairs2, _, _ = t.boost_screen(0, m, 0, m, threshold=30.0, max_hits=1 << 20, extremes=False)
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
import numpy as np
same = n_hits2 == n_hits and np.array_equal(np.sort(hits, order=["snp_a", "snp_b"]), np.sort(hits2, order=["snp_a", "snp_b"]))
out = {"metric": "SNP pairs/s, BOOST KSA pre-screen", "value": n_pairs / best, "unit": "pairs/s",
       "n_snps": m, "n_samples": n, "pairs": n_pairs, "seconds": best, "seconds_every_measure_fp64": all_fp64,
       "same_hit_list": bool(same), "prep_seconds": prep,
       "hits_over_30": n_hits, "max_measure": vmax, "min_measure": vmin,
       "popcount_path": "4 corners + margins" if args.missing_q16 == 0 else "9 cells",
       "matrix_cores": bool(t.get_option(gw.capi.OPT_BOOST_MATRIX_CORES)),
       "and_popc_per_s": n_pairs / best * (n / 32.0) * (4 if args.missing_q16 == 0 else 9)}
print(json.dumps(out))
t.close()
