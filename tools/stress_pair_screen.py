#!/usr/bin/env python3
"""Randomised cross-check of the pair screen's kernel families on one GPU (not part of the test suite; run it after a change of the
FP4 tile code): random cohort sizes, SNP counts, missing-call rates, block calls at odd offsets and thresholds -- the FP4 kernels
(the default), the int8 kernels (GW_OPT_BOOST_FP4 = 0) and the vector-ALU kernels (GW_OPT_BOOST_MATRIX_CORES = 0) must list the same
pairs with bit-identical measures and the same pair tally, and the pre-filter's error model must hold (no trips).
    python3 tools/stress_pair_screen.py [--seconds 240] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libgwaspp_amd as gw  # noqa: E402
from libgwaspp_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=240.0)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t_end = time.time() + a.seconds
cases = calls = 0
order = ["snp_a", "snp_b"]
while time.time() < t_end:
    n = int(rng.choice([rng.integers(40, 400), rng.integers(400, 4000), rng.integers(4000, 30000)]))
    m = int(rng.choice([rng.integers(2, 70), rng.integers(70, 300), rng.integers(300, 900)]))
    missing = int(rng.choice([0, 0, 0, 300, 2000]))
    with gw.GwTable(n, m) as t:
        t.simulate(int(rng.integers(1, 1 << 30)), int(rng.integers(0, 1000)), missing, exact=bool(rng.integers(0, 2)))
        t.simulate_phenotype(int(rng.integers(1, 1 << 30)))
        cases += 1
        for _ in range(4):
            a0 = int(rng.integers(0, m)); na = int(rng.integers(1, m - a0 + 1))
            b0 = int(rng.integers(0, m)); nb = int(rng.integers(1, m - b0 + 1))
            if rng.integers(0, 3) == 0:
                a0, na, b0, nb = 0, m, 0, m
            thr = float(rng.choice([30.0, 25.0, 12.0, 3.0, -1.0e9]))
            got = {}
            for name, mfma, fp4 in (("fp4", 1, 1), ("int8", 1, 0), ("valu", 0, 0)):
                t.set_option(capi.OPT_BOOST_MATRIX_CORES, mfma)
                t.set_option(capi.OPT_BOOST_FP4, fp4)
                hits, n_hits, n_pairs, *_ = t.boost_screen(a0, na, b0, nb, threshold=thr, max_hits=1 << 20, extremes=False)
                got[name] = (np.sort(hits, order=order), n_hits, n_pairs, t.last_boost_kernel())
            ref = got["valu"]
            for name in ("fp4", "int8"):
                g = got[name]
                if g[1:3] != ref[1:3] or not np.array_equal(g[0], ref[0]):
                    print(f"MISMATCH {name} vs valu: n {n} m {m} missing {missing} block ({a0},{na},{b0},{nb}) thr {thr}: "
                          f"{g[1:]} vs {ref[1:]}", flush=True)
                    sys.exit(1)
            calls += 1
        trips = t.get_option(capi.OPT_BOOST_PREFILTER_TRIPS)
        if trips:
            print(f"pre-filter error model tripped {trips} times: n {n} m {m} missing {missing}", flush=True)
            sys.exit(1)
print(f"stress ok: {cases} tables, {calls} block calls x 3 kernel families, seed {a.seed}")
