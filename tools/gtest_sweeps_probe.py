#!/usr/bin/env python3
"""How many proportional-fitting sweeps does BOOST stage 2 (k_gtest, computeGTest epistasis_func.cpp:508-704) take per pair on
config 5's hits?  Needs the variant build that returns the sweep count in place of the z value:
    tools/build_variant_file.sh sweeps gw_boost.hip -DGW_GTEST_COUNT_SWEEPS
    GW_LIBGWSCAN=libgwaspp_amd/exp/libgwscan_sweeps.so python3 tools/gtest_sweeps_probe.py
(round 4, MI355X: median 4, 90 % under 3 008, max 27 312 of 22 096 pairs -- the kernel's time is the slowest pair's chain)"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libgwaspp_amd as gw
from libgwaspp_amd import capi
n, m = 10000, 100000
t = gw.GwTable(n, m)
t.simulate(0x5EED, 0, 0, exact=True); t.simulate_phenotype(0x5EED)
hits, n_hits, n_pairs, *_ = t.boost_screen(0, m, 0, m, threshold=30.0, max_hits=1 << 18, extremes=False)
for rep in range(2):
    t0 = time.perf_counter(); meas, sw, uns = t.boost_gtest(hits); dt = time.perf_counter() - t0
print("pairs", len(hits), "gtest wall ms", dt * 1e3, "unsettled", uns)
sw = sw.astype(np.int64)
print("sweeps: median", int(np.median(sw)), "p90", int(np.percentile(sw, 90)), "p99", int(np.percentile(sw, 99)), "max", int(sw.max()), "sum", int(sw.sum()))
print("top 10:", np.sort(sw)[-10:])
