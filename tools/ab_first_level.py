#!/usr/bin/env python3
"""Same-process A/B of GW_OPT_BOOST_FIRST_LEVEL on the config-5 pair screen (all pairs of M SNPs x N samples, exact-count
input as bench.py's, hits to a device list, HIP events), plus the identity of the hit lists and the run-time check tally.
    python tools/ab_first_level.py --snps 100000 --samples 10000
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import libgwaspp_amd as gw  # noqa: E402
from libgwaspp_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--snps", type=int, default=100_000)
ap.add_argument("--samples", type=int, default=10_000)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--threshold", type=float, default=30.0)
args = ap.parse_args()
m, n = args.snps, args.samples
t = gw.GwTable(n, m)
t.simulate(0x5EED, 0, 0, exact=True)
t.simulate_phenotype(0x5EED)
t.boost_screen(0, min(m, 64), 0, min(m, 64))
cap = 1 << 18
lst = capi.DeviceBuffer(capi.BOOST_LIST_HEADER_BYTES + 16 * cap)
s = t.stream()
res, lists, names = {}, {}, {}
for rnd in range(args.rounds):
    for first in (0, 1):
        t.set_option(capi.OPT_BOOST_FIRST_LEVEL, first)
        t.boost_list_init(lst.ptr, s)
        t.timer_start(s)
        t.boost_screen_device(0, m, 0, m, lst.ptr, cap, args.threshold, s)
        ms = t.timer_stop(s)
        res.setdefault(first, []).append(ms)
        hits, n_hits, n_pairs, aux0 = gw.parse_boost_list(lst.download(s), cap)
        lists[first] = (np.sort(hits, order=["snp_a", "snp_b"]), n_hits, n_pairs, aux0)
        names[first] = t.last_boost_kernel()
for first, v in sorted(res.items()):
    mn = min(v)
    same = bool(np.array_equal(lists[first][0], lists[0][0]) and lists[first][1:] == lists[0][1:])
    print(f"first level {first} | min {mn:8.3f} ms | median {float(np.median(v)):8.3f} ms | {lists[first][2] / mn / 1e6:7.2f} G pairs/s | "
          f"hits {lists[first][1]} outside the error model {lists[first][3]} same list {same} | {names[first]}", flush=True)
