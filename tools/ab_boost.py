#!/usr/bin/env python3
"""Same-process timing of the config-5 pair screen (all pairs of M SNPs x N samples, exact-count input as bench.py's, hits
to a device list, HIP events around the launch) under one or several settings of ONE option, alternating, plus the
identity of the hit lists.  (Round 3's tools/ab_stagger.py, which swept the retired GW_OPT_BOOST_STAGGER_US.)
    python tools/ab_boost.py --snps 100000 --samples 10000                       # the default kernel, 3 rounds
    python tools/ab_boost.py --option first_level --values 0,1
    GW_LIBGWSCAN=libgwaspp_amd/exp/libgwscan_x.so python tools/ab_boost.py       # a variant build (tools/build_variant_file.sh)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import libgwaspp_amd as gw  # noqa: E402
from libgwaspp_amd import capi  # noqa: E402

OPTIONS = {"first_level": capi.OPT_BOOST_FIRST_LEVEL, "fp4": capi.OPT_BOOST_FP4, "prefilter": capi.OPT_BOOST_PREFILTER,
           "matrix_cores": capi.OPT_BOOST_MATRIX_CORES, "candidates": capi.OPT_BOOST_CANDIDATES}
ap = argparse.ArgumentParser()
ap.add_argument("--snps", type=int, default=100_000)
ap.add_argument("--samples", type=int, default=10_000)
ap.add_argument("--missing-q16", type=int, default=0)
ap.add_argument("--threshold", type=float, default=30.0)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--option", choices=sorted(OPTIONS), default=None)
ap.add_argument("--values", default="")
ap.add_argument("--fp4", type=int, default=1, help="GW_OPT_BOOST_FP4 for the whole run")
args = ap.parse_args()
m, n = args.snps, args.samples
t = gw.GwTable(n, m)
t.simulate(0x5EED, 0, args.missing_q16, exact=True)
t.simulate_phenotype(0x5EED)
t.set_option(capi.OPT_BOOST_FP4, args.fp4)
t.boost_screen(0, min(m, 64), 0, min(m, 64))
cap = 1 << 18
lst = capi.DeviceBuffer(capi.BOOST_LIST_HEADER_BYTES + 16 * cap)
s = t.stream()
values = [int(x) for x in args.values.split(",")] if args.option else [None]
res, lists, names = {}, {}, {}
for rnd in range(args.rounds):
    for v in values:
        if args.option:
            t.set_option(OPTIONS[args.option], v)
        t.boost_list_init(lst.ptr, s)
        t.timer_start(s)
        t.boost_screen_device(0, m, 0, m, lst.ptr, cap, args.threshold, s)
        ms = t.timer_stop(s)
        res.setdefault(v, []).append(ms)
        hits, n_hits, n_pairs, aux0 = gw.parse_boost_list(lst.download(s), cap)
        lists[v] = (np.sort(hits, order=["snp_a", "snp_b"]), n_hits, n_pairs, aux0)
        names[v] = t.last_boost_kernel()
base = lists[values[0]]
ops = base[2] * 4.0 * n * 2.0 * (1 if args.missing_q16 == 0 else 9 / 4)
for v in values:
    mn = min(res[v])
    same = bool(np.array_equal(lists[v][0], base[0]) and lists[v][1:3] == base[1:3])
    label = f"{args.option} {v}" if args.option else "default"
    print(f"{label:>16} | min {mn:8.3f} ms | median {float(np.median(res[v])):8.3f} ms | {base[2] / mn / 1e6:7.2f} G pairs/s | "
          f"{ops / (mn * 1e-3) / 1e15:5.2f} POP/s = {ops / (mn * 1e-3) / 1e16 * 100:5.1f} % of the FP4 peak | hits {lists[v][1]} "
          f"outside the error model / candidates lost {lists[v][3]} same list {same} | {names[v]}", flush=True)
