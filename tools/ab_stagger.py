#!/usr/bin/env python3
"""Same-process sweep of GW_OPT_BOOST_STAGGER_US on the config-5 pair screen (all pairs of M SNPs x N samples, hits to
a device list, HIP events), plus the identity of the hit lists.
    python tools/ab_stagger.py --snps 100000 --samples 10000 --us 0,20,35,50,65,80
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
ap.add_argument("--missing-q16", type=int, default=0)
ap.add_argument("--us", default="0,20,35,50,65,80")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--lut", type=int, default=0, help="GW_OPT_BOOST_LUT")
ap.add_argument("--fp4", type=int, default=1, help="GW_OPT_BOOST_FP4")
args = ap.parse_args()
m, n = args.snps, args.samples
t = gw.GwTable(n, m)
t.simulate(0x5EED, 0, args.missing_q16)
t.simulate_phenotype(0x5EED)
t.set_option(capi.OPT_BOOST_LUT, args.lut)
t.set_option(capi.OPT_BOOST_FP4, args.fp4)
t.boost_screen(0, min(m, 64), 0, min(m, 64))
cap = 1 << 18
lst = capi.DeviceBuffer(capi.BOOST_LIST_HEADER_BYTES + 16 * cap)
s = t.stream()
res, lists = {}, {}
for rnd in range(args.rounds):
    for us in [int(x) for x in args.us.split(",")]:
        t.set_option(capi.OPT_BOOST_STAGGER_US, us)
        t.boost_list_init(lst.ptr, s)
        t.timer_start(s)
        t.boost_screen_device(0, m, 0, m, lst.ptr, cap, 30.0, s)
        ms = t.timer_stop(s)
        res.setdefault(us, []).append(ms)
        hits, n_hits, n_pairs, aux0 = gw.parse_boost_list(lst.download(s), cap)
        lists[us] = (np.sort(hits, order=["snp_a", "snp_b"]), n_hits, n_pairs, aux0)
base = lists[min(lists)]
macs = base[2] * 4.0 * n
for us, v in sorted(res.items()):
    mn = min(v)
    same = bool(np.array_equal(lists[us][0], base[0]) and lists[us][1:] == base[1:])
    print(f"stagger {us:4d} us | min {mn:8.3f} ms | median {float(np.median(v)):8.3f} ms | {base[2] / mn / 1e6:7.2f} G pairs/s | "
          f"{2 * macs / (mn * 1e-3) / 1e12 / 5000 * 100:5.1f} % of int8 peak | hits {lists[us][1]} same list {same} | {t.last_boost_kernel()}",
          flush=True)
