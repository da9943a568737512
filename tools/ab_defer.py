#!/usr/bin/env python3
"""Same-process A/B of GW_OPT_DEFER_RECORDS (records written in one burst at the end of the launch vs between the
reads), and a bit-for-bit comparison of the two forms' records.
    python tools/ab_defer.py --samples 2000,10000,20000 --snps 1000000 --rounds 5 --reps 10
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import libgwaspp_amd as gw  # noqa: E402
from libgwaspp_amd import capi  # noqa: E402
from bench import SEED  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--samples", default="2000,10000,20000")
ap.add_argument("--snps", type=int, default=1_000_000)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--missing", type=int, default=0)
args = ap.parse_args()

for n in [int(x) for x in args.samples.split(",")]:
    m = args.snps
    t = gw.GwTable(n, m)
    t.simulate(SEED, 0, args.missing)
    t.simulate_phenotype(SEED)
    bufs = [torch.zeros(m * 96, dtype=torch.uint8, device="cuda") for _ in range(2)]
    alg = m * (n / 4.0)
    res = {0: [], 1: []}
    kern = {}
    for rnd in range(args.rounds):
        for d in (0, 1):
            t.set_option(capi.OPT_DEFER_RECORDS, d)
            base = bufs[d].data_ptr()
            t.scan_case_control_device(0, m, base, base + m * 32, None)
            t.timer_start(None)
            for _ in range(args.reps):
                t.scan_case_control_device(0, m, base, base + m * 32, None)
            res[d].append(t.timer_stop(None) / args.reps)
            kern[d] = t.last_scan_kernel()
            assert t.get_option(capi.OPT_DEFER_RECORDS) == d, "the defer form is not available for this shape"
    torch.cuda.synchronize()
    same = bool(torch.equal(bufs[0], bufs[1]))
    for d in (0, 1):
        mn, med = min(res[d]), float(np.median(res[d]))
        print(f"N={n:7d} M={m:8d} defer={d} | median {med:7.4f} ms | min {mn:7.4f} ms | {alg/mn/1e6:6.0f} GB/s | "
              f"{alg/mn/1e6/80:5.1f} % | {kern[d]}", flush=True)
    print(f"N={n:7d} records identical: {same}", flush=True)
    if not same:
        sys.exit(1)
    t.close()
    del bufs
