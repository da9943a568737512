#!/usr/bin/env python3
"""Kernel-shape sweep of the case/control scan on one resident table
(interleaved rounds in ONE process, HIP-event timing per launch group).
    python tools/sweep_scan.py --workload c3 --rounds 3 --reps 5
"""
import argparse
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402  -- before libgwscan.so: torch ships its own libamdhip64 and must load it first
import libgwaspp_amd as gw  # noqa: E402
from libgwaspp_amd import capi  # noqa: E402
from bench import WORKLOADS, SEED  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3")
ap.add_argument("--snps", type=int, default=0)
ap.add_argument("--samples", type=int, default=0, help="with --snps: a table of this size instead of a named workload")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--unroll", default="2,4,8")
ap.add_argument("--threads", default="256,512,1024")
ap.add_argument("--per-cu", default="0")
ap.add_argument("--lds", default="1")
ap.add_argument("--nt", default="1")
ap.add_argument("--pipe", default="-1", help="-1 auto, 0 row-at-a-time, 1 continuous-stream")
ap.add_argument("--compl", default="1", help="0 never, 1 auto, 2 whenever possible")
ap.add_argument("--reserve", default="0", help="CUs the persistent grid leaves free")
ap.add_argument("--stats", type=int, default=1)
args = ap.parse_args()

EXTRA = {"n250k": (250_000, 400_000, ""), "n50k": (50_000, 1_000_000, ""), "n1m": (1_000_000, 200_000, "")}
n, m, _ = {**WORKLOADS, **EXTRA}[args.workload]
if args.snps:
    m = args.snps
if args.samples:
    n = args.samples
t = gw.GwTable(n, m)
t.simulate(SEED, 0, 0)
t.simulate_phenotype(SEED)
buf = torch.empty(m * 96, dtype=torch.uint8, device="cuda")
base = buf.data_ptr()
alg = m * (n / 4.0)

lst = lambda s: [int(x) for x in s.split(",")]
combos = list(itertools.product(lst(args.unroll), lst(args.threads), lst(args.per_cu), lst(args.lds), lst(args.nt), lst(args.pipe), lst(args.compl), lst(args.reserve)))
res = {c: [] for c in combos}
kern = {}
for rnd in range(args.rounds):
    for c in combos:
        u, th, pc, lds, nt, pipe, cm, rsv = c
        t.set_option(capi.OPT_RESERVE_CUS, rsv)
        t.set_option(capi.OPT_PIPELINE, pipe)
        t.set_option(capi.OPT_COMPLEMENT, cm)
        t.set_option(capi.OPT_UNROLL, u)
        t.set_option(capi.OPT_BLOCK_THREADS, th)
        t.set_option(capi.OPT_BLOCKS_PER_CU, pc)
        t.set_option(capi.OPT_MASK_IN_LDS, lds)
        t.set_option(capi.OPT_NONTEMPORAL, nt)
        t.scan_case_control_device(0, m, base, base + m * 32 if args.stats else None, None)
        t.timer_start(None)
        for _ in range(args.reps):
            t.scan_case_control_device(0, m, base, base + m * 32 if args.stats else None, None)
        ms = t.timer_stop(None) / args.reps
        res[c].append(ms)
        kern[c] = t.last_scan_kernel()
probe = []
for _ in range(3):
    t.timer_start(None)
    nb = t.stream_read_probe(None)
    probe.append(nb / (t.timer_stop(None) * 1e-3) / 1e9)
print(f"workload {args.workload}: {n} x {m}, algorithmic {alg/1e9:.2f} GB/launch, probe {max(probe):.0f} GB/s")
print("unroll thr perCU lds nt pipe compl | median ms | min ms | GB/s(min) | %peak | kernel")
for c, v in sorted(res.items(), key=lambda kv: min(kv[1])):
    med, mn = float(np.median(v)), min(v)
    print(f"{c[0]:6d} {c[1]:4d} {c[2]:5d} {c[3]:3d} {c[4]:2d} {c[5]:4d} {c[6]:5d} rsv{c[7]:3d} | {med:9.3f} | {mn:7.3f} | {alg/mn/1e6:8.0f} | {alg/mn/1e6/80:5.1f} | {kern[c]}")
# the plain genotype distribution (no masks) on the same table: short-row kernel on / off
dist_buf = torch.empty(m * 16, dtype=torch.uint8, device="cuda")
for mode in (1, 0):
    t.set_option(capi.OPT_SHORT_ROWS, mode)
    t.scan_distribution_device(0, m, dist_buf.data_ptr(), None)
    best = 1e9
    for _ in range(args.reps):
        t.timer_start(None)
        t.scan_distribution_device(0, m, dist_buf.data_ptr(), None)
        best = min(best, t.timer_stop(None))
    print(f"distribution scan, short-row kernel {'on ' if mode else 'off'}: {best:7.3f} ms  {alg/best/1e6:8.0f} GB/s  {alg/best/1e6/80:5.1f} %")
t.close()
