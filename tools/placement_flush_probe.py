#!/usr/bin/env python3
"""Companion of tools/placement_pmc.py: the three speed levels of the short-row scan per (table, record buffer) pair, measured
(a) as bench.py does -- launches back to back -- and (b) with the caches emptied before every launch (a 768 MB device
memset in front of it: the 256 MB Infinity Cache and the L2s hold nothing of the previous launch), HIP events around the
scan launch only.  If the levels are a property of what launch k leaves behind for launch k + 1 (its 96 MB of records
in the Infinity Cache, aliasing with the table's lines), (b) shows one level.
    python3 tools/placement_flush_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--torch" in sys.argv:
    import torch  # noqa: F401  (before the first call into libgwscan.so, see capi.py)
import libgwaspp_amd as gw  # noqa: E402
from libgwaspp_amd import capi  # noqa: E402

n, m = 10_000, 1_000_000
tabs = []
for _ in range(3):
    t = gw.GwTable(n, m)
    t.simulate(0x5EED, 0, 0, exact=True)
    t.simulate_phenotype(0x5EED)
    tabs.append(t)
# record buffers: two plain hipMalloc allocations (gw_device_malloc) and, with --torch, two from torch's caching allocator
bufs = [capi.DeviceBuffer(m * 96) for _ in range(2)]
names = ["hipMalloc", "hipMalloc"]
if "--torch" in sys.argv:
    import torch
    keep = [torch.empty(m * 96, dtype=torch.uint8, device="cuda") for _ in range(2)]

    class _T:
        def __init__(self, x):
            self.ptr = x.data_ptr()
    bufs += [_T(x) for x in keep]
    names += ["torch", "torch"]
else:
    bufs += [capi.DeviceBuffer(m * 96) for _ in range(2)]
    names += ["hipMalloc", "hipMalloc"]
junk = capi.DeviceBuffer(768 << 20)
print("# record buffers:", ", ".join(f"{k} {nm} {b.ptr:#x}" for k, (nm, b) in enumerate(zip(names, bufs))))
print("# ms per launch; a = back to back (20 launches, min of 3 rounds), b = every launch behind a 768 MB memset (mean / min of 12)")
for ti, t in enumerate(tabs):
    s = t.stream()
    row = []
    for bi, b in enumerate(bufs):
        t.scan_case_control_device(0, m, b.ptr, b.ptr + m * 32, None)
        best = 1e9
        for _ in range(3):
            t.timer_start(None)
            for _ in range(20):
                t.scan_case_control_device(0, m, b.ptr, b.ptr + m * 32, None)
            best = min(best, t.timer_stop(None) / 20)
        single = []
        for _ in range(12):
            junk.memset(0, s)
            t.timer_start(None)
            t.scan_case_control_device(0, m, b.ptr, b.ptr + m * 32, None)
            single.append(t.timer_stop(None))
        row.append((best, sum(single) / len(single), min(single)))
    print(f"table {ti}  " + "   ".join(f"a {a:.4f} b {bm:.4f}/{bn:.4f}" for a, bm, bn in row), flush=True)
