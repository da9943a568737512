#!/usr/bin/env python3
"""The pair screen at genome-wide size on ONE GPU: all 4.999995 x 10^11 pairs of 1 000 000 SNPs x 10 000 samples (synthetic,
bench.py's generator) in one gw_boost_screen_device call -- row blocks through one candidate list (GW_OPT_BOOST_ROWS_PER_LAUNCH
auto).  (round 4, MI355X: 6.02 s = 83 x 10^9 pairs/s, 2 211 317 pairs over 30, no candidate lost)"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libgwaspp_amd as gw
from libgwaspp_amd import capi
n, m = 10_000, 1_000_000
t = gw.GwTable(n, m)
t0 = time.time(); t.simulate(0x5EED, 0, 0, exact=True); t.simulate_phenotype(0x5EED); print(f"table generated in {time.time()-t0:.2f} s", flush=True)
cap = 1 << 22
lst = capi.DeviceBuffer(capi.BOOST_LIST_HEADER_BYTES + 16 * cap)
s = t.stream()
t0 = time.time(); t.boost_screen(0, 64, 0, 64); print(f"selectCaseControl + margins + tiled copy {time.time()-t0:.2f} s", flush=True)
t.boost_list_init(lst.ptr, s)
t.timer_start(s)
t.boost_screen_device(0, m, 0, m, lst.ptr, cap, 30.0, s)
ms = t.timer_stop(s)
hits, n_hits, n_pairs, aux0 = gw.parse_boost_list(lst.download(s), cap)
print(f"all {n_pairs} pairs of {m} SNPs x {n} samples: {ms:.1f} ms = {n_pairs/ms/1e6:.1f} G pairs/s, hits over 30: {n_hits}, candidates lost / outside the model: {aux0}, trips {t.get_option(capi.OPT_BOOST_PREFILTER_TRIPS)}, kernel {t.last_boost_kernel()}")
