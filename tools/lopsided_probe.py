#!/usr/bin/env python3
"""The pair screen on a lopsided biobank-shaped cohort: 200 000 samples, 20 % cases (40 000 < 65 536: the kernel that parks the
case counts, two-piece first level), 20 000 SNPs -- time, and the FP4 kernel against the int8 kernel on a block."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libgwaspp_amd as gw
from libgwaspp_amd import capi
n, m = 200_000, 20_000
t = gw.GwTable(n, m)
t.simulate(0x5EED, 0, 0, exact=True)
rng = np.random.default_rng(3)
is_case = rng.random(n) < 0.2
bits_c = np.zeros(t.geometry.stream_blocks * 16, dtype=np.uint8); bits_c[:n] = is_case
bits_t = np.zeros_like(bits_c); bits_t[:n] = ~is_case
pack = lambda b: np.packbits(b.reshape(-1, 16)[:, ::-1], axis=1).view(">u2").astype(np.uint16).ravel()
t.set_case_control(pack(bits_c), pack(bits_t), int(is_case.sum()), int((~is_case).sum()))
cap = 1 << 18
lst = capi.DeviceBuffer(capi.BOOST_LIST_HEADER_BYTES + 16 * cap)
s = t.stream()
t.boost_screen(0, 64, 0, 64)
best = 1e9
for _ in range(3):
    t.boost_list_init(lst.ptr, s); t.timer_start(s)
    t.boost_screen_device(0, m, 0, m, lst.ptr, cap, 30.0, s)
    best = min(best, t.timer_stop(s))
hits, n_hits, n_pairs, aux0 = gw.parse_boost_list(lst.download(s), cap)
ops = n_pairs * 4.0 * n * 2.0
print(f"{n} samples ({int(is_case.sum())} cases) x {m} SNPs: {best:.2f} ms = {n_pairs/best/1e6:.2f} G pairs/s = {ops/best/1e12:.2f} POP/s, hits {n_hits}, lost/outside {aux0}, {t.last_boost_kernel()}")
res = {}
for fp4 in (1, 0):
    t.set_option(capi.OPT_BOOST_FP4, fp4)
    h, nh, npairs, *_ = t.boost_screen(100, 3000, 5000, 4000, threshold=25.0, max_hits=1 << 18, extremes=False)
    res[fp4] = (np.sort(h, order=["snp_a", "snp_b"]), nh, npairs, t.last_boost_kernel())
print("block 3000 x 4000, FP4 == int8:", res[1][1:3] == res[0][1:3] and np.array_equal(res[1][0], res[0][0]), res[1][3], "|", res[0][3], "hits", res[1][1], "trips", t.get_option(capi.OPT_BOOST_PREFILTER_TRIPS))
