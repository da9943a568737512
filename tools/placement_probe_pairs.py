import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import libgwaspp_amd as gw
n, m = 10_000, 1_000_000
def timeit(t, base):
    t.scan_case_control_device(0, m, base, base + m * 32, None)
    best = 1e9
    for _ in range(3):
        t.timer_start(None)
        for _ in range(20): t.scan_case_control_device(0, m, base, base + m * 32, None)
        best = min(best, t.timer_stop(None) / 20)
    return best
tabs = []
for k in range(3):
    t = gw.GwTable(n, m); t.simulate(0x5EED, 0, 0, exact=True); t.simulate_phenotype(0x5EED); tabs.append(t)
bufs = [torch.empty(m * 96 + (4 << 20), dtype=torch.uint8, device="cuda") for _ in range(4)]
for ti, t in enumerate(tabs):
    row = []
    for b in bufs:
        for off in (0, 4096, 1 << 20):
            row.append(timeit(t, b.data_ptr() + off))
    print("table", ti, " ".join(f"{x:.4f}" for x in row))
