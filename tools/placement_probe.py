import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import libgwaspp_amd as gw
from libgwaspp_amd import capi
n, m = 10_000, 1_000_000
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0
junk = torch.empty(pre << 20, dtype=torch.uint8, device="cuda") if pre else None   # shift the heap by `pre` MB
t = gw.GwTable(n, m); t.simulate(0x5EED, 0, 0, exact=True); t.simulate_phenotype(0x5EED)
buf = torch.empty(m * 96, dtype=torch.uint8, device="cuda"); base = buf.data_ptr()
res = []
for rnd in range(3):
    t.scan_case_control_device(0, m, base, base + m * 32, None)
    t.timer_start(None)
    for _ in range(20): t.scan_case_control_device(0, m, base, base + m * 32, None)
    res.append(t.timer_stop(None) / 20)
pr = []
for _ in range(3):
    t.timer_start(None); nb = t.stream_read_probe(None); pr.append(nb / (t.timer_stop(None) * 1e-3) / 1e9)
print(f"pre {pre:5d} MB  buf {base:#x}  ms {min(res):.4f} {max(res):.4f}  frac {2.5e9/min(res)/1e6/8000:.3f}  probe {max(pr):.0f} GB/s")
