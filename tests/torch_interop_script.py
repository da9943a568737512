"""Run as a script by tests/test_gpu_torch_interop.py (own process: torch must be imported before
libgwscan.so is loaded).  Device-pointer forms of the scan with torch-owned buffers and streams, two
tables in flight on two streams, and host-upload throughput."""
import json
import sys
import time

import numpy as np
import torch  # first: torch ships its own libamdhip64

import libgwaspp_amd as gw

assert torch.cuda.is_available() and gw.device_count() > 0
out = {}
n, m = 20_000, 4000
tables = []
for k in range(2):
    t = gw.GwTable(n, m)
    t.simulate(0x5EED + k, 1000 * k, 200)
    t.simulate_phenotype(0x5EED + k)
    tables.append(t)
want = [t.scan_case_control() for t in tables]           # host-pointer form (table's own stream)
want_dist = [t.scan_distribution() for t in tables]

streams = [torch.cuda.Stream(), torch.cuda.Stream()]
bufs = [torch.zeros(m * 96, dtype=torch.uint8, device="cuda") for _ in range(2)]
dists = [torch.zeros(m * 4, dtype=torch.int32, device="cuda") for _ in range(2)]
torch.cuda.synchronize()                                   # the fills ran on torch's default stream, the scans will not
for rep in range(3):                                       # interleave launches of both tables
    for k, t in enumerate(tables):
        with torch.cuda.stream(streams[k]):
            base = bufs[k].data_ptr()
            t.scan_case_control_device(0, m, base, base + m * 32, streams[k].cuda_stream)
            t.scan_distribution_device(0, m, dists[k].data_ptr(), streams[k].cuda_stream)
torch.cuda.synchronize()
ok = True
for k in range(2):
    raw = bufs[k].cpu().numpy()
    counts = raw[: m * 32].view(np.uint32).reshape(m, 8)
    stats = raw[m * 32:].view(np.float64).reshape(m, 8)
    ok &= bool(np.array_equal(counts, want[k][0]) and np.array_equal(stats, want[k][1]))
    ok &= bool(np.array_equal(dists[k].cpu().numpy().view(np.uint32).reshape(m, 4), want_dist[k]))
out["device_pointer_forms_match_host_forms"] = ok
# counts-only form writes nothing to the stats pointer
buf = torch.full((m * 96,), 0xAB, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()                                   # (same: the fill must have landed before the scan writes)
tables[0].scan_case_control_device(0, m, buf.data_ptr(), None, streams[0].cuda_stream)
torch.cuda.synchronize()
raw = buf.cpu().numpy()
out["counts_only_leaves_stats_untouched"] = bool(np.all(raw[m * 32:] == 0xAB) and
                                                 np.array_equal(raw[: m * 32].view(np.uint32).reshape(m, 8), want[0][0]))
# host -> HBM hand-over of packed rows (pageable numpy memory), both reference packings
rows = tables[0].download_rows(0, m)
big = np.tile(rows, (8, 1))
with gw.GwTable(n, big.shape[0]) as t:
    t0 = time.perf_counter()
    t.upload_rows(0, big)
    dt = time.perf_counter() - t0
    out["upload_stream_rows_GBps"] = big.nbytes / dt / 1e9
    back = t.download_rows(0, big.shape[0])
    out["upload_download_round_trip"] = bool(np.array_equal(back, big))
for t in tables:
    t.close()
print(json.dumps(out))
sys.exit(0 if all(v for k, v in out.items() if isinstance(v, bool)) else 1)
