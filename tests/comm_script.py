"""Run as a script (tests/test_gpu_shard.py): the C-ABI's RCCL communicator with a world of one, in a process of its
own that never imports torch -- libgwscan.so and librccl.so then share ROCm's HIP / HSA runtime.  (A process that
imports torch FIRST shares torch's bundled runtime instead; mixing the two orders leaves RCCL without a device.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libgwaspp_amd as gw  # noqa: E402
from libgwaspp_amd import capi  # noqa: E402

RECORD = 96
n, m = 10_000, 5000
with gw.GwTable(n, m) as t:
    t.simulate(0x5EED, 0, 0)
    t.simulate_phenotype(0x5EED)
    want_c, want_s = t.scan_case_control()
    comm = gw.GwComm(0, 0, 1, gw.comm_new_id())
    send, recv = capi.DeviceBuffer(m * RECORD), capi.DeviceBuffer(m * RECORD)
    s = t.stream()
    t.scan_case_control_device(0, m, send.ptr, send.ptr + m * 32, s)
    comm.gather(send.ptr, m * RECORD, recv.ptr, 0, s)     # behind the scan, on the table's stream
    comm.synchronize(s)
    raw = recv.download(s)
    assert np.array_equal(raw[:m * 32].view(np.uint32).reshape(m, 8), want_c)
    assert np.array_equal(raw[m * 32:].view(np.float64).reshape(m, 8), want_s)
    try:
        comm.gather(send.ptr, 16, recv.ptr, 3, s)          # root outside the world
        raise SystemExit("a root outside the world was accepted")
    except gw.GwError as e:
        assert e.code == -1
    comm.close()
    send.close()
    recv.close()
print("comm_script ok")
