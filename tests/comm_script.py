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
    # all-gather / broadcast with a world of one: a copy / nothing to do
    comm.allgather(send.ptr, m * RECORD, recv.ptr, s)
    comm.broadcast(send.ptr, m * RECORD, 0, s)
    comm.synchronize(s)
    assert np.array_equal(recv.download(s), send.download(s))
    # gw_table_replicate from the only rank: the rows stay what they were
    before = t.download_rows(0, 64)
    t.replicate(comm, 0)
    assert np.array_equal(t.download_rows(0, 64), before)
    # the device-resident screen, block by block into one list == the host form over the same blocks
    k, cap = 3000, 1 << 14
    lst = capi.DeviceBuffer(capi.BOOST_LIST_HEADER_BYTES + 16 * cap)
    t.boost_list_init(lst.ptr, s)
    blocks = gw.shard_pair_blocks(k, 0, 1, 512) + []
    parts = [(0, 1000), (1000, 2000)]
    for a, na in parts:
        t.boost_screen_device(a, na, a, k - a, lst.ptr, cap, 30.0, s)
    comm.allgather(lst.ptr, lst.nbytes, lst.ptr, s)         # in place in a world of one
    comm.synchronize(s)
    hits, n_hits, n_pairs, aux0 = gw.parse_boost_list(lst.download(s), cap)
    want, wh, wp, _, _ = t.boost_screen(0, k, 0, k, threshold=30.0, max_hits=cap, extremes=False)
    assert blocks == [(0, k)] and aux0 == 0 and n_pairs == wp == k * (k - 1) // 2 and n_hits == wh > 0
    assert np.array_equal(np.sort(hits, order=["snp_a", "snp_b"]), np.sort(want, order=["snp_a", "snp_b"]))
    assert t.last_boost_kernel() == "gw::k_boost_mfma_fp4<true, 2, 2, 1, false>"
    lst.close()
    comm.close()
    send.close()
    recv.close()
print("comm_script ok")
