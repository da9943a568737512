"""The N > 1 path with REAL HIP scans: two ranks, both on device 0 (the GPU box has one GPU;
RCCL refuses two ranks on one device, so the process group is gloo and the records cross through host
memory), each rank generates and scans its own contiguous SNP shard of ONE table with the HIP kernel, the
96-byte records are gathered to rank 0 with the same shard.* helpers bench.py uses, and rank 0's result is
compared byte for byte with a single-process scan of the whole table.  Needs a real MI355X: pytest -m gpu."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import libgwaspp_amd as gw

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORD = 96


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if gw.device_count() == 0:
        pytest.fail("no HIP device visible: the gpu tests must run on the GPU box")


def _worker(rank, world, rdzv, total, n, seed, missing, n_chunks, q):
    sys.path.insert(0, ROOT)
    import torch                      # before the first call into libgwscan.so (see capi.py)
    import torch.distributed as dist
    import libgwaspp_amd as gw
    from libgwaspp_amd import shard
    # rendezvous through a file under the test's tmp_path: nobody picks a port, so nobody can lose one
    dist.init_process_group("gloo", init_method="file://" + rdzv, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        begin, end = shard.snp_range(total, rank, world)
        n_local = end - begin
        cap = max(shard.snp_range(total, r, world)[1] - shard.snp_range(total, r, world)[0] for r in range(world))
        bounds = shard.chunk_bounds(cap, n_chunks)
        stream = torch.cuda.Stream()
        gathered = []
        with gw.GwTable(n, n_local, device=0) as t:
            t.simulate(seed, begin, missing)           # global SNP index = shard begin + row
            t.simulate_phenotype(seed)
            kernel = None
            for c in range(len(bounds) - 1):
                rows_cap = bounds[c + 1] - bounds[c]
                lo, hi = min(bounds[c], n_local), min(bounds[c + 1], n_local)
                # zero-filled ON the scan's stream: filled on torch's default stream it would race with the scan (the fill
                # can land after the records; seen once on the GPU box as zeros in a gathered chunk)
                with torch.cuda.stream(stream):
                    buf = torch.zeros(rows_cap * RECORD, dtype=torch.uint8, device="cuda")
                base = buf.data_ptr()
                if hi > lo:
                    t.scan_case_control_device(lo, hi - lo, base, base + rows_cap * 32, stream.cuda_stream)
                    kernel = t.last_scan_kernel()
                stream.synchronize()
                out, work = shard.gather_records(dist, buf.cpu(), rank, world, dst=0, async_op=True)
                gathered.append((out, work))
            for _, w in gathered:
                w.wait()
        if rank == 0:
            q.put(([[x.numpy() for x in out] for out, _ in gathered], bounds, kernel))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,total,missing,n_chunks", [(10_000, 20_001, 655, 1), (100_000, 3_001, 0, 2),
                                                       (500_000, 1_501, 0, 2)])
def test_two_ranks_scan_their_shards_on_the_gpu_and_gather(n, total, missing, n_chunks, tmp_path):
    import torch.multiprocessing as mp
    from libgwaspp_amd import shard
    world, seed = 2, 0x5EED
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    rdzv = str(tmp_path / "rdzv")
    procs = [ctx.Process(target=_worker, args=(r, world, rdzv, total, n, seed, missing, n_chunks, q))
             for r in range(world)]
    for p in procs:
        p.start()
    try:
        result = q.get(timeout=600)
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    chunks, bounds, kernel = result
    assert kernel.startswith("gw::k_scan_cc")
    # the single-process scan of the whole table
    with gw.GwTable(n, total) as t:
        t.simulate(seed, 0, missing)
        t.simulate_phenotype(seed)
        counts, stats = t.scan_case_control()
    seen = 0
    for r in range(world):
        b, e = shard.snp_range(total, r, world)
        for c in range(len(bounds) - 1):
            cap = bounds[c + 1] - bounds[c]
            lo, hi = min(bounds[c], e - b), min(bounds[c + 1], e - b)
            raw = chunks[c][r]
            got_c = raw[:cap * 32].view(np.uint32).reshape(cap, 8)[:hi - lo]
            got_s = raw[cap * 32:].view(np.float64).reshape(cap, 8)[:hi - lo]
            assert np.array_equal(got_c, counts[b + lo:b + hi])
            assert np.array_equal(got_s, stats[b + lo:b + hi])     # same kernel, same rows: bit-identical
            seen += hi - lo
    assert seen == total


def _torchrun(extra, nproc=2):
    """The launcher binds its own rendezvous socket (c10d store on port 0, held by the agent for the whole job and
    handed to the workers as MASTER_PORT): no port number exists before a socket is bound to it."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--rdzv-backend=c10d", "--rdzv-endpoint=127.0.0.1:0", "--local-addr", "127.0.0.1",
           os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    return subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)


def test_bench_gpus_2_on_a_one_gpu_box_fails_loudly_before_any_gpu_call():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has a GPU per rank")
    r = _torchrun(["--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "has no GPU of its own" in r.stderr and "nothing was launched" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_multi_rank_logic_rehearsed_on_one_gpu():
    """bench.py's own N = 2 step loop (config 4 sharding, ragged shards, two pieces per step, three record
    buffers per piece, eight CUs left free) with both ranks on device 0 and a gloo gather: rank 0 checks the gathered records of the
    last step against one scan of the whole table."""
    r = _torchrun(["--rehearse-on-one-gpu", "--snps", "40001", "--steps", "4", "--warmup", "1", "--chunks", "2",
                   "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["n_snps_total"] == 40001 and line["config"]["n_snps_per_gpu"] == 20001
    assert line["config"]["workload"].startswith("REHEARSAL ON ONE GPU")
    assert line["config"]["rehearsal_gathered_equals_single_scan"] is True
    assert line["roofline"]["kernel"].startswith("gw::k_scan_cc_pipe")
    assert line["config"]["kernel_shape"]["reserved_cus"] == 8


def test_c_abi_communicator_with_one_rank():
    """gw_comm_* (RCCL loaded at run time) with a world of one: the gather is the root's own device copy, enqueued
    behind the scan on the table's stream (tests/comm_script.py, in a process of its own without torch)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "comm_script.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "comm_script ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_cpp_host_sharded_scan_with_one_rank(tmp_path):
    """gwas_hip --assoc --world 1: the C++ host's sharded path (gw_shard_range, gw_comm_create, scan on the table's
    stream, gw_comm_gather, download) prints what the plain --assoc prints."""
    from golden_util import PERL_DIR
    cli = os.path.join(ROOT, "libgwaspp_amd", "bin", "gwas_hip")
    if not os.path.exists(cli):
        subprocess.run(["make", "-C", os.path.join(ROOT, "libgwaspp_amd", "host"), "-s"], check=True)
    name = "cc_m24_n1000"
    base = [cli, "-g", os.path.join(PERL_DIR, name + ".tped"), "-p", os.path.join(PERL_DIR, name + ".tfam")]
    plain = subprocess.run(base + ["--assoc"], check=True, capture_output=True, text=True).stdout
    shard = subprocess.run(base + ["--world", "1", "--rank", "0", "--comm-id", str(tmp_path / "id"), "--assoc"], check=True,
                           capture_output=True, text=True).stdout
    rows = lambda text: [ln for ln in text.splitlines() if ln.count("\t") == 15]
    assert len(rows(plain)) == 24 and rows(shard) == rows(plain)


def test_bench_exchange_through_the_c_abi_on_one_gpu():
    """bench.py --exchange capi at N = 1: the C-ABI gather (side stream, events, three record buffers) runs inside every
    step; rank 0's own slice arrives intact."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--exchange", "capi", "--workload", "c2", "--snps", "60000",
                        "--steps", "5", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["exchange"].startswith("gw_comm_gather")
    assert line["config"]["capi_gather_own_slice_intact"] is True


def test_bench_line_keeps_the_contract():
    """One JSON line with the keys the driver reads (a small workload; the default run is the 125 GB one)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c1", "--steps", "5", "--warmup", "2",
                        "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2
    assert d["unit"] == "SNPs/s" and d["higher_is_better"] is True and d["data"] == "synthetic"
    assert d["config"]["workload"].startswith("c1") and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.05 < rf["frac"] < 1.0
    assert rf["kernel"].startswith("gw::k_scan_cc") and "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "SNPs/s" and cb["value"] > 0 and cb["gpu_parity_on_sample"] is True
    assert abs(d["value"] - d["config"]["n_snps_total"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9


def test_bench_exchange_through_torch_rccl_with_one_rank():
    """bench.py --dist-with-one-rank: the calls of the N > 1 path (init_process_group("nccl"), async dist.gather to
    rank 0 with three record buffers, Work.wait() on the scan stream, barriers, all_reduce of the time) on the one GPU
    this box has; the gathered records of the last step equal what was scanned."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dist-with-one-rank", "--workload", "c2", "--snps", "60000",
                        "--steps", "6", "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["dist_with_one_rank"] == {"backend": "nccl", "gathered_equals_scanned": True}
    assert "torch.distributed gather (RCCL)" in line["config"]["exchange"]


# ---- config 5: the pair screen sharded over ranks (SURVEY.md 8e "C5") ---------------------------------------------
def test_bench_c5_two_ranks_rehearsed_on_one_gpu():
    """bench.py --workload c5 with two ranks on device 0 (gloo): each rank screens the row blocks gw_shard_pair_blocks
    deals it, hits go to a device list (gw_boost_screen_device), the lists are all-gathered; rank 0 requires the union
    of the two lists to be the single-process screen of the whole triangle, measures bit-identical."""
    r = _torchrun(["--workload", "c5", "--rehearse-on-one-gpu", "--snps", "6000", "--steps", "2", "--warmup", "1",
                   "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["unit"] == "pairs/s"
    assert line["config"]["pairs"] == 6000 * 5999 // 2
    assert line["config"]["rehearsal_union_equals_single_screen"] is True
    assert line["config"]["hits_over_30"] > 0
    assert line["roofline"]["kernel"] == "gw::k_boost_mfma_fp4<true, 2, 2, 1, false>"


def test_bench_c5_exchange_through_the_c_abi_on_one_gpu():
    """--exchange capi at N = 1: gw_comm_allgather (a world of one copies) behind the screens on the same stream."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c5", "--snps", "5000", "--exchange", "capi",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["exchange"].startswith("gw_comm_allgather")
    assert line["config"]["pairs"] == 5000 * 4999 // 2 and line["config"]["hits_over_30"] > 0


def test_cpp_host_sharded_boost_with_one_rank(tmp_path):
    """gwas_hip --test-boost-epi --world 1: computeBoostSharded (gw_comm_create, gw_table_replicate, gw_shard_pair_blocks,
    gw_boost_screen_device, gw_comm_allgather, stage 2 on rank 0) prints the report lines of the plain computeBoost."""
    import numpy as np
    cli = os.path.join(ROOT, "libgwaspp_amd", "bin", "gwas_hip")
    n, m = 2000, 60
    rng = np.random.default_rng(5)
    status = rng.integers(0, 2, size=n)
    q = rng.uniform(0.1, 0.5, size=(m, 1))
    a = (rng.random((m, n)) < q).astype(np.uint8) + (rng.random((m, n)) < q).astype(np.uint8)
    for i, j in ((3, 17), (20, 31), (40, 59)):
        copy = (status == 1) & (rng.random(n) < 0.7)
        a[j, copy] = a[i, copy]
    sym = np.array(["A A", "A C", "C C"])
    with open(tmp_path / "d.tped", "w") as f:
        for r in range(m):
            f.write(f"0 rs{r} 0 {r} " + " ".join(sym[a[r]]) + "\n")
    with open(tmp_path / "d.tfam", "w") as f:
        for s in range(n):
            f.write(f"Fam_{s} Ind_{s} Pat_{s} Mat_{s} x {status[s]}\n")
    base = [cli, "-g", str(tmp_path / "d.tped"), "-p", str(tmp_path / "d.tfam")]
    plain = subprocess.run(base + ["--test-boost-epi"], check=True, capture_output=True, text=True).stdout
    shard = subprocess.run(base + ["--world", "1", "--rank", "0", "--comm-id", str(tmp_path / "id"), "--comm-nonce", "77",
                                   "--test-boost-epi"], check=True, capture_output=True, text=True).stdout
    keep = lambda text: [ln for ln in text.splitlines() if ln.count("\t") == 6 or ln.startswith("Located ")]
    assert len(keep(plain)) >= 4 and keep(shard) == keep(plain)
