#!/usr/bin/env python3
"""bench.py -- SNPs/s of the per-SNP case/control association scan on MI355X.

A "step" is one pass of the hot path (fused counts + MAF + HWE + chi-square
kernel) over the whole genotype table of the workload, inputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c1|c2|c3|c4|n10k|c5]

N = 1 (default): BASELINE.json configs[2] ("c3"): 500 000 samples x 1 000 000
SNPs (125 GB of 2-bit genotypes) on one GPU -- the configuration the north-star
target (>= 70 % of the HBM-read roofline) is quoted on.

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
one rank per GPU, and runs BASELINE.json configs[3] ("c4"): the SAME 1 M-SNP
table sharded by contiguous SNP ranges (rank r holds and scans
shard.snp_range(1 000 000, r, N); strong scaling), and the one exchange step of
the path -- the gather of the 96-byte per-SNP records to rank 0 over RCCL/xGMI
-- inside every timed step (it runs on RCCL's stream under the next step's
scan; every step's records have reached rank 0 before the clock stops).
`--scaling weak` keeps the full-size table on every rank instead and says so in
config.workload.  There is no silent fallback anywhere: a collective that
fails ends the run non-zero.

Prints ONE JSON line on rank 0 (see the task contract): value = SNPs all ranks
processed / max-over-ranks wall time; "roofline" for the scan kernel from HIP
events on the launching stream; "cpu_baseline" = the CPU oracle's faithful
single-thread port timed on this host on a bounded sample of the same table
(+ the whole of configs[0], 10 k x 50 k, on the CPU path).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n_samples, n_snps in the whole job, text)
    "c1": (10_000, 50_000, "c1: 10k samples x 50k SNPs case/control chi2 scan (plumbing size)"),
    "c2": (100_000, 500_000, "c2: 100k samples x 500k SNPs, per-SNP 3x3 contingency + chi2"),
    "c3": (500_000, 1_000_000, "c3: 500k samples x 1M SNPs, MAF + HWE + case/control chi2 combined scan"),
    "c4": (500_000, 1_000_000, "c4: 500k samples x 1M SNPs SNP-sharded across the GPUs, RCCL gather over xGMI"),
    # not a BASELINE.json config: the size of a typical cohort (short rows -> the sub-wave-per-row kernel)
    "n10k": (10_000, 1_000_000, "10k samples x 1M SNPs, MAF + HWE + case/control chi2 combined scan (short rows)"),
    # configs[4]: all SNP pairs; BASELINE.json gives no sample count for it -> 10 k (5 k / 5 k), stated
    "c5": (10_000, 100_000, "c5: pairwise SNPxSNP epistasis, BOOST KSA pre-screen over 100k SNPs (10k samples)"),
}
MFMA_I8_PEAK_TOPS = 5000.0  # dense int8 MFMA: 2x the ~2.5 PFLOP/s bf16 peak (MI355X_MICROARCH.md, matrix cores)
MFMA_FP4_PEAK_TOPS = 10000.0  # dense FP4 / FP6 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4): 4x bf16 per clock (same table)
MFMA_FP4_STREAM_TOPS = 8900.0   # measured: a pure stream of v_mfma_scale_f32_32x32x64_f8f6f4 (profiles/r03_mfma_rate_probe.txt)
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E nominal, /opt/skills/guides/MI355X_MICROARCH.md
SEED = 0x5EED
RECORD_BYTES = 32 + 64  # counts uint32[8] + stats float64[8] per SNP
KERNEL_SOURCE = os.path.join(ROOT, "libgwaspp_amd", "csrc", "gw_kernels.hip")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None,
                    help="default: c3 on one GPU, c4 (= c3's table sharded) on several")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = the workload's SNPs are sharded over the ranks (BASELINE config 4); "
                         "weak = every rank scans a full-size table of its own")
    ap.add_argument("--snps", type=int, default=0, help="override the job's SNP count (debug)")
    ap.add_argument("--chunks", type=int, default=1, help="scan/gather pieces per step")
    ap.add_argument("--reserve-cus", type=int, default=-1,
                    help="compute units the persistent scan leaves free (default: 0 on one GPU, 8 when N > 1 so that "
                         "RCCL's kernel never has to wait for, or push aside, a scan workgroup)")
    ap.add_argument("--buffers", type=int, default=3, help="N > 1: record buffers per piece (gathers in flight)")
    ap.add_argument("--collective", choices=["gather", "all_gather"], default="gather",
                    help="N > 1: RCCL gather to rank 0 (grouped send/recv, one xGMI link per peer) or all_gather")
    ap.add_argument("--exchange", choices=["torch", "capi"], default="torch",
                    help="who gathers the records: torch.distributed (RCCL through PyTorch, default) or the C-ABI's own "
                         "RCCL communicator (gw_comm_gather, the path a C++ host uses; also runs at N = 1, where the "
                         "gather is the root's own device copy)")
    ap.add_argument("--dist-with-one-rank", action="store_true",
                    help="N = 1 debugging aid: initialise torch.distributed / RCCL with a world of one and run the exchange "
                         "step through it (gather to self, barriers, all_reduce) -- exercises the N > 1 code path's calls "
                         "on one GPU; the JSON line says so")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 debugging aid for a one-GPU box: every rank opens device 0, the process group is gloo "
                         "and the records travel through host memory -- exercises the sharding / buffering logic, "
                         "NOT a measurement (the JSON line says so)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements (c2, n10k, c5)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline work")
    ap.add_argument("--unroll", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--mask-lds", type=int, default=-1)
    ap.add_argument("--nt", type=int, default=-1)
    return ap.parse_args()


def kernel_source_sha16():
    """sha256[:16] over every kernel source of the library (libgwaspp_amd/csrc/*.hip, *.inc, *.h, sorted by name): the stamp
    a committed PMC figure carries (tools/collect_pmc.py computes the same)."""
    import glob
    h = hashlib.sha256()
    d = os.path.dirname(KERNEL_SOURCE)
    for path in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.inc")) + glob.glob(os.path.join(d, "*.h"))):
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def git_head():
    """HEAD of the tree this run was built from: asked from git where the checkout is (the build container), otherwise
    read from libgwaspp_amd/GIT_HEAD, which __graft_entry__.build() / the csrc Makefile stamp at build time -- the GPU
    box receives a snapshot without .git."""
    try:
        h = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True,
                           timeout=10).stdout.strip()
        if h:
            return h
    except Exception:
        pass
    try:
        with open(os.path.join(ROOT, "libgwaspp_amd", "GIT_HEAD")) as f:
            return (f.read().strip() or None)
    except OSError:
        return None


def pmc_traffic(workload, kernel_name):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json) -- only if
    that figure was taken for the kernel instantiation that ran here AND from the kernel source as it is
    now (each entry is stamped with both); otherwise (None, why)."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            e = json.load(f).get(workload)
    except Exception as ex:
        return None, f"profiles/pmc_traffic.json unreadable: {ex}"
    if not e:
        return None, f"no PMC entry for workload {workload}"
    if e.get("kernel") != kernel_name:
        return None, f"PMC entry is for {e.get('kernel')}, this run launched {kernel_name}"
    if e.get("kernel_source_sha16") != kernel_source_sha16():
        return None, (f"PMC entry was taken at kernel source {e.get('kernel_source_sha16')} (git {e.get('git')}), "
                      f"the source is now {kernel_source_sha16()}")
    return e.get("hbm_bytes_per_launch"), f"profiles/pmc_traffic.json[{workload}], git {e.get('git')}"


def host_threads():
    # the GPU box gives a one-GPU job a 16-core share of its host (os.cpu_count() shows all of it)
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    return max(1, min(threads, 16))


def cpu_baseline(table, n_samples, seconds):
    """The oracle's faithful port of the reference CPU path (one thread, 64 Ki
    LUT popcount, the word loop of compressed_genotype_table5.cpp:609-657),
    timed on this host on the first rows of the SAME synthetic table, plus a
    parity check of the GPU results on that sample."""
    import numpy as np
    from oracle import oracle  # the reported baseline / checker -- never the measured path

    g = oracle.geometry_stream(n_samples)
    bytes_per_snp = n_samples / 4.0
    # ~1 GB/s per core for the faithful port: size the sample for `seconds` of work in 3 passes
    sample = int(max(256, min(table.n_snps, (seconds / 3.0) * 1.0e9 / bytes_per_snp)))
    rows = table.download_rows(0, sample)
    cs, ct, nc, nt = table.get_case_control()
    t0 = time.perf_counter()
    passes, counts, stats = 0, None, None
    while passes < 2 or time.perf_counter() - t0 < seconds * 0.8:
        counts, stats = oracle.scan_stream_cc(rows, g, cs, ct, nc, nt)
        passes += 1
        if passes >= 50:
            break
    dt = (time.perf_counter() - t0) / passes
    gpu_counts, gpu_stats = table.scan_case_control(0, sample)
    parity = bool(np.array_equal(gpu_counts, counts)) and bool(
        np.allclose(gpu_stats, stats, rtol=1e-10, atol=1e-12))
    threads = host_threads()
    t1 = time.perf_counter()
    mt_passes = 0
    while mt_passes < 2 or time.perf_counter() - t1 < 2.0:
        oracle.scan_stream_cc(rows, g, cs, ct, nc, nt, threads=threads)
        mt_passes += 1
    dt_mt = (time.perf_counter() - t1) / mt_passes
    return {
        "value": sample / dt, "unit": "SNPs/s", "cores": 1, "kind": "port",
        "sample": f"first {sample} SNPs x {n_samples} samples of the same synthetic table, "
                  f"{passes} passes, faithful single-thread LUT-popcount port",
        "GBps": sample * bytes_per_snp / dt / 1e9,
        "gpu_parity_on_sample": parity,
        "best_effort_all_cores": {"value": sample / dt_mt, "unit": "SNPs/s", "cores": threads,
                                  "GBps": sample * bytes_per_snp / dt_mt / 1e9,
                                  "note": "hardware popcount, pthreads over SNP ranges"},
    }


def cpu_baseline_c1(seconds=4.0):
    """BASELINE.json configs[0] as stated: the WHOLE 10 k x 50 k case/control scan on the CPU path
    (simulate_data.cpp-shaped input generated on the GPU and downloaded in the reference's stream
    packing), faithful single-thread LUT port and all host cores; the GPU scan of the same table is
    checked against it and timed beside it."""
    import numpy as np
    import libgwaspp_amd as gw
    from oracle import oracle

    n, m, text = WORKLOADS["c1"]
    g = oracle.geometry_stream(n)
    with gw.GwTable(n, m) as t:
        t.simulate(SEED, 0, 0, exact=True)
        t.simulate_phenotype(SEED)
        rows = t.download_rows(0, m)
        cs, ct, nc, nt = t.get_case_control()
        gpu_counts, gpu_stats = t.scan_case_control()
        import torch
        buf = torch.empty(m * RECORD_BYTES, dtype=torch.uint8, device="cuda")
        base = buf.data_ptr()
        for _ in range(5):
            t.scan_case_control_device(0, m, base, base + m * 32, None)
        t.synchronize()
        t.timer_start(None)
        for _ in range(50):
            t.scan_case_control_device(0, m, base, base + m * 32, None)
        gpu_ms = t.timer_stop(None) / 50
        gpu_kernel = t.last_scan_kernel()
    t0 = time.perf_counter()
    passes = 0
    while passes < 2 or time.perf_counter() - t0 < seconds:
        counts, stats = oracle.scan_stream_cc(rows, g, cs, ct, nc, nt)
        passes += 1
    dt = (time.perf_counter() - t0) / passes
    threads = host_threads()
    t1 = time.perf_counter()
    mt = 0
    while mt < 3 or time.perf_counter() - t1 < 1.0:
        oracle.scan_stream_cc(rows, g, cs, ct, nc, nt, threads=threads)
        mt += 1
    dt_mt = (time.perf_counter() - t1) / mt
    parity = bool(np.array_equal(gpu_counts, counts)) and bool(np.allclose(gpu_stats, stats, rtol=1e-10, atol=1e-12))
    return {"workload": text, "value": m / dt, "unit": "SNPs/s", "cores": 1, "kind": "port",
            "sample": f"all {m} SNPs x {n} samples, {passes} passes, faithful single-thread LUT-popcount port",
            "seconds_per_scan": dt, "GBps": m * (n / 4.0) / dt / 1e9, "gpu_parity": parity,
            "best_effort_all_cores": {"value": m / dt_mt, "unit": "SNPs/s", "cores": threads,
                                      "seconds_per_scan": dt_mt, "GBps": m * (n / 4.0) / dt_mt / 1e9},
            "gpu_same_table": {"ms_per_scan": gpu_ms, "value": m / (gpu_ms * 1e-3), "unit": "SNPs/s",
                               "achieved_GBps": m * (n / 4.0) / (gpu_ms * 1e-3) / 1e9,
                               "frac": m * (n / 4.0) / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "kernel": gpu_kernel,
                               "note": "0.125 GB table, resident in the 256 MB Infinity Cache"}}


def dropin_walk(name):
    """What a libgwaspp maintainer gets: the reference's own harness loop inline_cc_maf (algorithms/maf_func.cpp:292-316)
    walking HipGenotypeTable's per-row getter over a table of the workload's shape generated in HBM
    (libgwaspp_amd/bin/gwas_hip --simulate N M --time-inline-cc-maf, the C++ host over the C-ABI), window prefetch on
    and off, beside the kernel rate of the same table; the walk's counts are checked against one batch scan."""
    n, m, text = WORKLOADS[name]
    cli = os.path.join(ROOT, "libgwaspp_amd", "bin", "gwas_hip")
    if not os.path.exists(cli):
        return {"workload": text, "error": "libgwaspp_amd/bin/gwas_hip is not built (make -C libgwaspp_amd/host)"}
    r = subprocess.run([cli, "--simulate", str(n), str(m), "--seed", str(SEED), "--time-inline-cc-maf"], capture_output=True,
                       text=True, timeout=900)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        return {"workload": text, "error": f"gwas_hip exit {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
    d = json.loads(line[-1])
    d["workload"] = text
    d["value"], d["unit"] = d["snps_per_s_prefetch"], "SNPs/s"
    d["walk_over_kernel"] = d["snps_per_s_prefetch"] / d["kernel_snps_per_s"]
    return d


def bench_pairs(args, rank, world, local_rank, dist, steps=None, warmup=None, with_cpu_baseline=True, comm=None):
    """--workload c5: the BOOST / KSA pre-screen over every SNP pair (computeBoost,
    algorithms/epistasis_func.cpp:349-506).  A step = one screen of the whole pair space; the
    triangle is sharded over the ranks by row blocks (gw_shard_pair_blocks; strong scaling: the pair
    count is fixed), every rank holds the whole table, each rank appends its hits to a fixed-capacity
    list in HBM (gw_boost_screen_device: nothing comes back to the host inside a step) and the lists are
    all-gathered at the end of the step (RCCL through torch.distributed, or gw_comm_allgather with
    --exchange capi)."""
    import numpy as np
    import torch
    import libgwaspp_amd as gw
    from libgwaspp_amd import capi

    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    rehearse = args.rehearse_on_one_gpu and world > 1
    n_samples, n_snps, text = WORKLOADS["c5"]
    if args.snps and args.workload == "c5":
        n_snps = args.snps
    table = gw.GwTable(n_samples, n_snps, device=local_rank)
    table.simulate(SEED, 0, 0, exact=True)   # every rank generates the same table (counter-based RNG); a job
    table.simulate_phenotype(SEED)           # with ingested rows replicates them with gw_table_replicate instead
    blocks = gw.shard_pair_blocks(n_snps, rank, world, block=2048)
    cap = 1 << 18
    list_bytes = capi.BOOST_LIST_HEADER_BYTES + 16 * cap
    d_list = torch.zeros(list_bytes, dtype=torch.uint8, device="cuda")
    d_all = torch.zeros(world * list_bytes, dtype=torch.uint8, device="cuda") if (world > 1 or comm is not None) else d_list
    torch.cuda.synchronize()   # the zero fills ran on torch's default stream; everything below runs on `stream`
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    sptr = stream.cuda_stream
    table.boost_screen(0, min(64, n_snps), 0, min(64, n_snps))  # compaction + margins (untimed setup)
    host_lists = {}

    def step():
        table.boost_list_init(d_list.data_ptr(), sptr)
        for a, na in blocks:
            table.boost_screen_device(a, na, a, n_snps - a, d_list.data_ptr(), cap, 30.0, sptr)  # as computeBoost: threshold 30
        if world > 1 or comm is not None:  # fixed-size hit lists: one all-gather per step, enqueued behind the screens
            if rehearse:  # gloo, through host memory: logic check only
                stream.synchronize()
                parts = [torch.empty(list_bytes, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(parts, d_list.cpu())
                host_lists["all"] = parts
            elif comm is not None:
                comm.allgather(d_list.data_ptr(), list_bytes, d_all.data_ptr(), sptr)
            else:
                dist.all_gather_into_tensor(d_all, d_list)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    table.timer_start(sptr)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    kernel_ms_total = table.timer_stop(sptr)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # what the last step left: every rank's header (hits, pairs, pairs outside the pre-filter's error model)
    if rehearse:
        raws = [p.numpy() for p in host_lists["all"]]
    else:
        whole = d_all.cpu().numpy()
        raws = [whole[r * list_bytes:(r + 1) * list_bytes] for r in range(world)]
    parsed = [capi.parse_boost_list(r, cap) for r in raws]
    n_hits = sum(p[1] for p in parsed)
    n_pairs = sum(p[2] for p in parsed)
    if any(p[3] for p in parsed) or any(p[1] > cap for p in parsed):
        raise SystemExit("bench.py: a hit list overflowed or the pre-filter's error model was violated "
                         f"(hits {[p[1] for p in parsed]}, capacity {cap}, outside-model pairs {[p[3] for p in parsed]})")
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kernel_name = table.last_boost_kernel()
    rehearsal_ok = None
    if rehearse and rank == 0:  # the union of the ranks' lists == the single-process screen of the whole triangle
        got = np.sort(np.concatenate([p[0] for p in parsed]), order=["snp_a", "snp_b"])
        want, wh, wp, _, _ = table.boost_screen(0, n_snps, 0, n_snps, threshold=30.0, max_hits=cap, extremes=False)
        want = np.sort(want, order=["snp_a", "snp_b"])
        rehearsal_ok = bool(wh == n_hits and wp == n_pairs and np.array_equal(got, want))
    out = None
    c5_traffic = (pmc_traffic("c5", kernel_name) if world == 1 and not args.snps and len(blocks) == 1
                  else (None, "PMC figure is per whole-triangle launch on one GPU"))
    if rank == 0:
        per_step = elapsed / steps
        # no missing calls in the synthetic table: 4 counted corners per pair and group, each a dot product
        # of 0/1 class vectors over the group's samples = n_samples MACs per corner -> int8 MFMA
        macs = n_pairs * 4.0 * n_samples
        fp4 = "fp4" in kernel_name   # the kernel the dispatch recorded decides which matrix-core peak prices the run
        peak = MFMA_FP4_PEAK_TOPS if fp4 else MFMA_I8_PEAK_TOPS
        out = {
            "metric": "SNP pairs/sec BOOST KSA pre-screen", "value": n_pairs / per_step, "unit": "pairs/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": per_step * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": ("fp4 (e2m1) x fp4 -> f32 (counts, exact), f64 (measure)" if fp4 else "i8 x i8 -> i32 (counts), f64 (measure)"),
            "data": "synthetic",
            "config": {"workload": ("REHEARSAL ON ONE GPU (gloo, host-staged exchange; not a measurement): " if rehearse else "") + text,
                       "n_samples": n_samples, "n_snps": n_snps, "pairs": n_pairs,
                       "hits_over_30": n_hits, "sharding": "row blocks of 2048 SNPs in snake order (gw_shard_pair_blocks), table replicated",
                       "exchange": (("gw_comm_allgather (C-ABI, ncclAllGather)" if comm is not None else
                                     "torch.distributed all_gather_into_tensor (RCCL)") +
                                    f" of the fixed-capacity device hit lists ({list_bytes} B per rank) inside every step"
                                    if (world > 1 or comm is not None) else "none"),
                       "rehearsal_union_equals_single_screen": rehearsal_ok,
                       "stats_dtype": "f64", "git": git_head()},
            "roofline": {"bound": "mfma", "achieved": 2.0 * macs / per_step / 1e12, "peak": peak,
                         "unit": "TOP/s (2 ops per multiply-add of a 0/1 class-vector dot product; wall time of the whole screen "
                                 "incl. the per-pair bound / measure epilogue), peak = dense " +
                                 ("FP4 MFMA (the operands are FP4: one sample per nibble)" if fp4 else "int8 MFMA"),
                         "frac": 2.0 * macs / per_step / 1e12 / peak,
                         "frac_of_int8_peak": 2.0 * macs / per_step / 1e12 / MFMA_I8_PEAK_TOPS,
                         # what a pure stream of this MFMA sustained on one MI355X (tools/mfma_rate_probe.hip,
                         # profiles/r03_mfma_rate_probe.txt): informational, `peak` stays the guide's dense figure
                         "mfma_stream_sustained_TOPs": (MFMA_FP4_STREAM_TOPS if fp4 else None),
                         "traffic": c5_traffic[0], "traffic_source": c5_traffic[1],
                         "kernel": kernel_name, "kernel_ms": kernel_ms_total / steps,
                         "launches_per_step": len(blocks),
                         "note": "algorithmic MACs only (sample padding to 1024 per group not counted); kernel_ms = HIP "
                                 "events on the launching stream around the timed steps / steps"},
            "cpu_baseline": None,
        }
        if world == 1 and with_cpu_baseline and not args.no_cpu_baseline:
            from oracle import oracle
            k = 1200  # ~7 x 10^5 pairs, a few seconds of one host core
            rows = table.download_rows(0, k)
            cs, ct, nc, nt = table.get_case_control()
            t1 = time.perf_counter()
            want = oracle.boost_all_pairs(rows, oracle.geometry_stream(n_samples), cs, ct, nc, nt)
            dt = time.perf_counter() - t1
            got = table.boost_measures(0, k, 0, k)
            flat = got[np.triu_indices(k, 1)]
            out["cpu_baseline"] = {"value": len(want) / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
                                   "sample": f"all pairs of the first {k} SNPs of the same table, oracle restatement "
                                             "of computeBoost's loop (LUT popcount, one thread)",
                                   # the measure is 2n x (a sum of ~40 terms that cancel to ~1e-4): its rounding
                                   # noise is absolute, ~n x 1e-15; device log vs libm log differ in the last ulp
                                   # (monomorphic SNPs give 0/0 = NaN in the reference's formula: NaN on both sides)
                                   "max_abs_diff_vs_gpu": float(np.nanmax(np.abs(flat - want))),
                                   "nan_pairs": int(np.isnan(want).sum()),
                                   "gpu_parity_on_sample": bool(np.array_equal(np.isnan(flat), np.isnan(want)) and
                                                                np.allclose(flat, want, rtol=1e-10, atol=1e-8,
                                                                            equal_nan=True))}
    table.close()
    if out is not None and world == 1 and not args.snps and not args.no_cpu_baseline:
        out["config"]["compute_boost"] = compute_boost_end_to_end(n_samples, n_snps, local_rank)
    return out


def compute_boost_end_to_end(n_samples, n_snps, device):
    """computeBoost as a user of the drop-in table runs it (algorithms/epistasis_func.cpp:349-506): selectCaseControl +
    computeMargins, the pre-screen of all pairs, stage 2 (computeGTest) -- through the C++ host (`gwas_hip --time-boost`,
    a child process with a table of its own; this process has closed its table).  Untimed side measurement: `value` is
    the screen.  The one-thread host form of stage 2 (round 3's) is timed beside the device form as its baseline."""
    cli = os.path.join(ROOT, "libgwaspp_amd", "bin", "gwas_hip")
    if not os.path.exists(cli):
        return {"error": f"{cli} is not built (make -C libgwaspp_amd/host)"}
    r = subprocess.run([cli, "--simulate", str(n_samples), str(n_snps), "--seed", hex(SEED), "--device", str(device), "--time-boost"],
                       capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": (r.stderr or r.stdout)[-400:]}
    d = json.loads(lines[-1])
    return {"compute_boost_end_to_end_ms": d["compute_boost_end_to_end_s"] * 1e3,
            "select_and_margins_ms": d["select_and_margins_s"] * 1e3, "screen_ms": d["screen_s"] * 1e3,
            "stage2_device_ms": d["stage2_device_s"] * 1e3,
            "stage2_one_host_thread_ms": d["stage2_one_host_thread_s"] * 1e3,
            "hits_over_30": d["hits_over_30"], "hits_over_30_after_stage2": d["hits_over_30_after_stage2"],
            "max_abs_diff_device_vs_host_stage2": d["max_abs_diff_device_vs_host_stage2"],
            "what": "wall clock of one computeBoost through HipGenotypeTable (second pass of two): host-side walk of the per-row "
                    "margins getter, gw_boost_screen incl. the copy of the hits, gw_boost_gtest incl. its copies"}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus) and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    import torch
    # before any GPU call: one rank per GPU means this node must show at least LOCAL_RANK + 1 devices
    # (torch.cuda.device_count() does not initialise the GPU)
    n_visible = torch.cuda.device_count()
    if n_visible == 0:
        raise SystemExit("bench.py: no GPU visible; the scan has no CPU fallback")
    rehearse = args.rehearse_on_one_gpu and world > 1
    if rehearse:
        local_rank = 0
    if local_rank >= n_visible:
        raise SystemExit(f"bench.py: rank {rank} (LOCAL_RANK {local_rank}) has no GPU of its own: this node shows "
                         f"{n_visible} device(s) for {world} ranks -- one rank per GPU, nothing was launched")
    import numpy as np  # noqa: F401
    import libgwaspp_amd as gw
    from libgwaspp_amd import capi, shard

    if gw.device_count() == 0:
        raise SystemExit("bench.py: no GPU visible; the scan has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    solo_dist = args.dist_with_one_rank and world == 1
    if world > 1 or solo_dist:
        import torch.distributed as dist
        # Under a launcher MASTER_ADDR / MASTER_PORT come from it (env://).  Started by hand with one rank there is no
        # launcher and no port worth picking: the store is a file in a directory of this process's own.
        rdzv_dir = None
        if "MASTER_PORT" in os.environ:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            init = {}
        elif world == 1:
            import atexit
            import shutil
            import tempfile
            rdzv_dir = tempfile.mkdtemp(prefix="gw_bench_rdzv_")
            atexit.register(shutil.rmtree, rdzv_dir, True)
            init = {"init_method": "file://" + os.path.join(rdzv_dir, "store")}
        else:
            raise SystemExit("bench.py: WORLD_SIZE > 1 without MASTER_PORT -- start the ranks with torch.distributed.run")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world, **init)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank), **init)

    comm = None
    if args.exchange == "capi" and not (args.rehearse_on_one_gpu and world > 1):
        # the C-ABI's own communicator: rank 0 makes the id, torch.distributed only carries it to the others
        if world > 1:
            idt = torch.zeros(capi.COMM_ID_BYTES, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(gw.comm_new_id()), dtype=torch.uint8))
            dist.broadcast(idt, src=0)
            comm_id = bytes(idt.cpu().numpy().tobytes())
        else:
            comm_id = gw.comm_new_id()
        comm = gw.GwComm(local_rank, rank, world, comm_id)

    workload = args.workload or ("c3" if world == 1 else "c4")
    if workload == "c5":
        out = bench_pairs(args, rank, world, local_rank, dist, comm=comm)
        if comm is not None:
            comm.close()
        if world > 1 or solo_dist:
            dist.barrier()
            dist.destroy_process_group()
        if out is not None:
            print(json.dumps(out))
        return

    n_samples, n_job, wl_text = WORKLOADS[workload]
    if args.snps:
        n_job = args.snps
    strong = world > 1 and args.scaling == "strong"
    if rehearse:
        wl_text = "REHEARSAL ON ONE GPU (gloo, host-staged exchange; not a measurement): " + wl_text
    if world > 1 and not strong:
        # the explicit weak-scaling form: a full-size table on every rank, global SNP index = rank * M + row
        wl_text = f"WEAK-SCALING variant (not a BASELINE config): every rank scans its own {n_job}-SNP table; " + wl_text
        begin, n_local, n_total = rank * n_job, n_job, n_job * world
    elif world > 1:
        begin, end = shard.snp_range(n_job, rank, world)  # BASELINE config 4: the job's SNPs sharded
        n_local, n_total = end - begin, n_job
    else:
        begin, n_local, n_total = 0, n_job, n_job
    n_cap = max(shard.snp_range(n_total, r, world)[1] - shard.snp_range(n_total, r, world)[0]
                for r in range(world)) if strong else n_local  # equal buffer sizes for the collective

    table = gw.GwTable(n_samples, n_local, device=local_rank)
    if args.unroll:
        table.set_option(capi.OPT_UNROLL, args.unroll)
    if args.threads:
        table.set_option(capi.OPT_BLOCK_THREADS, args.threads)
    if args.blocks_per_cu:
        table.set_option(capi.OPT_BLOCKS_PER_CU, args.blocks_per_cu)
    if args.mask_lds >= 0:
        table.set_option(capi.OPT_MASK_IN_LDS, args.mask_lds)
    if args.nt >= 0:
        table.set_option(capi.OPT_NONTEMPORAL, args.nt)
    reserve_cus = args.reserve_cus if args.reserve_cus >= 0 else (8 if world > 1 else 0)
    table.set_option(capi.OPT_RESERVE_CUS, reserve_cus)
    t_setup = time.perf_counter()
    table.simulate(SEED, begin, 0, exact=True)  # rows of ONE table: row r of this rank = global SNP begin + r
    table.simulate_phenotype(SEED)           # identical masks on every rank (same seed)
    t_setup = time.perf_counter() - t_setup

    # result buffers (torch = device memory + streams + RCCL plumbing).  A step is cut into n_chunks pieces;
    # each piece has `n_buf` record buffers used round-robin over the steps, so a scan only ever waits for
    # the gather that used its buffer n_buf steps ago.
    n_chunks = max(1, args.chunks)
    bounds = shard.chunk_bounds(n_cap, n_chunks)
    n_chunks = len(bounds) - 1
    exchanging = world > 1 or comm is not None or solo_dist
    n_buf = max(1, args.buffers) if exchanging else 1
    bufs, gathers, allgather_out = [], [], []
    recv_flat, events = [], []
    for c in range(n_chunks):
        cap = bounds[c + 1] - bounds[c]
        bufs.append([torch.empty(cap * RECORD_BYTES, dtype=torch.uint8, device="cuda") for _ in range(n_buf)])
        if (world > 1 or solo_dist) and args.collective == "gather" and rank == 0 and not rehearse and comm is None:
            gathers.append([[torch.empty(cap * RECORD_BYTES, dtype=torch.uint8, device="cuda") for _ in range(world)]
                            for _ in range(n_buf)])
        if comm is not None:
            recv_flat.append([torch.empty(world * cap * RECORD_BYTES, dtype=torch.uint8, device="cuda") if rank == 0 else None
                              for _ in range(n_buf)])
            events.append([None] * n_buf)
        if (world > 1 or solo_dist) and args.collective == "all_gather":
            allgather_out.append([torch.empty(world * cap * RECORD_BYTES, dtype=torch.uint8, device="cuda")
                                  for _ in range(n_buf)])
    # An explicit (non-default) torch stream: the scans are launched on it through the C-ABI and RCCL
    # orders each gather behind it.  (The default stream's handle is 0, which the C-ABI reads as "the
    # table's own stream" -- RCCL would then not wait for the scan.)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    sptr = stream.cuda_stream
    assert sptr != 0
    side = torch.cuda.Stream() if comm is not None else None  # the C-ABI gather's stream (torch's RCCL has its own)

    def scan_piece(c, b):
        lo, hi = min(bounds[c], n_local), min(bounds[c + 1], n_local)  # a shard one row short scans one row less
        cap = bounds[c + 1] - bounds[c]
        base = bufs[c][b].data_ptr()
        if hi > lo:
            table.scan_case_control_device(lo, hi - lo, base, base + cap * 32, sptr)

    # the exchange step: per-SNP records to rank 0.  gather = grouped send/recv in RCCL, every peer's
    # slice over its own xGMI link.  An exception here ends the run (no fallback).
    rehearsal_last = {}

    def exchange(c, b):
        if comm is not None:  # gw_comm_gather on a side stream, behind this piece's scan
            side.wait_stream(stream)
            cap = bounds[c + 1] - bounds[c]
            comm.gather(bufs[c][b].data_ptr(), cap * RECORD_BYTES, recv_flat[c][b].data_ptr() if rank == 0 else None, 0,
                        side.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(side)
            events[c][b] = ev
            return None
        if rehearse:  # gloo, through host memory, synchronous: logic check only
            stream.synchronize()
            got, _ = shard.gather_records(dist, bufs[c][b].cpu(), rank, world, dst=0)
            if rank == 0:
                rehearsal_last[c] = got
            return None
        if args.collective == "gather":
            if solo_dist:  # shard.gather_records short-cuts a world of one; here the collective itself is wanted
                return dist.gather(bufs[c][b], gathers[c][b], dst=0, async_op=True)
            _, w = shard.gather_records(dist, bufs[c][b], rank, world, dst=0, async_op=True,
                                        out=gathers[c][b] if rank == 0 else None)
            return w
        return dist.all_gather_into_tensor(allgather_out[c][b], bufs[c][b], async_op=True)

    pending = [[None] * n_buf for _ in range(n_chunks)]
    step_no = [0]

    def step():
        b = step_no[0] % n_buf
        step_no[0] += 1
        for c in range(n_chunks):
            if pending[c][b] is not None:
                pending[c][b].wait()  # GPU-side: the scan stream waits for the gather that last read this buffer
                pending[c][b] = None
            if comm is not None and events[c][b] is not None:
                stream.wait_event(events[c][b])
                events[c][b] = None
            scan_piece(c, b)
            if exchanging:
                pending[c][b] = exchange(c, b)

    def drain():
        for c in range(n_chunks):
            for b in range(n_buf):
                if pending[c][b] is not None:
                    pending[c][b].wait()
                    pending[c][b] = None
                if comm is not None and events[c][b] is not None:
                    stream.wait_event(events[c][b])
                    events[c][b] = None

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1 or solo_dist:
        dist.barrier()
    torch.cuda.synchronize()
    table.timer_start(sptr)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()  # every step's records have reached rank 0 before the clock stops
    kernel_ms_total = table.timer_stop(sptr)  # HIP events on the launching stream; syncs on the stop event
    torch.cuda.synchronize()
    if world > 1 or solo_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1 or solo_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    launches = args.steps * n_chunks
    if world > 1 or solo_dist:
        # with the exchange in the loop the compute stream also waits for RCCL; time the scan
        # kernel alone (same launches, no exchange) for the roofline figure
        torch.cuda.synchronize()
        table.timer_start(sptr)
        for _ in range(args.steps):
            for c in range(n_chunks):
                scan_piece(c, 0)
        kernel_ms_total = table.timer_stop(sptr)

    out = None
    if rank == 0:
        geo = table.geometry
        kernel_ms = kernel_ms_total / max(1, launches)
        alg_bytes_per_launch = (n_local / n_chunks) * (n_samples / 4.0)
        achieved = alg_bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        kernel_name = table.last_scan_kernel()  # recorded by the dispatch that launched it
        # empirical streaming-read ceiling on the same buffer
        probe_ms = []
        for _ in range(3):
            table.timer_start(None)
            nbytes = table.stream_read_probe(None)
            probe_ms.append(table.timer_stop(None))
        probe_gbps = nbytes / (min(probe_ms) * 1e-3) / 1e9 if nbytes else None
        shape = {"unroll": table.get_option(capi.OPT_UNROLL),
                 "mask_in_lds": table.get_option(capi.OPT_MASK_IN_LDS),
                 "block_threads": table.get_option(capi.OPT_BLOCK_THREADS),
                 "blocks_per_cu": table.get_option(capi.OPT_BLOCKS_PER_CU),
                 "nontemporal": table.get_option(capi.OPT_NONTEMPORAL),
                 "continuous_stream": table.get_option(capi.OPT_PIPELINE),
                 "single_mask": table.get_option(capi.OPT_COMPLEMENT),
                 "sub_wave_rows": table.get_option(capi.OPT_SHORT_ROWS),
                 "reserved_cus": table.get_option(capi.OPT_RESERVE_CUS)}
        traffic, traffic_note = (pmc_traffic(workload, kernel_name) if n_chunks == 1 and world == 1 and not args.snps
                                 else (None, "PMC figure is per whole-table launch on one GPU"))
        out = {
            "metric": "SNPs/sec case/control chi2 scan",
            "value": n_total / (elapsed / args.steps),
            "unit": "SNPs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if (strong or world == 1) else "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": wl_text, "n_samples": n_samples, "n_snps_per_gpu": n_local,
                "n_snps_total": n_total, "table_GB_per_gpu": geo.device_table_bytes / 1e9,
                "algorithmic_bytes_per_snp": n_samples / 4.0,
                "input": f"restated data/simulate_data.cpp (affy6 MAF spectrum, exactly floor(p * 2N) minor alleles per SNP "
                         f"in a random arrangement as its shuffle gives, coin-flip phenotype), "
                         f"counter-based RNG seed {SEED:#x}, generated in HBM in {t_setup:.2f} s",
                "sharding": (f"contiguous SNP ranges of one {n_total}-SNP table, one rank per GPU (rank 0: "
                             f"[{begin}, {begin + n_local}))" if strong else
                             "a full-size table per rank" if world > 1 else "single GPU"),
                "exchange": ((f"gw_comm_gather (C-ABI, grouped ncclSend/ncclRecv)" if comm is not None else
                              f"torch.distributed {args.collective} (RCCL)") +
                             f" of {RECORD_BYTES} B/SNP to rank 0 inside every step, {n_chunks} piece(s) per step, "
                             f"{n_buf} buffers per piece (gathers run under the following scans)" +
                             (" -- ONE RANK, debugging aid" if solo_dist else "")) if exchanging else "none",
                "kernel_shape": shape, "stats_dtype": "f64", "git": git_head(),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_note,
                "kernel": kernel_name, "kernel_ms": kernel_ms, "launches_timed": launches,
                "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                "stream_read_probe_GBps": probe_gbps,
                "frac_of_probe": (achieved / probe_gbps) if probe_gbps else None,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(table, n_samples, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
    if rehearse and rank == 0:
        import numpy as np
        with gw.GwTable(n_samples, n_total, device=0) as whole:
            whole.simulate(SEED, 0 if strong else 0, 0, exact=True)
            whole.simulate_phenotype(SEED)
            counts, stats = whole.scan_case_control()
        ok = True
        for c in range(n_chunks):
            cap = bounds[c + 1] - bounds[c]
            for r in range(world):
                b0 = shard.snp_range(n_total, r, world)[0] if strong else r * n_job
                nl = (shard.snp_range(n_total, r, world)[1] - b0) if strong else n_job
                lo, hi = min(bounds[c], nl), min(bounds[c + 1], nl)
                raw = rehearsal_last[c][r].numpy()
                gc = raw[:cap * 32].view(np.uint32).reshape(cap, 8)[:hi - lo]
                gs = raw[cap * 32:].view(np.float64).reshape(cap, 8)[:hi - lo]
                if strong:
                    ok &= bool(np.array_equal(gc, counts[b0 + lo:b0 + hi]) and np.array_equal(gs, stats[b0 + lo:b0 + hi]))
        out["config"]["rehearsal_gathered_equals_single_scan"] = ok if strong else None
    if comm is not None:
        if rank == 0 and out is not None:  # the last step's records as gathered: rank 0's own slice equals what it scanned
            cap0 = bounds[1] - bounds[0]
            b_last = (step_no[0] - 1) % n_buf
            own = recv_flat[0][b_last][:cap0 * RECORD_BYTES]
            out["config"]["capi_gather_own_slice_intact"] = bool(torch.equal(own, bufs[0][b_last]))
        comm.close()
    if solo_dist and out is not None:  # what the gather to self delivered in the last step
        b_last = (step_no[0] - 1) % n_buf
        out["config"]["dist_with_one_rank"] = {
            "backend": dist.get_backend(), "gathered_equals_scanned": bool(torch.equal(gathers[0][b_last][0], bufs[0][b_last]))}
    if world > 1 or solo_dist:
        dist.barrier()
        dist.destroy_process_group()
    table.close()
    if out is not None and world == 1 and workload == "c3" and not args.snps:
        # side measurements, each on its own table after the main one was freed
        if not args.no_cpu_baseline:
            out["cpu_baseline"]["c1"] = cpu_baseline_c1()
        if not args.no_extra:
            # each side workload in a process of its own, after this one has freed its 125 GB: the same command line a
            # user would run (`bench.py --workload c2`), and a table allocated from an untouched heap -- inside this
            # process, right after the big table was freed, the 2.5 GB table of n10k measured 10 % slower
            def side(name, steps, warmup, cpu):
                cmd = [sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", str(steps), "--warmup", str(warmup),
                       "--no-extra"] + ([] if cpu else ["--no-cpu-baseline"])
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                if r.returncode != 0 or not line:
                    return {"workload": WORKLOADS[name][2], "error": f"exit {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
                d = json.loads(line[-1])
                rf = d["roofline"]
                e = {"workload": d["config"]["workload"], "ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"],
                     "steps": d["steps"], "frac": rf["frac"], "bound": rf["bound"], "kernel": rf["kernel"],
                     "kernel_ms": rf.get("kernel_ms"), "traffic": rf.get("traffic"), "traffic_source": rf.get("traffic_source"),
                     "command": "python bench.py " + " ".join(cmd[2:])}
                if rf["bound"] == "hbm":
                    e["achieved_GBps"] = rf["achieved"]
                    e["algorithmic_bytes_per_launch"] = rf["algorithmic_bytes_per_launch"]
                else:
                    e.update({"pairs": d["config"]["pairs"], "hits_over_30": d["config"]["hits_over_30"],
                              "mfma_TOPs": rf["achieved"], "peak_TOPs": rf["peak"],
                              "frac_of_int8_peak": rf.get("frac_of_int8_peak"), "dtype": d["dtype"],
                              "cpu_baseline": d.get("cpu_baseline")})
                    cb = d["config"].get("compute_boost")
                    if cb:   # computeBoost end to end through the drop-in table (margins + screen + stage 2 on the device)
                        e["compute_boost"] = cb
                        e["compute_boost_end_to_end_ms"] = cb.get("compute_boost_end_to_end_ms")
                return e
            extra = {"c2": side("c2", args.steps, args.warmup, False),
                     "n10k": side("n10k", args.steps, args.warmup, False),
                     "dropin_walk_c2": dropin_walk("c2"),
                     "c5": side("c5", 3, 1, not args.no_cpu_baseline)}
            out["extra_workloads"] = extra
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
