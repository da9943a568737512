#!/usr/bin/env python3
"""Why does the short-row scan (N = 10 000 x 1 M SNPs, k_scan_cc_sub<16, 5, true, false>) run at three reproducible
levels (0.406 / 0.430 / 0.448 ms) depending on WHICH table allocation meets WHICH record buffer
(profiles/r03_placement_n10k.txt)?  Round 4: hardware counters per (table, buffer) pair.

    python3 tools/placement_pmc.py            # driver: one rocprofv3 pass per counter list, each over the child below
    python3 tools/placement_pmc.py --child    # three tables x four record buffers, 8 launches per pair

Every pass is a process of its own, i.e. a placement of its own: the comparison is BETWEEN THE PAIRS OF ONE PASS -- kernel
duration (rocprofv3 --kernel-trace of the same run) beside the counters of the same dispatches.  Address-translation
counters (TCP_UTCL1_*: the vector L1's TLB; GRBM_UTCL2_BUSY), the L2's traffic to the memory side and its stalls
(TCC_EA0_*), L2 hit / miss.  -> gpurun_out/<tag>/placement_pmc.json + .txt (copied to profiles/<tag>_placement_pmc.* when kept)."""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, M, LAUNCHES = 10_000, 1_000_000, 8
PASSES = [
    ["TCP_UTCL1_REQUEST_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_UTCL1_TRANSLATION_MISS_sum",
     "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"],
    ["TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum", "TCP_UTCL1_STALL_MULTI_MISS_sum", "TCP_UTCL1_STALL_INFLIGHT_MAX_sum", "GRBM_UTCL2_BUSY"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_STALL_sum", "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_TAG_STALL_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"],
]


def child():
    sys.path.insert(0, ROOT)
    import libgwaspp_amd as gw
    from libgwaspp_amd import capi
    tabs = []
    for _ in range(3):
        t = gw.GwTable(N, M)
        t.simulate(0x5EED, 0, 0, exact=True)
        t.simulate_phenotype(0x5EED)
        tabs.append(t)
    bufs = [capi.DeviceBuffer(M * 96) for _ in range(4)]
    for ti, t in enumerate(tabs):
        for bi, b in enumerate(bufs):
            t.scan_case_control_device(0, M, b.ptr, b.ptr + M * 32, None)      # warm (not counted: dropped by the driver)
            t.timer_start(None)
            for _ in range(LAUNCHES):
                t.scan_case_control_device(0, M, b.ptr, b.ptr + M * 32, None)
            ms = t.timer_stop(None) / LAUNCHES
            print(f"PAIR table {ti} buffer {bi} ptr {b.ptr:#x} events_ms {ms:.4f}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--tag", default="r04")
    a = ap.parse_args()
    if a.child:
        return child()
    out_dir = os.path.join(ROOT, "gpurun_out", a.tag)
    os.makedirs(out_dir, exist_ok=True)
    report, text = [], []
    for k, counters in enumerate(PASSES):
        d = f"/tmp/placement_pmc_{k}"
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", "python3",
               os.path.abspath(__file__), "--child"]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
        pairs = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("PAIR")]
        cfiles = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        kfiles = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not cfiles or not kfiles or len(pairs) != 12:
            text.append(f"pass {k} ({' '.join(counters)}) FAILED: rc {r.returncode} {(r.stderr or r.stdout)[-300:]}")
            continue
        # dispatches of the scan kernel in order: 12 pairs x (1 warm + LAUNCHES)
        dur = {}
        with open(kfiles[0], newline="") as fh:
            for row in csv.DictReader(fh):
                if "k_scan_cc_sub" in row["Kernel_Name"]:
                    dur[int(row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
        vals = {}
        with open(cfiles[0], newline="") as fh:
            for row in csv.DictReader(fh):
                if "k_scan_cc_sub" in row["Kernel_Name"]:
                    vals.setdefault(int(row["Dispatch_Id"]), {})[row["Counter_Name"]] = float(row["Counter_Value"])
        ids = sorted(dur)
        per = 1 + LAUNCHES
        if len(ids) != 12 * per:
            text.append(f"pass {k}: {len(ids)} scan dispatches seen, expected {12 * per}")
            continue
        rows = []
        for pi, p in enumerate(pairs):
            mine = ids[pi * per + 1:(pi + 1) * per]
            row = {"table": int(p[2]), "buffer": int(p[4]), "events_ms_in_this_run": float(p[8]),
                   "kernel_ms": sum(dur[i] for i in mine) / len(mine)}
            for c in counters:
                row[c] = sum(vals.get(i, {}).get(c, 0.0) for i in mine) / len(mine)
            rows.append(row)
        report.append({"counters": counters, "pairs": rows})
        text.append(f"# pass {k}: rocprofv3 --pmc {' '.join(counters)} --kernel-trace; per (table, buffer) pair: mean of {LAUNCHES} launches")
        text.append(f"{'table':>5} {'buf':>3} {'kernel_ms':>9} " + " ".join(f"{c[:34]:>34}" for c in counters))
        for row in sorted(rows, key=lambda x: x["kernel_ms"]):
            text.append(f"{row['table']:>5} {row['buffer']:>3} {row['kernel_ms']:>9.4f} " + " ".join(f"{row[c]:>34.4g}" for c in counters))
        fast, slow = min(rows, key=lambda x: x["kernel_ms"]), max(rows, key=lambda x: x["kernel_ms"])
        text.append("  slowest / fastest pair: time x%.3f; " % (slow["kernel_ms"] / fast["kernel_ms"]) +
                    "; ".join(f"{c} x{(slow[c] / fast[c]) if fast[c] else float('nan'):.3f}" for c in counters))
        shutil.rmtree(d, ignore_errors=True)
    with open(os.path.join(out_dir, "placement_pmc.json"), "w") as fh:
        json.dump({"workload": f"{N} samples x {M} SNPs, three tables x four record buffers, {LAUNCHES} launches per pair",
                   "passes": report}, fh, indent=1)
    with open(os.path.join(out_dir, "placement_pmc.txt"), "w") as fh:
        fh.write("\n".join(text) + "\n")
    print("\n".join(text))


if __name__ == "__main__":
    main()
