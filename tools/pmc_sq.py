#!/usr/bin/env python3
"""SQ counters of the scan kernels on one table (run on the GPU box):
    python3 tools/pmc_sq.py --samples 10000 --snps 1000000 [--tag n10k]
runs `rocprofv3 --pmc <SQ counters>` (one pass, nothing else on the rocprofv3 line) over tools/sweep_scan.py with
every shape option on auto and prints, per kernel, the wave-cycle budget: parked (s_waitcnt / barrier), issue-stalled,
issuing, VALU share, instructions per row.  Output also to gpurun_out/r02/pmc_sq_<tag>.json."""
import argparse, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, required=True)
ap.add_argument("--snps", type=int, required=True)
ap.add_argument("--extra", default="", help="more sweep_scan.py options, e.g. '--pipe 1 --threads 256'")
ap.add_argument("--tag", default="x")
ap.add_argument("--counters", default="", help="custom passes: 'A,B,C;D,E' (one rocprofv3 pass per ;-group) instead of the SQ set")
ap.add_argument("--stats", default="1")
a = ap.parse_args()
PASSES = [["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVES"],
          ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_VMEM", "SQ_WAIT_INST_LDS"]]
if a.counters:
    PASSES = [g.split(",") for g in a.counters.split(";")]
out = {}
for k, counters in enumerate(PASSES):
    d = f"/tmp/pmc_sq_{a.tag}_{k}"
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "tools", "sweep_scan.py"),
           "--samples", str(a.samples), "--snps", str(a.snps), "--unroll", "0", "--threads", "0", "--per-cu", "0", "--lds", "-1",
           "--stats", a.stats, "--rounds", "1", "--reps", "3"] + a.extra.split()
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-1500:] + r.stderr[-1500:])
        raise SystemExit("rocprofv3 failed")
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].replace("void ", "").split("(")[0]
            if not name.startswith("gw::k_scan"):
                continue
            e = out.setdefault(name, {})
            c = e.setdefault(row["Counter_Name"], [0.0, 0])
            c[0] += float(row["Counter_Value"]); c[1] += 1
res = {}
for name, e in out.items():
    v = {k: s / n for k, (s, n) in e.items()}
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    res[name] = {**v, "launches": e["SQ_WAVE_CYCLES"][1] if "SQ_WAVE_CYCLES" in e else 0,
                 "parked_frac": v.get("SQ_WAIT_ANY", 0) / wc, "issue_stall_frac": v.get("SQ_WAIT_INST_ANY", 0) / wc,
                 "issuing_frac": v.get("SQ_ACTIVE_INST_ANY", 0) / wc, "valu_frac": v.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                 "valu_per_row": v.get("SQ_INSTS_VALU", 0) / a.snps, "vmem_rd_per_row": v.get("SQ_INSTS_VMEM_RD", 0) / a.snps,
                 "salu_per_row": v.get("SQ_INSTS_SALU", 0) / a.snps}
os.makedirs(os.path.join(ROOT, "gpurun_out", "r02"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r02", f"pmc_sq_{a.tag}.json"), "w") as fh:
    json.dump({"samples": a.samples, "snps": a.snps, "kernels": res}, fh, indent=1)
if a.counters:
    for name, e in out.items():
        print(name, {k: round(s_ / n_, 1) for k, (s_, n_) in e.items()})
    raise SystemExit(0)
for name, r in res.items():
    print(name, {k: (round(x, 4) if isinstance(x, float) else x) for k, x in r.items() if k.endswith(("frac", "row")) or k == "launches"})
