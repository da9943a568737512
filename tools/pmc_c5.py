#!/usr/bin/env python3
"""SQ counters of the config-5 pair screen itself (run on the GPU box):
    python3 tools/pmc_c5.py [--snps 100000 --samples 10000] [--tag r03]
`rocprofv3 --pmc <counters>` (separate passes, nothing else on the rocprofv3 line, the program directly after --) over
tools/ab_boost.py (the default kernel); per k_boost_* kernel the sums over all SEs and what follows from them:
matrix-core busy share, vector-ALU issue share, VALU instructions per MFMA, wave-cycle budget.
-> gpurun_out/<tag>/c5_pmc.json (copied to profiles/<tag>_c5_pmc.json when kept)."""
import argparse, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--snps", type=int, default=100_000)
ap.add_argument("--samples", type=int, default=10_000)
ap.add_argument("--missing-q16", type=int, default=0)
ap.add_argument("--tag", default="r04")
a = ap.parse_args()
PASSES = [["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
           "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU"],
          ["SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_I8", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_LDS",
           "SQ_WAVES", "GRBM_GUI_ACTIVE"]]
out, notes = {}, []
for k, counters in enumerate(PASSES):
    d = f"/tmp/pmc_c5_{a.tag}_{k}"
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", "python3",
           os.path.join(ROOT, "tools", "ab_boost.py"), "--snps", str(a.snps), "--samples", str(a.samples),
           "--missing-q16", str(a.missing_q16), "--rounds", "2"]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        notes.append(f"pass {k} ({' '.join(counters)}) failed: {(r.stderr or r.stdout)[-400:]}")
        continue
    with open(files[0], newline="") as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].replace("void ", "").split("(")[0]
            if not name.startswith("gw::k_boost"):
                continue
            e = out.setdefault(name, {})
            c = e.setdefault(row["Counter_Name"], [0.0, 0])
            c[0] += float(row["Counter_Value"]); c[1] += 1
    shutil.rmtree(d, ignore_errors=True)
res = {}
pairs = a.snps * (a.snps - 1) // 2
for name, e in out.items():
    v = {k: s / n for k, (s, n) in e.items()}
    launches = max(n for _, n in e.values())
    wc = v.get("SQ_WAVE_CYCLES") or 1.0
    mfma = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 32.0    # v_mfma_i32_32x32x32_i8: 32 busy cycles each
    d = {"launches_seen": launches,
         "wave_time_parked": v.get("SQ_WAIT_ANY", 0) / wc, "wave_time_issue_stalled": v.get("SQ_WAIT_INST_ANY", 0) / wc,
         "wave_time_issuing": v.get("SQ_ACTIVE_INST_ANY", 0) / wc, "wave_time_issuing_valu": v.get("SQ_ACTIVE_INST_VALU", 0) / wc,
         "valu_insts_per_mfma": (v.get("SQ_INSTS_VALU", 0) / mfma) if mfma else None,
         "valu_insts_per_pair": v.get("SQ_INSTS_VALU", 0) / pairs, "mfma_per_pair": mfma / pairs}
    res[name] = {**v, "derived": d}
path = os.path.join(ROOT, "gpurun_out", a.tag, "c5_pmc.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
with open(path, "w") as fh:
    json.dump({"command": "rocprofv3 --pmc <one pass per list> --output-format csv -- python3 tools/ab_boost.py --snps %d --samples %d --missing-q16 %d --rounds 2"
               % (a.snps, a.samples, a.missing_q16), "passes": PASSES, "notes": notes,
               "units": "sums over all SEs / XCDs per launch (averaged over the launches seen); SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_* in units of 4 cycles; "
                        "SQ_VALU_MFMA_BUSY_CYCLES in cycles (32 per v_mfma_i32_32x32x32_i8)",
               "pairs_per_launch": pairs, "kernels": res}, fh, indent=1)
for name, r in res.items():
    print(name, json.dumps(r["derived"]))
for n in notes:
    print("NOTE", n)
