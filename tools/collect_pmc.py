#!/usr/bin/env python3
"""HBM traffic per scan launch from rocprofv3 PMC counters, stamped with the kernel it was taken for.

On the GPU box (inside a gpurun command; one counter per pass, nothing but --pmc on the rocprofv3 line, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for FETCH_SIZE / WRITE_SIZE):

    python3 tools/collect_pmc.py run c3          # -> gpurun_out/pmc/<workload>/{FETCH_SIZE,WRITE_SIZE}.csv + entry.json

Back in the build container:

    python3 tools/collect_pmc.py merge c3 [tag]  # -> profiles/pmc_traffic.json[c3], profiles/<tag>_<wl>_pmc_*.csv

An entry records the kernel instantiation (as bench.py reads it from gw_last_scan_kernel), the sha256[:16] of
libgwaspp_amd/csrc/gw_kernels.hip and the git commit; bench.py reports roofline.traffic only while both still
match the run, null otherwise."""
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCE = os.path.join(ROOT, "libgwaspp_amd", "csrc", "gw_kernels.hip")
BENCH = {"c3": ["--workload", "c3"], "c2": ["--workload", "c2"], "n10k": ["--workload", "n10k"],
         "c1": ["--workload", "c1"], "c5": ["--workload", "c5"]}
SHAPES = {"c3": (500_000, 1_000_000), "c2": (100_000, 500_000), "n10k": (10_000, 1_000_000), "c1": (10_000, 50_000),
          "c5": (10_000, 100_000)}
PREFIX = {"c5": "gw::k_boost"}  # the kernels of a workload (default: the scan's)


def sha16():
    sys.path.insert(0, ROOT)
    import bench
    return bench.kernel_source_sha16()   # every kernel source of the library, one stamp


def norm(name):
    name = name.strip()
    if name.startswith("void "):
        name = name[5:]
    return name.split("(")[0]


def scan_rows(path, prefix="gw::k_scan_cc"):
    vals = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = norm(row["Kernel_Name"])
            if k.startswith(prefix):
                vals.setdefault(k, []).append(float(row["Counter_Value"]))
    return vals


def run(workload):
    out = os.path.join(ROOT, "gpurun_out", "pmc", workload)
    os.makedirs(out, exist_ok=True)
    entry = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, "raw_" + counter)
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", "python3",
               os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra"] + BENCH[workload]
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            sys.stderr.write(r.stdout[-2000:] + r.stderr[-2000:])
            raise SystemExit(f"rocprofv3 --pmc {counter} failed ({r.returncode})")
        bench = json.loads(line[-1])
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            raise SystemExit("no counter_collection.csv under " + d)
        shutil.copy(files[0], os.path.join(out, counter + ".csv"))
        vals = scan_rows(files[0], PREFIX.get(workload, "gw::k_scan_cc"))
        kernel = bench["roofline"]["kernel"]
        if kernel not in vals:
            raise SystemExit(f"kernel {kernel!r} not in the PMC file (has {sorted(vals)})")
        entry[counter + "_KB"] = sum(vals[kernel]) / len(vals[kernel])
        entry["kernel"] = kernel
        entry["launches_" + counter] = len(vals[kernel])
        shutil.rmtree(d, ignore_errors=True)
    n, m = SHAPES[workload]
    # gfx950: FETCH_SIZE (KB) counts 64 B per 128-B request of a 16-B/lane stream -> x2; WRITE_SIZE exact
    entry["hbm_bytes_per_launch"] = 2.0 * entry["FETCH_SIZE_KB"] * 1024.0 + entry["WRITE_SIZE_KB"] * 1024.0
    entry["correction"] = ("gfx950: FETCH_SIZE counts 64 B per 128-B request of a 16-B/lane stream -> x2 "
                           "(MI355X_MICROARCH.md, HBM); KB -> x1024; WRITE_SIZE exact")
    # c5 is bound by the matrix cores, not by HBM: its table (250 MB) is read M / 128 times over, mostly from L2 and the
    # Infinity Cache; the entry records what reached HBM
    entry["algorithmic_bytes_per_launch"] = m * (n / 4.0)
    entry["kernel_source_sha16"] = sha16()
    entry["command"] = "rocprofv3 --pmc <counter> --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra " + " ".join(BENCH[workload])
    with open(os.path.join(out, "entry.json"), "w") as f:
        json.dump(entry, f, indent=1)
    print(json.dumps(entry))


def merge(workload, tag):
    src = os.path.join(ROOT, "gpurun_out", "pmc", workload)
    with open(os.path.join(src, "entry.json")) as f:
        entry = json.load(f)
    if entry["kernel_source_sha16"] != sha16():
        raise SystemExit("the kernel source changed since the PMC passes were taken; collect again")
    entry["git"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True,
                                  text=True).stdout.strip()
    names = []
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        dst = os.path.join(ROOT, "profiles", f"{tag}_{workload}_pmc_{counter}.csv")
        shutil.copy(os.path.join(src, counter + ".csv"), dst)
        names.append(os.path.relpath(dst, ROOT))
    entry["source"] = ", ".join(names) + " (separate --pmc passes)"
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    with open(p) as f:
        allv = json.load(f)
    allv[workload] = entry
    with open(p, "w") as f:
        json.dump(allv, f, indent=1)
    print("merged", workload, entry["kernel"], entry["hbm_bytes_per_launch"] / entry["algorithmic_bytes_per_launch"])


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "run":
        run(sys.argv[2])
    elif len(sys.argv) >= 3 and sys.argv[1] == "merge":
        merge(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "r03")
    else:
        raise SystemExit(__doc__)
