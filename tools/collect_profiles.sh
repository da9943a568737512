#!/bin/bash
# Run on the GPU box (inside one gpurun call): the rocprofv3 summaries and bench lines committed under profiles/.
#   bash tools/collect_profiles.sh [tag]      -> gpurun_out/<tag>/...
# The program goes directly after `--` on every rocprofv3 line (no env / bash -c hop), PMC passes carry nothing but --pmc.
tag=${1:-r04}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
root=$PWD
stats() {  # name, bench args...
  name=$1; shift
  (cd /tmp && rm -rf /tmp/prof_$name && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 $root/bench.py "$@" > $out/${name}_bench_under_rocprof.json 2> $out/${name}_rocprof.err)
  cp $(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1) $out/${name}_kernel_stats.csv
  rm -rf /tmp/prof_$name
}
# 1. kernel trace + stats of the default bench command (c3) and of the side workloads, each with the bench line of that run
stats c3 --steps 20 --warmup 3 --no-cpu-baseline --no-extra
stats c2 --workload c2 --steps 20 --warmup 3 --no-cpu-baseline
stats n10k --workload n10k --steps 20 --warmup 3 --no-cpu-baseline
stats c5 --workload c5 --steps 10 --warmup 2 --no-cpu-baseline
# 2. PMC traffic (separate passes, nothing but --pmc)
for w in c3 c2 n10k c5; do python3 tools/collect_pmc.py run $w > $out/pmc_$w.log 2>&1; done
# 3. SQ counters of the c5 launch itself
python3 tools/pmc_c5.py --tag $tag > $out/pmc_c5_sq.log 2>&1
# 4. the plain default bench line (what the driver runs)
python3 bench.py > $out/c3_bench.json 2> $out/c3_bench.err
tail -c 300 $out/pmc_c3.log; head -3 $out/c3_kernel_stats.csv; head -3 $out/c2_kernel_stats.csv; head -3 $out/n10k_kernel_stats.csv; head -3 $out/c5_kernel_stats.csv; tail -c 600 $out/c3_bench.json
