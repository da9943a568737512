#!/bin/bash
# Run on the GPU box (inside one gpurun call): the rocprofv3 summaries and bench lines committed under profiles/.
#   bash tools/collect_profiles.sh [tag]      -> gpurun_out/<tag>/...
tag=${1:-r02}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
root=$PWD
# 1. kernel trace + stats of the default bench command (c3), and the bench line printed inside that run
(cd /tmp && rm -rf /tmp/prof_c3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $out/c3_bench_under_rocprof.json 2> $out/c3_rocprof.err)
cp $(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1) $out/c3_kernel_stats.csv
# 2. the same for c5 (pair path)
(cd /tmp && rm -rf /tmp/prof_c5 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python3 $root/bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $out/c5_bench_under_rocprof.json 2> $out/c5_rocprof.err)
cp $(find /tmp/prof_c5 -name "*kernel_stats.csv" | head -1) $out/c5_kernel_stats.csv
# 3. PMC traffic (separate passes, nothing but --pmc): c3 and the short-row side workload
python3 tools/collect_pmc.py run c3 > $out/pmc_c3.log 2>&1
python3 tools/collect_pmc.py run n10k > $out/pmc_n10k.log 2>&1
# 4. the plain default bench line
python3 bench.py > $out/c3_bench.json 2> $out/c3_bench.err
tail -c 400 $out/pmc_c3.log; head -4 $out/c3_kernel_stats.csv; head -3 $out/c5_kernel_stats.csv; tail -c 300 $out/c3_bench.json
