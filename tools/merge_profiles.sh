#!/bin/bash
# Container side, after `gpurun -- bash tools/collect_profiles.sh <tag>` has merged gpurun_out/<tag>/ back:
# stamp the PMC entries into profiles/pmc_traffic.json and copy the judged summaries into profiles/.
#   bash tools/merge_profiles.sh [tag]
tag=${1:-r04}
for w in c3 c2 n10k c5; do python3 tools/collect_pmc.py merge $w $tag | tail -1; done
for w in c3 c2 n10k c5; do
  cp gpurun_out/$tag/${w}_kernel_stats.csv profiles/${tag}_${w}_kernel_stats.csv
  cp gpurun_out/$tag/${w}_bench_under_rocprof.json profiles/${tag}_${w}_bench_under_rocprof.json
done
cp gpurun_out/$tag/c5_pmc.json profiles/${tag}_c5_pmc_fp4.json
echo "now: gpurun -- 'mkdir -p gpurun_out/$tag && python bench.py > gpurun_out/$tag/c3_bench.json' and copy it to profiles/${tag}_c3_bench.json"
