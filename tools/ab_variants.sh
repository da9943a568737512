#!/bin/bash
# Times the config-5 pair screen with the product library and with each variant build under libgwaspp_amd/exp/
# (tools/build_variant_file.sh), one process each, twice over (alternating, so that a drift of the box shows).
#   tools/ab_variants.sh <out file> [ab_boost.py arguments ...]
out=$1; shift
: > $out
for rep in 1 2; do
  echo "default: $(python3 tools/ab_boost.py "$@" 2>&1 | tail -1)" >> $out
  for so in libgwaspp_amd/exp/libgwscan_*.so; do
    tag=$(basename $so .so); tag=${tag#libgwscan_}
    echo "$tag: $(GW_LIBGWSCAN=$so python3 tools/ab_boost.py "$@" 2>&1 | tail -1)" >> $out
  done
done
cat $out
