#!/bin/bash
# Builds libgwaspp_amd/exp/libgwscan_<tag>.so: the product library with ONE translation unit compiled under extra -D flags
# (kernel experiments; select with GW_LIBGWSCAN=<path>).  usage: tools/build_variant_file.sh <tag> <file.hip> -DGW_X=1 ...
set -e
cd "$(dirname "$0")/../libgwaspp_amd/csrc"
tag=$1; file=$2; shift; shift
base=$(basename $file .hip)
mkdir -p build ../exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -x hip "$@" -c $file -o build/${base}_$tag.o.var
objs=$(ls build/*.o | grep -v "build/${base}.o" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $objs build/${base}_$tag.o.var -ldl -o ../exp/libgwscan_$tag.so
echo built ../exp/libgwscan_$tag.so
