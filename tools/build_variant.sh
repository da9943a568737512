#!/bin/bash
# Builds libgwaspp_amd/exp/libgwscan_<tag>.so: the product library with gw_kernels.hip compiled under extra -D flags
# (kernel experiments; select with GW_LIBGWSCAN=<path>).  usage: tools/build_variant.sh <tag> -DGW_X=1 ...
set -e
cd "$(dirname "$0")/../libgwaspp_amd/csrc"
tag=$1; shift
mkdir -p build ../exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip "$@" -c gw_kernels.hip -o build/gw_kernels_$tag.o
objs=$(ls build/*.o | grep -v "gw_kernels" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $objs build/gw_kernels_$tag.o -o ../exp/libgwscan_$tag.so
echo built ../exp/libgwscan_$tag.so
