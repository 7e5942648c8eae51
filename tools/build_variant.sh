#!/bin/bash
# Builds another librt_amd.so with extra compiler flags for rt_kernels.hip (e.g. -DPAIR_LEAVES=4u) as _ab/librt_<name>.so,
# for A/B runs on one box (tools/ab_pair.sh, tools/ab_bench.py).  usage: tools/build_variant.sh <name> <flags...>
set -e
name=$1; shift
cd "$(dirname "$0")/../raytracing_amd/csrc"
make -s -j4
mkdir -p ../../_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function "$@" -c rt_kernels.hip -o /tmp/rt_kernels_$name.o
objs=$(ls _build/*.o | grep -v rt_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../_ab/librt_$name.so /tmp/rt_kernels_$name.o $objs -lz -ldl -pthread
echo built _ab/librt_$name.so
