#!/bin/bash
# build an A/B variant of the library from the same sources: tools/ab_build.sh <name> [-DFLAG ...]  ->  tools/ab/<name>.so
# (compare on one box with tools/ab_bench.sh <name1>.so <name2>.so)
cd "$(dirname "$0")/.." && mkdir -p tools/ab
n=$1; shift
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-honor-nans "$@" -shared -fPIC -o tools/ab/$n.so transport_se_amd/csrc/tse_api.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
