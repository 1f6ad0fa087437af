#!/bin/bash
# A/B/C comparison of library builds on one box: tools/ab_bench.sh libA.so libB.so ...  (two interleaved rounds)
set -e
cd "$(dirname "$0")/.."
for round in 1 2; do
  for l in "$@"; do
    TSE_LIB=$PWD/tools/ab/$l timeout -k 10 200 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>>gpurun_out/ab_bench.err | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$l', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items()})"
  done
done
