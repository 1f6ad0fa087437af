#!/bin/bash
# A/B comparison of environment switches of ONE build on one box: tools/ab_env.sh "VAR=1" "VAR2=1" ...  ("-" = no switch);
# two interleaved rounds; prints ms/step, per-kernel ms and the state checksum of each variant
cd "$(dirname "$0")/.."
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    env $e timeout -k 10 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>>gpurun_out/ab_bench.err | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items() if b}, d['state_checksum'])" || exit 1
  done
done
