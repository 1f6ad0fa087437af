#!/bin/bash
# tools/ab_env.sh "VAR=val ..." "VAR=val ..." : bench.py under different environments on one box (two interleaved rounds)
cd "$(dirname "$0")/.."
for round in 1 2; do
  for e in "$@"; do
    env $e timeout -k 10 200 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$e', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items()})"
  done
done
