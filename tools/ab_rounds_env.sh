#!/bin/bash
# like ab_env.sh but N interleaved rounds: tools/ab_rounds_env.sh N "VAR=1" ...  ("-" = no switch)
cd "$(dirname "$0")/.."
n=$1; shift
for round in $(seq $n); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    env $e timeout -k 10 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items() if b}, (d.get('placement') or {}).get('chosen'), (d.get('placement') or {}).get('write_GBs'), d['state_checksum'])" || exit 1
  done
done
