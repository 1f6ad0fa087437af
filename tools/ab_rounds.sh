#!/bin/bash
# same-box comparison of the round-2 tree (tools/ab/r02tree = `git archive 98d1d21`, built there) with the current tree:
# two interleaved rounds of bench.py --steps 12 --warmup 3 --no-cpu-baseline each
cd "$(dirname "$0")/.."
for round in 1 2; do
  for t in tools/ab/r02tree .; do
    (cd $t && timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$t', round(d['ms_per_step'],2), '%.4g' % d['value'], {a:round(b,2) for a,b in k.items() if b})") || exit 1
  done
done
