#!/bin/bash
# this round's tree against an earlier one on the same box: tools/ab_rounds.sh <path to the other tree> [rounds=3]
# (the other tree: git archive <commit> | tar -x -C tools/ab/<name>, then build its library there)
cd "$(dirname "$0")/.."
other=$1; n=${2:-3}
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items() if b})"; }
for r in $(seq $n); do
  (cd $other && timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | line other) || exit 1
  timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | line this || exit 1
done
