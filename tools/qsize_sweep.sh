#!/bin/bash
# per-kernel ms per step against the tracer count (ne120, one GPU): the intercept is what a step costs before the first tracer
# (per-(element, level) prologues, level fields, tables), the slope the cost of a tracer
cd "$(dirname "$0")/.."
for q in 1 2 4 8 16 35; do
  timeout -k 10 300 python bench.py --qsize $q --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print($q, round(d['ms_per_step'],3), {a:round(b,3) for a,b in k.items() if b})" || exit 1
done
