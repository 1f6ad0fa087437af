#!/bin/bash
# per-kernel ms per step against the number of elements (one GPU, no exchange): is what a small rank loses a constant per launch or a rate?
cd "$(dirname "$0")/.."
for ne in 30 42 60 84 120; do
  timeout -k 10 300 python bench.py --ne $ne --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; n=d['config']['elements_per_gpu']; print($ne, n, round(d['ms_per_step'],3), round(d['ms_per_step']/n*86400,2), {a:round(b/n*86400,2) for a,b in k.items() if b})" || exit 1
done
