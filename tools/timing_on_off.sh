#!/bin/bash
# what the per-kernel HIP-event timing inside bench.py's timed region costs: the same run with and without it, interleaved, twice
# usage (on the GPU box): tools/timing_on_off.sh > profiles/rNN_timing_on_off.txt
cd "$(dirname "$0")/.."
echo "bench.py --steps 20 --warmup 5 --no-cpu-baseline, ne120/72L/q35, one MI355X: ms per tracer step with the per-kernel event timing of the"
echo "timed region on (the default: 2 hipEventRecord per kernel group and launch, resolved after the region) and off (TSE_BENCH_KERNEL_TIMING=0)"
for round in 1 2 3; do
  for t in 1 0; do
    TSE_BENCH_KERNEL_TIMING=$t timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel timing %s: %.3f ms/step  (checksum %d)' % ('on ' if $t else 'off', d['ms_per_step'], d['state_checksum']))" || exit 1
  done
done
