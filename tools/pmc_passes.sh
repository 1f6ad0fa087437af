#!/bin/bash
# separate rocprofv3 --pmc passes over a short bench run (never combined with trace domains other than --kernel-trace)
# usage: tools/pmc_passes.sh <outdir-under-gpurun_out> "<COUNTERS pass 1>" "<COUNTERS pass 2>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -o p$i -- python3 $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline > $out/p$i.log 2>&1 || exit 1
  echo "pass $i ($ctrs) done"
done
