#!/bin/bash
# A/B in the rank rehearsal of one hooks-build switch (here TSE_AB_CHAIN; TSE_AB_SPLIT_STREAMS the same way): ranks 0 and 2 of 8, rank 1 of 4, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
export TSE_LIB=$R/transport_se_amd/libtransport_se_hip_hooks.so
for round in 1 2; do
  for ab in 0 1; do
    for spec in "8 0" "8 2" "4 1"; do
      set -- $spec
      TSE_AB_CHAIN=$ab python3 $R/tools/rank_rehearsal.py --ne 120 --qsize 35 --world $1 --rank $2 --cycles 6 2>/dev/null | grep '^{' | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('chain=$ab', 'world', d['world'], 'rank', d['rank'], 'ms_per_step', d['ms_per_step'], d['kernel_ms_per_step_timing_mode'])"
    done
  done
done
