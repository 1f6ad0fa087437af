#!/bin/bash
# DESIGN section 4's per-rank table: every rank of the 8-rank partition of ne120/q35, one rank of the 4- and of the 2-rank partition,
# each alone on the one GPU with its neighbour slots in RCCL loopback (tools/rank_rehearsal.py); one JSON object per line.
# usage (on the GPU box): tools/rehearsal_table.sh <out.jsonl>      (about 5 minutes)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=${1:-$R/gpurun_out/rank_rehearsal.jsonl}
: > $out
python3 $R/tools/rank_rehearsal.py --ne 120 --qsize 35 --world 8 --cycles 2 >> $out || exit 1
python3 $R/tools/rank_rehearsal.py --ne 120 --qsize 35 --world 4 --rank 1 --cycles 2 >> $out || exit 1
python3 $R/tools/rank_rehearsal.py --ne 120 --qsize 35 --world 2 --rank 0 --cycles 2 >> $out || exit 1
python3 $R/tools/rank_rehearsal.py --ne 120 --qsize 35 --world 1 --rank 0 --cycles 2 >> $out || exit 1
echo "rehearsal_table: done -> $out"
