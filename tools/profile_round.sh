#!/bin/bash
# one profiling session for profiles/: bench line, rocprofv3 kernel-trace stats of the same command, separate PMC passes (traffic, SQ).
# usage (on the GPU box): tools/profile_round.sh <tag>   -> gpurun_out/<tag>/{bench.json,kernel_stats.csv,pmc_traffic.json,sq_counters.json}
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-prof}; out=$R/gpurun_out/$tag; mkdir -p $out
cd $R && python3 bench.py --steps 12 --warmup 3 > $out/bench.json 2> $out/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $out/kt.log 2>&1 || exit 1
db=$(ls $out/kt/*.db 2>/dev/null | head -1)
if [ -n "$db" ]; then python3 $R/tools/kstats.py $db $out/kernel_stats.csv > /dev/null; else cp $(ls $out/kt/*kernel_stats.csv | head -1) $out/kernel_stats.csv; fi
echo "kernel trace done"
bash $R/tools/pmc_passes.sh $tag/pmc "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" > $out/pmc.log 2>&1 || exit 1
python3 $R/tools/pmc_traffic.py $out/pmc 120 35 1 $out/pmc_traffic.json > /dev/null
echo "traffic passes done"
bash $R/tools/pmc_passes.sh $tag/sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
     "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" > $out/sq.log 2>&1 || exit 1
python3 $R/tools/sq_summary.py $out/sq 120 35 1 $out/sq_counters.json > /dev/null
rm -rf $out/kt/*.db $out/pmc/p*/ $out/sq/p*/ 2>/dev/null
echo "profile_round: done -> $out"
