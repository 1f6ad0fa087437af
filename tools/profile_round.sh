#!/bin/bash
# one profiling session for profiles/: bench line, rocprofv3 kernel-trace stats of the same command, separate PMC passes (traffic, SQ).
# usage (on the GPU box): tools/profile_round.sh <tag> [round]   -> gpurun_out/<tag>/{bench.json,kernel_stats.csv,pmc_traffic.json,sq_counters.json[,bench_filled.json,bench_driver_filled.json]}
# (run python -m pytest tests -m gpu first: the ne120 DCMIP 1-1 test leaves the L2 record of the build under gpurun_out/)
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
# with a round tag (e.g. r03): put the summaries where bench.py looks for them (profiles/<round>_final_*; this box's copy of the
# repository is scratch -- commit the files that come back under gpurun_out/) and print the bench lines again, now with the counter
# traffic and the L2 record of this very build in them
if [ -n "$2" ]; then
  cp $out/kernel_stats.csv $R/profiles/$2_final_kernel_stats_ne120_q35.csv
  cp $out/pmc_traffic.json $R/profiles/$2_final_pmc_traffic_ne120_q35.json
  cp $out/sq_counters.json $R/profiles/$2_final_sq_counters_ne120_q35.json
  [ -f $R/gpurun_out/l2_dcmip11_ne120.json ] && cp $R/gpurun_out/l2_dcmip11_ne120.json $R/profiles/$2_l2_dcmip11_ne120.json
  cd $R && python3 bench.py --steps 12 --warmup 3 > $out/bench_filled.json 2>> $out/bench.err || exit 1
  python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_filled.json 2>> $out/bench.err || exit 1
fi
echo "profile_round: done -> $out"
