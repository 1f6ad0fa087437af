#!/bin/bash
# sample clocks / power / temperature while the benchmark runs (is the box throttling?): tools/clock_watch.sh  -> stdout
cd "$(dirname "$0")/.."
python bench.py --steps 60 --warmup 3 --no-cpu-baseline > /tmp/cw_bench.json 2>/dev/null &
pid=$!
sleep 20
for i in $(seq 12); do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.5
done
wait $pid
tail -1 /tmp/cw_bench.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {a:round(b,2) for a,b in d['kernel_ms_per_step'].items() if b})"
