#!/bin/bash
# sample clocks / power / temperature while the benchmark runs (is the box throttling?): tools/clock_watch.sh  -> stdout
cd "$(dirname "$0")/.."
python bench.py --steps 120 --warmup 3 --no-cpu-baseline > /tmp/cw_bench.json 2>/dev/null &
pid=$!
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)" | sed -e 's/GPU\[0\]\s*: //' -e 's/clock level: //' -e 's/Temperature (Sensor \(\w*\)) (C)/T\1/' -e 's/Current Socket Graphics Package Power (W)/W/' | tr -s ' ' | tr '\n' ';'
  echo
  sleep 1
done
tail -1 /tmp/cw_bench.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {a:round(b,2) for a,b in d['kernel_ms_per_step'].items() if b})"
