#!/bin/bash
# Samples rocm-smi (power, clocks, temperature) while bench.py runs: is the step power/clock limited?
out=${1:-gpurun_out/power_trace.txt}
mkdir -p "$(dirname "$out")"
python bench.py --steps 900 --warmup 5 > gpurun_out/power_bench.json 2>/dev/null &
pid=$!
t0=$(date +%s.%N)
while kill -0 $pid 2>/dev/null; do
  t=$(echo "$(date +%s.%N) - $t0" | bc)
  s=$(rocm-smi --showpower --showclocks --showtemp --showuse 2>/dev/null | grep -E "Package Power|sclk|GPU use|junction" | sed -e 's/GPU\[0\]\t*: //' | tr '\n' ';')
  echo "t=$t $s"
  sleep 0.7
done > "$out"
wait $pid
tail -1 gpurun_out/power_bench.json | cut -c1-160
cat "$out"
