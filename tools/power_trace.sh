#!/bin/bash
# Samples rocm-smi (power, clocks, temperature) while bench.py runs: is the step power / clock limited?
# usage (GPU box, repo root): tools/power_trace.sh [outfile] [workload] [steps]
out=${1:-gpurun_out/power_trace.txt}; WL=${2:-swinir_x8}; STEPS=${3:-1500}
mkdir -p "$(dirname "$out")"
rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | sed -e 's/GPU\[0\]\t*: //' | tr '\n' ';' > "$out"; echo >> "$out"
python bench.py --workload $WL --steps $STEPS --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/power_bench.json 2>/dev/null &
pid=$!
sleep 12
for i in $(seq 1 12); do
  kill -0 $pid 2>/dev/null || break
  s=$(rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|GPU use" | sed -e 's/GPU\[0\]\t*: //' | tr '\n' ';')
  echo "busy: $s" >> "$out"
  sleep 1
done
wait $pid
tail -1 gpurun_out/power_bench.json | cut -c1-120
cat "$out"
