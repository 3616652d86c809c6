#!/bin/bash
# On the GPU box, from the repo root: the rocprofv3 summary of the default bench.py run and the HBM-traffic PMC
# passes behind profiles/ (copy gpurun_out/prof_refresh/* into profiles/ afterwards: tools/parse_traffic.py,
# tools/prof_summary.py).
set -e
ROOT=$(pwd)
OUT=gpurun_out/prof_refresh
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/run" -- python3 "$ROOT/bench.py" > "$ROOT/$OUT/bench.json" 2> "$ROOT/$OUT/bench.err" || true
cd "$ROOT"
python3 tools/prof_summary.py "$OUT/run" 24 > "$OUT/rocprofv3_stats.txt"
cp $(find "$OUT/run" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
bash tools/collect_traffic.sh "$OUT/traffic"
python3 tools/parse_traffic.py "$OUT/traffic" "$OUT/hbm_traffic_per_kernel.json" > "$OUT/hbm_traffic_top.txt" 2> "$OUT/parse.err" || true
rm -rf "$OUT/run" "$OUT/traffic"
tail -1 "$OUT/bench.json" | cut -c1-300
head -12 "$OUT/rocprofv3_stats.txt"
cat "$OUT/hbm_traffic_top.txt"
du -sh "$OUT"
