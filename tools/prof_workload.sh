#!/bin/bash
# On the GPU box, from the repo root: rocprofv3 kernel-trace summary of one bench workload.
#   tools/prof_workload.sh <workload> <outdir>      (e.g. edsr_x8 gpurun_out/prof_e8)
set -e
WL=${1:-swinir_x8}; OUT=${2:-gpurun_out/prof_$WL}
ROOT=$(pwd)
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/run" -- python3 "$ROOT/bench.py" --workload $WL --train-only > "$ROOT/$OUT/bench.json" 2> "$ROOT/$OUT/bench.err" || true
cd "$ROOT"
python3 tools/prof_summary.py "$OUT/run" 40 > "$OUT/rocprofv3_stats.txt"
cp $(find "$OUT/run" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
rm -rf "$OUT/run"
