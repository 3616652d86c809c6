#!/bin/bash
# On the GPU box, from the repo root: the kernel sequence of ONE step (durations, gaps) of a bench workload.
# usage: tools/step_trace.sh <workload> <outdir> [extra bench flags]
WL=${1:-swinir_x8}
OUT=${2:-gpurun_out/trace_$WL}
shift 2
ROOT=$(pwd)
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$OUT/run" -- python3 "$ROOT/bench.py" --workload $WL --train-only --steps 6 --warmup 3 "$@" > "$ROOT/$OUT/bench.json" 2> "$ROOT/$OUT/bench.err" || true
cd "$ROOT"
python3 tools/step_trace.py "$(find "$OUT/run" -name '*kernel_trace.csv' | head -1)" > "$OUT/step.txt"
rm -rf "$OUT/run"
tail -1 "$OUT/bench.json" | cut -c1-160
tail -5 "$OUT/step.txt"
