#!/bin/bash
# On the GPU box, from the repo root: rocprofv3 kernel-trace summary of one tape net's training steps (tools/tape_train_probe.py).
#   tools/prof_tape_train.sh <ACT|OmniSR|GRL> <outdir>
set -e
NET=${1:-GRL}; OUT=${2:-gpurun_out/prof_train_$NET}
ROOT=$(pwd)
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/run" -- python3 "$ROOT/tools/tape_train_probe.py" $NET 8 8 > "$ROOT/$OUT/probe.log" 2> "$ROOT/$OUT/probe.err" || true
cd "$ROOT"
python3 tools/prof_summary.py "$OUT/run" 30 > "$OUT/rocprofv3_stats.txt"
rm -rf "$OUT/run"
tail -1 "$OUT/probe.log"
