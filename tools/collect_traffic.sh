#!/bin/bash
# HBM traffic of the dominant kernel, per MI355X_MICROARCH.md "HBM": FETCH_SIZE and
# WRITE_SIZE in SEPARATE --pmc passes (TCC slots), kernel-trace only.
# usage (on the GPU box, from the repo root): tools/collect_traffic.sh <outdir>
set -e
OUT=${1:-gpurun_out/traffic}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$ROOT/$OUT/$c" -- \
    python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > "$ROOT/$OUT/$c.log" 2>&1
done
