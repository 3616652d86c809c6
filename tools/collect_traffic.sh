#!/bin/bash
# HBM traffic per kernel of one bench workload, per MI355X_MICROARCH.md "HBM": FETCH_SIZE and
# WRITE_SIZE in SEPARATE --pmc passes (TCC slots), kernel-trace only.
# usage (on the GPU box, from the repo root): tools/collect_traffic.sh <outdir> [workload]
set -e
OUT=${1:-gpurun_out/traffic}
WL=${2:-swinir_x8}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export SRHIP_SWIN_SIDE_WGRAD=0     # launches in order on one stream: per-kernel counters are attributable (see refresh_profiles.sh)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$ROOT/$OUT/$c" -- \
    python3 "$ROOT/bench.py" --workload $WL --steps 2 --warmup 1 --train-only --no-roofline > "$ROOT/$OUT/$c.log" 2>&1
done
