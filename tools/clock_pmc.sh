#!/bin/bash
# Effective shader clock per kernel during a bench workload (MI355X_MICROARCH.md "DVFS give-back": effective clock ~=
# GRBM_GUI_ACTIVE / 8 / kernel wall time, the counter is summed over the 8 XCDs; reliable on dispatches >= 0.3 ms, read
# as an upper bound on shorter ones).  usage (GPU box, repo root): tools/clock_pmc.sh <outdir> [workload] [steps]
set -e
OUT=${1:-gpurun_out/clock}; WL=${2:-swinir_x8}; STEPS=${3:-6}
ROOT=$(pwd)
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/$OUT/run" -- \
  python3 "$ROOT/bench.py" --workload $WL --steps $STEPS --warmup 2 --no-cpu-baseline --no-roofline > "$ROOT/$OUT/bench.log" 2>&1
cd "$ROOT"
python3 tools/parse_clock.py "$OUT/run" "$OUT/clock_per_kernel.json"
rm -rf "$OUT/run"
