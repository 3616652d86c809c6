#!/bin/bash
# On the GPU box, from the repo root: everything behind profiles/rNN_* for one workload --
#   bench line under rocprofv3 --kernel-trace --stats, its per-kernel summary, and the HBM-traffic PMC passes.
# usage: tools/refresh_profiles.sh <workload> <outdir>     then copy <outdir>/* into profiles/ with the round prefix
set -e
WL=${1:-swinir_x8}
OUT=${2:-gpurun_out/prof_$WL}
ROOT=$(pwd)
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# per-kernel figures are taken with every launch IN ORDER on one stream (SwinIR: a layer's weight gradients otherwise run on a
# side stream beside the next layer's chain, and kernels that share the chip have no duration of their own); bench.py probes
# its roofline the same way
export SRHIP_SWIN_SIDE_WGRAD=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/run" -- python3 "$ROOT/bench.py" --workload $WL --train-only > "$ROOT/$OUT/bench.json" 2> "$ROOT/$OUT/bench.err" || true
cd "$ROOT"
python3 tools/prof_summary.py "$OUT/run" 30 > "$OUT/rocprofv3_stats.txt"
cp $(find "$OUT/run" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
bash tools/collect_traffic.sh "$OUT/traffic" $WL
python3 tools/parse_traffic.py "$OUT/traffic" "$OUT/hbm_traffic_per_kernel.json" > "$OUT/hbm_traffic_top.txt" 2> "$OUT/parse.err" || true
rm -rf "$OUT/run" "$OUT/traffic"
tail -1 "$OUT/bench.json" | cut -c1-200
head -10 "$OUT/rocprofv3_stats.txt"
cat "$OUT/hbm_traffic_top.txt"
