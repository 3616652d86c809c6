#!/bin/bash
# Matrix-core busy share of the window-attention kernels (north_star: >= 30 % MFMA utilisation on W-MSA):
# SQ counters in their own pass, kernel-trace only, the program directly after "--".
# usage (GPU box, repo root): tools/wattn_pmc.sh <outdir>   -> <outdir>/wattn_pmc.json
set -e
OUT=${1:-gpurun_out/wattn_pmc}
ROOT=$(pwd)
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d "$ROOT/$OUT/sq" -- python3 "$ROOT/tools/mb_attn_once.py" > "$ROOT/$OUT/sq.log" 2>&1 || true
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/$OUT/grbm" -- python3 "$ROOT/tools/mb_attn_once.py" > "$ROOT/$OUT/grbm.log" 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -- python3 "$ROOT/tools/mb_attn_once.py" > "$ROOT/$OUT/trace.log" 2>&1 || true
cd "$ROOT"
python3 tools/parse_wattn_pmc.py "$OUT" > "$OUT/wattn_pmc.json"
cat "$OUT/wattn_pmc.json"
rm -rf "$OUT/sq" "$OUT/grbm" "$OUT/trace"
