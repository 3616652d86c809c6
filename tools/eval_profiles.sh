#!/bin/bash
# On the GPU box, from the repo root: rocprofv3 --kernel-trace --stats of one evaluation forward loop per network of the sweep
# (x8, batch 8, 512x512 HR patches) -> gpurun_out/eval_prof/<net>_x8.txt (top kernels); copy into profiles/ with the round prefix.
ROOT=$(pwd)
OUT="$ROOT/gpurun_out/eval_prof"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pair in swinir:SWINIR EDSR_LIIF:EDSR_LIIF VDSR:VDSR DRRN:DRRN SRCNN:SRCNN MSLapSRN:MSLAPSR MemNet:MemNet DBPN:DBPN SRFBN:SRFBN \
            ProSR:PROSR ENLCN:ENLCN NLSN:NLSN DFCAN:DFCAN ACT:ACT OmniSR:OmniSR GRL:GRL; do
  n=${pair%%:*}; m=${pair##*:}
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/run_$n" -- python3 "$ROOT/tools/mb_eval_net.py" $n $m 8 3 > "$OUT/$n.log" 2>&1 < /dev/null
  { grep patches "$OUT/$n.log"; python3 "$ROOT/tools/prof_summary.py" "$OUT/run_$n" 12 < /dev/null | cut -c1-170; } > "$OUT/${n}_x8.txt"
  rm -rf "$OUT/run_$n" "$OUT/$n.log"
  head -3 "$OUT/${n}_x8.txt"
done
