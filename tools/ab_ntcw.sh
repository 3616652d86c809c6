#!/bin/bash
# same-box A/B of the conv variants in the training step: W fragments straight from global memory (k_ntcw: 64-pixel x 192-column
# tiles, SwinIR; k_ntcw2: 128-pixel x 64-column tiles, EDSR and the other 64-channel nets) against W through LDS (k_ntb)
# usage: tools/ab_ntcw.sh [workload] [NTCW|NTCW2]
WL=${1:-swinir_x8}; SW=${2:-NTCW}
for i in 1 2 3; do
  for v in 0 1; do
    env SRHIP_$SW=$v python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$WL $SW=$v', round(d['value'],1))"
  done
done
