#!/bin/bash
# same-box A/B of the Linear GEMM variants in the training step (bench.py):
#   SRHIP_NTW=0 (k_ntp: W through LDS) | SRHIP_NTW=1 without / with the per-block rotation of the K walk (k_ntw)
for i in 1 2 3; do
  for v in "SRHIP_NTW=0" "SRHIP_NTW=1 SRHIP_NTW_ROT=0" "SRHIP_NTW=1 SRHIP_NTW_ROT=1"; do
    env $v python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['roofline']['avg_launch_us'],2))"
  done
done
