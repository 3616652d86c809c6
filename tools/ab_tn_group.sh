#!/bin/bash
# grouped Linear weight gradients: XCD-aware (slice, tile) order (SRHIP_TN_GROUP_XCD, default 1) x fp16x2 / three products
# (SRHIP_TN_F16X2_LINEAR), same box: parity tests with both on, then the SwinIR step in the four arms
SRHIP_TN_F16X2_LINEAR=1 timeout 900 python -m pytest tests/test_gpu_bx3.py tests/test_gpu_swinir.py -q -x 2>&1 | tail -3
for i in 1 2 3; do for x in 0 1; do for v in 0 1; do
  SRHIP_TN_GROUP_XCD=$x SRHIP_TN_F16X2_LINEAR=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('GROUP_XCD=$x F16X2_LINEAR=$v', round(d['value'],1), 'loss', d['config'].get('final_loss'))"
done; done; done
