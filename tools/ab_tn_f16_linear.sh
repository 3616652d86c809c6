#!/bin/bash
# Linear weight gradients (192-column tiles of the grouped TN kernel) on two fp16 planes / three products
# (SRHIP_TN_F16X2_LINEAR=1, tnb_body_h) against bf16x3 / six, same box: parity tests first, then the SwinIR step
SRHIP_TN_F16X2_LINEAR=1 timeout 1500 python -m pytest tests/test_gpu_bx3.py tests/test_gpu_swinir.py tests/test_gpu_fullsize.py tests/test_gpu_mlp_fused.py -q -x 2>&1 | tail -12
for i in 1 2 3; do for v in 0 1; do
  SRHIP_TN_F16X2_LINEAR=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TN_F16X2_LINEAR=$v', round(d['value'],1), 'loss', d['config'].get('final_loss'))"
done; done
