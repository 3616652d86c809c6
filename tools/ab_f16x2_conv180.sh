#!/bin/bash
# SwinIR's 180-column 3x3 convs on two fp16 planes / three products (k_nhcw) against bf16x3 / six (k_ntcw,
# SRHIP_F16X2_CONV180=0), same box: parity tests first, then the training step
timeout 1500 python -m pytest tests/test_gpu_swinir.py tests/test_gpu_fullsize.py tests/test_gpu_fallback_kernels.py tests/test_gpu_amp.py -q -x 2>&1 | tail -4
for i in 1 2 3; do for v in 0 1; do
  SRHIP_F16X2_CONV180=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('CONV180=$v', round(d['value'],1), 'loss', d['config'].get('final_loss'), 'eval', d['config'].get('eval_patches_per_s_one_gpu'), d['config'].get('eval_amp_patches_per_s_one_gpu'))"
done; done
