#!/bin/bash
# fp16x2 / three products in the three-tap conv weight-gradient kernels (default) against bf16x3 / six products
# (SRHIP_TN_F16X2=0), same box: parity tests first, then the EDSR steps
timeout 1500 python -m pytest tests/test_gpu_fallback_kernels.py tests/test_gpu_bx3.py tests/test_gpu_ps2.py tests/test_gpu_edsr_api.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -q -x 2>&1 | tail -6
for w in edsr_x8 edsr_x4; do for i in 1 2 3; do for v in 0 1; do
  SRHIP_TN_F16X2=$v python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w TN_F16X2=$v', round(d['value'],1), 'loss', d['config'].get('final_loss'))"
done; done; done
