#!/bin/bash
# fp16x2 on the wider convs (64 -> 256 with the fused PixelShuffle, its data gradient, 128 / 256-column convs) against
# bf16x3 there (SRHIP_F16X2_CONV_WIDE=0), same box: parity tests first, then the EDSR steps
timeout 1500 python -m pytest tests/test_gpu_ps2.py tests/test_gpu_edsr_api.py tests/test_gpu_fullsize.py tests/test_gpu_fallback_kernels.py tests/test_gpu_bx3.py -q -x 2>&1 | tail -4
for w in edsr_x8 edsr_x4; do for i in 1 2 3; do for v in 0 1; do
  SRHIP_F16X2_CONV_WIDE=$v python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w WIDE=$v', round(d['value'],1), 'loss', d['config'].get('final_loss'))"
done; done; done
