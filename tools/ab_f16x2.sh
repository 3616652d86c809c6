#!/bin/bash
# the two-plane fp16 / three-product kernels against the bf16x3 / six-product ones, same box:
#   Linear GEMMs (default on, SRHIP_F16X2=0 switches it off): SwinIR parity tests with it off, then the step
#   64-column convs (experiment, SRHIP_F16X2_CONV=1): EDSR / VDSR / MSLapSRN parity tests with it on, then the EDSR steps
# usage: tools/ab_f16x2.sh [gemm|conv]
if [ "${1:-gemm}" = gemm ]; then
  SRHIP_F16X2=0 timeout 900 python -m pytest tests/test_gpu_swinir.py tests/test_gpu_fullsize.py -q -x -k "swinir or readme or tiny" 2>&1 | tail -2
  for i in 1 2 3; do for v in 0 1; do
    SRHIP_F16X2=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('F16X2=$v', round(d['value'],1), round(d['roofline']['avg_launch_us'],2), 'loss', d['config'].get('final_loss'))"
  done; done
else
  SRHIP_F16X2_CONV=1 timeout 1500 python -m pytest tests/test_gpu_edsr_api.py tests/test_gpu_fullsize.py tests/test_gpu_ps2.py tests/test_gpu_mslapsrn.py tests/test_gpu_memnet.py -q 2>&1 | tail -4
  for w in edsr_x4 edsr_x8; do for i in 1 2 3; do for v in 0 1; do
    SRHIP_F16X2_CONV=$v python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w F16X2_CONV=$v', round(d['value'],1), 'loss', d['config'].get('final_loss'))"
  done; done; done
fi
