#!/bin/bash
mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "attention" > gpurun_out/r3d/test_attn.log 2>&1; echo "rc=$?" >> gpurun_out/r3d/test_attn.log
tail -25 gpurun_out/r3d/test_attn.log | cut -c1-400
timeout 300 python tools/mb_attn.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3d/mb_attn.log
for r in 1 2; do
  SRHIP_WATTN_F16=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 30 --warmup 5 > gpurun_out/r3d/bench_off_$r.json 2>/dev/null
  SRHIP_WATTN_F16=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 30 --warmup 5 > gpurun_out/r3d/bench_on_$r.json 2>/dev/null
done
grep -h -o '"value": [0-9.]*\|"final_loss": [0-9.]*\|eval_patches_per_s_one_gpu": [0-9.]*' gpurun_out/r3d/bench_*.json
