#!/bin/bash
mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "attention" > gpurun_out/r3d/test_attn.log 2>&1; echo "rc=$?" >> gpurun_out/r3d/test_attn.log
tail -4 gpurun_out/r3d/test_attn.log | cut -c1-300
timeout 300 python tools/mb_attn.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3d/mb_attn.log
