#!/bin/bash
mkdir -p gpurun_out/r3i
timeout 1800 python -m pytest tests/test_gpu_tape_nets.py tests/test_gpu_amp.py -q -s > gpurun_out/r3i/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3i/tests.log
grep "worst gradient\|AssertionError\|Error\|passed\|failed\|amp vs" gpurun_out/r3i/tests.log | cut -c1-220 | tail -30
timeout 1500 python tools/eval_sweep.py --batch 8 --iters 5 --nets DBPN,SRFBN,ProSR --out gpurun_out/r3i/eval_new.json > gpurun_out/r3i/eval_new.log 2>&1; echo "rc=$?" >> gpurun_out/r3i/eval_new.log
grep -v amdgpu.ids gpurun_out/r3i/eval_new.log | tail -20 | cut -c1-200
