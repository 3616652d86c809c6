#!/bin/bash
mkdir -p gpurun_out/r3f
timeout 2400 python tools/eval_sweep.py --batch 8 --iters 5 --nets DBPN,SRFBN,ProSR --out gpurun_out/r3f/eval_new.json > gpurun_out/r3f/eval_new.log 2>&1; echo "rc=$?" >> gpurun_out/r3f/eval_new.log
grep -v amdgpu.ids gpurun_out/r3f/eval_new.log | tail -25 | cut -c1-220
