#!/bin/bash
# round-3 call 1: parity of the fused fp16x2 MLP kernels, microbench, same-box A/B of the training step
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_mlp_f16.py -x -q > gpurun_out/r3a/test_mlp.log 2>&1; echo "test rc=$?" >> gpurun_out/r3a/test_mlp.log
timeout 300 python tools/mb_mlp_f16.py > gpurun_out/r3a/mb_mlp.log 2>&1
for r in 1 2 3; do
  SRHIP_MLP_F16=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > gpurun_out/r3a/bench_off_$r.json 2>gpurun_out/r3a/bench_off_$r.err
  SRHIP_MLP_F16=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > gpurun_out/r3a/bench_on_$r.json 2>gpurun_out/r3a/bench_on_$r.err
done
timeout 900 python -m pytest tests/test_gpu_swinir.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r3a/test_swinir.log 2>&1; echo "test rc=$?" >> gpurun_out/r3a/test_swinir.log
tail -5 gpurun_out/r3a/test_mlp.log; cat gpurun_out/r3a/mb_mlp.log; grep -h -o '"value": [0-9.]*' gpurun_out/r3a/bench_*.json; tail -5 gpurun_out/r3a/test_swinir.log
