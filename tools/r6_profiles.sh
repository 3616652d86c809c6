#!/bin/bash
# Round 6: everything behind profiles/r06_* in one GPU call (repo root on the GPU box).
for WL in swinir_x8 edsr_x8; do
  bash tools/refresh_profiles.sh $WL gpurun_out/prof_$WL > gpurun_out/r6_refresh_$WL.log 2>&1
  bash tools/step_mfma_pmc.sh gpurun_out/mfma_$WL $WL > gpurun_out/r6_mfma_$WL.log 2>&1
done
python bench.py > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err
tail -c 400 gpurun_out/r6_bench_default.json
