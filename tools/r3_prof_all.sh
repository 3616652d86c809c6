#!/bin/bash
# round-3 profile refresh on the GPU box: SwinIR x8 (headline), EDSR x8 / x4 (secondary) -> gpurun_out/prof_<workload>/
for wl in swinir_x8 edsr_x8 edsr_x4; do
  bash tools/refresh_profiles.sh $wl gpurun_out/prof_$wl 2>&1 | tail -14
done
