#!/bin/bash
# Same-box per-kernel comparison of two builds in one step: tools/cmp_trace.sh <workload> <libA.so|-> <libB.so|->
WL=$1; A=$2; B=$3
for T in A B; do
  L=$([ $T = A ] && echo $A || echo $B)
  if [ "$L" = "-" ]; then unset SRHIP_LIB; else export SRHIP_LIB=$(pwd)/sr-caco-2_amd/lib/$L; fi
  bash tools/step_trace.sh $WL gpurun_out/trace_$T > /dev/null
  grep "^#" gpurun_out/trace_$T/step.txt | head -${4:-14} > gpurun_out/cmp_$T.txt
done
paste gpurun_out/cmp_A.txt gpurun_out/cmp_B.txt | cut -c1-150
