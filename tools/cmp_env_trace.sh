#!/bin/bash
# Same-box per-kernel comparison of two environments in one step (experiments build):
#   tools/cmp_env_trace.sh <workload> "VAR=a" "VAR=b" [lines]
WL=$1; A=$2; B=$3
export SRHIP_LIB=$(pwd)/sr-caco-2_amd/lib/libsrhip_exp.so
for T in A B; do
  E=$([ $T = A ] && echo "$A" || echo "$B")
  ( [ -n "$E" ] && export $E; bash tools/step_trace.sh $WL gpurun_out/trace_$T > /dev/null )
  grep "^#" gpurun_out/trace_$T/step.txt | head -${4:-14} > gpurun_out/cmp_$T.txt
done
paste gpurun_out/cmp_A.txt gpurun_out/cmp_B.txt | cut -c1-150
