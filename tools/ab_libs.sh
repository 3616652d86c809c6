#!/bin/bash
# Same-box comparison of several BUILDS of libsrhip in the training step (headline number only):
#   bash tools/ab_libs.sh <workload> <rounds> <steps> lib1.so lib2.so ...     (paths under sr-caco-2_amd/lib/; "-" = the in-tree default)
WL=$1; N=$2; K=$3; shift 3
for i in $(seq 1 $N); do
  for L in "$@"; do
    if [ "$L" = "-" ]; then unset SRHIP_LIB; else export SRHIP_LIB=$(pwd)/sr-caco-2_amd/lib/$L; fi
    python bench.py --workload $WL --steps $K --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --train-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['value'],1), 'patches/s', round(d['ms_per_step'],3), 'ms')"
  done
done
