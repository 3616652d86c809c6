#!/bin/bash
# Same-box A/B of two BUILDS of libsrhip in the training step, headline number only (no secondary legs, no CPU baseline):
#   bash tools/ab_step.sh [workload] [rounds] [steps]      baseline = sr-caco-2_amd/lib/libsrhip_base.so (tools/ab_lib.sh)
BASE=$(pwd)/sr-caco-2_amd/lib/libsrhip_base.so
WL=${1:-swinir_x8}; N=${2:-3}; K=${3:-40}
one() { python bench.py --workload $WL --steps $K --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --train-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'patches/s', round(d['ms_per_step'],3), 'ms')"; }
for i in $(seq 1 $N); do
  SRHIP_LIB=$BASE one base
  one "new "
done
