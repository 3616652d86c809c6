#!/bin/bash
# same-box A/B of two BUILDS of libsrhip in the training step: the in-tree library against sr-caco-2_amd/lib/libsrhip_base.so
# (build the baseline from another commit, copy it there; .so files travel to the GPU box, they are not in git)
BASE=$(pwd)/sr-caco-2_amd/lib/libsrhip_base.so
WL=${1:-swinir_x8}
for i in 1 2 3; do
  SRHIP_LIB=$BASE python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base', round(d['value'],1), round(d['roofline']['avg_launch_us'],2))"
  python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new ', round(d['value'],1), round(d['roofline']['avg_launch_us'],2))"
done
