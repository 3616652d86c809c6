#!/bin/bash
# A/B of two builds on ONE box: put the other build at sr-caco-2_amd/lib/libsrhip_old.so, then
#   bash tools/ab_lib.sh [rounds]      (bench.py --no-roofline --no-cpu-baseline, alternating)
L=sr-caco-2_amd/lib; n=${1:-3}
val() { python bench.py --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])"; }
cp $L/libsrhip.so $L/new.so
for i in $(seq 1 $n); do
  cp $L/new.so $L/libsrhip.so; a=$(val)
  cp $L/libsrhip_old.so $L/libsrhip.so; b=$(val)
  echo "new $a   old $b"
done
cp $L/new.so $L/libsrhip.so
