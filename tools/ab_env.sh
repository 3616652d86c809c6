#!/bin/bash
# A/B of one environment switch on ONE box: bash tools/ab_env.sh VAR=value [rounds]   (bench.py --no-roofline --no-cpu-baseline)
kv=$1; n=${2:-3}
for i in $(seq 1 $n); do
  a=$(python bench.py --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  b=$(env $kv python bench.py --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  echo "default $a   $kv $b"
done
