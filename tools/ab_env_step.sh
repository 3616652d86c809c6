#!/bin/bash
# Same-box sweep of C-side experiment switches in the training step (experiments build):
#   bash tools/ab_env_step.sh "VAR=a VAR2=b" "VAR=c" ...     each argument = one arm's environment; arm "" = defaults
export SRHIP_LIB=$(pwd)/sr-caco-2_amd/lib/libsrhip_exp.so
for r in 1 2; do
  for arm in "" "$@"; do
    v=$(env $arm python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --train-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))")
    echo "[$arm] $v"
  done
done
