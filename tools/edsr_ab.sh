#!/bin/bash
# EDSR-baseline training throughput (main.py on synthetic patches): one environment switch on / off
#   bash tools/edsr_ab.sh SRHIP_BX3_MIN_CH_NT=100000
kv=${1:-SRHIP_CONV_XCD_F32=1}
for s in 8 4 2; do
  echo -n "x$s default: "
  python sr-caco-2_amd/main.py --method EDSR_LIIF --net_type EDSR_LIIF --scale $s --h_size 512 --batch_size 8 --max_iters 30 2>&1 | grep -i "patches/s" | tail -1
  echo -n "x$s $kv: "
  env $kv python sr-caco-2_amd/main.py --method EDSR_LIIF --net_type EDSR_LIIF --scale $s --h_size 512 --batch_size 8 --max_iters 30 2>&1 | grep -i "patches/s" | tail -1
done
