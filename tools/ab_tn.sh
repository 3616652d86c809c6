#!/bin/bash
# grouped weight-gradient microbenchmark, new vs sr-caco-2_amd/lib/libsrhip_old.so on one box
L=sr-caco-2_amd/lib
cp $L/libsrhip.so $L/new.so
for i in 1 2 3; do
  cp $L/new.so $L/libsrhip.so; python tools/mb_tn_roles.py 2>&1 | grep "DBG=0" | sed "s/^/new /"
  cp $L/libsrhip_old.so $L/libsrhip.so; python tools/mb_tn_roles.py 2>&1 | grep "DBG=0" | sed "s/^/old /"
done
cp $L/new.so $L/libsrhip.so
