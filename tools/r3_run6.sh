#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_mlp_f16.py -x -q -m gpu 2>&1 | tail -5
SRHIP_LIB=$(pwd)/sr-caco-2_amd/lib/libsrhip_exp.so timeout 300 python tools/mb_mlp_phases.py 2>&1 | grep -A 16 forward | cut -c1-100
timeout 300 python tools/mb_mlp_f16.py 2>&1 | tail -5
