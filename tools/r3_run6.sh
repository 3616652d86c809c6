#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_wmsa_f16.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python tools/mb_wmsa_f16.py 2>&1 | tail -7
SRHIP_LIB=$(pwd)/sr-caco-2_amd/lib/libsrhip_exp.so timeout 300 python tools/mb_wmsa_phases.py 2>&1 | tail -12
