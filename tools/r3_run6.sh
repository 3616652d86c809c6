#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_wmsa_f16.py -x -q -m gpu -k "attention or wmsa" 2>&1 | tail -3
SRHIP_LIB=$(pwd)/sr-caco-2_amd/lib/libsrhip_exp.so timeout 300 python tools/mb_attn_phases.py 2>&1 | tail -3
timeout 300 python tools/mb_attn.py 2>&1 | tail -12
