#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_mlp_f16.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py --train-only 2>&1 | tail -1 | cut -c1-200
SRHIP_FRONT_QKV=0 timeout 600 python bench.py --train-only 2>&1 | tail -1 | cut -c1-200
timeout 600 python bench.py --train-only 2>&1 | tail -1 | cut -c1-200
SRHIP_FRONT_QKV=0 timeout 600 python bench.py --train-only 2>&1 | tail -1 | cut -c1-200
