#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_mlp_f16.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_swinir.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --train-only 2>&1 | tail -1 | cut -c1-300
SRHIP_CHAIN_PROJ=0 timeout 600 python bench.py --train-only 2>&1 | tail -1 | cut -c1-300
