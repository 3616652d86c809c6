#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_tape_nets.py -x -q -m gpu -k "nlsn" 2>&1 | tail -5
timeout 900 python tools/eval_sweep.py --nets NLSN --out gpurun_out/r03_eval_sweep_nlsn.json 2>&1 | tail -6 | cut -c1-200
