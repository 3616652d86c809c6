#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_swinir.py -x -q -m gpu -k "ddp or graph" 2>&1 | tail -25
