#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_amp.py -x -q -m gpu -s 2>&1 | grep -i "amp vs\|passed\|failed" | head
