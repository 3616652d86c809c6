#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_tape_nets.py -x -q -m gpu -k "dfcan" 2>&1 | tail -25
