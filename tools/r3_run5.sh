#!/bin/bash
mkdir -p gpurun_out/r3e
timeout 1500 python -m pytest tests/test_gpu_tape_nets.py -q -s > gpurun_out/r3e/test_tape.log 2>&1; echo "rc=$?" >> gpurun_out/r3e/test_tape.log
grep "worst gradient\|AssertionError: (\|passed\|failed" gpurun_out/r3e/test_tape.log | cut -c1-250
