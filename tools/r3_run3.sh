#!/bin/bash
mkdir -p gpurun_out/r3c
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r3c/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3c/gpu_tests.log
tail -15 gpurun_out/r3c/gpu_tests.log
