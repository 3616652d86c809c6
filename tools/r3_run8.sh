#!/bin/bash
mkdir -p gpurun_out/r3h
timeout 900 python tools/amp_gate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3h/amp_gate.log
