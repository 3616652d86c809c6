#!/bin/bash
mkdir -p gpurun_out/r3b
timeout 600 python tools/dbg_graph_nan.py > gpurun_out/r3b/dbg3.log 2>&1
grep -v amdgpu.ids gpurun_out/r3b/dbg3.log | cut -c1-1500
