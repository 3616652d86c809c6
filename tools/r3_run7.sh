#!/bin/bash
mkdir -p gpurun_out/r3g
timeout 1500 python -m pytest tests/test_lowres.py tests/test_patch_sampler.py tests/test_eval_fixture.py -q -x > gpurun_out/r3g/test_f2.log 2>&1; echo "rc=$?" >> gpurun_out/r3g/test_f2.log
tail -30 gpurun_out/r3g/test_f2.log | cut -c1-260
