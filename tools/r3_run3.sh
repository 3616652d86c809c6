#!/bin/bash
mkdir -p gpurun_out/r3c
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r3c/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3c/gpu_tests.log
tail -12 gpurun_out/r3c/gpu_tests.log
timeout 600 python bench.py > gpurun_out/r3c/bench_default.json 2> gpurun_out/r3c/bench_default.err; tail -3 gpurun_out/r3c/bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/r3c/bench_default.json'))
print(d['value'], d['dtype'], d['whole_step']); print(json.dumps(d['config'].get('secondary'), indent=1)); print({k:d['roofline'][k] for k in ('frac','achieved','avg_launch_us','launches','share_of_probed_time')}); print(d['cpu_baseline'])"
