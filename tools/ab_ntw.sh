set -x
timeout 900 python -m pytest tests/test_gpu_bx3.py tests/test_gpu_kernels.py -q -x 2>&1 | tail -5
for i in 1 2; do
  SRHIP_NTW=0 python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NTW=0', d['value'], d['roofline']['avg_launch_us'])"
  SRHIP_NTW=1 python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NTW=1', d['value'], d['roofline']['avg_launch_us'])"
done
