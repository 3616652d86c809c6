# same-box A/B of the SwinIR conv variants in the training step: SRHIP_NTCW=0 (k_ntb<1,3>: W through LDS) vs 1 (k_ntcw)
timeout 600 python -m pytest tests/test_gpu_bx3.py tests/test_gpu_kernels.py -q -x -k "conv" 2>&1 | tail -2
for i in 1 2; do
  for v in 0 1; do
    SRHIP_NTCW=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NTCW=$v', round(d['value'],1))"
  done
done
