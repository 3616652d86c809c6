# experiment: the Linear GEMMs of SwinIR on two fp16 planes / three products with per-row block exponents (k_nth,
# SRHIP_F16X2=1) against the default bf16x3 / six products (k_ntw): parity of the README-config tests, then the step
SRHIP_F16X2=1 timeout 900 python -m pytest tests/test_gpu_swinir.py tests/test_gpu_fullsize.py -q -x -k "swinir or readme or tiny" 2>&1 | tail -4
for i in 1 2 3; do
  for v in 0 1; do
    SRHIP_F16X2=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('F16X2=$v', round(d['value'],1), round(d['roofline']['avg_launch_us'],2), 'loss', d['config'].get('final_loss'))"
  done
done
