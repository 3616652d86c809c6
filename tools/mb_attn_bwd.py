#!/usr/bin/env python3
"""Attention-core backward (srhip_window_attention_bwd_f16x2) at the README shape, cold rotating operands, partial
bias-gradient tiles left unreduced (the training step's form).  HIP-event time per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops
B, H, W, C, heads = 8, 64, 64, 180, 6
T = B * H * W
dev = "cuda"
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
table = torch.randn(225, heads, device=dev) * 0.02
bF = torch.empty(heads, 64, 64, device=dev); bG = torch.empty(heads, 64, 64, device=dev)
ops.bias_expand_f16(table, bF, bG)
qs = [torch.randn(T, 3 * C, device=dev) for _ in range(6)]
das = [torch.randn(T, C, device=dev) for _ in range(6)]
dqs = [torch.empty(T, 3 * C, device=dev) for _ in range(6)]
parts = torch.empty(ops.wattn_dbias_ws(B, H, W, heads), device=dev)
modes = [0]
if hasattr(ops.lib, "srhip_wattn2_debug_mode"):      # experiment build: ablations
    modes = [0, 1, 2, 4, 8, 3, 5, 6]
for mode in modes:
  if len(modes) > 1:
    ops.lib.srhip_wattn2_debug_mode(mode)
    print("mode", mode, "(1 no arithmetic | 2 no stores | 4 no row loads | 8 no touches)")
  for shift in (0, 4):
    it = [0]
    def b16():
        it[0] += 1; i = it[0] % 6
        ops.window_attention_bwd_f16(qs[i], das[i], dqs[i], bF, bG, None, B, H, W, C, heads, shift, parts=parts)
    print("  ", end="")
    print(f"shift {shift}: attention backward fp16x2 {timeit(b16):6.1f} us per launch (cold operands, no reducer)")
