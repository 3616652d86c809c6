#!/usr/bin/env python3
"""Ablation timing of the grouped bf16x3 TN kernel (env SRHIP_TN_DBG bits: 1 no loads, 2 no split/LDS store, 4 no MFMA)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops
T, C, hid, dev = 32768, 180, 360, "cuda"
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
g = torch.randn(T, C, device=dev); dqkv = torch.randn(T, 3 * C, device=dev); dh = torch.randn(T, hid, device=dev)
x = torch.randn(T, C, device=dev); hb = torch.randn(T, hid, device=dev)
st = torch.stack([x.mean(1), 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)], 1).contiguous()
pr = [dict(dY=g, X=hb, dW=torch.empty(C, hid, device=dev), db=torch.empty(C, device=dev), b_mode=2),
      dict(dY=dh, X=x, dW=torch.empty(hid, C, device=dev), db=torch.empty(hid, device=dev), b_mode=1, ln_stats=st),
      dict(dY=g, X=x, dW=torch.empty(C, C, device=dev), db=torch.empty(C, device=dev)),
      dict(dY=dqkv, X=x, dW=torch.empty(3 * C, C, device=dev), db=torch.empty(3 * C, device=dev), b_mode=1, ln_stats=st)]
plain = [dict(q, b_mode=0, ln_stats=None) for q in pr]
for dbg in [0, 4, 2, 1, 3, 6, 7]:
    os.environ["SRHIP_TN_DBG"] = str(dbg)
    print(f"dbg={dbg}: with prologues {timeit(lambda: ops.linear_wgrad_grouped(pr)):7.1f} us   plain {timeit(lambda: ops.linear_wgrad_grouped(plain)):7.1f} us   (incl ~85 us reducers)")
