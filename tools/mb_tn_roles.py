#!/usr/bin/env python3
"""Grouped bf16x3 weight-gradient kernel: full vs consumers idle (SRHIP_TN_DBG=1) vs producers idle (=2).
Run after one normal launch (the debug instantiations reuse its LDS reservation)."""
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
x = torch.randn(T, C, device=dev); gh = torch.randn(T, hid, device=dev)
st = torch.stack([x.mean(1), 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)], 1).contiguous()
s2 = torch.ones(8, device=dev)
pr = [dict(dY=dqkv, X=x, dW=torch.empty(3 * C, C, device=dev), db=torch.empty(3 * C, device=dev), b_mode=1, ln_stats=st),
      dict(dY=g, X=gh, dW=torch.empty(C, hid, device=dev), db=torch.empty(C, device=dev), a_rowscale=s2, a_rowscale_rows=4096),
      dict(dY=dh, X=x, dW=torch.empty(hid, C, device=dev), db=torch.empty(hid, device=dev), b_mode=1, ln_stats=st),
      dict(dY=g, X=x, dW=torch.empty(C, C, device=dev), db=torch.empty(C, device=dev), a_rowscale=s2, a_rowscale_rows=4096)]
for dbg in ("0", "1", "2"):
    os.environ["SRHIP_TN_DBG"] = dbg
    print(f"SRHIP_TN_DBG={dbg}: {timeit(lambda: ops.linear_wgrad_grouped(pr)):7.1f} us (launch + ~21 us reducer)")
os.environ["SRHIP_TN_DBG"] = "3"
ops.linear_wgrad_grouped(pr)
torch.cuda.synchronize()
st = ops.SCRATCH.bufs["tng_part"][:16].cpu().tolist()
for w in range(4):
    o = st[4 * w:4 * w + 4]
    print(f"producer wave {w} ({'B' if w & 1 else 'A'} operand, half {w >> 1}): cycles per 32-token chunk: "
          f"store {o[0]:7.0f}  load-issue {o[1]:7.0f}  barrier wait {o[2]:7.0f}   (chunks {o[3]:.0f})")
