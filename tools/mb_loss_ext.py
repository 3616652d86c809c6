#!/usr/bin/env python3
"""Optional MasterLoss terms at the benchmark's 8 x 512 x 512: microseconds per fused value+gradient launch and
the HBM rate over the algorithmic bytes (pred + target read, gradient written: 25.2 MB)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops
p, t = torch.rand(8, 1, 512, 512, device="cuda"), torch.rand(8, 1, 512, 512, device="cuda")
g, out = torch.empty_like(p), torch.zeros(1, device="cuda")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
mb = 3 * p.numel() * 4 / 1e6
rows = [("l1", lambda: ops.loss_l1l2(p, t, 0, 1.0, None, g, out)),
        ("charbonnier", lambda: ops.loss_pointwise(p, t, 2, 1.0, 1e-9, None, g, out)),
        ("ssim19", lambda: ops.ssim_loss(p, t, 19, 1.0, g, out))]
for kind, ksz in (("grad", 3), ("laplace", 3), ("lv", 3), ("lv", 5), ("lv", 7)):
    for cn in (False, True):
        for norm in (2, 1):
            rows.append((f"{'norm_' if cn else ''}{kind}{ksz if kind == 'lv' else ''}_l{norm}",
                         lambda kind=kind, ksz=ksz, cn=cn, norm=norm: ops.loss_stencil(p, t, kind, 1.0, norm, ksz, cn, g, out)))
for name, fn in rows:
    us = timeit(fn)
    print(f"{name:18s} {us:8.1f} us   {mb / us:6.2f} TB/s algorithmic")
