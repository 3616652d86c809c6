#!/usr/bin/env python3
"""EDSR's 64 -> 64 3x3 conv at B=8, 128x128 (the x4 body): exact-f32 MFMA kernels vs the bf16x3 ones
(forward / data gradient and weight gradient)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (B, H, W, Ci, Co) in ((8, 128, 128, 64, 64), (8, 128, 128, 64, 256), (8, 64, 64, 64, 64)):
    x = torch.randn(B, H, W, Ci, device=dev); dy = torch.randn(B, H, W, Co, device=dev)
    wp = torch.randn(9, Co, Ci, device=dev) * 0.05; b = torch.randn(Co, device=dev)
    y = torch.empty(B, H, W, Co, device=dev)
    wb = ops.split_bf16x3(wp)
    fl = 2.0 * B * H * W * Ci * Co * 9
    t32 = timeit(lambda: ops.conv3x3(x, wp, b, Co, out=y)); y32 = y.clone()
    tbx = timeit(lambda: ops.conv3x3(x, wb, b, Co, out=y))
    print(f"conv {Ci}->{Co} {B}x{H}x{W}: f32 {t32:7.1f} us ({fl/t32*1e-6:6.1f} TF/s)  bx3 {tbx:7.1f} us ({fl/tbx*1e-6:6.1f} TF/s)  "
          f"maxdiff {(y - y32).abs().max().item():.1e}")
    dW, db = torch.empty(Co, Ci, 3, 3, device=dev), torch.empty(Co, device=dev)
    old = ops.BX3_MIN_CHANNELS
    ops.BX3_MIN_CHANNELS = 100000
    tw32 = timeit(lambda: ops.conv3x3_wgrad(dy, x, dW, db)); d32 = dW.clone()
    ops.BX3_MIN_CHANNELS = 1
    twbx = timeit(lambda: ops.conv3x3_wgrad(dy, x, dW, db))
    ops.BX3_MIN_CHANNELS = old
    print(f"   wgrad: f32 {tw32:7.1f} us ({fl/tw32*1e-6:6.1f} TF/s)  bx3 {twbx:7.1f} us ({fl/twbx*1e-6:6.1f} TF/s)  "
          f"rel maxdiff {((dW - d32).abs().max() / d32.abs().max()).item():.1e}")
