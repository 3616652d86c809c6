#!/usr/bin/env python3
"""EDSR body conv (64 -> 64 channels, B x 64 x 64 pixels, ReLU epilogue) on k_nhcw2<2>: launch time with cold rotating
operands and, on an experiments build (SRHIP_LIB=.../libsrhip_exp.so), wall-clock stamps of every wave of every block."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops
B, H, W, C = 8, int(os.environ.get("HW", "64")), int(os.environ.get("HW", "64")), 64
dev = "cuda"
w = torch.randn(C, C, 3, 3, device=dev) * 0.05
bias = torch.randn(C, device=dev) * 0.1
P = ops.Bx3(9 * C, C, dev)
tb = ops.PrepTable(); tb.conv(w, P); tb.build(dev).run()
xs = [torch.randn(B, H, W, C, device=dev) for _ in range(8)]
ys = [torch.empty(B, H, W, C, device=dev) for _ in range(8)]
def run(i): ops.conv3x3(xs[i % 8], P, bias, C, out=ys[i % 8], epi=1)
for i in range(10): run(i)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(200): run(i)
b.record(); torch.cuda.synchronize()
print(f"conv 64->64 at {B}x{H}x{W}: {a.elapsed_time(b) / 200 * 1e3:.2f} us per launch (back-to-back, rotating operands)")
if hasattr(ops.lib, "srhip_nhcw2_debug_buffer"):
    nblk = B * (H // 4) * (W // 16)
    dbg = torch.zeros(nblk, 4, 16, dtype=torch.int64, device=dev)
    fn = ops.lib.srhip_nhcw2_debug_buffer; fn.argtypes = [ctypes.c_void_p]
    fn(dbg.data_ptr()); run(3); torch.cuda.synchronize(); fn(None)
    d = dbg.cpu().double() * 0.01
    d = d[d[:, 0, 12] > 0]                      # blocks that ran (8-row tiles on large images: half the rows of the buffer)
    nblk = d.shape[0]
    t0 = d[:, :, 0].min()
    names = {0: "start", 1: "prologue: index math, first loads issued", 2: "chunk 0: halo arrived, maxima exchanged", 3: "chunk 0 staged",
             4: "chunk 0: 9 taps", 6: "chunk 1: halo arrived", 7: "chunk 1 staged", 8: "chunk 1: 9 taps", 10: "barrier", 11: "re-layout",
             12: "epilogue (bias, ReLU, stores)"}
    st = d[:, 0, 0] - t0
    print(f"{nblk} blocks; starts: median {st.median():.2f} us, max {st.max():.2f}; ends: median {(d[:, 0, 12] - t0).median():.2f}, max {(d[:, :, 12] - t0).max():.2f}")
    prev = 0
    for k in sorted(names):
        if k == 0: continue
        print(f"  {names[k]:45s} +{(d[:, :, k] - d[:, :, prev]).mean():6.2f} us   (at {(d[:, :, k] - t0).mean():6.2f})")
        prev = k
