#!/usr/bin/env python3
"""Which blocks of the strip-form conv weight gradient ran a second time in an EDSR x8 training step, and why?  Reads the
per-block words behind the batched launch's partial sums (flag, the staging waves' column maxima).  GPU box, repo root."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
sys.path.insert(0, ROOT)
import ctypes
import torch
from srhip import ops
from srhip.train import TrainStep, Optimizer
from dlib.models.network_edsr_liif import EDSR_LIIF
import bench

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = EDSR_LIIF(scale=8).to(dev).train()
ts = TrainStep(net, [("l1", 1.0)])
ts.opt = Optimizer(ts.fp, "adam", lr=2e-4, wd=1e-4)
lr_img, hr_img = bench.synth_batch(8, 8, dev, seed=2008)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    ts.step(lr_img, hr_img)
torch.cuda.synchronize()
n, B, H, W, C = 33, 8, 64, 64, 64
S, per = ctypes.c_int(0), ctypes.c_long(0)
ops.call("srhip_conv3x3_wgrad_batched_plan", n, B, H, W, C, C, ctypes.addressof(S), ctypes.addressof(per))
S = S.value
part = ops.SCRATCH.bufs["tnb_part"]
aux = part[n * S * 9 * C * C:][: n * S * 320].view(n, S, 320).cpu()
flags = aux[:, :, 0].view(torch.int32)
mx = aux[:, :, 64:].view(n, S, 4, 64)
dy = torch.maximum(mx[:, :, 0], mx[:, :, 1])
x = torch.maximum(mx[:, :, 2], mx[:, :, 3])
print("S", S, "flagged blocks", int((flags != 0).sum()), "of", n * S)
for name, m in (("dY", dy), ("X", x)):
    tile = m.amax(-1, keepdim=True)
    rel = (m / tile.clamp_min(1e-300))
    print(name, "columns == 0:", int((m == 0).sum()), " < 1e-4 of the tile's max:", int(((rel < 1e-4) & (m > 0)).sum()),
          " < 1e-2:", int(((rel < 1e-2) & (m > 0)).sum()), "of", m.numel())
    print("   per item (flagged slices / zero cols / cols < 1e-2 of tile max):",
          [(int((flags[k] != 0).sum()), int((m[k] == 0).sum()), int(((rel[k] < 1e-2) & (m[k] > 0)).sum())) for k in range(n)])
# details of the flagged blocks: every column's maximum relative to the tile's
for k in range(n):
    for sl in range(S):
        if flags[k, sl] != 0:
            print(f"item {k} slice {sl}: why (staging waves dY lo / dY hi / X lo / X hi; 1 = overflow, 2 = below resolution):",
                  aux[k, sl, 1:5].view(torch.int32).tolist())
            for name, m in (("dY", dy[k, sl]), ("X", x[k, sl])):
                t = m.max().item()
                srt = torch.sort(m)[0]
                nz = srt[srt > 0]
                print(f"item {k} slice {sl} {name}: tile max {t:.3e}, zero cols {int((m == 0).sum())}, smallest nonzero / tile max "
                      f"{(nz[0] / t).item():.2e} {(nz[1] / t).item():.2e} {(nz[2] / t).item():.2e}, largest {(srt[-1] / t).item():.2f} {(srt[-2] / t).item():.2f}")
