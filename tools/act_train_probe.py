"""ACT (registry configuration) training step at the README batch: B = 8, 64 x 64 -> 512 x 512, L1 + SGD; prints ms per step.
usage: python tools/act_train_probe.py [scale] [batch]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
from dlib.models.network_act import ACT  # noqa: E402
from srhip.train import TrainStep  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.manual_seed(0)
net = ACT(upscale=scale, in_chans=1).cuda().train()
ts = TrainStep(net, [("l1", 1.0)])
x, t = torch.rand(B, 1, 64, 64).cuda(), torch.rand(B, 1, 64 * scale, 64 * scale).cuda()
losses = []
for i in range(6):
    if i == 2:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    ts.step(x, t)
    losses.append(ts.loss_values()[0])
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 4 * 1e3
print(f"ACT x{scale} B={B}: {ms:.1f} ms per training step = {B / ms * 1e3:.1f} patches/s; loss {losses[0]:.5f} -> {losses[-1]:.5f}; "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
