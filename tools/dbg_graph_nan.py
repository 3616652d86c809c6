#!/usr/bin/env python3
"""debug: which gradient goes bad first when the README step is replayed from a hipGraph after NaN-filled buffers were freed"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
os.chdir(ROOT)
from srhip import ops
from srhip.train import TrainStep, Optimizer
from dlib.models.network_swinir import SwinIR

junk = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(int(os.environ.get("POISON", "30")))]
del junk
net = SwinIR(upscale=8, in_chans=1, img_size=64, window_size=8, depths=[6, 6, 6, 6], embed_dim=180,
             num_heads=[6, 6, 6, 6], mlp_ratio=2, upsampler="pixelshuffledirect").cuda().train()
ts = TrainStep(net, [("l1", 1.0)])
ts.opt = Optimizer(ts.fp, "sgd", lr=0.01, momentum=0.9, nesterov=True, wd=0.0)
lr_img, hr_img = torch.rand(8, 1, 64, 64).cuda(), torch.rand(8, 1, 512, 512).cuda()
names = ts.fp.names
for i in range(8):
    ts.step_graph(lr_img, hr_img)
    torch.cuda.synchronize()
    l = float(ts.loss_buf[1])
    gmax = {k: float(ts.fp.gviews[k].abs().max()) for k in names}
    worst = sorted(gmax.items(), key=lambda kv: -(kv[1] if kv[1] == kv[1] else 1e99))[:6]
    pmax = float(ts.fp.flat.abs().max())
    print(f"step {i} loss {l:.6g} max|param| {pmax:.4g} worst grads {[(k, f'{v:.3g}') for k, v in worst]}", flush=True)
    bad = [k for k, v in gmax.items() if not (v < 1e3)]
    if bad:
        print("  bad grads:", bad[:30])
        for k, v in sorted(net.engine.bufs.d.items()):
            if v.dtype == torch.float32:
                m = float(v.abs().max())
                if not (m < 1e6):
                    print("   buffer", k, m)
        for k, v in ops.SCRATCH.bufs.items():
            if v.dtype in (torch.float32, torch.float64):
                m = float(v.abs().max())
                if not (m < 1e6):
                    print("   scratch", k, m, v.numel())
        break
