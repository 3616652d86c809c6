#!/usr/bin/env python3
"""debug: the README graph step after the mlp_f16 tests ran in the same process (NaN-filled freed buffers)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import pytest, torch
os.chdir(ROOT)
pytest.main(["tests/test_gpu_mlp_f16.py", "-q", "-x"])
from srhip import ops
from srhip.train import TrainStep, Optimizer
from dlib.models.network_swinir import SwinIR

def finite_report(net, tag):
    bad = []
    for k, v in sorted(net.engine.bufs.d.items()):
        if v.dtype == torch.float32 and not torch.isfinite(v).all():
            bad.append((k, int((~torch.isfinite(v)).sum()), v.numel()))
    print(tag, "non-finite buffers:", bad[:40], flush=True)

for mode in ("graph", "eager"):
    net = SwinIR(upscale=8, in_chans=1, img_size=64, window_size=8, depths=[6, 6, 6, 6], embed_dim=180,
                 num_heads=[6, 6, 6, 6], mlp_ratio=2, upsampler="pixelshuffledirect").cuda().train()
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "sgd", lr=0.01, momentum=0.9, nesterov=True, wd=0.0)
    lr_img, hr_img = torch.rand(8, 1, 64, 64).cuda(), torch.rand(8, 1, 512, 512).cuda()
    for i in range(13):
        (ts.step_graph if mode == "graph" else ts.step)(lr_img, hr_img)
        torch.cuda.synchronize()
        l = float(ts.loss_buf[1])
        print(mode, i, l, flush=True)
        if l != l:
            finite_report(net, f"{mode} step {i}")
            print(" params finite:", bool(torch.isfinite(ts.fp.flat).all()), "grads finite:", bool(torch.isfinite(ts.fp.grad).all()))
            dp = net.engine.saved["dp"]
            print(" dp finite:", None if dp is None else bool(torch.isfinite(dp).all()), None if dp is None else dp[:4])
            break
