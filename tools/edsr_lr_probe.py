#!/usr/bin/env python3
"""EDSR x8 training throughput as a function of the learning rate (same kernels, same launches): do the operands' VALUES
move the step time?  GPU box, repo root: python tools/edsr_lr_probe.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
sys.path.insert(0, ROOT)
import torch
from srhip.train import TrainStep, Optimizer
from dlib.models.network_edsr_liif import EDSR_LIIF
import bench

dev = torch.device("cuda:0")
for lr in (2e-4, 0.0, 2e-4, 2e-6):
    torch.manual_seed(0)
    net = EDSR_LIIF(scale=8).to(dev).train()
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "adam", lr=lr, wd=1e-4)
    lr_img, hr_img = bench.synth_batch(8, 8, dev, seed=2008)
    for _ in range(5):
        ts.step_graph(lr_img, hr_img)
    torch.cuda.synchronize()
    out = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            ts.step_graph(lr_img, hr_img)
        torch.cuda.synchronize()
        out.append(8 * 20 / (time.perf_counter() - t0))
    print(f"lr {lr:g}: " + " ".join(f"{v:7.1f}" for v in out) + f" patches/s   loss {ts.loss_values()[0]:.5f}")
