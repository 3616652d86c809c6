#!/usr/bin/env python3
"""CPU enqueue time vs GPU time of one training step (is the step launch-bound?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd")); sys.path.insert(0, ROOT)
import torch
from dlib.models.network_swinir import SwinIR
from srhip.train import TrainStep, Optimizer
import bench
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = SwinIR(upscale=8, in_chans=1, img_size=64, window_size=8, depths=[6, 6, 6, 6], embed_dim=180,
             num_heads=[6, 6, 6, 6], mlp_ratio=2, upsampler="pixelshuffledirect").to(dev).train()
ts = TrainStep(net, [("l1", 1.0)])
ts.opt = Optimizer(ts.fp, "sgd", lr=0.01, momentum=0.9, nesterov=True, wd=0.0,
                   scheduler={"type": "MyStepLR", "step_size": 30, "gamma": 0.5, "min_lr": 1e-4})
lr_img, hr_img = bench.synth_batch(8, 8, dev, seed=1000)
for _ in range(3): ts.step(lr_img, hr_img)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n): ts.step(lr_img, hr_img)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"cpu enqueue {1e3*(t1-t0)/n:.2f} ms/step   total {1e3*(t2-t0)/n:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): ts.step(lr_img, hr_img)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
