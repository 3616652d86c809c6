"""Microbenchmark of the per-step weight preparation (k_prep_table) by job family, SwinIR README shapes (24 blocks):
forward Linear planes, transposed Linear planes, conv packs, bias images, folded biases.  GPU box only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
from srhip import ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(540, 180), (180, 180), (360, 180), (180, 360)]


def timeit(tab, name, n=30):
    tab.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        tab.run()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) / n * 1000:8.1f} us  jobs {tab.n} blocks {tab.blocks}", flush=True)


for tr in (False, True):
    t = ops.PrepTable()
    for _ in range(24):
        for (N, K) in shapes:
            W = torch.randn(N, K, device=dev)
            rows, kd = (K, N) if tr else (N, K)
            t.linear(W, ops.Bx3(rows, kd, dev), gamma=torch.randn(K, device=dev) if N != 180 or K == 360 else None, transpose=tr, f16=True)
    timeit(t.build(dev), "linear f16 " + ("transposed" if tr else "forward"))
for dg in (False, True):
    t = ops.PrepTable()
    for _ in range(7):
        w = torch.randn(180, 180, 3, 3, device=dev)
        t.conv(w, ops.Bx3(9 * 180, 180, dev), data_grad=dg)
    timeit(t.build(dev), "conv 180 " + ("data-grad" if dg else "forward"))
t = ops.PrepTable()
for _ in range(24):
    tb = torch.randn(225, 6, device=dev)
    im = [torch.empty(6, 64, 64, device=dev) for _ in range(4)]
    t.bias_expand(tb, im[0], im[1], 6, im[2], im[3])
timeit(t.build(dev), "bias images")
t = ops.PrepTable()
for _ in range(24):
    for (N, K) in [(540, 180), (360, 180)]:
        t.fold_bias(torch.randn(N, K, device=dev), torch.randn(N, device=dev), torch.randn(K, device=dev), torch.empty(N, device=dev))
timeit(t.build(dev), "folded biases")
