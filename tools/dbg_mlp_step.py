#!/usr/bin/env python3
"""debug: fused fp16x2 MLP inside the README training step (eager and graph), call counts, losses, NaN hunt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops
from srhip.train import TrainStep, Optimizer
from dlib.models.network_swinir import SwinIR

cnt = {"f": 0, "b": 0}
_f, _b = ops.mlp_fwd_f16, ops.mlp_bwd_f16
def wf(*a, **k):
    cnt["f"] += 1
    return _f(*a, **k)
def wb(*a, **k):
    cnt["b"] += 1
    return _b(*a, **k)
ops.mlp_fwd_f16, ops.mlp_bwd_f16 = wf, wb

def make():
    torch.manual_seed(0)
    net = SwinIR(upscale=8, in_chans=1, img_size=64, window_size=8, depths=[6, 6, 6, 6], embed_dim=180,
                 num_heads=[6, 6, 6, 6], mlp_ratio=2, upsampler="pixelshuffledirect").cuda().train()
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "sgd", lr=0.01, momentum=0.9, nesterov=True, wd=0.0)
    return net, ts

if os.environ.get("POISON"):      # freed blocks full of NaN: the caching allocator hands them to the step's buffers
    junk = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(int(os.environ["POISON"]))]
    del junk
torch.manual_seed(5)
lr_img, hr_img = torch.rand(8, 1, 64, 64).cuda(), torch.rand(8, 1, 512, 512).cuda()
for mode in ("eager", "graph"):
    for fuse in (1, 0):
        net, ts = make()
        net.engine.fuse_mlp_h = bool(fuse)
        cnt["f"] = cnt["b"] = 0
        torch.manual_seed(77)
        out = []
        for i in range(int(os.environ.get('NSTEP', '16'))):
            (ts.step if mode == "eager" else ts.step_graph)(lr_img, hr_img)
            out.append(ts.loss_buf.clone())
        torch.cuda.synchronize()
        print(mode, "fuse", fuse, "calls", cnt, "losses", [f"{float(o[1]):.6f}" for o in out],
              "param nan:", bool(torch.isnan(ts.fp.flat).any()), flush=True)
        if fuse and mode == "eager":
            sv = net.engine.saved
            for bi, (t, st1, qkv, a, x1, st2, h) in enumerate(sv["blocks"]):
                bad = [n for n, v in (("t", t), ("st1", st1), ("qkv", qkv), ("a", a), ("x1", x1), ("st2", st2), ("h", h))
                       if v is not None and not torch.isfinite(v).all()]
                if bad:
                    print("  first non-finite saved tensors at block", bi, bad)
                    break
            g = net.engine.bufs.d
            for k in sorted(g):
                if k.startswith("g.") and not torch.isfinite(g[k]).all():
                    print("  non-finite grad buffer", k, int((~torch.isfinite(g[k])).sum()))
