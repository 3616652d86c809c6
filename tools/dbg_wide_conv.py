#!/usr/bin/env python3
"""debug: 3x3 conv forward / data gradient / weight gradient at very wide channel counts (the expanded transposed /
strided convs of DBPN / SRFBN: 64 -> 1024 / 4096 and back) against float64"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch, torch.nn.functional as F
from srhip import ops
torch.manual_seed(0)
def rel(a, b): return ((a.double().cpu() - b).abs().max() / b.abs().max()).item()
for (B, H, W, Ci, Co) in [(2, 16, 16, 64, 256), (2, 16, 16, 64, 1024), (2, 16, 16, 1024, 64), (2, 8, 8, 64, 4096), (2, 8, 8, 4096, 64),
                          (2, 64, 64, 128, 64), (2, 16, 16, 256, 64)]:
    x = torch.randn(B, Ci, H, W); w = torch.randn(Co, Ci, 3, 3) / (3 * Ci ** 0.5); b = torch.randn(Co) * 0.1
    dy = torch.randn(B, Co, H, W)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    y = F.conv2d(xr, wr, br, padding=1); y.backward(dy.double())
    xn = x.permute(0, 2, 3, 1).contiguous().cuda(); dyn = dy.permute(0, 2, 3, 1).contiguous().cuda(); wc = w.cuda(); bc = b.cuda()
    wp = torch.empty(9, Co, Ci).cuda(); wpt = torch.empty(9, Ci, Co).cuda()
    ops.pack_conv_weight(wc, wp, wpt)
    yo = ops.conv3x3(xn, wp, bc, Co)
    dxo = ops.conv3x3(dyn, wpt, None, Ci)
    dW = torch.empty(Co, Ci, 3, 3).cuda(); db = torch.empty(Co).cuda()
    ops.conv3x3_wgrad(dyn, xn, dW, db)
    print(f"Ci {Ci:5d} Co {Co:5d} {H}x{W}: fwd {rel(yo.permute(0,3,1,2), y.detach()):.2e}  dx {rel(dxo.permute(0,3,1,2), xr.grad):.2e}  "
          f"dW {rel(dW, wr.grad):.2e}  db {rel(db, br.grad):.2e}", flush=True)
# 1x1 conv = GEMM + TN at 128 -> 64 on 8192 tokens
for (T, Ci, Co) in [(8192, 128, 64), (8192, 384, 64), (2048, 64, 64)]:
    x = torch.randn(T, Ci); w = torch.randn(Co, Ci) / Ci ** 0.5; dy = torch.randn(T, Co)
    dWr = dy.double().t() @ x.double()
    dW = torch.empty(Co, Ci).cuda(); db = torch.empty(Co).cuda()
    ops.linear_wgrad(dy.cuda(), x.cuda(), dW, db)
    wT = torch.empty(Ci, Co).cuda(); ops.transpose(w.cuda(), wT)
    dx = ops.gemm_nt(dy.cuda(), wT, None)
    print(f"1x1 T {T} Ci {Ci} Co {Co}: dW {rel(dW, dWr):.2e}  db {rel(db, dy.double().sum(0)):.2e}  dx {rel(dx, dy.double() @ w.double()):.2e}")
