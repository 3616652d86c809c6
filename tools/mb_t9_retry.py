#!/usr/bin/env python3
"""Does the block-level exponent retry of the nine-tap conv weight gradient fire on a given kind of data?  (a launch whose
blocks retry takes up to twice as long).  GPU box, repo root: python tools/mb_t9_retry.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops

B, H, W, Cin, Cout = 8, 128, 128, 64, 64
torch.manual_seed(0)
kinds = {
    "randn": lambda s: torch.randn(*s, device="cuda"),
    "randn*1e-7": lambda s: torch.randn(*s, device="cuda") * 1e-7,
    "relu(randn)": lambda s: torch.relu(torch.randn(*s, device="cuda")),
    "randn*chanscale(1e-3..1)": lambda s: torch.randn(*s, device="cuda") * torch.logspace(-3, 0, s[-1], device="cuda"),
    "rows grow x100": lambda s: torch.randn(*s, device="cuda") * torch.linspace(1, 100, s[1], device="cuda")[None, :, None, None],
    "first row zero": lambda s: torch.randn(*s, device="cuda") * (torch.arange(s[1], device="cuda") % 16 != 0)[None, :, None, None],
}
for kx, fx in kinds.items():
    X = fx((B, H, W, Cin))
    dY = fx((B, H, W, Cout))
    dW, db = torch.empty(Cout, Cin, 3, 3, device="cuda"), torch.empty(Cout, device="cuda")
    for _ in range(3):
        ops.conv3x3_wgrad(dY, X, dW, db)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ops.conv3x3_wgrad(dY, X, dW, db)
    b.record()
    torch.cuda.synchronize()
    print(f"{kx:28s} {a.elapsed_time(b) * 100.0:8.1f} us/launch")
