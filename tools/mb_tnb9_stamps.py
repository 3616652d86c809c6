#!/usr/bin/env python3
"""Wall-clock stamps of k_tnb9's matrix waves (experiments build, SRHIP_TN_DBG=3): loop start / loop end / done per wave of
blocks 0..3, in 10-ns units.  usage: SRHIP_LIB=.../libsrhip_exp.so SRHIP_TN_DBG=3 python tools/mb_tnb9_stamps.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops
for (B, H, W, Cout, Cin, ps2) in ((8, 256, 256, 256, 64, True), (8, 64, 64, 64, 64, False)):
    X = torch.randn(B, H, W, Cin, device="cuda")
    dY = torch.randn(B, 2 * H, 2 * W, Cout // 4, device="cuda") if ps2 else torch.randn(B, H, W, Cout, device="cuda")
    dW, db = torch.empty(Cout, Cin, 3, 3, device="cuda"), torch.empty(Cout, device="cuda")
    for _ in range(3):
        ops.conv3x3_wgrad(dY, X, dW, db, ps2=ps2)
    torch.cuda.synchronize()
    cs = ops.SCRATCH.bufs["tn_colsum"]
    raw = cs[:4 * 8 * 4 * 2].view(torch.int64).cpu().view(4, 8, 4)
    print(f"{B}x{H}x{W} {Cin}->{Cout}: per matrix wave (block, wave): loop us, epilogue us, chunks")
    for b in range(4):
        print("  block", b, [(round((raw[b, w, 1] - raw[b, w, 0]).item() / 100.0, 1), round((raw[b, w, 2] - raw[b, w, 1]).item() / 100.0, 1),
                              int(raw[b, w, 3])) for w in range(8)])
    t0 = min(raw[b, w, 0].item() for b in range(4) for w in range(8))
    print("  starts (us after the first):", [round((raw[b, 0, 0].item() - t0) / 100.0, 1) for b in range(4)])
