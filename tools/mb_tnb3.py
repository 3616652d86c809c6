#!/usr/bin/env python3
"""Microbenchmark of the 64-wide conv weight-gradient kernel (k_tnb3: three taps per block) at EDSR x8's shapes, with the role
ablations of the experiments build (SRHIP_LIB=.../libsrhip_exp.so, SRHIP_TN_DBG=0|1|2: all | no MFMAs | no staging).
usage (GPU box, repo root): SRHIP_LIB=$PWD/sr-caco-2_amd/lib/libsrhip_exp.so python tools/mb_tnb3.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    from srhip import ops
    out = []
    for (B, H, W, Cout, Cin, ps2) in ((8, 64, 64, 64, 64, False), (8, 256, 256, 256, 64, True), (8, 128, 128, 256, 64, True)):
        X = torch.randn(B, H, W, Cin, device="cuda")
        dY = torch.randn(B, 2 * H, 2 * W, Cout // 4, device="cuda") if ps2 else torch.randn(B, H, W, Cout, device="cuda")
        dW, db = torch.empty(Cout, Cin, 3, 3, device="cuda"), torch.empty(Cout, device="cuda")
        for _ in range(3):
            ops.conv3x3_wgrad(dY, X, dW, db, ps2=ps2)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.conv3x3_wgrad(dY, X, dW, db, ps2=ps2)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 100.0
        gf = 18.0 * B * H * W * Cout * Cin / 1e9
        out.append(f"{B}x{H}x{W} {Cin}->{Cout}{' ps2' if ps2 else ''}: {us:8.1f} us/launch (+reducer)  {3 * gf / us:6.1f} TF/s fp16-equivalent")
    print(f"SRHIP_TN_DBG={os.environ.get('SRHIP_TN_DBG', '0')}: " + " | ".join(out))
else:
    for dbg in ("0", "1", "2"):
        env = dict(os.environ, SRHIP_TN_DBG=dbg)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, check=False)
