#!/usr/bin/env python3
"""Conv weight gradients of channel counts other than 64 (SwinIR's 180 -> 180 and 180 -> 64, 128, 256, 192, 96) in the strip form with partly empty 64-column tiles
against the one-tap-per-block kernels: experiments build, SRHIP_TN_T9S_RAGGED=1|0.  GPU box, repo root."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    from srhip import ops
    out = []
    for (B, H, W, Cout, Cin) in ((8, 64, 64, 180, 180), (8, 64, 64, 64, 180), (8, 64, 64, 128, 128), (8, 128, 128, 128, 128), (8, 64, 64, 256, 256), (8, 64, 64, 192, 192), (8, 64, 64, 96, 96)):
        X, dY = torch.randn(B, H, W, Cin, device="cuda"), torch.randn(B, H, W, Cout, device="cuda")
        dW, db = torch.empty(Cout, Cin, 3, 3, device="cuda"), torch.empty(Cout, device="cuda")
        for _ in range(3):
            ops.conv3x3_wgrad(dY, X, dW, db)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.conv3x3_wgrad(dY, X, dW, db)
        b.record()
        torch.cuda.synchronize()
        out.append(f"{Cin}->{Cout}: {a.elapsed_time(b) * 50.0:7.1f} us")
    print(f"SRHIP_TN_T9S_RAGGED={os.environ.get('SRHIP_TN_T9S_RAGGED', '1')}: " + " | ".join(out) + "  (launch + reducer)")
else:
    for m in ("1", "0"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, SRHIP_TN_T9S_RAGGED=m), check=False)
