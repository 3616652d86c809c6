import os, sys, torch
sys.path.insert(0, "sr-caco-2_amd")
from srhip import ops
import torch.nn.functional as F
torch.manual_seed(0)
for (B, H, W, Ci, Co) in ((2, 64, 64, 180, 64), (2, 64, 64, 64, 64), (2, 64, 64, 64, 180), (2, 16, 16, 180, 180), (2, 24, 40, 180, 64), (8, 64, 64, 64, 64), (1, 20, 20, 64, 64), (2, 64, 64, 64, 256)):
    x = torch.randn(B, H, W, Ci, device="cuda"); w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05; b = torch.randn(Co, device="cuda") * 0.1
    P = ops.Bx3(9 * Co, Ci, "cuda"); tb = ops.PrepTable(); tb.conv(w, P); tb.build("cuda").run()
    y = ops.conv3x3(x, P, b, Co)
    print("   fmt", P.fmt, end=" ")
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    e = (y.double() - ref).abs()
    print(f"{B}x{H}x{W} {Ci}->{Co}: max abs err {e.max().item():.3e}  (|y| max {ref.abs().max().item():.2f}), mean {e.mean().item():.3e}, bad pixels (>1e-4) {(e > 1e-4).sum().item()}", flush=True)
