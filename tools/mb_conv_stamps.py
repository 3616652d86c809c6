import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sr-caco-2_amd"))
import torch
from srhip import ops
os.environ["SRHIP_NT_DBG"] = "64"
dev = "cuda"
x = torch.randn(8, 64, 64, 180, device=dev); wp = torch.randn(9, 180, 180, device=dev) * 0.02
b = torch.randn(180, device=dev); y = torch.zeros(8, 64, 64, 180, device=dev)
wb = ops.split_bf16x3(wp)
for _ in range(3): ops.conv3x3(x, wb, b, 180, out=y)
torch.cuda.synchronize()
o = y.flatten()[:16].cpu().tolist()
for blk, v in (("first", o[:8]), ("mid", o[8:])):
    n = max(v[6], 1)
    print(f"conv 180->180 {blk:5s} block: prologue {v[0]:6.0f} | per (chunk,tap) iteration: barrier1 {v[1]/n:6.0f} stage {v[2]/n:6.0f} barrier2 {v[3]/n:6.0f} load-issue {v[4]/n:6.0f} mfma {v[5]/n:6.0f}  (iterations {v[6]:.0f}; 36 MFMA = 1152 cycles)")
