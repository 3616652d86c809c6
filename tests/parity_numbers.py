#!/usr/bin/env python3
"""Actual parity margins on the README SwinIR golden (tests/golden/g4): pixel MAE, max error, PSNR shift.
Not a pytest file (the gates are asserted in test_gpu_swinir.py); lives under tests/ because it uses the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import sr_oracle as O
from dlib.models.network_swinir import SwinIR
g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(ROOT, "tests/golden/g4_swinir_readme.npz")).items() if v.dtype.kind in "fiu"}
cfg = O.swinir_config()
sd = O.swinir_init_state_dict(cfg, seed=0)
net = SwinIR(upscale=8, in_chans=1, img_size=64, window_size=8, depths=[6, 6, 6, 6], embed_dim=180,
             num_heads=[6, 6, 6, 6], mlp_ratio=2, upsampler="pixelshuffledirect")
net.load_state_dict(sd, strict=True)
net = net.cuda().eval()
with torch.no_grad():
    y = net(g["x"].cuda()).cpu()
d = (y - g["y"]).abs()
tgt = torch.rand(1, 1, 512, 512, generator=torch.Generator().manual_seed(1))
ps = lambda a, b: O.metric_psnr(O.tensor2uint82float(a), O.tensor2uint82float(b), 8)
print(f"SRHIP_MM={os.environ.get('SRHIP_MM', 'bx3')}: pixel MAE {d.mean().item():.3e}  max {d.max().item():.3e}  "
      f"|dPSNR| {(ps(y, tgt) - ps(g['y'], tgt)).abs().max().item():.2e} dB   (gates: 1e-5, -, 0.01)")
