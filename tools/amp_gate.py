#!/usr/bin/env python3
"""--amp gate per network: PSNR of the reduced-precision forward (leading fp16 / bf16 plane, one product) against the
f32-accurate forward on 8 synthetic 512 x 512 patches; the gate of tests/test_gpu_amp.py is 0.01 dB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, torch.nn.functional as F
import sr_oracle as O
from srhip import ops

def psnr(a, b, border):
    return O.metric_psnr(O.tensor2uint82float(a), O.tensor2uint82float(b), border)

def nets():
    from dlib.models.network_vdsr import VDSR
    from dlib.models.network_drrn import DRRN
    from dlib.models.network_mslapsr import MSLapSRN
    from dlib.models.network_memnet import MemNet
    v = VDSR(in_chans=1, upscale=2); v.load_state_dict(O.vdsr_init_state_dict(1, seed=2)); yield "VDSR x2", v, 2
    d = DRRN(in_chans=1, upscale=2, num_residual_units=25); d.load_state_dict(O.drrn_init_state_dict(1, seed=3)); yield "DRRN x2", d, 2
    m = MSLapSRN(upscale=4, in_chans=1); m.load_state_dict(O.mslapsrn_init_state_dict(4, seed=5)); yield "MSLapSRN x4", m, 4
    mn = MemNet(in_chans=1, upscale=2, num_memory_blocks=6, num_residual_blocks=6)
    mn.load_state_dict(O.memnet_init_state_dict(6, 6, seed=7)); yield "MemNet x2", mn, 2

gen = torch.Generator().manual_seed(11)
for name, net, s in nets():
    net = net.cuda().eval()
    B = 2 if name.startswith("MemNet") else 8
    hr = (torch.rand(B, 1, 512, 512, generator=gen) * 255).round() / 255
    x = F.interpolate(hr, scale_factor=1.0 / s, mode="bicubic").clamp(0, 1).cuda()
    xi = net.prepare_input(x)[0]
    with torch.no_grad():
        y32 = net.engine.forward(xi, None, save=False).clone().cpu().reshape(hr.shape)
        with ops.amp_inference(True):
            y16 = net.engine.forward(xi, None, save=False).clone().cpu().reshape(hr.shape)
    gap = (psnr(y32, hr, s) - psnr(y16, hr, s)).abs().max().item()
    print(f"{name}: forced single-product forward vs f32-accurate: MAE {(y32 - y16).abs().mean().item():.2e}, PSNR gap {gap:.4f} dB", flush=True)
