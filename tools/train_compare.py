#!/usr/bin/env python3
"""100 optimisation steps of the README SwinIR (B=4) under the three matmul paths from the same seed: loss trajectories
and final parameters of the default path (Linear GEMMs: two fp16 planes / three products; the rest bf16x3) and of the
all-bf16x3 path (SRHIP_F16X2=0) against the exact-f32 MFMA path."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    mode = sys.argv[1]
    os.environ["SRHIP_MM"] = "f32" if mode == "f32" else "bx3"
    os.environ["SRHIP_F16X2"] = "0" if mode == "bx3_six_products" else "1"
    sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd")); sys.path.insert(0, ROOT)
    import torch
    from dlib.models.network_swinir import SwinIR
    from srhip.train import TrainStep, Optimizer
    import bench
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = SwinIR(upscale=8, in_chans=1, img_size=64, window_size=8, depths=[6, 6, 6, 6], embed_dim=180,
                 num_heads=[6, 6, 6, 6], mlp_ratio=2, upsampler="pixelshuffledirect", drop_path_rate=0.0).to(dev).train()
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "adam", lr=2e-4, wd=0.0)
    losses = []
    for i in range(100):
        lr_img, hr_img = bench.synth_batch(4, 8, dev, seed=100 + i % 5)
        ts.step(lr_img, hr_img)
        if i % 10 == 9:
            losses.append(ts.loss_values()[0])
    flat = ts.fp.flat.double()
    print(json.dumps({"mode": mode, "losses": losses, "pnorm": flat.norm().item(),
                      "checksum": flat[::997].sum().item()}))
    torch.save(ts.fp.flat.cpu(), f"/tmp/params_{mode}.pt")
else:
    out = {}
    for mode in ("f32", "bx3_six_products", "default"):
        r = subprocess.run([sys.executable, __file__, mode], capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        out[mode] = json.loads(line)
        print(line)
    import torch
    a = torch.load("/tmp/params_f32.pt").double()
    for mode in ("bx3_six_products", "default"):
        b = torch.load(f"/tmp/params_{mode}.pt").double()
        print(f"{mode}: after 100 Adam steps max |dparam| = {(a - b).abs().max().item():.3e}  rel L2 = {((a - b).norm() / a.norm()).item():.3e}")
        print(f"  loss |f32 - {mode}| per checkpoint:", [f"{abs(x - y):.2e}" for x, y in zip(out['f32']['losses'], out[mode]['losses'])])
