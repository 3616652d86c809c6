"""Training step of a registry tape net (ACT | OmniSR | GRL | ...) at the README batch: B = 8, 64 x 64 -> 512 x 512, L1 + SGD;
prints ms per step, the loss before / after, peak memory.
usage: python tools/tape_train_probe.py [ACT|OmniSR|GRL] [scale] [batch] [graph]     (graph: TrainStep.step_graph, one hipGraph replay per step)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
from srhip.train import TrainStep  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ACT"
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
graph = len(sys.argv) > 4 and sys.argv[4] == "graph"
torch.manual_seed(0)
if name == "ACT":
    from dlib.models.network_act import ACT
    net = ACT(upscale=scale, in_chans=1)
elif name == "OmniSR":
    from dlib.models.network_omni_sr import OmniSR
    net = OmniSR(input_shape=1, upscale=scale)
elif name == "GRL":
    from dlib.models.network_grl import GRL          # the registry's options (select_network.py:70-90)
    net = GRL(upscale=scale, img_size=64, in_chans=1, window_size=8, depths=[4, 4, 8, 8, 8, 4, 4], embed_dim=180,
              num_heads_window=[3] * 7, num_heads_stripe=[3] * 7, mlp_ratio=2, qkv_proj_type="linear", anchor_proj_type="avgpool",
              anchor_window_down_factor=2, out_proj_type="linear", conv_type="1conv", upsampler="pixelshuffle",
              local_connection=True)
else:
    raise SystemExit(f"unknown net {name}")
net = net.cuda().train()
ts = TrainStep(net, [("l1", 1.0)])
x, t = torch.rand(B, 1, 64, 64).cuda(), torch.rand(B, 1, 64 * scale, 64 * scale).cuda()
losses = []
n_warm = 4 if graph else 2
for i in range(n_warm + 3):
    if i == n_warm:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    (ts.step_graph if graph else ts.step)(x, t)
    losses.append(ts.loss_values()[0])
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
print(f"{name} x{scale} B={B}{' (step_graph)' if graph else ''}: {ms:.1f} ms per training step = {B / ms * 1e3:.1f} patches/s; loss {losses[0]:.5f} -> {losses[-1]:.5f}; "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
