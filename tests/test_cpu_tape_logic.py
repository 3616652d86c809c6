"""Host logic of the ACT training tape (srhip/act_engine.py::_forward_tape + the token-matrix ops of srhip/tape.py) with torch
stand-ins for the kernels (tests/emul_ops.py): the graph wiring, every backward closure's accumulation and the parameter names
against the REFERENCE's autograd (tests/golden/g45_act_grad.npz).  The same comparison with the real kernels:
tests/test_gpu_tape_nets.py::test_act_training_step_gradients_vs_reference_golden."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_act_tape_wiring_against_reference_gradients(monkeypatch):
    import emul_ops
    import sr_oracle as O
    from dlib.models.network_act import ACT
    emul_ops.install(monkeypatch)
    z = np.load(os.path.join(ROOT, "tests", "golden", "g45_act_grad.npz"))
    g = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("x2/")}
    net = ACT(upscale=2, in_chans=1, n_feats=16, n_resgroups=4, n_resblocks=2, reduction=4, n_heads=4, n_layers=8, n_fusionblocks=4)
    layout = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    net.load_state_dict(O.seeded_state_dict(layout, int(g["seed"])), strict=True)
    net.train()
    x, tgt = g["x"], g["tgt"]
    eng = net.engine
    y = eng.forward(x[:, 0].contiguous(), None, save=True)
    assert (y - g["y"]).abs().max().item() <= 2e-5 * g["y"].abs().max().item()
    dy = torch.sign(y - tgt) / y.numel()
    grads = {k: torch.full_like(p, float("nan")) for k, p in net.named_parameters()}
    eng.backward(dy, grads)
    n = 0
    for k, got in grads.items():
        if "grad/" + k in g:
            ref = g["grad/" + k]
            e = ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
        elif "gslice/" + k in g:
            ref, sums = g["gslice/" + k], g["gsum/" + k]
            e = ((got[:2] - ref).abs().max() / sums[2].float().clamp_min(1e-30)).item()
            assert abs(got.double().sum().item() - sums[0].item()) <= 1e-4 * sums[1].item(), k
            assert abs(got.double().abs().sum().item() - sums[1].item()) <= 1e-4 * sums[1].item(), k
        else:                                   # a parameter the forward does not reach
            assert float(got.abs().max()) == 0.0, k
            continue
        assert e <= 2e-4, (k, e)
        n += 1
    assert n == int(g["n_grads"])


def test_omnisr_tape_wiring_against_reference_gradients(monkeypatch):
    """srhip/omnisr_engine.py::_forward_tape: window / grid attention with the bias table, both channel attentions, MBConv,
    the gated feed-forwards, ESA -- against tests/golden/g47_omnisr_grad.npz."""
    import emul_ops
    import sr_oracle as O
    from dlib.models.network_omni_sr import OmniSR
    emul_ops.install(monkeypatch)
    z = np.load(os.path.join(ROOT, "tests", "golden", "g47_omnisr_grad.npz"))
    g = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("x2/")}
    net = OmniSR(input_shape=1, upscale=2, num_feat=16, res_num=2, block_num=1)
    layout = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    net.load_state_dict(O.seeded_state_dict(layout, int(g["seed"])), strict=True)
    net.train()
    x, tgt = g["x"], g["tgt"]
    eng = net.engine
    y = eng.forward(x[:, 0].contiguous(), None, save=True)
    assert (y - g["y"]).abs().max().item() <= 2e-5 * g["y"].abs().max().item()
    dy = torch.sign(y - tgt) / y.numel()
    grads = {k: torch.full_like(p, float("nan")) for k, p in net.named_parameters()}
    eng.backward(dy, grads)
    n = 0
    for k, got in grads.items():
        ref = g["grad/" + k]
        e = ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
        assert e <= 2e-4, (k, e)
        n += 1
    assert n == int(g["n_grads"])


def test_grl_tape_wiring_against_reference_gradients(monkeypatch):
    """srhip/grl_engine.py::_forward_tape: cosine window attention (shifted: bias + mask per window), the anchored stripe
    attention both ways, logit scales (one over the clamp), the CPB MLPs, the local conv branch -- against
    tests/golden/g48_grl_grad.npz."""
    import emul_ops
    import sr_oracle as O
    from dlib.models.network_grl import GRL
    emul_ops.install(monkeypatch)
    z = np.load(os.path.join(ROOT, "tests", "golden", "g48_grl_grad.npz"))
    g = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("x2/")}
    net = GRL(upscale=2, img_size=16, in_chans=1, window_size=8, mlp_ratio=2, qkv_proj_type="linear", anchor_proj_type="avgpool",
              anchor_window_down_factor=2, out_proj_type="linear", conv_type="1conv", upsampler="pixelshuffle",
              local_connection=True, depths=[2, 2], embed_dim=36, num_heads_window=[3, 3], num_heads_stripe=[3, 3],
              drop_path_rate=0.0)
    layout = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    net.load_state_dict(O.grl_state_dict(layout, int(g["seed"]), 16), strict=True)
    net.train()
    x, tgt = g["x"], g["tgt"]
    eng = net.engine
    y = eng.forward(x[:, 0].contiguous(), None, save=True)
    assert (y - g["y"]).abs().max().item() <= 2e-5 * g["y"].abs().max().item()
    dy = torch.sign(y - tgt) / y.numel()
    grads = {k: torch.full_like(p, float("nan")) for k, p in net.named_parameters()}
    eng.backward(dy, grads)
    n = 0
    for k, got in grads.items():
        if "grad/" + k in g:
            ref = g["grad/" + k]
            e = ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
        else:
            ref, sums = g["gslice/" + k], g["gsum/" + k]
            e = ((got[:2] - ref).abs().max() / sums[2].float().clamp_min(1e-30)).item()
            assert abs(got.double().sum().item() - sums[0].item()) <= 1e-4 * sums[1].item(), k
        assert e <= (2e-3 if k.endswith("logit_scale") else 2e-4), (k, e)
        n += 1
    assert n == int(g["n_grads"])
    # stochastic depth (network_grl.py:1058-1066): multipliers of one leave the image as it is, zeros drop both branches
    x3 = x[:, 0].contiguous()
    y1 = eng._forward_tape(x3, torch.ones(8, 2)).clone()          # (the tape's buffers are persistent: clone what is kept)
    y0 = eng._forward_tape(x3, torch.zeros(8, 2)).clone()
    assert torch.allclose(y1, g["y"], atol=2e-5 * g["y"].abs().max().item()) and (y0 - y1).abs().max().item() > 1e-3
    net.drop_probs = [0.0, 0.1, 0.2, 0.3]
    dp = net.sample_drop_path(2, "cpu")
    assert dp.shape == (8, 2) and set(dp[:2].flatten().tolist()) == {1.0} and net.eval().sample_drop_path(2, "cpu") is None


def test_swinir_general_window_tape_wiring_against_reference_gradients(monkeypatch):
    """srhip/swinir_tape_engine.py (window_size 4, qk_scale 0.3, ape, shifted odd blocks, 'pixelshuffle' tail) against
    tests/golden/g49_swinir_window4.npz: state_dict layout, eval forward, every parameter gradient."""
    import emul_ops
    from dlib.models.network_swinir import SwinIR
    emul_ops.install(monkeypatch)
    z = np.load(os.path.join(ROOT, "tests", "golden", "g49_swinir_window4.npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    net = SwinIR(upscale=2, in_chans=1, img_size=16, window_size=4, depths=[2, 2], embed_dim=60, num_heads=[6, 6], mlp_ratio=2,
                 upsampler="pixelshuffle", drop_path_rate=0.0, qk_scale=0.3, ape=True)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd/")}
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd, strict=True)
    eng = net.engine
    assert type(eng).__name__ == "SwinIRTapeEngine"
    x = g["x"][:, 0].contiguous()
    y_eval = eng.forward(x, None, save=False)
    assert (y_eval - g["y_eval"]).abs().max().item() <= 2e-5
    y = eng.forward(x, None, save=True)
    dy = torch.sign(y - g["target"]) / y.numel()
    grads = {k: torch.full_like(p, float("nan")) for k, p in net.named_parameters()}
    eng.backward(dy, grads)
    for k, got in grads.items():
        ref = g["grad/" + k]
        e = ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()
        assert e <= 2e-4, (k, e)


import pytest  # noqa: E402


@pytest.mark.parametrize("name,kw", [
    ("g43_swinir_rgb_direct", dict(upscale=2, in_chans=3, depths=[2], num_heads=[6], upsampler="pixelshuffledirect", img_range=2.0)),
    ("g44_swinir_rgb_pixelshuffle", dict(upscale=2, in_chans=3, depths=[2], num_heads=[6], upsampler="pixelshuffle", img_range=2.0)),
    ("g27_swinir_3conv", dict(upscale=4, in_chans=1, depths=[2, 2], num_heads=[6, 6], upsampler="pixelshuffledirect",
                              resi_connection="3conv")),
])
def test_swinir_tape_engine_rgb_and_3conv_against_reference_gradients(monkeypatch, name, kw):
    """The general tape graph on configurations the fused engine also runs (8 x 8 windows): RGB input / output convs and the
    '3conv' residual convs as im2col GEMMs -- reference goldens g43 / g44 / g27."""
    import emul_ops
    import torch.nn.functional as F
    from dlib.models.network_swinir import SwinIR
    from srhip.swinir_tape_engine import SwinIRTapeEngine
    emul_ops.install(monkeypatch)
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    net = SwinIR(img_size=16, window_size=8, embed_dim=60, mlp_ratio=2, drop_path_rate=0.0, **kw)
    net.load_state_dict({k[3:]: v for k, v in g.items() if k.startswith("sd/")}, strict=True)
    eng = SwinIRTapeEngine(net)
    x, rng, ci = g["x"], float(kw.get("img_range", 1.0)), kw["in_chans"]
    if ci == 1:
        xi, mean = x[:, 0].contiguous(), 0.0
    else:
        mean = net.mean
        xi = F.pad(((x - mean) * rng).permute(0, 2, 3, 1), (0, 4 - ci)).contiguous()
    y = eng.forward(xi, None, save=True) / rng + mean
    assert (y - g["y_eval"]).abs().max().item() <= 2e-5
    dy = torch.sign(y - g["target"]) / y.numel() / rng
    grads = {k: torch.full_like(p, float("nan")) for k, p in net.named_parameters()}
    eng.backward(dy.contiguous(), grads)
    for k, got in grads.items():
        ref = g["grad/" + k]
        e = ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()
        assert e <= 2e-4, (k, e)


def test_swinir_tape_engine_droppath_multipliers_against_the_oracle(monkeypatch):
    """DropPath on the general tape graph (row 2 i of the multipliers on block i's attention branch, 2 i + 1 on its MLP branch,
    network_swinir.py:334-335): forward and gradients against the oracle run with the same per-sample multipliers."""
    import emul_ops
    import sr_oracle as O
    from dlib.models.network_swinir import SwinIR
    emul_ops.install(monkeypatch)
    cfg = O.swinir_config(upscale=2, in_chans=1, img_size=16, window_size=4, depths=(2, 2), embed_dim=60, num_heads=(6, 6),
                          mlp_ratio=2, upsampler="pixelshuffledirect", drop_path_rate=0.2, qk_scale=None)
    sd = O.swinir_init_state_dict(cfg, seed=5)
    net = SwinIR(upscale=2, in_chans=1, img_size=16, window_size=4, depths=[2, 2], embed_dim=60, num_heads=[6, 6], mlp_ratio=2,
                 upsampler="pixelshuffledirect", drop_path_rate=0.2)
    net.load_state_dict(sd, strict=True)
    gen = torch.Generator().manual_seed(6)
    x, tgt = torch.rand(3, 1, 16, 16, generator=gen), torch.rand(3, 1, 32, 32, generator=gen)
    dp = torch.tensor([[0.0, 1.25, 1.25], [1.25, 1.25, 0.0], [1.0, 1.0, 1.0], [1.25, 0.0, 1.25],
                       [1.5, 1.5, 0.0], [0.0, 1.5, 1.5], [2.0, 0.0, 0.0], [1.0, 1.0, 1.0]])
    eng = net.engine
    y = eng.forward(x[:, 0].contiguous(), dp, save=True)
    sdo = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith("attn_mask") else v) for k, v in sd.items()}
    yo = O.swinir_forward(sdo, x, cfg, dp_scales=dp.view(4, 2, 3))
    assert (y - yo.detach()).abs().max().item() <= 2e-5
    (yo - tgt).abs().mean().backward()
    grads = {k: torch.full_like(p, float("nan")) for k, p in net.named_parameters()}
    eng.backward((torch.sign(y - tgt) / y.numel()).contiguous(), grads)
    for k, got in grads.items():
        ref = sdo[k].grad
        assert ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item() <= 2e-4, k
