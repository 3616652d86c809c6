"""GPU end-to-end parity of the HIP SwinIR (dlib.models.network_swinir) against
the golden fixtures generated from the real reference and against the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import sr_oracle as O  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# parameter / input gradients relative to the tensor's largest entry: <= 10x the margins measured on
# the MI355X (2.2e-6 on the README configuration, 9e-7 on the tiny net)
GRAD_GATE = 2e-5


def load(name):
    z = np.load(os.path.join(G, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiu" else z[k]) for k in z.files}


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(scope="module")
def SwinIR():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from dlib.models.network_swinir import SwinIR as cls
    return cls


def tiny(SwinIR, dpr=0.0):
    return SwinIR(upscale=8, in_chans=1, img_size=16, window_size=8, depths=[2, 2], embed_dim=60,
                  num_heads=[6, 6], mlp_ratio=2, upsampler="pixelshuffledirect", drop_path_rate=dpr)


def readme(SwinIR, dpr=0.1):
    return SwinIR(upscale=8, in_chans=1, img_size=64, window_size=8, depths=[6, 6, 6, 6], embed_dim=180,
                  num_heads=[6, 6, 6, 6], mlp_ratio=2, upsampler="pixelshuffledirect",
                  drop_path_rate=dpr)


def psnr(a, b):
    return O.metric_psnr(O.tensor2uint82float(a), O.tensor2uint82float(b), 8)


def test_state_dict_layout_matches_reference(SwinIR):
    g = load("g4_swinir_readme")
    net = readme(SwinIR)
    sd = net.state_dict()
    assert list(sd.keys()) == list(g["keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["shapes"])
    assert sum(p.numel() for p in net.parameters()) == 7865884


def test_tiny_forward_backward_vs_reference_golden(SwinIR):
    g = load("g3_swinir_tiny")
    net = tiny(SwinIR)
    net.load_state_dict(sub(g, "sd/"), strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(g["x"].cuda()).cpu()
    err = (y - g["y_eval"]).abs().max().item()
    assert err <= 1e-5, f"eval forward max err {err}"      # gate is pixel MAE 1e-5; max is stricter
    # training-mode forward + backward, drop path neutralised as in the golden
    net.train()
    for b in net.swin_blocks():
        b.drop_prob = 0.0
    x = g["x"].cuda().requires_grad_(True)
    yt = net(x)
    (yt - g["target"].cuda()).abs().mean().backward()
    assert (x.grad.cpu() - g["dx"]).abs().max() <= GRAD_GATE * g["dx"].abs().max()
    worst = 0.0
    for k, p in net.named_parameters():
        ref = g["grad/" + k]
        e = (p.grad.cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        worst = max(worst, e)
        assert e <= GRAD_GATE, f"grad {k}: rel err {e:.2e}"
    print("worst param-grad rel err", worst)


def test_tiny_forced_droppath_and_padded(SwinIR):
    g = load("g3_swinir_tiny")
    net = tiny(SwinIR, 0.5)
    net.load_state_dict(sub(g, "sd/"), strict=True)
    net = net.cuda().train()
    dp = g["dp"].reshape(-1, 2).cuda().contiguous()      # [blocks,2(branch),B] -> [2*blocks, B]
    with torch.no_grad():
        y = net(g["x"].cuda(), dp=dp).cpu()
    assert (y - g["y_dp"]).abs().max() <= 1e-5
    net.eval()
    with torch.no_grad():
        y = net(g["x_pad"].cuda()).cpu()
    assert y.shape == g["y_pad"].shape
    assert (y - g["y_pad"]).abs().max() <= 1e-5


@pytest.fixture(params=["separate", "fused_mlp"])
def mlp_path(request, monkeypatch):
    """README-configuration tests run on both forms of the MLP half of a block: the separate Linear
    launches (SRHIP_MLP_F16=0) and the fused kernels of mlp_f16.hip (default)."""
    monkeypatch.setenv("SRHIP_MLP_F16", "1" if request.param == "fused_mlp" else "0")
    return request.param


def test_readme_config_forward_vs_reference_golden(SwinIR, mlp_path):
    g = load("g4_swinir_readme")
    cfg = O.swinir_config()
    sd = O.swinir_init_state_dict(cfg, seed=0)
    net = readme(SwinIR)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(g["x"].cuda()).cpu()
    assert net.engine.fuse_mlp_h == (mlp_path == "fused_mlp")
    mae = (y - g["y"]).abs().mean().item()
    assert mae <= 1e-5, f"pixel MAE {mae}"
    tgt = torch.rand(1, 1, 512, 512, generator=torch.Generator().manual_seed(1))
    assert (psnr(y, tgt) - psnr(g["y"], tgt)).abs().max() <= 0.01
    # padded-eval path of the trainer: 72x72 (81 windows, mask from indices)
    xp = torch.rand(1, 1, 72, 72, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        yp = net(xp.cuda()).cpu()
        yo = O.swinir_forward(sd, xp, cfg)
    assert (yp - yo).abs().mean() <= 1e-5


def test_readme_config_train_step_grads_vs_oracle(SwinIR, mlp_path):
    cfg = O.swinir_config(drop_path_rate=0.0)
    sd = O.swinir_init_state_dict(cfg, seed=0)
    net = readme(SwinIR, 0.0)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    gen = torch.Generator().manual_seed(3)
    x, tgt = torch.rand(2, 1, 64, 64, generator=gen), torch.rand(2, 1, 512, 512, generator=gen)
    y = net(x.cuda())
    (y - tgt.cuda()).abs().mean().backward()
    sdo = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith("attn_mask")
               else v) for k, v in sd.items()}
    yo = O.swinir_forward(sdo, x, cfg)
    (yo - tgt).abs().mean().backward()
    assert (y.detach().cpu() - yo.detach()).abs().mean() <= 1e-5
    worst = ("", 0.0)
    for k, p in net.named_parameters():
        ref = sdo[k].grad
        e = (p.grad.cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        if e > worst[1]:
            worst = (k, e)
    print("worst grad", worst)
    assert worst[1] <= GRAD_GATE, worst


def test_pixelshuffle_upsampler_vs_reference_golden_and_oracle(SwinIR):
    """upsampler 'pixelshuffle' -- the registry default (utils_init_default_args.py:23): conv 180->64 +
    LeakyReLU(0.01) as a conv epilogue, log2(s) x [conv 64->256 + PixelShuffle(2)], conv 64->1
    (network_swinir.py:862-868,937-942).  Tiny x4 net against the reference golden g20 (forward, dL/dx, all
    gradients; state_dict keys / order), README trunk x8 against the oracle."""
    g = load("g20_swinir_pixelshuffle")
    net = SwinIR(upscale=4, in_chans=1, img_size=16, window_size=8, depths=[2, 2], embed_dim=60,
                 num_heads=[6, 6], mlp_ratio=2, upsampler="pixelshuffle", drop_path_rate=0.0)
    assert list(net.state_dict().keys()) == list(sub(g, "sd/").keys())
    net.load_state_dict(sub(g, "sd/"), strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(g["x"].cuda()).cpu()
    assert y.shape == g["y_eval"].shape and (y - g["y_eval"]).abs().max() <= 1e-5
    net.train()
    x = g["x"].cuda().requires_grad_(True)
    (net(x) - g["target"].cuda()).abs().mean().backward()
    assert (x.grad.cpu() - g["dx"]).abs().max() <= GRAD_GATE * g["dx"].abs().max()
    worst = ("", 0.0)
    for k, p in net.named_parameters():
        ref = g["grad/" + k]
        e = (p.grad.cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        gate = 1e-4 if k.endswith(".bias") and ("upsample" in k or "conv_" in k) else GRAD_GATE
        assert e <= gate, f"grad {k}: rel err {e:.2e}"
        worst = max(worst, (k, e), key=lambda t: t[1])
    print("pixelshuffle tiny: worst grad", worst)
    # README trunk, x8 (three upsampling stages up to 512 x 512 x 64), against the oracle
    cfg = O.swinir_config(upsampler="pixelshuffle", drop_path_rate=0.0)
    sd = O.swinir_init_state_dict(cfg, seed=5)
    big = SwinIR(upscale=8, in_chans=1, img_size=64, window_size=8, depths=[6, 6, 6, 6], embed_dim=180,
                 num_heads=[6, 6, 6, 6], mlp_ratio=2, upsampler="pixelshuffle", drop_path_rate=0.0)
    big.load_state_dict(sd, strict=True)
    big = big.cuda().train()
    gen = torch.Generator().manual_seed(6)
    xb, tb = torch.rand(1, 1, 64, 64, generator=gen), torch.rand(1, 1, 512, 512, generator=gen)
    yb = big(xb.cuda())
    (yb - tb.cuda()).abs().mean().backward()
    sdo = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith("attn_mask")
               else v) for k, v in sd.items()}
    yo = O.swinir_forward(sdo, xb, cfg)
    (yo - tb).abs().mean().backward()
    assert (yb.detach().cpu() - yo.detach()).abs().mean() <= 1e-5
    assert (psnr(yb.detach().cpu(), tb) - psnr(yo.detach(), tb)).abs().max() <= 0.01
    worst = ("", 0.0)
    for k, p in big.named_parameters():
        ref = sdo[k].grad
        e = (p.grad.cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        worst = max(worst, (k, e), key=lambda t: t[1])
    print("pixelshuffle README x8: worst grad", worst)
    assert worst[1] <= 1e-4, worst      # bias sums over 262144 pixels: fp32 summation order (see g20)


def test_nearest_conv_upsampler_vs_reference_golden_and_oracle(SwinIR):
    """upsampler 'nearest_conv' (x4; network_swinir.py:874-885,948-961): conv 180->64 + LeakyReLU(0.01), 2 x [nearest
    x2 (index kernel + its adjoint), conv 64->64, LeakyReLU(0.2)], conv_hr + LeakyReLU(0.2), conv_last.  Tiny net against
    the reference golden g25 (forward, dL/dx, all gradients; state_dict keys / order), README trunk against the oracle."""
    g = load("g25_swinir_nearest_conv")
    net = SwinIR(upscale=4, in_chans=1, img_size=16, window_size=8, depths=[2, 2], embed_dim=60,
                 num_heads=[6, 6], mlp_ratio=2, upsampler="nearest_conv", drop_path_rate=0.0)
    assert list(net.state_dict().keys()) == list(sub(g, "sd/").keys())
    net.load_state_dict(sub(g, "sd/"), strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(g["x"].cuda()).cpu()
    assert y.shape == g["y_eval"].shape and (y - g["y_eval"]).abs().max() <= 1e-5
    net.train()
    x = g["x"].cuda().requires_grad_(True)
    (net(x) - g["target"].cuda()).abs().mean().backward()
    # four LeakyReLU layers at up to 64 x 96 pixels behind the trunk: an activation within f32 rounding of zero takes the
    # other slope than in the reference and moves everything upstream by ~1 / pixels (measured: 2e-5 on every tensor
    # from conv_hr down, 3e-7 on conv_last.weight above it) -- tensor-wise relative L2 gate on this tiny fixture, the
    # entry-wise gate on the README trunk below
    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    assert l2(x.grad.cpu(), g["dx"]) <= 2e-3
    worst = ("", 0.0)
    for k, p in net.named_parameters():
        e = l2(p.grad.cpu(), g["grad/" + k])
        assert e <= 1e-3, f"grad {k}: relative L2 error {e:.2e}"      # bias-table sums cancel: 2.5e-4 with one flipped pixel
        worst = max(worst, (k, e), key=lambda t: t[1])
    print("nearest_conv tiny: worst grad", worst)
    cfg = O.swinir_config(upscale=4, upsampler="nearest_conv", drop_path_rate=0.0)
    sd = O.swinir_init_state_dict(cfg, seed=7)
    big = SwinIR(upscale=4, in_chans=1, img_size=64, window_size=8, depths=[6, 6, 6, 6], embed_dim=180,
                 num_heads=[6, 6, 6, 6], mlp_ratio=2, upsampler="nearest_conv", drop_path_rate=0.0)
    big.load_state_dict(sd, strict=True)
    big = big.cuda().train()
    gen = torch.Generator().manual_seed(8)
    xb, tb = torch.rand(1, 1, 64, 64, generator=gen), torch.rand(1, 1, 256, 256, generator=gen)
    yb = big(xb.cuda())
    (yb - tb.cuda()).abs().mean().backward()
    sdo = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith("attn_mask")
               else v) for k, v in sd.items()}
    yo = O.swinir_forward(sdo, xb, cfg)
    (yo - tb).abs().mean().backward()
    assert (yb.detach().cpu() - yo.detach()).abs().mean() <= 1e-5
    worst, worst_tab = ("", 0.0), ("", 0.0)
    for k, p in big.named_parameters():
        ref = sdo[k].grad
        e = (p.grad.cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        if k.endswith("relative_position_bias_table"):
            worst_tab = max(worst_tab, (k, e), key=lambda t: t[1])
        else:
            worst = max(worst, (k, e), key=lambda t: t[1])
    print("nearest_conv README trunk x4: worst grad", worst, "worst bias table", worst_tab)
    assert worst[1] <= 1e-4, worst
    # the bias-table gradients are sums of 4096 signed window entries: the float32 ORACLE is 3e-4 away from float64 on
    # them (tests/test_gpu_fullsize.py gates them against the float64 oracle)
    assert worst_tab[1] <= 2e-3, worst_tab


def test_3conv_residual_connection_vs_reference_golden_and_oracle(SwinIR):
    """resi_connection '3conv' (network_swinir.py:545-552, 851-858): conv C -> C/4 + LeakyReLU(0.2), conv1x1 + LeakyReLU(0.2),
    conv C/4 -> C in front of every residual connection, run on the '1conv' kernels with the C/4 channels zero-padded
    (15 -> 16 on the exact-f32 kernels of the tiny net, 45 -> 64 on the bf16x3 kernels of the README trunk).  Tiny net
    against the reference golden g27 (forward, dL/dx, all gradients; state_dict keys / order), README trunk against
    the oracle."""
    g = load("g27_swinir_3conv")
    net = SwinIR(upscale=4, in_chans=1, img_size=16, window_size=8, depths=[2, 2], embed_dim=60,
                 num_heads=[6, 6], mlp_ratio=2, upsampler="pixelshuffledirect", resi_connection="3conv", drop_path_rate=0.0)
    assert list(net.state_dict().keys()) == list(sub(g, "sd/").keys())
    net.load_state_dict(sub(g, "sd/"), strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(g["x"].cuda()).cpu()
    assert y.shape == g["y_eval"].shape and (y - g["y_eval"]).abs().max() <= 1e-5
    net.train()
    x = g["x"].cuda().requires_grad_(True)
    (net(x) - g["target"].cuda()).abs().mean().backward()
    # tensor-wise relative L2 on the tiny fixture (measured: 2.7e-6 worst; a LeakyReLU decision that flips under f32
    # rounding would show as ~1e-4, see the nearest_conv test)
    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    assert l2(x.grad.cpu(), g["dx"]) <= 2e-4
    worst = ("", 0.0)
    for k, p in net.named_parameters():
        e = l2(p.grad.cpu(), g["grad/" + k])
        assert e <= 2e-4, f"grad {k}: relative L2 error {e:.2e}"
        worst = max(worst, (k, e), key=lambda t: t[1])
    print("3conv tiny: worst grad", worst)
    cfg = O.swinir_config(upscale=8, resi_connection="3conv", drop_path_rate=0.0)
    sd = O.swinir_init_state_dict(cfg, seed=9)
    big = SwinIR(upscale=8, in_chans=1, img_size=64, window_size=8, depths=[6, 6, 6, 6], embed_dim=180,
                 num_heads=[6, 6, 6, 6], mlp_ratio=2, upsampler="pixelshuffledirect", resi_connection="3conv",
                 drop_path_rate=0.0)
    assert big.engine.c4 == 45 and big.engine.c4p == 64
    big.load_state_dict(sd, strict=True)
    big = big.cuda().train()
    # A LeakyReLU pre-activation within f32 rounding of zero may take the other branch than the oracle's: ONE such pixel
    # shows as ~4e-5 in a weight gradient of this trunk (both branches are correct f32 results; which inputs have such a
    # pixel depends on the kernels' summation order, which round 5 changed per block).  So: two batches; every one within
    # the flip-tolerant bound, and at least one at the tight one (measured 1.9e-6) -- an error of the kernels fails both.
    tight = []
    for seed in (11, 12):       # (of seeds 10 .. 15 only 10 has such a pixel under the round-5 kernels: 4.4e-5 / 2.7e-4)
        gen = torch.Generator().manual_seed(seed)
        xb, tb = torch.rand(1, 1, 64, 64, generator=gen), torch.rand(1, 1, 512, 512, generator=gen)
        for p in big.parameters():
            p.grad = None
        yb = big(xb.cuda())
        (yb - tb.cuda()).abs().mean().backward()
        sdo = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith("attn_mask")
                   else v) for k, v in sd.items()}
        yo = O.swinir_forward(sdo, xb, cfg)
        (yo - tb).abs().mean().backward()
        assert (yb.detach().cpu() - yo.detach()).abs().mean() <= 1e-5
        worst, worst_tab = ("", 0.0), ("", 0.0)
        for k, p in big.named_parameters():
            ref = sdo[k].grad
            e = (p.grad.cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
            if k.endswith("relative_position_bias_table"):
                worst_tab = max(worst_tab, (k, e), key=lambda t: t[1])
            else:
                worst = max(worst, (k, e), key=lambda t: t[1])
        print("3conv README trunk x8, batch", seed, ": worst grad", worst, "worst bias table", worst_tab)
        assert worst[1] <= 2e-4, worst               # a flipped LeakyReLU pixel or two
        assert worst_tab[1] <= 5e-4, worst_tab       # measured 6.8e-7 (float32 summation order of 4096 signed entries); 2.7e-4 behind a flipped pixel
        tight.append(max(worst[1], 0.1 * worst_tab[1]))
    assert min(tight) <= 2e-5, tight                 # no flip: measured 1.9e-6


def test_absolute_position_embedding_vs_reference_golden(SwinIR):
    """ape=True (network_swinir.py:812-815, 918-919): the learned [1, img_size^2, C] table added to the tokens after
    patch_embed.norm, its gradient the batch sum of the token gradient.  Reference golden g42: state_dict keys / order
    (the table comes first), eval forward, dL/dx, every gradient; an input that is not img_size x img_size fails as
    the reference's broadcast does."""
    g = load("g42_swinir_ape")
    net = SwinIR(upscale=2, in_chans=1, img_size=16, window_size=8, depths=[2, 2], embed_dim=60,
                 num_heads=[6, 6], mlp_ratio=2, upsampler="pixelshuffledirect", drop_path_rate=0.0, ape=True)
    assert list(net.state_dict().keys()) == list(sub(g, "sd/").keys())
    assert "absolute_pos_embed" in net.no_weight_decay()
    net.load_state_dict(sub(g, "sd/"), strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(g["x"].cuda()).cpu()
        with pytest.raises(RuntimeError):
            net(torch.rand(1, 1, 16, 24).cuda())
    assert y.shape == g["y_eval"].shape and (y - g["y_eval"]).abs().max() <= 1e-5
    net.train()
    x = g["x"].cuda().requires_grad_(True)
    (net(x) - g["target"].cuda()).abs().mean().backward()
    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    assert l2(x.grad.cpu(), g["dx"]) <= 2e-4
    worst = ("", 0.0)
    for k, p in net.named_parameters():
        e = l2(p.grad.cpu(), g["grad/" + k])
        assert e <= 2e-4, f"grad {k}: relative L2 error {e:.2e}"
        worst = max(worst, (k, e), key=lambda t: t[1])
    print("ape tiny: worst grad", worst, "table", l2(net.absolute_pos_embed.grad.cpu(), g["grad/absolute_pos_embed"]))


@pytest.mark.parametrize("name,ups", [("g43_swinir_rgb_direct", "pixelshuffledirect"),
                                      ("g44_swinir_rgb_pixelshuffle", "pixelshuffle")])
def test_three_image_channels_vs_reference_golden(SwinIR, name, ups):
    """in_chans=3 (network_swinir.py:722-727, 934-935, 968): the RGB mean and img_range around the network, conv_first 3 -> C
    and conv_last 64 -> 3 on the exact-f32 conv kernel with the image channels zero-padded to 4.  Reference goldens
    g43 / g44: keys / order, eval forward, dL/dx, every gradient; the fused training step (srhip/train.py) lands on the
    same gradients."""
    from srhip.train import TrainStep
    g = load(name)
    net = SwinIR(upscale=2, in_chans=3, img_size=16, window_size=8, depths=[2], embed_dim=60, num_heads=[6], mlp_ratio=2,
                 upsampler=ups, drop_path_rate=0.0, img_range=2.0)
    assert list(net.state_dict().keys()) == list(sub(g, "sd/").keys())
    net.load_state_dict(sub(g, "sd/"), strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(g["x"].cuda()).cpu()
    assert y.shape == g["y_eval"].shape and (y - g["y_eval"]).abs().max() <= 1e-5
    net.train()
    x = g["x"].cuda().requires_grad_(True)
    (net(x) - g["target"].cuda()).abs().mean().backward()
    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    assert l2(x.grad.cpu(), g["dx"]) <= 2e-4
    worst = ("", 0.0)
    for k, p in net.named_parameters():
        e = l2(p.grad.cpu(), g["grad/" + k])
        assert e <= 2e-4, f"grad {k}: relative L2 error {e:.2e}"
        worst = max(worst, (k, e), key=lambda t: t[1])
    print(name, "worst grad", worst)
    # the fused step (y / img_range + mean in front of the loss): loss and gradients of the same batch
    from srhip.train import Optimizer
    net.load_state_dict(sub(g, "sd/"), strict=True)
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "sgd", lr=1e-12, momentum=0.0, nesterov=False, wd=0.0)
    ts.step(g["x"].cuda(), g["target"].cuda())
    torch.cuda.synchronize()
    ref_loss = (g["y_eval"] - g["target"]).abs().mean().item()
    assert abs(ts.loss_buf[1].item() - ref_loss) <= 1e-5 * max(1.0, ref_loss), (ts.loss_buf, ref_loss)      # [total, terms...]
    for k, _ in net.named_parameters():
        assert l2(ts.fp.gviews[k].cpu(), g["grad/" + k]) <= 2e-4, k


def test_three_image_channels_nearest_conv_vs_oracle(SwinIR):
    """in_chans=3 with the 'nearest_conv' tail (x4): forward and every gradient against the oracle's autograd (the oracle
    is pinned to the reference on this tail by g25 and on RGB by g43 / g44)."""
    cfg = O.swinir_config(upscale=4, in_chans=3, img_size=16, window_size=8, depths=(2,), embed_dim=60, num_heads=(6,),
                          mlp_ratio=2, upsampler="nearest_conv", drop_path_rate=0.0)
    sd = O.swinir_init_state_dict(cfg, seed=91)
    net = SwinIR(upscale=4, in_chans=3, img_size=16, window_size=8, depths=[2], embed_dim=60, num_heads=[6], mlp_ratio=2,
                 upsampler="nearest_conv", drop_path_rate=0.0)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    # (LeakyReLU decisions within f32 rounding of zero: see the '3conv' test -- two batches, each within the flip-tolerant
    # bound, one at least at the tight one)
    tight = []
    for seed in (95, 96):       # (of seeds 92 .. 103, 92 / 93 / 94 have such pixels under the round-5 kernels: 4e-4 .. 2e-3)
        gen = torch.Generator().manual_seed(seed)
        x, t = torch.rand(2, 3, 16, 16, generator=gen), torch.rand(2, 3, 64, 64, generator=gen)
        for p in net.parameters():
            p.grad = None
        xg = x.cuda().requires_grad_(True)
        y = net(xg)
        (y - t.cuda()).abs().mean().backward()
        sdo = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith("attn_mask") else v)
               for k, v in sd.items()}
        xo = x.clone().requires_grad_(True)
        yo = O.swinir_forward(sdo, xo, cfg)
        (yo - t).abs().mean().backward()
        assert (y.detach().cpu() - yo.detach()).abs().max() <= 1e-5
        errs = [l2(xg.grad.cpu(), xo.grad)] + [l2(p.grad.cpu(), sdo[k].grad) for k, p in net.named_parameters()]
        print("nearest_conv RGB, batch", seed, ": worst relative L2", max(errs))
        assert max(errs) <= 3e-3, (seed, max(errs))
        tight.append(max(errs))
    assert min(tight) <= 2e-4, tight


def test_without_patch_norm_and_qkv_bias_vs_reference_golden(SwinIR):
    """patch_norm=False, qkv_bias=False (network_swinir.py:799-803, 104): the tokens are conv_first's output, the folded qkv
    bias is W beta alone.  Reference golden g46: keys / order (no patch_embed.*, no attn.qkv.bias), eval forward, dL/dx, every
    gradient through the autograd path and through the fused step."""
    from srhip.train import TrainStep, Optimizer
    g = load("g46_swinir_plain_embed")
    net = SwinIR(upscale=2, in_chans=1, img_size=16, window_size=8, depths=[2], embed_dim=60, num_heads=[6], mlp_ratio=2,
                 upsampler="pixelshuffledirect", drop_path_rate=0.0, patch_norm=False, qkv_bias=False)
    assert list(net.state_dict().keys()) == list(sub(g, "sd/").keys())
    net.load_state_dict(sub(g, "sd/"), strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(g["x"].cuda()).cpu()
    assert y.shape == g["y_eval"].shape and (y - g["y_eval"]).abs().max() <= 1e-5
    net.train()
    x = g["x"].cuda().requires_grad_(True)
    (net(x) - g["target"].cuda()).abs().mean().backward()
    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    assert l2(x.grad.cpu(), g["dx"]) <= 2e-4
    for k, p in net.named_parameters():
        assert l2(p.grad.cpu(), g["grad/" + k]) <= 2e-4, k
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "sgd", lr=1e-12, momentum=0.0, nesterov=False, wd=0.0)
    ts.step(g["x"].cuda(), g["target"].cuda())
    for k, _ in net.named_parameters():
        assert l2(ts.fp.gviews[k].cpu(), g["grad/" + k]) <= 2e-4, k


def test_general_window_and_qk_scale_on_the_tape_graph_vs_reference_golden(SwinIR):
    """window_size=4, qk_scale=0.3 (network_swinir.py:102, 232-236), ape, shifted odd blocks, 'pixelshuffle' tail: the general
    tape graph (srhip/swinir_tape_engine.py; the fused engine owns 8 x 8 windows).  Reference golden g49: keys / order, eval
    forward, every gradient through the autograd path and through the fused step; DropPath multipliers scale the branches."""
    from srhip.train import TrainStep, Optimizer
    g = load("g49_swinir_window4")
    net = SwinIR(upscale=2, in_chans=1, img_size=16, window_size=4, depths=[2, 2], embed_dim=60, num_heads=[6, 6], mlp_ratio=2,
                 upsampler="pixelshuffle", drop_path_rate=0.0, qk_scale=0.3, ape=True)
    assert list(net.state_dict().keys()) == list(sub(g, "sd/").keys())
    net.load_state_dict(sub(g, "sd/"), strict=True)
    net = net.cuda().eval()
    assert type(net.engine).__name__ == "SwinIRTapeEngine"
    with torch.no_grad():
        y = net(g["x"].cuda()).cpu()
    assert y.shape == g["y_eval"].shape and (y - g["y_eval"]).abs().max() <= 1e-5
    net.train()
    (net(g["x"].cuda()) - g["target"].cuda()).abs().mean().backward()
    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    worst = ("", 0.0)
    for k, p in net.named_parameters():
        e = l2(p.grad.cpu(), g["grad/" + k])
        assert e <= 2e-4, f"grad {k}: relative L2 error {e:.2e}"
        worst = max(worst, (k, e), key=lambda t: t[1])
    print("window 4 / qk_scale: worst grad", worst)
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "sgd", lr=1e-12, momentum=0.0, nesterov=False, wd=0.0)
    ts.step(g["x"].cuda(), g["target"].cuda())
    for k, _ in net.named_parameters():
        assert l2(ts.fp.gviews[k].cpu(), g["grad/" + k]) <= 2e-4, k
    # DropPath multipliers: zeros on every branch leave the blocks as identities
    dp0 = torch.zeros(8, 2, device="cuda")
    with torch.no_grad():
        xi, _, _ = net.prepare_input(g["x"].cuda())
        y0 = net.engine.forward(xi, dp0, save=False)
        y1 = net.engine.forward(xi, torch.ones(8, 2, device="cuda"), save=False)
    assert not torch.equal(y0, y1) and (y1.cpu() - g["y_eval"]).abs().max() <= 1e-5


def test_tape_graph_rgb_3conv_nearest_conv_window4_vs_oracle(SwinIR):
    """Everything the fused engine does not take at once -- window 4, qk_scale, RGB, '3conv', 'nearest_conv' x4, no patch norm,
    no qkv bias -- on the general tape graph against the oracle's autograd (the oracle is pinned to the reference on each of
    these by g25, g27, g43, g44, g46, g49)."""
    cfg = O.swinir_config(upscale=4, in_chans=3, img_size=16, window_size=4, depths=(2,), embed_dim=60, num_heads=(6,), mlp_ratio=2,
                          upsampler="nearest_conv", resi_connection="3conv", drop_path_rate=0.0, qk_scale=0.25, patch_norm=False,
                          qkv_bias=False, img_range=2.0)
    sd = O.swinir_init_state_dict(cfg, seed=95)
    net = SwinIR(upscale=4, in_chans=3, img_size=16, window_size=4, depths=[2], embed_dim=60, num_heads=[6], mlp_ratio=2,
                 upsampler="nearest_conv", resi_connection="3conv", drop_path_rate=0.0, qk_scale=0.25, patch_norm=False,
                 qkv_bias=False, img_range=2.0)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    assert type(net.engine).__name__ == "SwinIRTapeEngine"
    gen = torch.Generator().manual_seed(96)
    x, tg = torch.rand(2, 3, 16, 20, generator=gen), torch.rand(2, 3, 64, 80, generator=gen)
    y = net(x.cuda())
    (y - tg.cuda()).abs().mean().backward()
    sdo = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith("attn_mask") else v)
           for k, v in sd.items()}
    yo = O.swinir_forward(sdo, x, cfg)
    (yo - tg).abs().mean().backward()
    assert (y.detach().cpu() - yo.detach()).abs().max() <= 1e-5
    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    for k, p in net.named_parameters():
        assert l2(p.grad.cpu(), sdo[k].grad) <= 2e-4, k


def test_step_graph_replays_the_eager_step_bit_for_bit(SwinIR):
    """TrainStep.step_graph (one hipGraph replay per step) against TrainStep.step (~70 launches for this net):
    20 steps from the same weights on a changing batch, SGD-Nesterov with a MyStepLR schedule that halves the rate
    every 3 iterations (the device-side learning rate must follow it) -- identical loss trajectories and
    parameters, bit for bit; then the README configuration at B=8: the host spends < 2 ms per step."""
    import time
    from srhip.train import TrainStep, Optimizer
    cfg = O.swinir_config(upscale=8, in_chans=1, img_size=16, window_size=8, depths=(2, 2), embed_dim=60,
                          num_heads=(6, 6), mlp_ratio=2, drop_path_rate=0.0)
    sd0 = O.trained_like_(O.swinir_init_state_dict(cfg, seed=81), 82, lin_scale=3.0)
    gen = torch.Generator().manual_seed(83)
    batches = [(torch.rand(2, 1, 16, 16, generator=gen).cuda(), torch.rand(2, 1, 128, 128, generator=gen).cuda())
               for _ in range(20)]
    runs = []
    for mode in ("eager", "graph", "eager"):
        net = tiny(SwinIR)
        net.load_state_dict(sd0, strict=True)
        net = net.cuda().train()
        ts = TrainStep(net, [("l1", 1.0)])
        ts.opt = Optimizer(ts.fp, "sgd", lr=0.05, momentum=0.9, nesterov=True, wd=0.0,
                           scheduler={"type": "MyStepLR", "step_size": 3, "gamma": 0.5, "min_lr": 1e-4})
        losses = []
        for lr_img, hr_img in batches:
            (ts.step if mode == "eager" else ts.step_graph)(lr_img, hr_img)
            losses.append(ts.loss_buf.clone())
        torch.cuda.synchronize()
        runs.append((torch.stack(losses).cpu(), ts.fp.flat.clone().cpu(), ts.opt.lr))
    assert runs[1][2] == runs[0][2] and runs[0][2] < 0.05 / 32
    assert torch.equal(runs[0][0], runs[1][0]), (runs[0][0][:, 0] - runs[1][0][:, 0]).abs().max()
    # parameters: bit for bit -- every reduction of the step is deterministic (the LayerNorm-affine gradients were the
    # last ones on float atomics; they are two-stage sums in a fixed order since round 3)
    d_graph = (runs[0][1] - runs[1][1]).abs().max().item()
    d_eager = (runs[0][1] - runs[2][1]).abs().max().item()
    print(f"step_graph vs step: max |param diff| {d_graph:.2e}; step vs step (second eager run): {d_eager:.2e}")
    assert torch.equal(runs[0][1], runs[2][1]), d_eager
    assert torch.equal(runs[0][1], runs[1][1]), d_graph
    assert runs[0][0][0, 1] != runs[0][0][-1, 1]          # it did train
    # README configuration, B = 8, DropPath live (masks drawn by captured generator ops): host time per step
    net = readme(SwinIR).cuda().train()
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "sgd", lr=0.01, momentum=0.9, nesterov=True, wd=0.0)
    lr_img, hr_img = torch.rand(8, 1, 64, 64).cuda(), torch.rand(8, 1, 512, 512).cuda()
    for _ in range(3):
        ts.step_graph(lr_img, hr_img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ts.step_graph(lr_img, hr_img)
    host_ms = (time.perf_counter() - t0) * 100.0
    torch.cuda.synchronize()
    total_ms = (time.perf_counter() - t0) * 100.0
    print(f"step_graph README B=8: host {host_ms:.2f} ms per step, device {total_ms:.2f} ms per step")
    assert host_ms < 2.0 and torch.isfinite(ts.loss_buf).all()


def test_step_graph_with_droppath_follows_the_per_step_seed(SwinIR):
    """ModelPlain replays the step from a hipGraph by default and the trainer re-seeds every rank with
    myseed + current_step before each iteration (utils_trainer.py:359-361, main.py): the captured torch.bernoulli of
    sample_drop_path must draw from the CURRENT seed at every replay, not from the capture-time one.  drop_path_rate 0.5
    on the tiny net (so that masks differ between steps with near certainty), re-seeded before every step: graph and eager
    trajectories are bit-identical, and consecutive steps do not repeat a mask."""
    from srhip.train import TrainStep, Optimizer
    cfg = O.swinir_config(upscale=8, in_chans=1, img_size=16, window_size=8, depths=(2, 2), embed_dim=60,
                          num_heads=(6, 6), mlp_ratio=2, drop_path_rate=0.5)
    sd0 = O.trained_like_(O.swinir_init_state_dict(cfg, seed=91), 92, lin_scale=3.0)
    gen = torch.Generator().manual_seed(93)
    batch = (torch.rand(4, 1, 16, 16, generator=gen).cuda(), torch.rand(4, 1, 128, 128, generator=gen).cuda())
    runs = []
    for mode in ("eager", "graph"):
        net = tiny(SwinIR, dpr=0.5)
        net.load_state_dict(sd0, strict=True)
        net = net.cuda().train()
        assert max(b.drop_prob for b in net.swin_blocks()) == 0.5
        ts = TrainStep(net, [("l1", 1.0)])
        ts.opt = Optimizer(ts.fp, "sgd", lr=0.01, momentum=0.9, nesterov=True, wd=0.0)
        losses = []
        for it in range(8):
            torch.manual_seed(1000 + it)                     # the trainer's per-iteration seed
            (ts.step if mode == "eager" else ts.step_graph)(*batch)
            losses.append(ts.loss_buf.clone())
        torch.cuda.synchronize()
        runs.append((torch.stack(losses).cpu(), ts.fp.flat.clone().cpu()))
    assert torch.equal(runs[0][0], runs[1][0]), (runs[0][0][:, 0] - runs[1][0][:, 0]).abs().max()
    assert torch.equal(runs[0][1], runs[1][1])
    # the same batch every step: a repeated mask would repeat the loss up to the (small) weight change; masks that differ
    # move it by far more -- at least one consecutive pair must differ visibly in the replayed run
    l = runs[1][0][:, 1]                                  # (loss_buf = [total (host-side), term 1, ...])
    assert (l[1:] - l[:-1]).abs().max() > 1e-3 * l.abs().max(), l


DDP_WORKER = r'''
import os, sys, socket, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, os.path.join(root, "sr-caco-2_amd")); sys.path.insert(0, os.path.join(root, "oracle"))
import sr_oracle as O
from dlib.models.network_swinir import SwinIR
from dlib.models.network_vdsr import VDSR
from srhip.train import TrainStep, Optimizer
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
cfg = O.swinir_config(upscale=8, in_chans=1, img_size=16, window_size=8, depths=(2, 2), embed_dim=60,
                      num_heads=(6, 6), mlp_ratio=2, drop_path_rate=0.0)
sd0 = O.trained_like_(O.swinir_init_state_dict(cfg, seed=91), 92, lin_scale=3.0)
gen = torch.Generator().manual_seed(93)
batches = [(torch.rand(2, 1, 16, 16, generator=gen).cuda(), torch.rand(2, 1, 128, 128, generator=gen).cuda()) for _ in range(3)]
out = {}
for mode in ("plain", "ddp"):
    os.environ["SRHIP_FORCE_DDP"] = "1" if mode == "ddp" else "0"
    net = SwinIR(upscale=8, in_chans=1, img_size=16, window_size=8, depths=[2, 2], embed_dim=60, num_heads=[6, 6],
                 mlp_ratio=2, upsampler="pixelshuffledirect", drop_path_rate=0.0)
    net.load_state_dict(sd0, strict=True)
    net = net.cuda().train()
    ts = TrainStep(net, [("l1", 1.0)], process_group=dist.group.WORLD if mode == "ddp" else None, world_size=1)
    ts.opt = Optimizer(ts.fp, "adam", lr=2e-4, wd=1e-4)
    assert ts.ddp == (mode == "ddp")
    for lr_img, hr_img in batches:
        ts.step(lr_img, hr_img)
    if mode == "ddp":      # every bucket went through RCCL exactly once per step, in backward order
        assert ts.reducer.log == [0, 1, 2] and len(ts.buckets) == 3, ts.reducer.log
        cover = sorted(ts.buckets)
        assert cover[0][0] == 0 and cover[-1][1] == ts.fp.total and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    torch.cuda.synchronize()
    out[mode] = (ts.fp.flat.clone(), ts.loss_buf.clone())
# the captured step under data parallelism: the bucket all-reduces ride in the hipGraph; same bits as the eager steps
os.environ["SRHIP_FORCE_DDP"] = "1"
net = SwinIR(upscale=8, in_chans=1, img_size=16, window_size=8, depths=[2, 2], embed_dim=60, num_heads=[6, 6],
             mlp_ratio=2, upsampler="pixelshuffledirect", drop_path_rate=0.0)
net.load_state_dict(sd0, strict=True)
net = net.cuda().train()
tg = TrainStep(net, [("l1", 1.0)], process_group=dist.group.WORLD, world_size=1)
tg.opt = Optimizer(tg.fp, "adam", lr=2e-4, wd=1e-4)
for lr_img, hr_img in batches:          # eager, capture + replay, replay
    tg.step_graph(lr_img, hr_img)
torch.cuda.synchronize()
assert tg._graph["g"] is not None
assert torch.equal(tg.fp.flat, out["ddp"][0]) and torch.equal(tg.loss_buf, out["ddp"][1]), \
    (tg.fp.flat - out["ddp"][0]).abs().max().item()
print("ddp graph ok")
d = (out["plain"][0] - out["ddp"][0]).abs().max().item()
# every reduction is deterministic (no float atomics left in the step).  Round 5: under data parallelism layer 0's Linear
# weight gradients leave in two or three grouped launches instead of one (their buckets are announced early), i.e. with
# another slice count: the same sums in another order -- the replicas of a run are still bit-identical to each other
# (and the captured step to the eager one, above); against the single-process step: rounding only
assert d <= 1e-6 * out["plain"][0].abs().max().item() and torch.allclose(out["plain"][1], out["ddp"][1], rtol=1e-6), d
# a single-bucket engine (VDSR): the bucket the engine used to announce itself must be reduced once
v = VDSR(in_chans=1, upscale=2)
v.load_state_dict(O.vdsr_init_state_dict(1, seed=2), strict=True)
v = v.cuda().train()
tv = TrainStep(v, [("l1", 1.0)], process_group=dist.group.WORLD, world_size=1)
tv.step(torch.rand(1, 1, 16, 16).cuda(), torch.rand(1, 1, 32, 32).cuda())
assert tv.reducer.log == [0], tv.reducer.log
dist.destroy_process_group()
print("ddp ok", d)
'''


def test_forced_ddp_single_rank_rccl_path_matches_plain_step(tmp_path):
    """a20 on the GPU: SRHIP_FORCE_DDP=1 takes the bucketed RCCL path (side stream, events, per-bucket
    all-reduce, flag MAX) with a one-rank nccl group; three Adam steps equal the plain step BIT FOR BIT, every bucket is reduced once per step in backward order.  (More
    than one rank needs more than one GPU: the driver's scaling run.)"""
    import socket
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "ddp_worker.py"
    script.write_text(DDP_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, str(script), root, str(port)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0 and "ddp ok" in p.stdout and "ddp graph ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


DDP_README_WORKER = r'''
import json, os, sys, torch, torch.distributed as dist
root = sys.argv[1]
for p in (os.path.join(root, "sr-caco-2_amd"), os.path.join(root, "oracle"), root):
    sys.path.insert(0, p)
os.environ["SRHIP_FORCE_DDP"] = "1"
from dlib.models.network_swinir import SwinIR
from srhip.train import TrainStep, Optimizer
import bench
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
torch.manual_seed(0)
net = SwinIR(upscale=8, in_chans=1, img_size=64, window_size=8, depths=[6, 6, 6, 6], embed_dim=180,
             num_heads=[6, 6, 6, 6], mlp_ratio=2, upsampler="pixelshuffledirect").cuda().train()
ts = TrainStep(net, [("l1", 1.0)], process_group=dist.group.WORLD, world_size=1)
ts.opt = Optimizer(ts.fp, "sgd", lr=0.01, momentum=0.9, nesterov=True, wd=0.0)
assert ts.ddp and len(ts.buckets) == 6      # layers 3, 2, 1; layer 0 in three (blocks 3-5 + conv, 1-2, 0 + head)
lr_img, hr_img = bench.synth_batch(8, 8, "cuda", seed=1000)
for _ in range(3):
    ts.step(lr_img, hr_img)
torch.cuda.synchronize()
ts.reducer.trace = True
# compute-stream marks: step start, backward end (recorded by the hook of the LAST bucket's announce = finish())
t_start, t_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t_start.record()
ts.step(lr_img, hr_img)
t_end.record()
torch.cuda.synchronize()
ev = ts.reducer.events
assert sorted(ev) == list(range(6)) and ts.reducer.log == list(range(6)), ts.reducer.log
rows = []
for i in range(6):
    lo, hi = ts.buckets[i]
    rows.append({"bucket": i, "mbytes": (hi - lo) * 4 / 1e6,
                 "announced_ms": t_start.elapsed_time(ev[i][0]), "allreduce_start_ms": t_start.elapsed_time(ev[i][1]),
                 "allreduce_end_ms": t_start.elapsed_time(ev[i][2])})
step_ms = t_start.elapsed_time(t_end)
out = {"what": "README SwinIR x8, B = 8, one rank (SRHIP_FORCE_DDP=1, RCCL group of one): per bucket, when the engine "
               "announced it on the compute stream and when its all-reduce started / ended on the side stream, ms after the "
               "step's first kernel; the step's last kernel (optimizer) ended at step_ms",
       "step_ms": step_ms, "total_mbytes": sum(r["mbytes"] for r in rows), "buckets": rows}
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(root, "gpurun_out", "r05_ddp_overlap.json"), "w"), indent=1)
# the schedule: buckets in backward-completion order, each all-reduce enqueued (and, with one rank, finished) while
# backward kernels of the layers behind it are still to come -- i.e. before the NEXT bucket is even announced, and all but
# the last one long before the step's end
for a, b in zip(rows, rows[1:]):
    assert a["announced_ms"] < b["announced_ms"], (a, b)
    assert a["allreduce_start_ms"] <= b["announced_ms"] + 0.05, (a, b)       # started before the next layer's backward ended
    assert a["allreduce_end_ms"] <= b["announced_ms"] + 0.5, (a, b)
assert rows[0]["announced_ms"] < 0.6 * step_ms and rows[4]["allreduce_end_ms"] < step_ms
assert abs(out["total_mbytes"] - 31.5) < 0.2, out["total_mbytes"]
# round 5: layer 0 in three buckets -- the one exchange a multi-GPU run cannot hide is the last bucket: block 0 + the head
assert rows[5]["mbytes"] <= 2.0, rows[5]
assert rows[3]["announced_ms"] <= step_ms - 0.8, (rows[3], step_ms)     # blocks 3-5 of layer 0: a millisecond before the end
# the ranges tile the flat gradient buffer exactly once
cover = sorted(ts.buckets)
assert cover[0][0] == 0 and all(a[1] == b[0] for a, b in zip(cover, cover[1:])) and cover[-1][1] == ts.fp.grad.numel()
dist.destroy_process_group()
print("ddp readme ok", json.dumps(out["buckets"]))
'''


def test_forced_ddp_at_readme_size_overlaps_buckets_with_backward(tmp_path):
    """Config 4's schedule at its real size on the one GPU there is: README SwinIR x8, B = 8, SRHIP_FORCE_DDP=1 (RCCL
    group of one): six buckets (31.5 MB; layer 0 in three, the last one 1.3 MB) in backward-completion order on the side
    stream, bucket i's all-reduce under way before the next bucket's backward has ended (HIP events on both streams;
    gpurun_out/r05_ddp_overlap.json)."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "ddp_readme_worker.py"
    script.write_text(DDP_README_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, str(script), root, str(port)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0 and "ddp readme ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_bench_self_launch_one_rank_rccl_path(tmp_path):
    """`python bench.py --gpus 1` with SRHIP_FORCE_DDP=1: the worker initialises RCCL, takes the bucketed path for every
    step of the README workload and prints the contract's JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(SRHIP_FORCE_DDP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                        "--train-only"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["n_ranks_seen"] == 1 and d["value"] > 100 and d["config"]["parallelism"] == "dp1"


CABI_WORKER = r'''
import os, sys, torch
root = sys.argv[1]
for p in (os.path.join(root, "sr-caco-2_amd"), os.path.join(root, "oracle"), root):
    sys.path.insert(0, p)
import sr_oracle as O
from dlib.models.network_swinir import SwinIR
from srhip.train import TrainStep, Optimizer, RcclComm
torch.cuda.set_device(0)
# the communicator by itself: one rank, a bucket summed in place (unchanged), the flag MAX-reduced, streams joined by events
comm = RcclComm(0, 1, RcclComm.unique_id())
side = torch.cuda.Stream()
g = torch.randn(1 << 20, device="cuda")
g0 = g.clone()
flag = torch.tensor([1], dtype=torch.int32, device="cuda")
comm.bucket(g[:1000], torch.cuda.current_stream(), side)
comm.bucket(g[1000:], torch.cuda.current_stream(), side)
comm.flag(flag, torch.cuda.current_stream(), side)
comm.wait(side, torch.cuda.current_stream())
g.mul_(2.0)                                  # on the compute stream, behind the exchange
torch.cuda.synchronize()
assert torch.equal(g, g0 * 2) and flag.item() == 1
comm.close()
# the training step on it (SRHIP_FORCE_DDP=1 SRHIP_COMM=cabi): bit for bit the step on torch.distributed's communicator
cfg = O.swinir_config(upscale=8, in_chans=1, img_size=16, window_size=8, depths=(2, 2), embed_dim=60, num_heads=(6, 6),
                      mlp_ratio=2, drop_path_rate=0.0)
sd0 = O.swinir_init_state_dict(cfg, seed=5)
gen = torch.Generator().manual_seed(6)
batches = [(torch.rand(2, 1, 16, 16, generator=gen).cuda(), torch.rand(2, 1, 128, 128, generator=gen).cuda()) for _ in range(3)]
out = {}
for mode in ("plain", "cabi"):
    os.environ["SRHIP_FORCE_DDP"] = "1" if mode == "cabi" else "0"
    os.environ["SRHIP_COMM"] = "cabi"
    net = SwinIR(upscale=8, in_chans=1, img_size=16, window_size=8, depths=[2, 2], embed_dim=60, num_heads=[6, 6],
                 mlp_ratio=2, upsampler="pixelshuffledirect", drop_path_rate=0.0)
    net.load_state_dict(sd0, strict=True)
    net = net.cuda().train()
    ts = TrainStep(net, [("l1", 1.0)], world_size=1)
    ts.opt = Optimizer(ts.fp, "adam", lr=2e-4, wd=1e-4)
    assert (ts.comm is not None) == (mode == "cabi")
    for lr_img, hr_img in batches:
        ts.step(lr_img, hr_img)
    torch.cuda.synchronize()
    if mode == "cabi":
        assert ts.reducer.log == [0, 1, 2], ts.reducer.log
    out[mode] = (ts.fp.flat.clone(), ts.loss_buf.clone())
d = (out["plain"][0] - out["cabi"][0]).abs().max().item()
assert d <= 1e-6 * out["plain"][0].abs().max().item() and torch.allclose(out["plain"][1], out["cabi"][1], rtol=1e-6), d
print("cabi ok", d)
'''


def test_cabi_rccl_communicator_single_rank(tmp_path):
    """srhip_allreduce_* (the collective behind the C-ABI, for a caller without PyTorch): RCCL
    resolved with dlopen, a one-rank communicator, buckets + flag on a side stream joined by events; then the training step
    on it (SRHIP_COMM=cabi) against the plain step."""
    import subprocess
    import sys
    script = tmp_path / "cabi_worker.py"
    script.write_text(CABI_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0 and "cabi ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


@pytest.mark.parametrize("comm", ["torch", "cabi"])
def test_bench_two_ranks_when_two_gpus_are_there(comm):
    """Config 4 beyond one rank: `python bench.py --gpus 2 --steps 3` launches two workers over RCCL /
    xGMI and rank 0's line says it saw two ranks.  Skipped on the one-GPU boxes of the test pool; there for the day a node
    appears.  comm = cabi: the exchange through srhip_allreduce_* (the id travels through torch.distributed's store)."""
    import json
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", SRHIP_COMM=comm)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--train-only"], capture_output=True, text=True, timeout=1200, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["global_batch"] == 16 and d["value"] > 100


def test_fused_step_with_img_range():
    """img_range != 1 (network_swinir.py:935,968: input x img_range, output / img_range) in the fused training step: the
    gradients of TrainStep equal those of the module path (net(x) + torch autograd around the same kernels)."""
    from dlib.models.network_swinir import SwinIR
    from srhip.train import TrainStep, Optimizer
    torch.manual_seed(3)
    kw = dict(upscale=2, in_chans=1, img_size=16, window_size=8, depths=[2], embed_dim=60, num_heads=[6], mlp_ratio=2,
              upsampler="pixelshuffledirect", img_range=4.0, drop_path_rate=0.0)
    net = SwinIR(**kw).cuda().train()
    x = torch.rand(2, 1, 16, 16, device="cuda")
    tgt = torch.rand(2, 1, 32, 32, device="cuda")
    y = net(x)
    (y - tgt).abs().mean().backward()
    ref = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    loss_ref = (y - tgt).abs().mean().item()
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "sgd", lr=0.0, momentum=0.0, nesterov=False, wd=0.0)
    ts.step(x, tgt)
    assert abs(ts.loss_values()[0] - loss_ref) <= 1e-6
    for k in ts.fp.names:
        a, b = ts.fp.gviews[k], ref[k]
        assert (a - b).abs().max().item() <= 1e-6 * max(1.0, b.abs().max().item()), k


def test_recompute_gelu_in_the_weight_gradient_matches_the_stored_form(SwinIR):
    """SRHIP_RECOMPUTE_GH=1 (round 6; opt-in because it measured slower): the fused MLP backward does not
    store gelu(h), the grouped weight-gradient launch recomputes it from the saved h in its operand prologue (b_mode 2, the
    forward's own packed x Phi(x), instantiation tnb_body_h<3, 2, true, true> with DropPath row scales / the generic one
    without).  Every parameter gradient of the tiny net and of a README-width net equals the stored form's to rounding."""
    from srhip.train import TrainStep, Optimizer
    for kw, dpr in ((dict(depths=[2, 2], embed_dim=60, num_heads=[6, 6]), 0.0), (dict(depths=[2], embed_dim=180, num_heads=[6]), 0.3)):
        torch.manual_seed(5)
        grads = {}
        for mode in ("0", "1"):
            os.environ["SRHIP_RECOMPUTE_GH"] = mode
            try:
                torch.manual_seed(11)
                net = SwinIR(upscale=8, in_chans=1, img_size=16, window_size=8, mlp_ratio=2, upsampler="pixelshuffledirect",
                             drop_path_rate=dpr, **kw).cuda().train()
                ts = TrainStep(net, [("l1", 1.0)])
                ts.opt = Optimizer(ts.fp, "sgd", lr=0.0, momentum=0.0, nesterov=False, wd=0.0)
                gen = torch.Generator().manual_seed(3)
                lr_img, hr_img = torch.rand(2, 1, 16, 16, generator=gen).cuda(), torch.rand(2, 1, 128, 128, generator=gen).cuda()
                torch.manual_seed(77)                      # the same DropPath masks in both runs
                ts.step(lr_img, hr_img)
                torch.cuda.synchronize()
                grads[mode] = {k: ts.fp.gviews[k].clone() for k in ts.fp.names}
            finally:
                os.environ.pop("SRHIP_RECOMPUTE_GH", None)
        for k in grads["0"]:
            a, b = grads["1"][k], grads["0"][k]
            assert (a - b).abs().max().item() <= 2e-6 * max(b.abs().max().item(), 1e-12), k
        assert any("fc2.weight" in k for k in grads["0"])


def test_side_stream_weight_gradients_match_the_in_order_form(SwinIR):
    """SRHIP_SWIN_SIDE_WGRAD=1 (round 6 experiment, opt-in): each RSTB layer's grouped weight-gradient launch, its reducers
    and the bias-table reductions run on a side stream beside the next layer's data-gradient chain, the per-layer operand
    buffers double by layer parity.  Four layers (so a buffer set is reused two layers on), eager and under graph replay:
    every parameter gradient is bit-identical to the in-order form's (same kernels, same fixed-order reductions)."""
    from srhip.train import TrainStep, Optimizer
    grads = {}
    for mode in ("0", "1", "1g"):
        os.environ["SRHIP_SWIN_SIDE_WGRAD"] = mode[0]
        try:
            torch.manual_seed(11)
            net = SwinIR(upscale=8, in_chans=1, img_size=16, window_size=8, mlp_ratio=2, upsampler="pixelshuffledirect",
                         drop_path_rate=0.0, depths=[2, 2, 2, 2], embed_dim=60, num_heads=[6, 6, 6, 6]).cuda().train()
            ts = TrainStep(net, [("l1", 1.0)])
            ts.opt = Optimizer(ts.fp, "sgd", lr=0.0, momentum=0.0, nesterov=False, wd=0.0)
            gen = torch.Generator().manual_seed(3)
            lr_img, hr_img = torch.rand(2, 1, 16, 16, generator=gen).cuda(), torch.rand(2, 1, 128, 128, generator=gen).cuda()
            if mode == "1g":
                for _ in range(3):
                    ts.step_graph(lr_img, hr_img)
            else:
                ts.step(lr_img, hr_img)
            torch.cuda.synchronize()
            grads[mode] = {k: ts.fp.gviews[k].clone() for k in ts.fp.names}
        finally:
            os.environ.pop("SRHIP_SWIN_SIDE_WGRAD", None)
    for mode in ("1", "1g"):
        for k in grads["0"]:
            assert torch.equal(grads[mode][k], grads["0"][k]), (mode, k)
