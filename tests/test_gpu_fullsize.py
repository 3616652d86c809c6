"""GPU parity at BASELINE.json's full workload sizes: one whole optimisation step
(forward + MasterLoss + backward + optimizer) of the fused TrainStep against the
oracle's autograd step on the same seeded inputs -- EDSR-baseline x2 256->512,
x4 128->512, x8 64->512 (16 blocks x 64 features, B=1; configs 1-2) and the SwinIR
README configuration at B=8 with a forced DropPath matrix (config 3, the bench
workload) -- and the trained-like-regime goldens (g18: saturating softmax, live
-100 shift mask, non-zero biases).

Gates: outputs pixel MAE <= 1e-5 and PSNR within 0.01 dB (north_star); parameter
gradients relative to the tensor's largest entry <= GRAD_GATE (about 10x the margins
measured on the MI355X, printed by every test with -s); parameters after the
update <= 2e-6 absolute."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import sr_oracle as O  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GRAD_GATE = 2e-5          # relative to max|grad| of the tensor; measured margins: see the prints


def load(name):
    z = np.load(os.path.join(G, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiu" else z[k]) for k in z.files}


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.set_num_threads(min(32, os.cpu_count() or 1))


def rel_err(a, ref):
    return (a - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)


def worst_l2(named_grads, ref_grads):
    """largest relative L2 error over the tensors: ||g - ref|| / ||ref||."""
    worst = ("", 0.0)
    for k, g in named_grads.items():
        r = ref_grads[k].double()
        e = ((g.detach().cpu().double() - r).norm() / (r.norm() + 1e-300)).item()
        if e > worst[1]:
            worst = (k, e)
    return worst


def worst_grad(named_grads, ref_grads):
    worst = ("", 0.0)
    for k, g in named_grads.items():
        e = rel_err(g.detach().cpu(), ref_grads[k])
        if e > worst[1]:
            worst = (k, e)
    return worst


def psnr_gap(y, yo, tgt, border):
    ps = lambda a: O.metric_psnr(O.tensor2uint82float(a), O.tensor2uint82float(tgt), border)
    return (ps(y) - ps(yo)).abs().max().item()


def synth(batch, scale, seed):
    """SURVEY 8d synthetic pair: H on the uint8 grid, L = clamp(bicubic_down(H))."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    hr = (torch.rand(batch, 1, 512, 512, generator=g) * 255).round() / 255
    lr = F.interpolate(hr, scale_factor=1.0 / scale, mode="bicubic").clamp(0, 1)
    return lr, hr


@pytest.mark.parametrize("scale,loss", [(2, "l1"), (4, "l1"), (8, "l2ssim")])
def test_edsr_full_size_train_step_vs_oracle(scale, loss):
    """Configs 1/2 of BASELINE.json on the HIP path: EDSR-baseline (network_nlsn.py:38-128 blocks wired as
    :355-369), 16 ResBlocks x 64 features, LR (512/s)^2 -> HR 512^2, B=1: forward, loss, every parameter
    gradient and the Adam update against the oracle.  x8 runs the README loss L2 + 5 SSIM(19)."""
    from dlib.models.network_edsr_liif import EDSR_LIIF
    from srhip.train import TrainStep, Optimizer
    cfg = O.edsr_config(upscale=scale)
    sd0 = O.edsr_init_state_dict(cfg, seed=50 + scale)
    net = EDSR_LIIF(scale=scale)
    net.load_state_dict(sd0, strict=True)
    net = net.cuda().train()
    terms = [("l1", 1.0)] if loss == "l1" else [("l2", 1.0), ("ssim", 5.0, 19)]
    ts = TrainStep(net, terms)
    ts.opt = Optimizer(ts.fp, "adam", lr=2e-4, wd=1e-4)
    lr_img, hr_img = synth(1, scale, seed=60 + scale)
    lb = ts.step(lr_img.cuda(), hr_img.cuda())
    y = net.engine.bufs.d["t.y"].detach().reshape(1, 1, 512, 512).cpu()
    grads = {k: v.clone() for k, v in ts.fp.gviews.items()}

    def oracle(dtype, masks=None):
        sd = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd0.items()}
        yo = O.edsr_forward(sd, lr_img.to(dtype), cfg, relu_masks=masks)
        tot, _ = O.master_loss(yo, hr_img.to(dtype), terms)
        tot.backward()
        return sd, yo.detach(), tot

    sdo, yo, tot = oracle(torch.float32)                   # the reference's own arithmetic
    sd64, _, _ = oracle(torch.float64)                     # the exact answer both fp32 computations approximate
    mae = (y - yo).abs().mean().item()
    gap = psnr_gap(y, yo, hr_img, scale)
    k, e = worst_grad(grads, {k: v.grad for k, v in sdo.items()})
    k64, e64 = worst_grad(grads, {k: v.grad.float() for k, v in sd64.items()})
    ko, eo = worst_grad({k: v.grad for k, v in sdo.items()}, {k: v.grad.float() for k, v in sd64.items()})
    lv = ts.loss_values()
    print(f"\nEDSR x{scale} {loss}: MAE {mae:.2e}, PSNR gap {gap:.2e} dB, loss {lv[0]:.6f} vs {tot.item():.6f}; "
          f"worst grad vs the fp32 oracle {k} {e:.2e}; vs the fp64 oracle {k64} {e64:.2e}; the fp32 oracle itself "
          f"vs fp64: {ko} {eo:.2e}")
    assert y.shape == (1, 1, 512, 512)
    assert mae <= 1e-5 and gap <= 0.01
    assert abs(lv[0] - tot.item()) <= 1e-5 * max(1.0, abs(tot.item()))
    # Gradient gates for a ReLU net.  A ReLU whose pre-activation lies within fp32 rounding of zero switches on in
    # one computation and off in the other -- ONE pixel of the 4096 .. 65536 of the body's feature maps then enters
    # or leaves a weight-gradient sum (measured: 1.3e-4 at 64x64; the reference's own fp32 result shows the same
    # against fp64 whenever it has such a pixel: 4.1e-5 at x4).  Against the free-running fp64 oracle only the
    # tensor-wise relative L2 error is gated (1e-4; such pixels show as a few 1e-5); every ENTRY is gated below
    # against an fp64 oracle run under the HIP run's own ReLU decisions.
    kl, el = worst_l2(grads, {k: v.grad for k, v in sd64.items()})
    # evidence for the mechanism: ReLU masks of the HIP run against an fp64 recomputation of the same
    # pre-activations from the HIP run's own block inputs
    import torch.nn.functional as F
    flips = 0
    for kb in (5, 11, 12):
        r_in, a = net.engine.saved["blocks"][kb]
        pre = F.conv2d(r_in.permute(0, 3, 1, 2).double().cpu(), sd0[f"body.{kb}.body.0.weight"].double(),
                       sd0[f"body.{kb}.body.0.bias"].double(), padding=1)
        flips += int(((pre > 0) != (a.permute(0, 3, 1, 2).cpu() > 0)).sum())
    print(f"  worst tensor-wise relative L2 error vs fp64: {kl} {el:.2e}; ReLU decisions that differ from an fp64 "
          f"recomputation in blocks 5, 11, 12: {flips} of {3 * 64 * (512 // scale) ** 2}")
    assert el <= 1e-4, (kl, el)
    # ENTRY-wise: against an fp64 oracle that takes the HIP run's own ReLU decisions (the saved activations' signs) --
    # the same piecewise-linear function on both sides, so no entry is excused: GRAD_GATE (2e-5) on every one
    masks = [(net.engine.saved["blocks"][kb][1].permute(0, 3, 1, 2).cpu() > 0) for kb in range(cfg["n_resblocks"])]
    sdm, ym, _ = oracle(torch.float64, masks)
    km, em = worst_grad(grads, {k: v.grad.float() for k, v in sdm.items()})
    print(f"  worst grad entry vs the fp64 oracle under the HIP run's ReLU decisions: {km} {em:.2e}; its output differs from "
          f"the free-running fp64 oracle's by {(ym.float() - y).abs().max().item():.1e} (HIP) ")
    assert em <= GRAD_GATE, (km, em)
    # Adam's first update is lr * g / (|g| + eps): where a gradient entry is ~0 its SIGN decides a full
    # +-lr step, so the update is checked from the HIP gradients themselves (the gradients are gated above)
    worst = 0.0
    with torch.no_grad():
        for k, p in net.named_parameters():
            po = sd0[k].clone()
            O.adam_step(po, grads[k].cpu(), torch.zeros_like(po), torch.zeros_like(po), 1, 2e-4, wd=1e-4)
            worst = max(worst, (p.detach().cpu() - po).abs().max().item())
    print(f"  worst parameter after the Adam step (oracle Adam on the same gradients): {worst:.2e}")
    assert worst <= 2e-6


def readme_net(dpr):
    from dlib.models.network_swinir import SwinIR
    return SwinIR(upscale=8, in_chans=1, img_size=64, window_size=8, depths=[6, 6, 6, 6], embed_dim=180,
                  num_heads=[6, 6, 6, 6], mlp_ratio=2, upsampler="pixelshuffledirect", drop_path_rate=dpr)


def forced_dp(cfg, batch, seed):
    """per block a (2, B) matrix of DropPath multipliers mask / keep (timm semantics)."""
    rates = O.swinir_drop_path_rates(cfg)
    g = torch.Generator().manual_seed(seed)
    out = []
    for r in rates:
        keep = 1.0 - r
        out.append(torch.bernoulli(torch.full((2, batch), keep), generator=g) / keep)
    return out


@pytest.mark.parametrize("batch,loss,regime", [(8, "l1", "trained"), (2, "l2ssim", "fresh")])
def test_swinir_readme_train_step_forced_droppath_vs_oracle(batch, loss, regime):
    """Config 3 (the bench workload): SwinIR README configuration, LR 64^2 -> HR 512^2, a forced DropPath
    matrix at the reference's rate 0.1 (network_swinir.py:821,334-335), B=8 with L1 and weights in a
    trained-like regime, B=2 with the README loss L2 + 5 SSIM(19) (README.md:152-159) on fresh weights;
    SGD-Nesterov.  Forward, loss values, all 330 parameter gradients, parameters after the update."""
    from srhip.train import TrainStep, Optimizer
    cfg = O.swinir_config(drop_path_rate=0.1)
    sd0 = O.swinir_init_state_dict(cfg, seed=70)
    if regime == "trained":
        O.trained_like_(sd0, 71, lin_scale=5.0, qk_scale=2.0)
    net = readme_net(0.1)
    net.load_state_dict(sd0, strict=True)
    net = net.cuda().train()
    terms = [("l1", 1.0)] if loss == "l1" else [("l2", 1.0), ("ssim", 5.0, 19)]
    ts = TrainStep(net, terms)
    ts.opt = Optimizer(ts.fp, "sgd", lr=0.01, momentum=0.9, nesterov=True, wd=0.0)
    lr_img, hr_img = synth(batch, 8, seed=72)
    dps = forced_dp(cfg, batch, seed=73)
    assert sum(float((m == 0).sum()) for m in dps) > 0, "no path dropped: pick another seed"
    dp_dev = torch.stack(dps).reshape(-1, batch).cuda().contiguous()       # [2*blocks, B]
    ts.step(lr_img.cuda(), hr_img.cuda(), dp=dp_dev)
    y = net.engine.bufs.d["t.y"].detach().cpu()
    grads = {k: v.clone() for k, v in ts.fp.gviews.items()}

    def oracle(dtype):
        sd = {k: ((v.to(dtype).clone().requires_grad_(True) if not k.endswith("attn_mask") else v.to(dtype))
                  if v.dtype == torch.float32 else v) for k, v in sd0.items()}
        yo = O.swinir_forward(sd, lr_img.to(dtype), cfg, dp_scales=[m.to(dtype) for m in dps])
        tot, _ = O.master_loss(yo, hr_img.to(dtype), terms)
        tot.backward()
        return sd, yo.detach(), tot

    sdo, yo, tot = oracle(torch.float32)                   # the reference's own arithmetic
    names = [k for k, v in sdo.items() if torch.is_tensor(v) and v.requires_grad]
    mae = (y - yo).abs().mean().item()
    gap = psnr_gap(y, yo, hr_img, 8)
    k, e = worst_grad(grads, {k: sdo[k].grad for k in names})
    # the same step in float64 = the exact answer both fp32 computations approximate
    sd64, yo64, _ = oracle(torch.float64)
    k64, e64 = worst_grad(grads, {k: sd64[k].grad.float() for k in names})
    ko, eo = worst_grad({k: sdo[k].grad for k in names}, {k: sd64[k].grad.float() for k in names})
    lv = ts.loss_values()
    print(f"\nSwinIR README B={batch} {loss} {regime}: MAE {mae:.2e}, PSNR gap {gap:.2e} dB, "
          f"loss {lv[0]:.6f} vs {tot.item():.6f}; worst grad vs the fp32 oracle {k} {e:.2e}; vs the fp64 oracle "
          f"{k64} {e64:.2e}; the fp32 oracle itself vs fp64: {ko} {eo:.2e}")
    assert mae <= 1e-5 and gap <= 0.01
    assert abs(lv[0] - tot.item()) <= 1e-5 * max(1.0, abs(tot.item()))
    # gate against the exact (fp64) gradients: within GRAD_GATE, or -- where cancellation makes fp32 itself
    # noisier than that (sums over 32768 tokens / 512 windows with saturated softmax) -- no further from the
    # exact value than 3x the reference's own fp32 result is
    assert e64 <= max(GRAD_GATE, 3.0 * eo), (k64, e64, eo)
    worst = 0.0
    with torch.no_grad():
        for k, p in net.named_parameters():
            po = sd0[k].clone()
            O.sgd_nesterov_step(po, grads[k].cpu(), torch.zeros_like(po), True, 0.01)
            worst = max(worst, (p.detach().cpu() - po).abs().max().item())
    print(f"  worst parameter after the SGD-Nesterov step (oracle update on the same gradients): {worst:.2e}")
    assert worst <= 2e-6


def test_trained_like_goldens_swinir_and_edsr():
    """g18 (generated from the REAL reference): weights scaled into a trained-like regime -- Linear x10,
    q/k rows x2.5 more, bias tables ~ N(0,1), non-zero biases / LayerNorm affine, EDSR convs x2 -- so
    the softmax saturates (mean max-prob 0.70) and the -100 mask decides probabilities
    (network_swinir.py:140-179, network_nlsn.py:72-128)."""
    from dlib.models.network_swinir import SwinIR
    from dlib.models.network_edsr_liif import EDSR_LIIF
    g = load("g18_trained_like")
    net = SwinIR(upscale=8, in_chans=1, img_size=16, window_size=8, depths=[2, 2], embed_dim=60,
                 num_heads=[6, 6], mlp_ratio=2, upsampler="pixelshuffledirect", drop_path_rate=0.0)
    net.load_state_dict(sub(g, "sd/"), strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(g["x"].cuda()).cpu()
    e_eval = (y - g["y_eval"]).abs().max().item()
    net.train()
    x = g["x"].cuda().requires_grad_(True)
    yt = net(x)
    (yt - g["target"].cuda()).abs().mean().backward()
    e_dx = rel_err(x.grad.cpu(), g["dx"])
    k, e = worst_grad({k: p.grad for k, p in net.named_parameters()}, sub(g, "grad/"))
    print(f"\ntrained-like SwinIR tiny: eval max err {e_eval:.2e} (|y| max {g['y_eval'].abs().max():.2f}), "
          f"dx {e_dx:.2e}, worst grad {k} {e:.2e}")
    assert e_eval <= 1e-5 * max(1.0, float(g["y_eval"].abs().max()))
    assert (yt.detach().cpu() - g["y_train"]).abs().max() <= 1e-5 * max(1.0, float(g["y_train"].abs().max()))
    assert e_dx <= GRAD_GATE and e <= 2 * GRAD_GATE, (k, e, e_dx)   # bias tables: sums with cancellation

    s, nb, nf = [int(v) for v in g["ecfg"]]
    enet = EDSR_LIIF(scale=s, n_resblocks=nb, n_feats=nf)
    enet.load_state_dict(sub(g, "esd/"), strict=True)
    enet = enet.cuda().train()
    ey = enet(g["ex"].cuda())
    (ey - g["etarget"].cuda()).abs().mean().backward()
    ee = rel_err(ey.detach().cpu(), g["ey"])
    k, e = worst_grad({k: p.grad for k, p in enet.named_parameters()}, sub(g, "egrad/"))
    print(f"trained-like EDSR: forward rel err {ee:.2e} (|y| max {g['ey'].abs().max():.1f}), worst grad {k} {e:.2e}")
    assert ee <= 1e-5 and e <= GRAD_GATE, (k, e, ee)


def test_module_path_with_stock_torch_optimizer_vs_oracle():
    """The drop-in nn.Module path trained by a stock torch.optim.SGD (no TrainStep, nobody calls
    weights_changed()): the derived operands -- LayerNorm-folded weights, bf16x3 planes, conv packs, bias
    images -- must follow the parameters; three steps against the oracle's autograd + the same optimizer."""
    from dlib.models.network_swinir import SwinIR
    from dlib.models.network_edsr_liif import EDSR_LIIF
    cfg = O.swinir_config(upscale=8, in_chans=1, img_size=16, window_size=8, depths=(2, 2), embed_dim=60,
                          num_heads=(6, 6), mlp_ratio=2, drop_path_rate=0.0)
    sd0 = O.trained_like_(O.swinir_init_state_dict(cfg, seed=31), 32, lin_scale=5.0)
    net = SwinIR(upscale=8, in_chans=1, img_size=16, window_size=8, depths=[2, 2], embed_dim=60,
                 num_heads=[6, 6], mlp_ratio=2, upsampler="pixelshuffledirect", drop_path_rate=0.0)
    net.load_state_dict(sd0, strict=True)
    net = net.cuda().train()
    ecfg = O.edsr_config(upscale=2, n_feats=16, n_resblocks=2)
    esd0 = O.edsr_init_state_dict(ecfg, seed=33)
    enet = EDSR_LIIF(scale=2, n_resblocks=2, n_feats=16)
    enet.load_state_dict(esd0, strict=True)
    enet = enet.cuda().train()
    gen = torch.Generator().manual_seed(34)
    for (model, sd_init, fwd, shp_in, shp_out) in (
            (net, sd0, lambda sd, x: O.swinir_forward(sd, x, cfg), (2, 1, 16, 16), (2, 1, 128, 128)),
            (enet, esd0, lambda sd, x: O.edsr_forward(sd, x, ecfg), (2, 1, 16, 24), (2, 1, 32, 48))):
        sdo = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith("attn_mask")
                   else v) for k, v in sd_init.items()}
        names = [k for k, v in sdo.items() if v.requires_grad]
        opt_hip = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
        opt_ref = torch.optim.SGD([sdo[k] for k in names], lr=0.05, momentum=0.9)
        for step in range(3):
            x, tgt = torch.rand(shp_in, generator=gen), torch.rand(shp_out, generator=gen)
            opt_hip.zero_grad()
            lh = (model(x.cuda()) - tgt.cuda()).abs().mean()
            lh.backward()
            opt_hip.step()
            opt_ref.zero_grad()
            lo = (fwd(sdo, x) - tgt).abs().mean()
            lo.backward()
            opt_ref.step()
            assert abs(lh.item() - lo.item()) <= 2e-6 * max(1.0, abs(lo.item())), (step, lh.item(), lo.item())
        for k, p in model.named_parameters():
            assert (p.detach().cpu() - sdo[k].detach()).abs().max() <= 5e-6, k
