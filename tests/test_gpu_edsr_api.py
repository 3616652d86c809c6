"""GPU parity: EDSR-baseline (reference blocks) and the dlib.loss / dlib.metrics /
ModelPlain surfaces, against the reference goldens and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import sr_oracle as O  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(G, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiu" else z[k]) for k in z.files}


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("scale", [2, 4, 8])
def test_edsr_small_fwd_bwd_vs_reference_golden(scale):
    from dlib.models.network_edsr_liif import EDSR_LIIF
    g = load(f"g2_edsr_x{scale}")
    s, nb, nf = [int(v) for v in g["cfg"]]
    net = EDSR_LIIF(scale=s, n_resblocks=nb, n_feats=nf, res_scale=float(g["res_scale"]))
    net.load_state_dict(sub(g, "sd/"), strict=True)
    net = net.cuda().train()
    x = g["x"].cuda().requires_grad_(True)
    y = net(x)
    assert (y.detach().cpu() - g["y"]).abs().max() <= 1e-5
    y.abs().mean().backward()
    for k, p in net.named_parameters():
        ref = g["grad/" + k]
        e = (p.grad.cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        assert e <= 2e-5, f"grad {k}: rel err {e:.2e}"      # <= 10x the measured margin
    # d loss / d input against the oracle
    sd = {k: v.clone() for k, v in sub(g, "sd/").items()}
    xo = g["x"].clone().requires_grad_(True)
    cfg = O.edsr_config(upscale=s, n_feats=nf, n_resblocks=nb, res_scale=float(g["res_scale"]))
    O.edsr_forward(sd, xo, cfg).abs().mean().backward()
    assert (x.grad.cpu() - xo.grad).abs().max() <= 2e-5 * xo.grad.abs().max()


def test_edsr_full_size_forward_vs_reference_golden():
    from dlib.models.network_edsr_liif import EDSR_LIIF
    g = load("g2b_edsr_full")
    for scale in (4, 8):
        cfg = O.edsr_config(upscale=scale)
        net = EDSR_LIIF(scale=scale)
        net.load_state_dict(O.edsr_init_state_dict(cfg, seed=scale), strict=True)
        net = net.cuda().eval()
        with torch.no_grad():
            y = net(g[f"x{scale}/x"].cuda()).cpu()
        assert (y - g[f"x{scale}/y"]).abs().mean() <= 1e-5
    # config 2 shape: x4, LR 128 -> HR 512, oracle comparison incl. PSNR gate
    cfg = O.edsr_config(upscale=4)
    sd = O.edsr_init_state_dict(cfg, seed=4)
    net = EDSR_LIIF(scale=4)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    x = torch.rand(1, 1, 128, 128, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        y = net(x.cuda()).cpu()
        yo = O.edsr_forward(sd, x, cfg)
    assert y.shape == (1, 1, 512, 512)
    assert (y - yo).abs().mean() <= 1e-5
    tgt = torch.rand(1, 1, 512, 512, generator=torch.Generator().manual_seed(6))
    ps = lambda a: O.metric_psnr(O.tensor2uint82float(a), O.tensor2uint82float(tgt), 4)
    assert (ps(y) - ps(yo)).abs().max() <= 0.01


@pytest.mark.parametrize("scale", [2, 4, 8])
def test_vdsr_fwd_bwd_vs_reference_golden(scale):
    """SURVEY f1, first of the plain CNNs: VDSR (network_vdsr.py) on the libsrhip conv kernels -- forward and
    every weight gradient against the reference's outputs; registry and state_dict contract."""
    from dlib.models.select_network import define_G
    from dlib.utils import constants
    g = load("g16_vdsr")
    pre = f"x{scale}/"
    args = type("A", (), {})()
    args.netG = {'net_type': constants.VDSR, 'VDSR_upscale': scale, 'VDSR_in_chans': 1}
    net = define_G(args)
    sd = O.vdsr_init_state_dict(1, seed=int(g[pre + "seed"]))
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(k, tuple(v.shape)) for k, v in sd.items()]
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    y = net(g[pre + "x"].cuda())
    assert (y.detach().cpu() - g[pre + "y"]).abs().mean() <= 1e-5
    assert (y.detach().cpu() - g[pre + "y"]).abs().max() <= 5e-5
    (y - g[pre + "target"].cuda()).abs().mean().backward()
    for i, (k, p) in enumerate(net.named_parameters()):
        gs = g[pre + "grad_sums"][i]
        assert abs(p.grad.double().sum().item() - float(gs[0])) <= 2e-4 * max(1e-3, float(gs[1])), k
        if pre + "grad/" + k in g:
            ref = g[pre + "grad/" + k]
            assert (p.grad.cpu() - ref).abs().max() <= 2e-3 * float(ref.abs().max()), k
    with torch.no_grad():
        net.eval()
        y2 = net(g[pre + "x"].cuda())
    assert (y2.cpu() - g[pre + "y"]).abs().max() <= 5e-5
    with pytest.raises(RuntimeError, match="GPU only"):
        net(g[pre + "x"])


@pytest.mark.parametrize("scale", [2, 4, 8])
def test_drrn_fwd_bwd_vs_reference_golden(scale):
    """DRRN (network_drrn.py; shared-weight recursive block, 3 and 25 units) -- forward and the four
    weight gradients (the shared ones summed over the applications) against the reference."""
    from dlib.models.select_network import define_G
    from dlib.utils import constants
    g = load("g17_drrn")
    pre = f"x{scale}/"
    seed, units = [int(v) for v in g[pre + "cfg"]]
    args = type("A", (), {})()
    args.netG = {'net_type': constants.DRRN, 'DRRN_upscale': scale, 'DRRN_in_chans': 1, 'DRRN_num_residual_units': units}
    net = define_G(args)
    sd = O.drrn_init_state_dict(1, seed=seed)
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(k, tuple(v.shape)) for k, v in sd.items()]
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    y = net(g[pre + "x"].cuda())
    ref = g[pre + "y"]
    assert (y.detach().cpu() - ref).abs().mean() <= 1e-5 * max(1.0, float(ref.abs().mean()))
    (y - g[pre + "target"].cuda()).abs().mean().backward()
    for k, p in net.named_parameters():
        gs = g[pre + "gsum/" + k]
        assert abs(p.grad.double().sum().item() - float(gs[0])) <= 2e-4 * max(1e-3, float(gs[1])), k
        if pre + "grad/" + k in g:
            r = g[pre + "grad/" + k]
            assert (p.grad.cpu() - r).abs().max() <= 2e-3 * float(r.abs().max()), k
    with torch.no_grad():
        net.eval()
        y2 = net(g[pre + "x"].cuda())
    assert (y2.cpu() - ref).abs().mean() <= 1e-5 * max(1.0, float(ref.abs().mean()))


def test_srcnn_fwd_bwd_vs_reference_golden_and_full_size_step():
    """SRCNN (network_srcnn.py:23-69, registry select_network.py:207-210) as token-matrix GEMMs on the
    bf16x3 kernels (5x5 layer through srhip_im2col_c1): forward and all six parameter gradients against the
    reference (g22); one fused optimisation step at 512 x 512 against the oracle."""
    from dlib.models.select_network import define_G
    from dlib.utils import constants
    from srhip.train import TrainStep, Optimizer
    g = load("g22_srcnn")
    args = type("A", (), {})()
    args.netG = {'net_type': constants.SRCNN, 'SRCNN_in_chans': 1}
    net = define_G(args)
    sd = O.srcnn_init_state_dict(1, seed=61, bias_std=0.05)
    sd["reconstruction.weight"] = sd["reconstruction.weight"] * 50.0
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(k, tuple(v.shape)) for k, v in sd.items()]
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    y = net(g["x"].cuda())
    assert (y.detach().cpu() - g["y"]).abs().max() <= 1e-5
    (y - g["target"].cuda()).abs().mean().backward()
    for k, p in net.named_parameters():
        ref = g["grad/" + k]
        e = (p.grad.cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
        assert e <= 2e-5, f"grad {k}: rel err {e:.2e}"
    with torch.no_grad():
        net.eval()
        assert (net(g["x"].cuda()).cpu() - g["y"]).abs().max() <= 1e-5
    with pytest.raises(RuntimeError, match="GPU only"):
        net(g["x"])
    # full size: 1 x 512 x 512 (the net runs at the target resolution), L1 + SGD-Nesterov, fused step
    net2 = define_G(args)
    sd2 = O.srcnn_init_state_dict(1, seed=63)
    net2.load_state_dict(sd2, strict=True)
    net2 = net2.cuda().train()
    ts = TrainStep(net2, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "sgd", lr=0.01, momentum=0.9, nesterov=True, wd=0.0)
    gen = torch.Generator().manual_seed(64)
    x, tgt = torch.rand(1, 1, 512, 512, generator=gen), torch.rand(1, 1, 512, 512, generator=gen)
    ts.step(x.cuda(), tgt.cuda())
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd2.items()}
    lo = O.loss_l1(O.srcnn_forward(sdo, x), tgt)
    lo.backward()
    assert abs(ts.loss_values()[0] - lo.item()) <= 1e-5
    with torch.no_grad():
        for k, p in net2.named_parameters():
            O.sgd_nesterov_step(sdo[k], sdo[k].grad, torch.zeros_like(sdo[k]), True, 0.01)
            assert (p.detach().cpu() - sdo[k].detach()).abs().max() <= 2e-6, k


def test_srcnn_training_step_walks_the_batch_in_groups():
    """A training batch larger than one GEMM call addresses (rows x 1024 < 2^29: B = 8 of 512 x 512 is four groups) is walked in
    groups of images, forward and backward, the weight gradients added up: with the limit forced down (three groups of one
    40 x 40 image) the gradients are those of the ungrouped step."""
    from dlib.models.network_srcnn import SRCNN
    from srhip.train import TrainStep, Optimizer
    gen = torch.Generator().manual_seed(70)
    x, tgt = torch.rand(3, 1, 40, 40, generator=gen).cuda(), torch.rand(3, 1, 40, 40, generator=gen).cuda()
    sd = O.srcnn_init_state_dict(1, seed=71, bias_std=0.05)
    got = []
    for rows_max in (None, 1600):
        net = SRCNN(in_chans=1)
        net.load_state_dict(sd, strict=True)
        net = net.cuda().train()
        if rows_max is not None:
            net.engine.rows_max = rows_max
        ts = TrainStep(net, [("l1", 1.0)])
        ts.opt = Optimizer(ts.fp, "sgd", lr=0.0, momentum=0.0, nesterov=False, wd=0.0)
        ts.step(x, tgt)
        got.append((ts.loss_values()[0], {k: ts.fp.gviews[k].clone() for k in ts.fp.names}))
        assert (net.engine.saved["per"] >= 3) if rows_max is None else (net.engine.saved["per"] == 1)      # images per group
    assert abs(got[0][0] - got[1][0]) <= 1e-7
    for k in got[0][1]:
        a, b = got[0][1][k], got[1][1][k]
        assert ((a - b).abs().max() / a.abs().max().clamp_min(1e-30)).item() <= 2e-6, k


def test_vdsr_full_size_forward_and_train_step():
    """VDSR at the benchmark patch (1 x 64 x 64 -> 512 x 512): forward against the oracle (MAE <= 1e-5, PSNR
    within 0.01 dB) and one fused optimisation step against the oracle's autograd + SGD-Nesterov step."""
    from dlib.models.network_vdsr import VDSR
    from srhip.train import TrainStep, Optimizer
    sd = O.vdsr_init_state_dict(1, seed=7)
    net = VDSR(in_chans=1, upscale=8)
    net.load_state_dict(sd)
    net = net.cuda().train()
    gen = torch.Generator().manual_seed(4)
    x = torch.rand(1, 1, 64, 64, generator=gen)
    tgt = torch.rand(1, 1, 512, 512, generator=gen)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yo = O.vdsr_forward(sdo, x, 8)
    with torch.no_grad():
        y = net(x.cuda()).cpu()
    assert (y - yo.detach()).abs().mean() <= 1e-5
    ps = lambda a: O.metric_psnr(O.tensor2uint82float(a), O.tensor2uint82float(tgt), 8)
    assert (ps(y) - ps(yo.detach())).abs().max() <= 0.01
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "sgd", lr=0.01, momentum=0.9, nesterov=True, wd=0.0)
    ts.step(x.cuda(), tgt.cuda())
    lo = O.loss_l1(yo, tgt)
    lo.backward()
    assert abs(ts.loss_values()[0] - lo.item()) <= 1e-5
    with torch.no_grad():
        for k in sdo:
            O.sgd_nesterov_step(sdo[k], sdo[k].grad, torch.zeros_like(sdo[k]), True, 0.01)
    for k, p in net.named_parameters():
        e = (p.detach().cpu() - sdo[k].detach()).abs().max().item()
        assert e <= 2e-6, f"{k}: {e}"


def test_dlib_loss_surface_vs_reference_golden():
    from dlib import loss as L
    from dlib import losses as L2mod
    assert L2mod.MasterLoss is L.MasterLoss
    g = load("g6_losses")
    cases = {"l1": ([("l1", 1.0)], None), "l2_ssim19": ([("l2", 1.0), ("ssim", 5.0, 19)], None),
             "l1_weighted": ([("l1", 1.0)], g["weight"]), "ssim11": ([("ssim", 1.0, 11)], None)}
    for name, (terms, w) in cases.items():
        m = L.MasterLoss(cuda_id=0)
        for t in terms:
            if t[0] == "l1":
                m.add(L.L1(cuda_id=0, lambda_=t[1]))
            elif t[0] == "l2":
                m.add(L.L2(cuda_id=0, lambda_=t[1]))
            else:
                s = L.NegativeSsim(cuda_id=0, lambda_=t[1])
                s.set_window_size(t[2])
                m.add(s)
        p = g["pred"].cuda().requires_grad_(True)
        v = m(epoch=0, y_pred=p, y_target=g["target"].cuda(),
              trg_per_pixel_weight=None if w is None else w.cuda(), model=None)
        v.backward()
        holder = torch.stack([h.detach().cpu().reshape(()) for h in m.l_holder])
        assert (holder - g[name + "/l_holder"]).abs().max() <= 2e-4 * g[name + "/l_holder"].abs().max(), name
        assert (p.grad.cpu() - g[name + "/grad"]).abs().max() <= 3e-4 * g[name + "/grad"].abs().max(), name
        assert m.n_holder == list(g[name + "/names"])


def _extra_term(L, term):
    kind, kw = term[0], dict(cuda_id=0, lambda_=term[1])
    if kind == "charbonnier":
        l = L.Charbonnier(**kw)
        l.set_eps(term[2])
    elif kind == "l2sum":
        l = L.L2Sum(**kw)
    else:
        cls = {"grad": L.ImageGradientLoss, "laplace": L.LaplacianFilterLoss, "lv": L.LocalVariationLoss,
               "norm_grad": L.NormImageGradientLoss, "norm_laplace": L.NormLaplacianFilterLoss,
               "norm_lv": L.NormLocalVariationLoss}[kind]
        l = cls(**kw)
        if kind.endswith("lv"):
            l.set_it(ksz=term[3], norm_str=L.NORM1 if term[2] == 1 else L.NORM2)
        else:
            l.set_it(norm_str=L.NORM1 if term[2] == 1 else L.NORM2)
    return l


def test_dlib_optional_loss_terms_vs_reference_golden():
    """Charbonnier, L2Sum and the local-variation family through the dlib.loss surface (one fused HIP kernel
    each) against the reference's values and gradients (tests/golden/g9_losses_extra.npz)."""
    from dlib import loss as L
    from test_oracle_golden import LOSS_EXTRA_CASES
    g = load("g9_losses_extra")
    for sn in "abc":
        for name, term in LOSS_EXTRA_CASES.items():
            m = L.MasterLoss(cuda_id=0)
            m.add(_extra_term(L, term))
            p = g[f"{sn}/pred"].cuda().requires_grad_(True)
            v = m(epoch=0, y_pred=p, y_target=g[f"{sn}/target"].cuda(), trg_per_pixel_weight=None, model=None)
            v.backward()
            rv, rg = g[f"{sn}/{name}/value"], g[f"{sn}/{name}/grad"]
            assert abs(float(v) - float(rv)) <= 2e-6 * max(1.0, abs(float(rv))), (sn, name, float(v), float(rv))
            err = (p.grad.cpu() - rg).abs().max().item()
            assert err <= 2e-6 * max(1.0, float(rg.abs().max())), (sn, name, err)
            assert m.n_holder == list(g[f"{sn}/{name}/names"])
            assert m.terms()[0][0] == term[0]


def test_dlib_bounded_prediction_and_sparsity_vs_reference_golden():
    """BoundedPrediction (ELB) and WeightsSparsityLoss through dlib.loss against the reference goldens; the
    barrier schedule and the reference's update_t quirk (dlib.loss.elb vs dlib.losses.elb)."""
    from dlib import loss as L
    from dlib.losses.elb import ELB
    g = load("g10_losses_elb")
    for name in ("rr_t1", "rr_t3upd", "raw_t1", "raw_t40upd"):
        lam, eps, rr, upd = [float(v) for v in g[name + "/cfg"]]
        l = L.BoundedPrediction(cuda_id=0, lambda_=lam, elb=ELB(init_t=1., max_t=10., mulcoef=1.01),
                                restore_range=bool(rr), color_max=255)
        l.set_eps(eps)
        m = L.MasterLoss(cuda_id=0)
        m.add(l)
        m.update_t()                                   # no-op for this term, as in the reference
        assert float(l.elb.get_t()) == 1.0
        for _ in range(int(upd)):
            l.elb.update_t()
        assert float(l.elb.get_t()) == float(g[name + "/t"])
        p = g["pred"].cuda().requires_grad_(True)
        v = m(epoch=0, y_pred=p, y_target=g["target"].cuda(), trg_per_pixel_weight=None, model=None)
        v.backward()
        rv, rg = g[name + "/value"], g[name + "/grad"]
        assert abs(float(v) - float(rv)) <= 5e-6 * max(1.0, abs(float(rv))), (name, float(v), float(rv))
        assert (p.grad.cpu() - rg).abs().max() <= 5e-6 * max(1.0, float(rg.abs().max())), name
        assert m.n_holder == list(g[name + "/names"])
    net = torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3), torch.nn.Linear(5, 3))
    with torch.no_grad():
        for i, q in enumerate(net.parameters()):
            q.copy_(g[f"ws/p{i}"])
    net = net.cuda()
    ws = L.WeightsSparsityLoss(cuda_id=0, lambda_=0.3)
    v = ws(epoch=0, y_pred=None, y_target=None, model=net)
    v.backward()
    assert abs(float(v) - float(g["ws/value"])) <= 1e-6
    for i, q in enumerate(net.parameters()):
        assert torch.equal(q.grad.cpu(), g[f"ws/g{i}"])


def test_dlib_local_moments_vs_reference_golden():
    """LocalMoments (KL between 3x3 patch Gaussians where the target patch is exactly flat) against the
    reference, incl. flat regions touching the border / corner and a 5 x 7 image; then at 8 x 512 x 512
    against the oracle on a crop."""
    from dlib import loss as L
    from srhip import ops
    g = load("g13_local_moments")
    for name in "ab":
        m = L.MasterLoss(cuda_id=0)
        m.add(L.LocalMoments(cuda_id=0, lambda_=0.7))
        p = g[f"{name}/pred"].cuda().requires_grad_(True)
        v = m(epoch=0, y_pred=p, y_target=g[f"{name}/target"].cuda(), trg_per_pixel_weight=None, model=None)
        v.backward()
        rv, rg = g[f"{name}/value"], g[f"{name}/grad"]
        assert abs(float(v) - float(rv)) <= 2e-6 * max(1.0, abs(float(rv))), (name, float(v), float(rv))
        assert (p.grad.cpu() - rg).abs().max() <= 2e-6 * max(1.0, float(rg.abs().max())), name
        assert m.n_holder == list(g[f"{name}/names"]) and m.terms() == [("local_moments", 0.7)]
    gen = torch.Generator().manual_seed(13)
    p = torch.rand(8, 1, 512, 512, generator=gen)
    t = torch.round(torch.rand(8, 1, 512, 512, generator=gen) * 255) / 255
    t[:, :, 100:300, 50:400] = 12.0 / 255
    t[3, :, :, :64] = 0.0
    gr = torch.empty_like(p).cuda()
    v = ops.loss_local_moments(p.cuda(), t.cuda(), 1.0, gr)
    po = p[3:4].clone().requires_grad_(True)
    vo = O.loss_local_moments(po, t[3:4], 1.0)
    vo.backward()
    assert (gr[3:4].cpu() * 8 - po.grad).abs().max() <= 2e-6 * po.grad.abs().max()
    pa = p.clone().requires_grad_(True)
    assert abs(float(v) - float(O.loss_local_moments(pa, t, 1.0))) <= 1e-5 * abs(float(v))


def test_dlib_histogram_match_vs_reference_golden():
    """HistogramMatch (NORM1 / NORM2) against the reference at the default sigma 1e5 (only the neighbouring bins
    of a pixel are non-zero in float32 -- the kernel visits just those) and at soft sigmas where many / all
    bins contribute; then the sparse evaluation against the dense oracle at 8 x 256 x 256."""
    from dlib import loss as L
    from dlib.losses.elb import ELB
    from srhip import ops
    g = load("g14_hist")
    for name in ("l2_default", "l1_default", "l2_soft", "l1_wide"):
        lam, norm, sigma = [float(v) for v in g[name + "/cfg"]]
        l = L.HistogramMatch(cuda_id=0, lambda_=lam, elb=ELB(), color_min=0, color_max=255)
        l.set_it(norm_str=L.NORM1 if norm == 1 else L.NORM2, sigma=sigma)
        m = L.MasterLoss(cuda_id=0)
        m.add(l)
        p = g["pred"].cuda().requires_grad_(True)
        v = m(epoch=0, y_pred=p, y_target=g["target"].cuda(), trg_per_pixel_weight=None, model=None)
        v.backward()
        rv, rg = g[name + "/value"], g[name + "/grad"]
        assert abs(float(v) - float(rv)) <= 2e-5 * abs(float(rv)), (name, float(v), float(rv))
        assert (p.grad.cpu() - rg).abs().max() <= 2e-4 * float(rg.abs().max()), (name, (p.grad.cpu() - rg).abs().max())
        assert m.n_holder == list(g[name + "/names"])
    gen = torch.Generator().manual_seed(21)
    p = torch.rand(8, 1, 256, 256, generator=gen)
    t = torch.round(torch.rand(8, 1, 256, 256, generator=gen) ** 2 * 255) / 255
    for norm, sigma in ((2, 1e5), (1, 5e3)):
        gr = torch.empty_like(p).cuda()
        v = ops.loss_hist(p.cuda(), t.cuda(), 1.0, norm, sigma, 256, gr)
        po = p[:2].clone().requires_grad_(True)                      # the dense oracle on two images
        pa = torch.cat([po, p[2:]])
        vo = O.loss_histogram_match(pa, t, 1.0, norm, sigma, 256)
        vo.backward()
        assert abs(float(v) - float(vo)) <= 2e-5 * abs(float(vo)), (norm, sigma, float(v), float(vo))
        assert (gr[:2].cpu() - po.grad).abs().max() <= 5e-4 * po.grad.abs().max(), (norm, sigma)


def test_dlib_kde_match_vs_reference_golden():
    """KDEMatch (NORM1 / NORM2) against the reference at the default bandwidth and a wide one; the sparse
    evaluation (bins within sqrt(104 * 2 bw) of the pixel) against the dense oracle at 8 x 256 x 256."""
    from dlib import loss as L
    from dlib.losses.elb import ELB
    from srhip import ops
    g = load("g15_kde")
    for name in ("l2_default", "l1_default", "l2_wide"):
        lam, norm, bw = [float(v) for v in g[name + "/cfg"]]
        l = L.KDEMatch(cuda_id=0, lambda_=lam, elb=ELB(), color_min=0, color_max=1)
        l.set_it(norm_str=L.NORM1 if norm == 1 else L.NORM2, kde_bw=bw, ndim=1, nbins=256)
        m = L.MasterLoss(cuda_id=0)
        m.add(l)
        p = g["pred"].cuda().requires_grad_(True)
        v = m(epoch=0, y_pred=p, y_target=g["target"].cuda(), trg_per_pixel_weight=None, model=None)
        v.backward()
        rv, rg = g[name + "/value"], g[name + "/grad"]
        assert abs(float(v) - float(rv)) <= 2e-5 * abs(float(rv)), (name, float(v), float(rv))
        assert (p.grad.cpu() - rg).abs().max() <= 2e-4 * float(rg.abs().max()), (name, (p.grad.cpu() - rg).abs().max())
        assert m.n_holder == list(g[name + "/names"])
    gen = torch.Generator().manual_seed(22)
    p = torch.rand(8, 1, 256, 256, generator=gen)
    t = torch.round(torch.rand(8, 1, 256, 256, generator=gen) ** 2 * 255) / 255
    gr = torch.empty_like(p).cuda()
    v = ops.loss_kde(p.cuda(), t.cuda(), 1.0, 2, 1. / 255. ** 2, 256, gr)
    po = p[:2].clone().requires_grad_(True)
    vo = O.loss_kde_match(torch.cat([po, p[2:]]), t, 1.0, 2, 1. / 255. ** 2, 256)
    vo.backward()
    assert abs(float(v) - float(vo)) <= 2e-5 * abs(float(vo)), (float(v), float(vo))
    assert (gr[:2].cpu() - po.grad).abs().max() <= 5e-4 * po.grad.abs().max()


def test_dlib_hist_kde_kl_and_bhattacharyya_vs_reference_golden():
    """The remaining metrics: HistogramMatch KL / BHATTACHARYYA and KDEMatch BHATTACHARYYA (dlib/loss/main.py:677-898)
    against the reference, the barrier parameter at its initial value and after 30 schedule updates; then through the
    fused training-step loss path."""
    from dlib import loss as L
    from dlib.losses.elb import ELB
    from srhip import ops
    g = load("g24_hist_kl_bh")
    for name in ("hist_kl", "hist_bh_t1", "hist_bh_t30", "hist_kl_soft", "kde_bh_t1", "kde_bh_t30"):
        lam, norm, sigma, tval, updates = [float(v) for v in g[name + "/cfg"]]
        e = ELB()
        for _ in range(int(updates)):
            e.update_t()
        assert abs(float(e.get_t()) - tval) <= 1e-6
        norm_str = {3: L.KL, 4: L.BH}[int(norm)]
        if name.startswith("hist"):
            l = L.HistogramMatch(cuda_id=0, lambda_=lam, elb=e, color_min=0, color_max=255)
            l.set_it(norm_str=norm_str, sigma=sigma)
        else:
            l = L.KDEMatch(cuda_id=0, lambda_=lam, elb=e, color_min=0, color_max=1)
            l.set_it(norm_str=norm_str, kde_bw=1. / 255. ** 2, ndim=1, nbins=256)
        m = L.MasterLoss(cuda_id=0)
        m.add(l)
        p = g["pred"].cuda().requires_grad_(True)
        v = m(epoch=0, y_pred=p, y_target=g["target"].cuda(), trg_per_pixel_weight=None, model=None)
        v.backward()
        rv, rg = g[name + "/value"], g[name + "/grad"]
        assert abs(float(v) - float(rv)) <= 2e-5 * abs(float(rv)), (name, float(v), float(rv))
        assert (p.grad.cpu() - rg).abs().max() <= 2e-4 * float(rg.abs().max()), (name, (p.grad.cpu() - rg).abs().max())
        # the same term as the fused step evaluates it (tuple from MasterLoss.terms())
        gr = torch.empty_like(g["pred"]).cuda()
        t = m.terms()[0]
        f = ops.loss_hist if t[0] == "hist" else ops.loss_kde
        v2 = f(g["pred"].cuda(), g["target"].cuda(), t[1], t[2], t[3], t[4], gr, elb_t=float(t[5].get_t()))
        assert abs(float(v2) - float(rv)) <= 2e-5 * abs(float(rv)) and torch.equal(gr, p.grad)


def test_optional_loss_terms_full_size_properties():
    """At the benchmark's 8 x 512 x 512: linearity of the plain local-variation terms in the difference
    (loss(pred, target) == loss(pred - target, 0)), zero loss / zero gradient at pred == target, tile-seam
    independence (a shifted crop gives the same interior gradient), and the fused step's accumulation."""
    from srhip import ops
    gen = torch.Generator().manual_seed(3)
    p = torch.rand(8, 1, 512, 512, generator=gen).cuda()
    t = torch.rand(8, 1, 512, 512, generator=gen).cuda()
    z = torch.zeros_like(p)
    for kind, ksz in (("grad", 3), ("laplace", 3), ("lv", 5)):
        for norm in (1, 2):
            g1, g2 = torch.empty_like(p), torch.empty_like(p)
            v1 = ops.loss_stencil(p, t, kind, 1.0, norm, ksz, False, g1)
            v2 = ops.loss_stencil(p - t, z, kind, 1.0, norm, ksz, False, g2)
            assert abs(float(v1) - float(v2)) <= 1e-5 * abs(float(v2))
            if norm == 2:
                assert (g1 - g2).abs().max() <= 1e-5 * g2.abs().max()
            g0 = torch.empty_like(p)
            v0 = ops.loss_stencil(p, p.clone(), kind, 1.0, norm, ksz, False, g0)
            assert float(v0) == 0.0 and float(g0.abs().max()) == 0.0
    # oracle on a crop whose interior is far from the borders: same gradient up to the mean's denominator
    pc, tc = p[:1, :, 100:180, 200:296].contiguous(), t[:1, :, 100:180, 200:296].contiguous()
    for kind, norm, ksz, cn in (("lv", 2, 7, False), ("grad", 1, 3, True), ("lv", 2, 3, True)):
        gfull = torch.empty_like(p)
        ops.loss_stencil(p, t, kind, 1.0, norm, ksz, cn, gfull)
        po = pc.cpu().clone().requires_grad_(True)
        O.loss_local_variation(po, tc.cpu(), kind, 1.0, norm, ksz, cn).backward()
        scale = pc.numel() / p.numel()
        a = gfull[0, 0, 100:180, 200:296].cpu()[8:-8, 8:-8]
        b = (po.grad[0, 0] * scale)[8:-8, 8:-8]
        assert (a - b).abs().max() <= 2e-6 * b.abs().max(), (kind, norm, ksz, cn)
    # accumulation flags (how TrainStep sums MasterLoss terms)
    g = torch.empty_like(p)
    out = torch.zeros(1, device="cuda")
    ops.loss_pointwise(p, t, 2, 0.5, 1e-6, None, g, out)
    ops.loss_stencil(p, t, "laplace", 2.0, 1, 3, False, g, out, grad_accum=True, loss_accum=True)
    ga, gb = torch.empty_like(p), torch.empty_like(p)
    va = ops.loss_pointwise(p, t, 2, 0.5, 1e-6, None, ga)
    vb = ops.loss_stencil(p, t, "laplace", 2.0, 1, 3, False, gb)
    assert abs(float(out) - float(va) - float(vb)) <= 1e-6 * abs(float(out))
    assert (g - ga - gb).abs().max() <= 1e-6 * g.abs().max()


def test_interpolate_bicubic_baseline_vs_reference_golden():
    """the Bicubic baseline object (utils_trainer.py:89-167) on stock PyTorch-ROCm, against the
    reference's CPU output; PSNR against a target agrees to 0.01 dB."""
    from dlib.utils.utils_trainer import Interpolate
    from dlib.utils import constants
    from dlib import metrics as M
    g = load("g11_interpolate")
    for name in "ab":
        x = g[f"{name}/x"]
        for s_ in (2, 4, 8):
            m = Interpolate(task=constants.SUPER_RES, scale=s_, scale_mode=constants.INTER_BICUBIC)
            m.feed_data({'l_im': x, 'h_im': g[f"{name}/x{s_}"]})
            m.set_eval_mode()
            m.test()
            vis = m.current_visuals()
            ref = g[f"{name}/x{s_}"]
            assert vis['E'].shape == ref.shape and vis['E'].is_cuda
            assert (vis['E'].cpu() - ref).abs().mean() <= 1e-5
            assert float(vis['E'].min()) >= 0.0 and float(vis['E'].max()) <= 1.0
            tgt = torch.rand(ref.shape, generator=torch.Generator().manual_seed(s_)).cuda()
            pa = M.mbatch_gpu_calculate_psnr(M.tensor2uint82float(vis['E']), M.tensor2uint82float(tgt), border=0)
            pb = M.mbatch_gpu_calculate_psnr(M.tensor2uint82float(ref.cuda()), M.tensor2uint82float(tgt), border=0)
            assert (pa - pb).abs().max() <= 0.01


def test_patch_assembly_bit_exact_vs_reference_golden():
    """f2: the device-side batch assembly (srhip_patch_gather) against the reference's crop / augment_img /
    uint2single / single2tensor3 outputs, bit for bit; at the benchmark's size against the oracle; and its
    error behaviour."""
    from srhip import ops
    g = load("g12_patches")
    hr = [g["hr0"].cuda().contiguous(), g["hr1"].cuda().contiguous()]
    lr = [g["lr0"].cuda().contiguous(), g["lr1"].cuda().contiguous()]
    ids, modes = g["ids"].tolist(), g["modes"].tolist()
    y0, x0, sf = g["y0"].tolist(), g["x0"].tolist(), int(g["sf"])
    batch = ops.train_batch(hr, lr, ids, y0, x0, modes, 16, sf)
    assert torch.equal(batch["h_im"].cpu(), g["h_im"]) and torch.equal(batch["l_im"].cpu(), g["l_im"])
    # benchmark size: 8 patches of 512 x 512 out of 1024 x 1280 tiles, every mode; augmentations are
    # permutations (sorted values equal) and mode pairs invert each other
    gen = torch.Generator().manual_seed(8)
    tiles = [torch.randint(0, 256, (1024, 1280), generator=gen, dtype=torch.uint8) for _ in range(3)]
    tl = [t.cuda() for t in tiles]
    ids8, m8 = [0, 1, 2, 0, 1, 2, 0, 1], list(range(8))
    yy, xx = [0, 512, 100, 37, 511, 256, 3, 400], [768, 0, 5, 700, 123, 64, 767, 333]
    out = ops.patch_gather(tl, ids8, yy, xx, m8, 512)
    ref = O.patch_batch([t.numpy() for t in tiles], ids8, yy, xx, m8, 512)
    assert torch.equal(out.cpu(), ref)
    base = ops.patch_gather(tl, ids8, yy, xx, [0] * 8, 512)
    assert torch.equal(out.flatten(1).sort(1).values, base.flatten(1).sort(1).values)
    with pytest.raises(RuntimeError, match="outside"):
        ops.patch_gather(tl, [0], [600], [0], [0], 512)
    with pytest.raises(RuntimeError, match="mode"):
        ops.patch_gather(tl, [0], [0], [0], [8], 512)


def test_dlib_metrics_surface_vs_reference_golden():
    from dlib import metrics as M
    from dlib.utils import utils_image
    assert utils_image.mbatch_gpu_calculate_psnr is M.mbatch_gpu_calculate_psnr
    g = load("g7_metrics")
    a = M.tensor2uint82float(g["pred"].cuda())
    b = M.tensor2uint82float(g["hr"].cuda())
    assert torch.equal(a.cpu(), g["a"]) and torch.equal(b.cpu(), g["b"])
    assert torch.equal(M.tensor2uint82float(g["corner"].cuda()).cpu(), g["corner_out"])
    border = int(g["border"])
    for th in (None, 4, 7, 10, 300):
        roi = None if th is None else (b >= th).float()
        tag = "noroi" if th is None else f"roi{th}"
        for nm, fn, tol in (("psnr", M.mbatch_gpu_calculate_psnr, 1e-9), ("mse", M.mbatch_gpu_calculate_mse, 0),
                            ("nrmse", M.mbatch_gpu_calculate_nrmse, 1e-12),
                            ("ssim", M.mbatch_gpu_calculate_ssim, 5e-5)):
            v = fn(a, b, border=border, roi=roi).cpu()
            assert (v.double() - g[f"{nm}/{tag}"].double()).abs().max() <= tol, (nm, tag)
    with pytest.raises(NotImplementedError):
        M.mbatch_gpu_calculate_psnr(a, b, border, roi=(torch.rand_like(b) > 0.5).float())
    sw = M.sweep(g["pred"].cuda(), g["hr"].cuda(), border, (4, 5, 6, 7, 8, 9, 10))
    assert sw["psnr"].shape == (3, 8) and torch.equal(sw["mse"][:, 0].cpu(), g["mse/noroi"])


class Args(dict):
    __getattr__ = dict.get


def tiny_args(opt="adam"):
    from dlib.utils import constants
    netG = {'net_type': constants.SWINIR, 'swinir_upscale': 8, 'swinir_in_chans': 1, 'swinir_img_size': 16,
            'swinir_window_size': 8, 'swinir_img_range': 1.0, 'swinir_depths': [2, 2], 'swinir_embed_dim': 60,
            'swinir_num_heads': [6, 6], 'swinir_mlp_ratio': 2,
            'swinir_upsampler': constants.US_PIXEL_SHUFFLE_DIRECT,
            'swinir_resi_connection': constants.R_CONNECTION_1CONV}
    train = {'l1': True, 'G_optimizer_type': opt, 'G_optimizer_lr': 2e-4 if opt == "adam" else 0.01,
             'G_optimizer_wd': 1e-4 if opt == "adam" else 0.0, 'G_scheduler_type': 'MyStepLR',
             'G_scheduler_step_size': 30, 'G_scheduler_gamma': 0.5, 'G_scheduler_min_lr': 1e-4}
    return Args(netG=netG, train=train, is_train=True)


@pytest.mark.parametrize("opt", ["adam", "sgd"])
def test_model_plain_steps_match_oracle_training(tmp_path, opt):
    """Two fused optimisation steps == two oracle (autograd + torch-semantics
    optimizer) steps, parameter for parameter; checkpoint round trip."""
    from dlib.models.select_model import define_model
    args = tiny_args(opt)
    args['outd'] = str(tmp_path)
    model = define_model(args)
    cfg = O.swinir_config(upscale=8, in_chans=1, img_size=16, window_size=8, depths=(2, 2), embed_dim=60,
                          num_heads=(6, 6), mlp_ratio=2, drop_path_rate=0.0)
    sd0 = O.swinir_init_state_dict(cfg, seed=9)
    model.netG.load_state_dict(sd0, strict=True)
    for b in model.netG.swin_blocks():
        b.drop_prob = 0.0
    model.init_train()
    gen = torch.Generator().manual_seed(12)
    batch = {'l_im': torch.rand(2, 1, 16, 16, generator=gen), 'h_im': torch.rand(2, 1, 128, 128, generator=gen)}
    sdo = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith("attn_mask")
               else v) for k, v in sd0.items()}
    names = [k for k, v in sdo.items() if v.requires_grad]
    st = {k: (torch.zeros_like(sdo[k]), torch.zeros_like(sdo[k])) for k in names}
    for step in range(2):
        model.feed_data(batch)
        model.optimize_parameters(epoch=0, current_step=step)
        model.update_learning_rate()
        lo = O.loss_l1(O.swinir_forward(sdo, batch['l_im'], cfg), batch['h_im'])
        for k in names:
            sdo[k].grad = None
        lo.backward()
        with torch.no_grad():
            for k in names:
                if opt == "adam":
                    O.adam_step(sdo[k], sdo[k].grad, st[k][0], st[k][1], step + 1, 2e-4, wd=1e-4)
                else:
                    O.sgd_nesterov_step(sdo[k], sdo[k].grad, st[k][0], step == 0, 0.01)
        assert abs(model.current_log()['G_loss'] - lo.item()) <= 1e-5
    assert model.check_finite()
    for k, p in model.netG.named_parameters():
        e = (p.detach().cpu() - sdo[k].detach()).abs().max().item()
        assert e <= 2e-6, f"{k}: {e}"
    # reference-format checkpoint round trip
    path = model.save(2)
    raw = torch.load(path)
    assert list(raw.keys()) == list(sd0.keys())
    model.feed_data(batch)
    model.test()
    e1 = model.current_visuals()['E'].clone()
    model.netG.load_state_dict(sd0)
    model.load_network(path, model.netG)
    model.test()
    assert torch.equal(model.current_visuals()['E'], e1)


@pytest.mark.parametrize("opt,graph", [("adam", False), ("sgd", True)])
def test_model_plain_clipgrad_and_ema_match_oracle_training(tmp_path, opt, graph):
    """G_optimizer_clipgrad + E_decay through the fused step (model_plain.py:350-361,393-394): three steps
    == the oracle's autograd + clip_grad_norm (pinned against torch's / the reference's calls by g50) + optimizer + update_E,
    parameter for parameter, eager and replayed from a hipGraph; <iter>_E.pth is netE's state_dict and resumes."""
    from dlib.models.select_model import define_model
    args = tiny_args(opt)
    args['outd'] = str(tmp_path)
    args['train_graph'] = graph
    max_norm, decay = 0.05 if opt == "adam" else 0.4, 0.9
    args['train'].update({'G_optimizer_clipgrad': max_norm, 'E_decay': decay})
    model = define_model(args)
    cfg = O.swinir_config(upscale=8, in_chans=1, img_size=16, window_size=8, depths=(2, 2), embed_dim=60,
                          num_heads=(6, 6), mlp_ratio=2, drop_path_rate=0.0)
    sd0 = O.swinir_init_state_dict(cfg, seed=9)
    model.netG.load_state_dict(sd0, strict=True)
    for b in model.netG.swin_blocks():
        b.drop_prob = 0.0
    model.init_train()
    gen = torch.Generator().manual_seed(12)
    batch = {'l_im': torch.rand(2, 1, 16, 16, generator=gen), 'h_im': torch.rand(2, 1, 128, 128, generator=gen)}
    sdo = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith("attn_mask")
               else v) for k, v in sd0.items()}
    names = [k for k, v in sdo.items() if v.requires_grad]
    st = {k: (torch.zeros_like(sdo[k]), torch.zeros_like(sdo[k])) for k in names}
    ema = {k: sdo[k].detach().clone() for k in names}
    clipped = []
    for step in range(3):
        model.feed_data(batch)
        model.optimize_parameters(epoch=0, current_step=step)
        lo = O.loss_l1(O.swinir_forward(sdo, batch['l_im'], cfg), batch['h_im'])
        for k in names:
            sdo[k].grad = None
        lo.backward()
        with torch.no_grad():
            total, coef = O.clip_grad_norm([sdo[k].grad for k in names], max_norm)
            clipped.append(float(coef) < 1.0)
            norm_dev, coef_dev = model.step_fn.clip_state.tolist()
            assert abs(norm_dev - float(total)) <= 2e-5 * float(total), (norm_dev, float(total))
            assert abs(coef_dev - float(coef)) <= 2e-5
            for k in names:
                if opt == "adam":
                    O.adam_step(sdo[k], sdo[k].grad, st[k][0], st[k][1], step + 1, 2e-4, wd=1e-4)
                else:
                    O.sgd_nesterov_step(sdo[k], sdo[k].grad, st[k][0], step == 0, 0.01)
            O.ema_update([ema[k] for k in names], [sdo[k] for k in names], decay)
    assert any(clipped), "the test's max_norm never clipped"
    for k, p in model.netG.named_parameters():
        e = (p.detach().cpu() - sdo[k].detach()).abs().max().item()
        assert e <= 2e-6, f"{k}: {e}"
    esd = model.step_fn.ema_state_dict()
    assert list(esd.keys()) == list(sd0.keys())
    for k in names:
        e = (esd[k] - ema[k]).abs().max().item()
        assert e <= 2e-6, f"netE {k}: {e}"
    assert torch.equal(esd["layers.0.residual_group.blocks.0.attn.relative_position_index"],
                       sd0["layers.0.residual_group.blocks.0.attn.relative_position_index"])
    # <iter>_E.pth beside <iter>_G.pth, E-current_model.pth / E-model.pth; a new model resumes the average from it
    model.save(3)
    raw = torch.load(os.path.join(model.save_dir, "3_E.pth"))
    assert list(raw.keys()) == list(sd0.keys()) and all(torch.equal(raw[k], esd[k]) for k in raw)
    model.save_current(str(tmp_path / "cur"))
    model.save_best(str(tmp_path / "best"), "model.pth")
    assert os.path.isfile(tmp_path / "cur" / "E-current_model.pth") and os.path.isfile(tmp_path / "best" / "E-model.pth")
    args2 = tiny_args(opt)
    args2['outd'] = str(tmp_path)
    args2['train'].update({'G_optimizer_clipgrad': max_norm, 'E_decay': decay})
    args2['netG']['checkpoint_path_netE'] = os.path.join(model.save_dir, "3_E.pth")
    m2 = define_model(args2)
    m2.init_train()
    e2 = m2.step_fn.ema_state_dict()
    assert all(torch.equal(e2[k], esd[k]) for k in names)
    # without E_decay there is no netE and no E file; clipgrad 0 leaves the step unclipped (clip_state absent)
    args3 = tiny_args(opt)
    args3['outd'] = str(tmp_path / "plain")
    m3 = define_model(args3)
    m3.init_train()
    assert m3.step_fn.ema_flat is None and m3.step_fn.clip_state is None
    m3.save(1)
    assert not os.path.isfile(os.path.join(m3.save_dir, "1_E.pth"))


def test_model_plain_refuses_what_it_does_not_implement(tmp_path):
    """--amp True in training, G_regularizer_*: NotImplementedError, never a silently different run."""
    from dlib.models.select_model import define_model
    args = tiny_args("adam")
    args['outd'] = str(tmp_path)
    args['amp'] = True
    model = define_model(args)
    model.init_train()
    model.feed_data({'l_im': torch.rand(2, 1, 16, 16), 'h_im': torch.rand(2, 1, 128, 128)})
    with pytest.raises(NotImplementedError, match="amp"):
        model.optimize_parameters(0, 0)
    model.test()                                   # evaluation under --amp stays available
    assert model.current_visuals()['E'].shape == (2, 1, 128, 128)
    for k in ("G_regularizer_orthstep", "G_regularizer_clipstep"):
        a = tiny_args("adam")
        a['outd'] = str(tmp_path)
        a['train'][k] = 10
        with pytest.raises(NotImplementedError, match=k):
            define_model(a)


def test_model_plain_step_with_optional_loss_terms(tmp_path):
    """MasterLoss = Charbonnier + 0.5 * LocalVariation(5, NORM1) + 2 * NormImageGradient(NORM2) through the
    fused step (config keys of utils_config.py:301-357): loss values and the updated parameters against the
    oracle's autograd step."""
    from dlib.models.select_model import define_model
    args = tiny_args("sgd")
    args['outd'] = str(tmp_path)
    args['train'].update({'l1': False, 'charbonnier': True, 'charbonnier_eps': 1e-6, 'loc_var': True,
                          'loc_var_ksz': 5, 'loc_var_norm': '1', 'loc_var_lambda': 0.5, 'norm_img_grad': True,
                          'norm_img_grad_type': '2', 'norm_img_grad_lambda': 2.0, 'boundpred': True,
                          'boundpred_eps': 2.0, 'boundpred_lambda': 0.25, 'w_sparsity': True,
                          'w_sparsity_lambda': 1e-4})
    model = define_model(args)
    cfg = O.swinir_config(upscale=8, in_chans=1, img_size=16, window_size=8, depths=(2, 2), embed_dim=60,
                          num_heads=(6, 6), mlp_ratio=2, drop_path_rate=0.0)
    sd0 = O.swinir_init_state_dict(cfg, seed=4)
    model.netG.load_state_dict(sd0, strict=True)
    for b in model.netG.swin_blocks():
        b.drop_prob = 0.0
    model.init_train()
    assert model.loss_fn.n_holder == ['master_loss', 'charbonnier', 'bounded_prediction',
                                      'norm_image_gradient_loss', 'local_variation_loss',
                                      'weights_sparsity_loss']         # the reference's order of addition
    terms = [(t[0], t[1], t[2], 1.0, t[4], t[5]) if t[0] == "boundpred" else t
             for t in model.loss_fn.terms() if t[0] != "w_sparsity"]
    gen = torch.Generator().manual_seed(2)
    batch = {'l_im': torch.rand(2, 1, 16, 16, generator=gen), 'h_im': torch.rand(2, 1, 128, 128, generator=gen)}
    sdo = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith("attn_mask")
               else v) for k, v in sd0.items()}
    names = [k for k, v in sdo.items() if v.requires_grad]
    model.feed_data(batch)
    model.optimize_parameters(epoch=0, current_step=0)
    tot, holder = O.master_loss(O.swinir_forward(sdo, batch['l_im'], cfg), batch['h_im'], terms)
    wsp = O.loss_weights_sparsity([sdo[k] for k in names], 1e-4)
    tot = tot + wsp
    holder = [tot] + holder[1:] + [wsp]
    tot.backward()
    model.current_log()
    got = [float(v) for v in model.loss_fn.l_holder]
    assert len(got) == len(holder) == 6
    for a, b in zip(got, holder):
        assert abs(a - float(b)) <= 1e-5 * max(1.0, abs(float(b))), (got, [float(h) for h in holder])
    with torch.no_grad():
        for k in names:
            O.sgd_nesterov_step(sdo[k], sdo[k].grad, torch.zeros_like(sdo[k]), True, 0.01)
    for k, p in model.netG.named_parameters():
        e = (p.detach().cpu() - sdo[k].detach()).abs().max().item()
        assert e <= 2e-6, f"{k}: {e}"


def test_model_plain_nonfinite_loss_skips_update(tmp_path):
    from dlib.models.select_model import define_model
    args = tiny_args("sgd")
    args['outd'] = str(tmp_path)
    model = define_model(args)
    model.init_train()
    before = model.step_fn.fp.flat.clone()
    bad = {'l_im': torch.rand(2, 1, 16, 16), 'h_im': torch.full((2, 1, 128, 128), float('nan'))}
    model.feed_data(bad)
    model.optimize_parameters(0, 0)
    assert not model.check_finite()
    assert torch.equal(model.step_fn.fp.flat, before)        # update skipped on the device
