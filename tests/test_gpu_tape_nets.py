"""DBPN and SRFBN on libsrhip (dlib.models.network_dbpn / network_srfbn: tape graphs over the libsrhip kernels,
srhip/tape.py) against the fixtures generated from the reference classes (g28_dbpn.npz, g29_srfbn.npz: outputs and
the gradient of every parameter) and against the oracle at the registry's default widths."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import sr_oracle as O  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# tiny fixtures (2 x 1 x 6 x 5 inputs): a PReLU decision that flips under f32 rounding moves the entries it feeds by ~1 /
# pixels of their size, so tensors are gated by their relative L2 error and their |gradient| sums, entries at 5e-5 of the
# tensor's largest one where no decision sits at the edge (see tests/test_gpu_mslapsrn.py for the same argument)
L2_GATE = 1e-4


def load(name):
    z = np.load(os.path.join(G, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiu" else z[k]) for k in z.files}


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def check_grads(named_grads, g):
    names = [str(n) for n in g["grad_names"]]
    sums = g["grad_sums"].numpy()
    assert [k for k, _ in named_grads] == names
    for i, (k, gk) in enumerate(named_grads):
        gk = gk.double().cpu()
        if gk.numel() == 1:
            # a PReLU slope's gradient is ONE sum of signed terms over a whole feature map: it may cancel to 1e-6 of its
            # terms, so it is held to an absolute error (1e-4 relative where it does not cancel)
            assert abs(gk.item() - sums[i][0]) <= max(1e-4 * abs(sums[i][0]), 5e-7), (k, gk.item(), sums[i][0])
            continue
        if "grad/" + k in g:
            ref = g["grad/" + k].double()
            assert ((gk - ref).norm() / ref.norm().clamp_min(1e-30)).item() <= L2_GATE, (k, ((gk - ref).norm() / ref.norm()).item())
        assert abs(gk.abs().sum().item() - sums[i][1]) <= 1e-4 * max(sums[i][1], 1e-6), (k, gk.abs().sum().item(), sums[i][1])


@pytest.mark.parametrize("scale", [2, 4, 8])
def test_dbpn_forward_and_gradients_vs_reference_golden(scale):
    from dlib.models.network_dbpn import DBPN
    allg = load("g28_dbpn")
    g = {k[len(f"x{scale}/"):]: v for k, v in allg.items() if k.startswith(f"x{scale}/")}
    cfg = dict(base_filter=16, feat=32, num_stages=2)
    sd = O.dbpn_init_state_dict(scale, 1, seed=int(g["seed"]), bias_std=0.05, **cfg)
    net = DBPN(upscale=scale, in_chans=1, **cfg)
    assert sorted((k, tuple(v.shape)) for k, v in net.state_dict().items()) == sorted((k, tuple(v.shape)) for k, v in sd.items())
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    y = net(g["x"].cuda())
    assert (y.detach().cpu() - g["y"]).abs().mean() <= 1e-5 and rel(y, g["y"]) < 1e-5
    (y - g["target"].cuda()).abs().mean().backward()
    check_grads([(k, p.grad) for k, p in net.named_parameters()], g)
    net.eval()
    with torch.no_grad():
        assert rel(net(g["x"].cuda()), g["y"]) < 1e-5
    # the registry's default net has the reference's state_dict keys
    assert sorted(DBPN(upscale=2, in_chans=1).state_dict().keys()) == sorted(str(k) for k in allg["state_dict_keys_default"])


@pytest.mark.parametrize("scale", [2, 3, 4, 8])
def test_srfbn_all_passes_and_curriculum_gradients_vs_reference_golden(scale):
    from dlib.models.network_srfbn import SRFBN
    allg = load("g29_srfbn")
    g = {k[len(f"x{scale}/"):]: v for k, v in allg.items() if k.startswith(f"x{scale}/")}
    cfg = dict(num_features=16, num_steps=3, num_groups=3)
    sd = O.srfbn_init_state_dict(scale, 1, cfg["num_features"], cfg["num_groups"], seed=int(g["seed"]))
    net = SRFBN(upscale=scale, in_chans=1, **cfg)
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(k, tuple(v.shape)) for k, v in sd.items()]
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    x, tgt = g["x"].cuda(), g["target"].cuda()
    eng = net.engine
    y = eng.forward(x[:, 0].contiguous(), None, save=True)
    outs = eng.all_outs
    assert len(outs) == cfg["num_steps"] and outs[-1].data_ptr() == y.data_ptr()
    for i, o in enumerate(outs):
        assert rel(o, g[f"y{i}"]) < 1e-5, i
    # the trainer's curriculum loss (model_plain.py:202-232): mean over the passes of mean |out - target|
    n = tgt.numel() * len(outs)
    ds = [torch.sign(o - tgt) / n for o in outs]
    grads = {k: torch.full_like(p, float("nan")) for k, p in net.named_parameters() if p.requires_grad}
    eng.backward(ds[-1], grads, d_inter=ds[:-1])
    check_grads([(k, grads[k]) for k, p in net.named_parameters() if p.requires_grad], g)
    # module path (forward only hands out the last prediction, the others through intermediate_outs)
    net.eval()
    with torch.no_grad():
        ye = net(x)
    assert rel(ye, g[f"y{cfg['num_steps'] - 1}"]) < 1e-5 and len(net.intermediate_outs) == cfg["num_steps"]
    assert [str(k) for k in allg["state_dict_keys_default"]] == list(SRFBN(upscale=2, in_chans=1).state_dict().keys())


@pytest.mark.parametrize("scale", [2, 4, 8])
def test_prosr_levels_and_multiscale_gradients_vs_reference_golden(scale):
    from dlib.models.network_prosr import ProSR
    allg = load("g30_prosr")
    g = {k[len(f"x{scale}/"):]: v for k, v in allg.items() if k.startswith(f"x{scale}/")}
    n = int(np.log2(scale))
    cfg = O.prosr_config(upscale=scale, num_init_features=32, bn_size=2, growth_rate=8, level_config=[[3, 2], [2], [2]][:n])
    sd = O.prosr_init_state_dict(cfg, seed=int(g["seed"]), bias_std=0.05)
    net = ProSR(upscale=scale, in_chans=1, num_init_features=32, bn_size=2, growth_rate=8, level_config=cfg["level_config"])
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(k, tuple(v.shape)) for k, v in sd.items()]
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    x, tgt = g["x"].cuda(), g["target"].cuda()
    eng = net.engine
    y = eng.forward(x[:, 0].contiguous(), None, save=True)
    outs = list(eng.intermediate_outs) + [y]
    assert len(outs) == n
    for i, o in enumerate(outs):
        assert rel(o, g[f"y{i}"]) < 1e-5, i
    # the trainer's multi-scale loss (model_plain.py:236-275): gradients of (sum of L1 against the resized target) / levels
    leaves = [o.detach().clone().requires_grad_(True) for o in outs]
    O.mslapsrn_loss(leaves[-1], leaves[:-1], tgt).backward()
    grads = {k: torch.full_like(p, float("nan")) for k, p in net.named_parameters()}
    eng.backward(leaves[-1].grad, grads, d_inter=[l.grad for l in leaves[:-1]])
    used = [str(k) for k in g["grad_names"]]
    check_grads([(k, grads[k]) for k in used], g)
    for k in grads:                 # init convs of the scales that were not requested: zero gradient
        if k not in used:
            assert k.startswith("init_conv_") and float(grads[k].abs().max()) == 0.0, k
    net.eval()
    with torch.no_grad():
        ye = net(x)
    assert rel(ye, g[f"y{n - 1}"]) < 1e-5 and len(net.intermediate_outs) == n - 1
    for sc in (2, 4, 8):
        assert list(ProSR(upscale=sc, in_chans=1, level_config=O.prosr_config(upscale=sc)["level_config"]).state_dict().keys()) \
            == [str(k) for k in allg[f"state_dict_keys_default_x{sc}"]]


@pytest.mark.parametrize("slopes", ["identity", "trained"])
@pytest.mark.parametrize("net_type,scale", [("DBPN", 4), ("SRFBN", 4), ("DBPN", 8), ("SRFBN", 2)])
def test_default_width_train_step_vs_oracle(net_type, scale, slopes):
    """The registry's default widths (DBPN: feat 256 / base 64 / 3 passes; SRFBN: 64 features, 6 groups, 4 passes) through
    the fused training step on a 16 x 16 patch: loss and every gradient against the oracle run in float64 (a PReLU slope's
    gradient is one cancelling sum over whole feature maps: the fp32 oracle itself is 1e-3 off on it)."""
    from srhip.train import TrainStep, Optimizer
    torch.manual_seed(3)
    x = torch.rand(2, 1, 16, 16)
    tgt = torch.rand(2, 1, 16 * scale, 16 * scale)
    if net_type == "DBPN":
        from dlib.models.network_dbpn import DBPN
        sd = O.dbpn_init_state_dict(scale, 1, seed=11, bias_std=0.02)
        net = DBPN(upscale=scale, in_chans=1)
        fwd = lambda s: [O.dbpn_forward(s, x, scale, 3)]
    else:
        from dlib.models.network_srfbn import SRFBN
        sd = O.srfbn_init_state_dict(scale, 1, seed=12)
        net = SRFBN(upscale=scale, in_chans=1)
        fwd = lambda s: O.srfbn_forward(s, x, scale, 4, 6)
    # A PReLU whose input sits within fp32 rounding of 0 takes the other slope in one of two fp32 computations; the output
    # is continuous there but the gradient through that pixel jumps by (1 - a), i.e. one pixel enters or leaves the weight
    # gradients upstream (measured here: single entries 4e-4 .. 8e-4 of the tensor's largest one at the default widths,
    # identically with the exact-f32 kernels, SRHIP_MM=f32 -- the same effect tests/test_gpu_fullsize.py documents for
    # EDSR's ReLUs).  So the entry-wise gate runs with all slopes = 1 (no jump: every conv / transposed / strided conv /
    # concatenation / weight-sharing path at full width, 5e-5), and the trained-like slopes are held tensor-wise.
    if slopes == "identity":
        for k in sd:
            if k.endswith("act.weight") or (k.endswith(".1.weight") and sd[k].numel() == 1):
                sd[k] = torch.ones(1)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "sgd", lr=0.0, momentum=0.0, nesterov=False, wd=0.0)        # lr 0: the gradients stay readable
    ts.step(x.cuda(), tgt.cuda())
    # the same step through the oracle in fp32: its own distance from the float64 run is the yardstick (deep recurrences:
    # SRFBN back-propagates through 4 passes x 6 groups; a PReLU decision within rounding of 0 flips in either fp32 run)
    sd32 = {k: (v.clone().requires_grad_(True) if not k.startswith(("sub_mean", "add_mean")) else v.clone()) for k, v in sd.items()}
    o32 = (lambda s_: [O.dbpn_forward(s_, x, scale, 3)])(sd32) if net_type == "DBPN" else O.srfbn_forward(sd32, x, scale, 4, 6)
    (sum((o - tgt).abs().mean() for o in o32) / len(o32)).backward()
    sdo = {k: (v.double().requires_grad_(True) if not k.startswith(("sub_mean", "add_mean")) else v.double()) for k, v in sd.items()}
    x, tgt = x.double(), tgt.double()
    outs = fwd(sdo)
    loss = sum((o - tgt).abs().mean() for o in outs) / len(outs)
    loss.backward()
    assert abs(ts.loss_values()[0] - loss.item()) <= 2e-6 * max(1.0, abs(loss.item()))
    worst = 0.0
    # a PReLU slope's gradient is one signed sum over whole feature maps that cancels to a small remainder: it is held to
    # an error relative to the LARGEST slope gradient of the net (what the terms of such a sum are scaled like)
    smax = max([abs(sdo[k].grad.item()) for k in ts.fp.names if sdo[k].grad.numel() == 1] + [0.0])
    for k in ts.fp.names:
        ref = sdo[k].grad
        got = ts.fp.gviews[k].double().cpu()
        if ref.numel() == 1:
            assert abs(got.item() - ref.item()) <= 2e-4 * abs(ref.item()) + 1e-3 * smax, (k, got.item(), ref.item(), smax)
            continue
        # the library's gradient gate: entries relative to the tensor's largest one (2e-5 on the hand-sequenced engines;
        # these graphs are 4 passes deep through shared weights, and from 64 channels on their weight gradients run on the
        # split-MFMA TN kernels whose precision is relative to an operand COLUMN's maximum)
        e = ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
        e32 = ((sd32[k].grad.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
        el2 = ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()
        worst = max(worst, e)
        worst_l2 = max(locals().get("worst_l2", 0.0), el2)
        if slopes == "identity":
            assert e <= max(3.0 * e32, 5e-5), (k, e, e32, el2)
        else:
            assert el2 <= 2e-3, (k, e, e32, el2)
    print(f"{net_type} x{scale}: loss {loss.item():.6f}, worst gradient error: {worst:.2e} of the tensor's largest entry, "
          f"{worst_l2:.2e} relative L2")


@pytest.mark.parametrize("scale", [2, 4, 8])
def test_enlcn_forward_vs_reference_golden(scale):
    """ENLCN (network_enlcn.py): narrow configuration of g33_enlcn.npz -- ENLCA blocks (1x1 embeddings, L2 normalisation,
    positive random features, linear attention with the normaliser as one more value column), ResBlocks with res_scale,
    the F -> 4F upsampler convs as four slices -- against the reference's own output; evaluation only (backward raises)."""
    from dlib.models.network_enlcn import ENLCN
    g = {k[len(f"x{scale}/"):]: v for k, v in load("g33_enlcn").items() if k.startswith(f"x{scale}/")}
    sd = O.enlcn_init_state_dict(scale, 1, 8, 64, seed=int(g["seed"]))
    net = ENLCN(upscale=scale, in_chans=1, n_resblock=8, n_feats=64)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    x, yref = g["x"], g["y"]
    with torch.no_grad():
        y = net(x.cuda()).cpu()
    assert (y - yref).abs().mean().item() <= 1e-5 and rel(y, yref) < 2e-5, rel(y, yref)


def test_enlcn_training_step_gradients_vs_reference_golden():
    """ENLCN trains: forward in training mode, L1 loss, every parameter gradient of the narrow
    configuration against the REFERENCE's own autograd (g39_enlcn_grad.npz: written by oracle/make_goldens.py::g_enlcn_grad
    from the imported reference net, the oracle's autograd asserted equal) -- ENLCA's backward (1x1 embeddings, L2
    normalisation, positive random features, linear attention with the normaliser column), ResBlocks, the F -> 4F upsampler
    convs run as four output slices (gradients into the matching rows).  Gate: the library's 2e-5 of a tensor's largest
    entry; ReLU decisions within rounding of zero are excused as in tests/test_gpu_fullsize.py by a 3x-the-fp32-oracle arm."""
    from dlib.models.network_enlcn import ENLCN
    from srhip.train import TrainStep, Optimizer
    scale = 2
    g = {k[len(f"x{scale}/"):]: v for k, v in load("g39_enlcn_grad").items() if k.startswith(f"x{scale}/")}
    sd = O.enlcn_init_state_dict(scale, 1, 8, 64, seed=int(g["seed"]))
    net = ENLCN(upscale=scale, in_chans=1, n_resblock=8, n_feats=64)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "sgd", lr=0.0, momentum=0.0, nesterov=False, wd=0.0)        # lr 0: the gradients stay readable
    x, tgt = g["x"], g["tgt"]
    ts.step(x.cuda(), tgt.cuda())
    assert abs(ts.loss_values()[0] - float(g["loss"])) <= 2e-6
    # yardstick for ReLU flips: the fp32 oracle's own distance from an fp64 run
    sd64 = {k: (v.double().requires_grad_(True) if v.dtype == torch.float32 and not k.startswith(("sub_mean", "add_mean"))
                else v) for k, v in sd.items()}
    (O.enlcn_forward(sd64, x.double(), scale, 8, 0.1) - tgt.double()).abs().mean().backward()
    worst, n = 0.0, 0
    for k in ts.fp.names:
        ref = g["grad/" + k].double()
        got = ts.fp.gviews[k].double().cpu()
        den = ref.abs().max().clamp_min(1e-30)
        e = ((got - ref).abs().max() / den).item()
        e32 = ((ref - sd64[k].grad).abs().max() / den).item()          # the reference's fp32 autograd vs fp64
        worst = max(worst, e)
        n += 1
        assert e <= max(2e-5, 3.0 * e32), (k, e, e32)
    assert n == 52
    print(f"ENLCN x{scale} training step: loss {ts.loss_values()[0]:.6f}, worst gradient error {worst:.2e} of a tensor's largest entry")


def test_main_cli_trains_enlcn(tmp_path):
    """`main.py --net_type ENLCN --max_iters 20`: the registry net through ModelPlain's step, loss finite and falling."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "sr-caco-2_amd", "main.py"), "--net_type", "ENLCN", "--method", "ENLCN",
                        "--task", "super-resolution", "--scale", "4", "--n_channels", "1", "--h_size", "128", "--batch_size", "2",
                        "--max_iters", "20", "--G_optimizer_lr", "1e-4", "--ENLCN_n_resblock", "8", "--ENLCN_n_feats", "64",
                        "--outd", str(tmp_path)], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    losses = [float(l.split("G_loss")[1].split()[0]) for l in p.stdout.splitlines() if "G_loss" in l]
    assert len(losses) == 2 and all(np.isfinite(losses)) and losses[1] < losses[0], losses
    assert os.path.isfile(os.path.join(str(tmp_path), "models", "20_G.pth"))


def test_enlcn_registry_default_width_vs_oracle():
    """The registry's net (32 ResBlocks, 256 features, five ENLCA blocks with 64-dim embeddings) at x4 on 24 x 20 inputs
    against the oracle; and --amp within the PSNR gate."""
    from dlib.models.network_enlcn import ENLCN
    sd = O.enlcn_init_state_dict(4, 1, seed=5)
    net = ENLCN(upscale=4, in_chans=1)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    gen = torch.Generator().manual_seed(9)
    x = torch.rand(2, 1, 24, 20, generator=gen)
    with torch.no_grad():
        yref = O.enlcn_forward(sd, x, 4)
        y = net(x.cuda()).cpu()
        net.amp = True
        ya = net(x.cuda()).cpu()
    assert (y - yref).abs().mean().item() <= 1e-5 and rel(y, yref) < 2e-5, rel(y, yref)
    mse = ((ya.clamp(0, 1) - yref.clamp(0, 1)) ** 2).mean().item()        # --amp: PSNR of the amp output against the f32 one
    assert mse < 1e-5, mse


@pytest.mark.parametrize("scale", [2, 4, 8])
def test_nlsn_forward_vs_reference_golden(scale):
    """NLSN (network_nlsn.py), narrow configuration of g34_nlsn.npz (x2: 720 tokens, 5 chunks; x4: 480 tokens, chunk
    padding 96; x8: 360 tokens, padding 72, three PixelShuffle(2) stages), fed the LSH rotations the reference drew.  (1) the hash codes agree with the reference's except where two
    rotated components tie to rounding; (2) with the oracle replaying THIS run's token order -- the reference leaves the
    order inside a hash bucket to torch.sort, here it is by token index -- the outputs agree to f32 rounding; (3) against
    the reference's own output (its order) the image differs only where bucket boundaries moved."""
    from dlib.models.network_nlsn import NLSN
    g = {k[len(f"x{scale}/"):]: v for k, v in load("g34_nlsn").items() if k.startswith(f"x{scale}/")}
    sd = O.nlsn_init_state_dict(scale, 1, 8, 64, seed=int(g["seed"]))
    net = NLSN(upscale=scale, in_chans=1, n_resblocks=8, n_feats=64)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    x, yref = g["x"], g["y"]
    rots = [g["rot0"], g["rot1"]]
    net.engine.rotations = [r.cuda() for r in rots]
    net.engine.taps = taps = []
    with torch.no_grad():
        y = net(x.cuda()).cpu()
    N, L = x.shape[0], x.shape[2] * x.shape[3]
    idx = []
    for a, tp in enumerate(taps):
        order = tp["order"].cpu()                                  # [N, n_hashes, L]: (group * buckets + code) << 20 | token
        nh = order.shape[1]
        tok = order & ((1 << 20) - 1)
        hb = int(g[f"codes{a}"].max().item()) // nh + 1
        hb += hb % 2
        code = (order >> 20) % hb
        for h in range(nh):                                        # ordered by code, ties by token
            key = code[:, h] * L + tok[:, h]
            assert bool((key[:, 1:] > key[:, :-1]).all())
            assert torch.equal(tok[:, h].sort(dim=1)[0], torch.arange(L).expand(N, L))
        mine = torch.empty(N, nh, L, dtype=torch.int64)
        mine.scatter_(2, tok, code + torch.arange(nh).view(1, nh, 1) * hb)
        agree = (mine.reshape(N, -1) == g[f"codes{a}"]).double().mean().item()
        assert agree >= 0.995, (a, agree)
        idx.append((tok + torch.arange(nh).view(1, nh, 1) * L).reshape(N, nh * L))
    yo = O.nlsn_forward(sd, x, scale, 8, 4, 144, 0.1, rotations=rots, indices=idx)
    assert (y - yo).abs().mean().item() <= 1e-5 and rel(y, yo) < 3e-5, rel(y, yo)
    assert (y - yref).abs().mean().item() <= 2e-3, (y - yref).abs().mean().item()


def test_nlsn_training_step_gradients_vs_reference_golden():
    """NLSN trains: forward in training mode, L1 loss, every parameter gradient of the narrow x4
    configuration (L = 480, chunk padding 96) against the REFERENCE's own autograd (g40_nlsn_grad.npz: written by
    oracle/make_goldens.py::g_nlsn_grad from the imported reference net, the oracle's autograd asserted equal), fed the LSH
    rotations the reference drew AND the token order its sort produced (the order inside a hash bucket is the sort
    implementation's; with it replayed the two runs are the same function).  Then, order computed here (own sort): the
    gradients against the oracle's fp64 autograd replaying THIS run's order.  Gate: the library's 2e-5 of a tensor's largest
    entry, ReLU decisions within rounding of zero excused by the 3x-the-fp32-reference arm."""
    from dlib.models.network_nlsn import NLSN
    from srhip.train import TrainStep, Optimizer
    scale = 4
    g = {k[len(f"x{scale}/"):]: v for k, v in load("g40_nlsn_grad").items() if k.startswith(f"x{scale}/")}
    sd = O.nlsn_init_state_dict(scale, 1, 8, 64, seed=int(g["seed"]))
    net = NLSN(upscale=scale, in_chans=1, n_resblocks=8, n_feats=64)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    x, tgt = g["x"], g["tgt"]
    N, L, nh = x.shape[0], x.shape[2] * x.shape[3], 4
    rots = [g["rot0"], g["rot1"]]
    ref_idx = [g["indices0"].long(), g["indices1"].long()]
    net.engine.rotations = [r.cuda() for r in rots]

    def sd64():
        return {k: (v.double().requires_grad_(True) if v.dtype == torch.float32 and not k.startswith(("sub_mean", "add_mean"))
                    else v) for k, v in sd.items()}

    def check(ts, ref_of, s64, what):
        worst, n = 0.0, 0
        for k in ts.fp.names:
            ref = ref_of(k).double()
            got = ts.fp.gviews[k].double().cpu()
            den = ref.abs().max().clamp_min(1e-30)
            e = ((got - ref).abs().max() / den).item()
            e32 = 0.0 if s64 is None else ((ref - s64[k].grad).abs().max() / den).item()
            worst, n = max(worst, e), n + 1
            assert e <= max(2e-5, 3.0 * e32), (what, k, e, e32)
        assert n == 50
        print(f"NLSN x{scale} training step ({what}): loss {ts.loss_values()[0]:.6f}, worst gradient error {worst:.2e}")

    # (1) the reference's order replayed
    net.engine.orders = [(i % L).view(N, nh, L).contiguous().cuda() for i in ref_idx]
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "sgd", lr=0.0, momentum=0.0, nesterov=False, wd=0.0)        # lr 0: the gradients stay readable
    ts.step(x.cuda(), tgt.cuda())
    assert abs(ts.loss_values()[0] - float(g["loss"])) <= 2e-6
    s64 = sd64()
    (O.nlsn_forward(s64, x.double(), scale, 8, 4, 144, 0.1, rotations=[r.double() for r in rots], indices=ref_idx)
     - tgt.double()).abs().mean().backward()
    check(ts, lambda k: g["grad/" + k], s64, "reference order")
    # (2) this library's own sort
    net.engine.orders = None
    net.engine.taps = taps = []
    ts.step(x.cuda(), tgt.cuda())
    idx = [((tp["order"].cpu() & ((1 << 20) - 1)) + torch.arange(nh).view(1, nh, 1) * L).reshape(N, nh * L) for tp in taps]
    assert len(idx) == 2
    s64 = sd64()
    (O.nlsn_forward(s64, x.double(), scale, 8, 4, 144, 0.1, rotations=[r.double() for r in rots], indices=idx)
     - tgt.double()).abs().mean().backward()
    check(ts, lambda k: s64[k].grad, None, "own order")


def test_main_cli_trains_nlsn(tmp_path):
    """`main.py --net_type NLSN --max_iters 20`: the registry net through ModelPlain's step (rotations drawn per call on the
    device), loss finite and falling."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "sr-caco-2_amd", "main.py"), "--net_type", "NLSN", "--method", "NLSN",
                        "--task", "super-resolution", "--scale", "4", "--n_channels", "1", "--h_size", "128", "--batch_size", "2",
                        "--max_iters", "20", "--G_optimizer_lr", "1e-4", "--NLSN_n_resblocks", "8", "--NLSN_n_feats", "64",
                        "--outd", str(tmp_path)], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    losses = [float(l.split("G_loss")[1].split()[0]) for l in p.stdout.splitlines() if "G_loss" in l]
    assert len(losses) == 2 and all(np.isfinite(losses)) and losses[1] < losses[0], losses
    assert os.path.isfile(os.path.join(str(tmp_path), "models", "20_G.pth"))


def test_nlsn_registry_default_width_vs_oracle():
    """The registry's net (32 ResBlocks, 256 features, five attention blocks with 64-dim matching embeddings, chunks of
    144) at x2 on 40 x 36 inputs, rotations drawn on the device, against the oracle replaying them and the order used."""
    from dlib.models.network_nlsn import NLSN
    sd = O.nlsn_init_state_dict(2, 1, seed=6)
    net = NLSN(upscale=2, in_chans=1)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    gen = torch.Generator().manual_seed(10)
    x = torch.rand(1, 1, 40, 36, generator=gen)
    net.engine.taps = taps = []
    with torch.no_grad():
        y = net(x.cuda()).cpu()
    N, L = 1, 40 * 36
    rots = [tp["rotations"].cpu() for tp in taps]
    idx = []
    for tp in taps:
        tok = tp["order"].cpu() & ((1 << 20) - 1)
        nh = tok.shape[1]
        idx.append((tok + torch.arange(nh).view(1, nh, 1) * L).reshape(N, nh * L))
    with torch.no_grad():
        yo = O.nlsn_forward(sd, x, 2, rotations=rots, indices=idx)
    assert len(taps) == 5 and (y - yo).abs().mean().item() <= 1e-5 and rel(y, yo) < 3e-5, rel(y, yo)


@pytest.mark.parametrize("scale", [2, 4, 8])
def test_dfcan_forward_vs_reference_golden(scale):
    """DFCAN (network_dfcan.py) against the reference's outputs of g35_dfcan.npz: GELU / sigmoid ops, the spectrum
    magnitude as a separable DFT (even and odd sizes through the quadrant swap), the channel gate, the 64 -> 64 s^2
    upsampling conv as 256-column slices."""
    from dlib.models.network_dfcan import DFCAN
    g = {k[len(f"x{scale}/"):]: v for k, v in load("g35_dfcan").items() if k.startswith(f"x{scale}/")}
    sd = O.dfcan_init_state_dict(scale, 1, seed=int(g["seed"]))
    net = DFCAN(input_shape=1, upscale=scale)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(g["x"].cuda()).cpu()
    assert (y - g["y"]).abs().mean().item() <= 1e-5 and rel(y, g["y"]) < 2e-5, rel(y, g["y"])


def test_dfcan_spectrum_magnitude_vs_torch_fft():
    """srhip_fft2_mag_pow_shift against torch.fft on NHWC data, sizes incl. odd and 256."""
    from srhip import ops
    gen = torch.Generator().manual_seed(3)
    for (B, H, W, C) in ((2, 16, 12, 64), (1, 15, 9, 64), (1, 256, 64, 64), (1, 40, 256, 8)):
        x = torch.randn(B, H, W, C, generator=gen)
        out = torch.empty(B, H, W, C, device="cuda")
        ops.fft2_mag_pow_shift(x.cuda(), out)
        xn = x.permute(0, 3, 1, 2).double()
        ref = O._dfcan_fftshift2d(torch.pow(torch.abs(torch.fft.fftn(xn, dim=(2, 3))) + 1e-8, 0.8)).permute(0, 2, 3, 1)
        assert rel(out, ref) < 2e-6, ((B, H, W, C), rel(out, ref))


@pytest.mark.parametrize("scale", [2, 4, 8])
def test_act_forward_vs_reference_golden(scale):
    """ACT (network_act.py), narrow configuration of g36_act.npz, against the reference's own output: 5 x 5 head convs as
    im2col + GEMM, tokens through srhip_unfold / srhip_fold (sizes that are not multiples of the token size leave uncovered
    pixels at zero), self- and cross-scale attention per (sample, head), RCAN groups with the channel gate, fusion blocks."""
    from dlib.models.network_act import ACT
    g = {k[len(f"x{scale}/"):]: v for k, v in load("g36_act").items() if k.startswith(f"x{scale}/")}
    cfg = dict(n_feats=16, n_resgroups=4, n_resblocks=2, reduction=4, n_heads=4, n_layers=8, n_fusionblocks=4)
    net = ACT(upscale=scale, in_chans=1, **cfg)
    sd = O.seeded_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], int(g["seed"]))
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(g["x"].cuda()).cpu()
    # (seeded random weights drive the features to |y| ~ 40: the gates are relative to the output's largest entry)
    assert (y - g["y"]).abs().mean().item() <= 1e-5 * g["y"].abs().max().item() and rel(y, g["y"]) < 3e-5, rel(y, g["y"])
    net.train()                             # the training forward (the tape graph) computes the same image
    yt = net(g["x"].cuda()).detach().cpu()
    assert rel(yt, g["y"]) < 3e-5, rel(yt, g["y"])


@pytest.mark.parametrize("M,C", [(40, 72), (37, 144), (4100, 1152), (9, 2048)])
def test_layernorm_rows_backward_kernel(M, C):
    """srhip_layernorm_rows_bwd against float64 autograd of F.layer_norm: dx, dgamma, dbeta; rows beyond one pass of the grid
    (M = 4100 > 4 x 512 blocks), widths up to the 2048 limit; twice the same bits (fixed summation order)."""
    from srhip import ops
    gen = torch.Generator().manual_seed(M + C)
    x = (torch.randn(M, C, generator=gen) * 2 + 0.5).cuda()
    dy = torch.randn(M, C, generator=gen).cuda()
    gamma, beta = (torch.rand(C, generator=gen) + 0.5).cuda(), torch.randn(C, generator=gen).cuda()
    y = ops.layernorm_rows(x, gamma, beta, torch.empty_like(x))
    outs = []
    for _ in range(2):
        dx, dg, db = torch.empty_like(x), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        ops.layernorm_rows_bwd(dy, x, gamma, dx, dg, db)
        outs.append((dx.clone(), dg.clone(), db.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    x64, g64, b64 = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y64 = torch.nn.functional.layer_norm(x64, (C,), g64, b64, 1e-5)
    assert (y.double() - y64).abs().max().item() <= 1e-5
    rx, rg, rb = torch.autograd.grad(y64, (x64, g64, b64), dy.double())
    for got, ref in zip(outs[0], (rx, rg, rb)):
        assert ((got.double() - ref).abs().max() / ref.abs().max()).item() <= 2e-6
    with pytest.raises(RuntimeError):
        ops.layernorm_rows_bwd(torch.zeros(4, 2052, device="cuda"), torch.zeros(4, 2052, device="cuda"), torch.ones(2052, device="cuda"),
                               torch.zeros(4, 2052, device="cuda"), torch.zeros(2052, device="cuda"), torch.zeros(2052, device="cuda"))


def test_act_training_step_gradients_vs_reference_golden():
    """ACT trains: forward in training mode, L1 loss, every parameter gradient of the narrow x2
    configuration against the REFERENCE's own autograd (g45_act_grad.npz, oracle/make_goldens.py::g_act_grad; tensors above
    8192 entries as two rows in full + the tensor's sum / sum of magnitudes, and the whole tensor against the oracle's fp64
    autograd) -- the attention products and the row softmax backward, F.fold / F.unfold as each other's adjoints, LayerNorm
    over 144 / 288 / 72 columns (srhip_layernorm_rows_bwd), GELU, the 5 x 5 head convs, RCAN's channel gate, the fusion
    blocks.  Gate 2e-5 of a tensor's largest entry; ReLU decisions within rounding of zero excused by the
    3x-the-fp32-oracle arm.  The same wiring with torch stand-ins for the kernels: tests/test_cpu_tape_logic.py."""
    from dlib.models.network_act import ACT
    from srhip.train import TrainStep, Optimizer
    scale = 2
    g = {k[len(f"x{scale}/"):]: v for k, v in load("g45_act_grad").items() if k.startswith(f"x{scale}/")}
    kw = dict(n_feats=16, n_resblocks=2, n_heads=4)
    net = ACT(upscale=scale, in_chans=1, n_resgroups=4, reduction=4, n_layers=8, n_fusionblocks=4, **kw)
    sd = O.seeded_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], int(g["seed"]))
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "sgd", lr=0.0, momentum=0.0, nesterov=False, wd=0.0)
    x, tgt = g["x"], g["tgt"]
    ts.step(x.cuda(), tgt.cuda())
    assert abs(ts.loss_values()[0] - float(g["loss"])) <= 2e-5 * max(1.0, float(g["loss"]))
    trainable = {k for k, p in net.named_parameters() if p.requires_grad}
    sd64 = {k: (v.double().requires_grad_(True) if k in trainable else v.double()) for k, v in sd.items()}
    (O.act_forward(sd64, x.double(), scale, **kw) - tgt.double()).abs().mean().backward()
    worst, n = ("", 0.0), 0
    for k in ts.fp.names:
        got = ts.fp.gviews[k].double().cpu()
        r64 = sd64[k].grad
        if "grad/" + k in g:
            ref = g["grad/" + k].double()
            den = ref.abs().max().clamp_min(1e-30)
            e = ((got - ref).abs().max() / den).item()
            e32 = ((ref - r64).abs().max() / den).item()
        elif "gslice/" + k in g:
            ref, sums = g["gslice/" + k].double(), g["gsum/" + k].double()
            den = sums[2].clamp_min(1e-30)
            e = ((got[:2] - ref).abs().max() / den).item()
            e32 = ((ref - r64[:2]).abs().max() / den).item()
            assert abs(got.sum().item() - sums[0].item()) <= 1e-4 * sums[1].item(), k
            assert abs(got.abs().sum().item() - sums[1].item()) <= 1e-4 * sums[1].item(), k
            e = max(e, ((got - r64).abs().max() / den).item() - e32)
        else:                                   # blocks past n_fusionblocks: the forward does not reach them
            assert float(got.abs().max()) == 0.0, k
            continue
        worst, n = max(worst, (k, e), key=lambda t: t[1]), n + 1
        assert e <= max(2e-5, 3.0 * e32), (k, e, e32)
    assert n == int(g["n_grads"])
    print(f"ACT x{scale} training step: loss {ts.loss_values()[0]:.6f}, worst gradient error {worst[1]:.2e} ({worst[0]}) of a tensor's largest entry")


def test_main_cli_trains_act(tmp_path):
    """`main.py --net_type ACT --max_iters 20`: the registry net through ModelPlain's step, loss finite and falling."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "sr-caco-2_amd", "main.py"), "--net_type", "ACT", "--method", "ACT",
                        "--task", "super-resolution", "--scale", "4", "--n_channels", "1", "--h_size", "96", "--batch_size", "2",
                        "--max_iters", "20", "--G_optimizer_lr", "1e-4", "--outd", str(tmp_path)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    losses = [float(l.split("G_loss")[1].split()[0]) for l in p.stdout.splitlines() if "G_loss" in l]
    assert len(losses) == 2 and all(np.isfinite(losses)) and losses[1] < losses[0], losses


def test_act_registry_default_width_vs_oracle():
    """The registry's net (64 features, 12 RCABs per group, 8 heads of 72: 576-dim tokens) at x2 on a 24 x 21 input."""
    from dlib.models.network_act import ACT
    net = ACT(upscale=2, in_chans=1)
    sd = O.seeded_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], 7)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    gen = torch.Generator().manual_seed(12)
    x = torch.rand(1, 1, 24, 21, generator=gen)
    with torch.no_grad():
        yo = O.act_forward(sd, x, 2)
        y = net(x.cuda()).cpu()
        net.amp = True                            # --amp: one product in the 3 x 3 convs and the Linears
        ya = net(x.cuda()).cpu()
    assert (y - yo).abs().mean().item() <= 1e-5 * yo.abs().max().item() and rel(y, yo) < 3e-5, rel(y, yo)
    assert not torch.equal(ya, y)
    mse = ((ya - yo) ** 2).mean().item() / max(1.0, yo.abs().max().item()) ** 2
    assert mse < 1e-5, mse


def test_unfold_fold_vs_torch():
    """srhip_unfold / srhip_fold against F.unfold / F.fold (channel-major token columns, overlap-add, uncovered pixels)."""
    import torch.nn.functional as F
    from srhip import ops
    gen = torch.Generator().manual_seed(4)
    for (B, H, W, C, k, s, pad) in ((2, 12, 15, 8, 3, 3, 0), (1, 14, 13, 4, 6, 3, 0), (2, 9, 10, 16, 5, 1, 2), (1, 64, 64, 32, 6, 3, 0)):
        x = torch.randn(B, C, H, W, generator=gen)
        ref = F.unfold(x, k, stride=s, padding=pad).permute(0, 2, 1)             # [B, T, C k k]
        T = ref.shape[1]
        tok = torch.empty(B * T, C * k * k, device="cuda")
        ops.unfold(x.permute(0, 2, 3, 1).contiguous().cuda(), C, k, s, pad, tok)
        assert torch.equal(tok.cpu().view(B, T, -1), ref)
        if pad == 0:
            t = torch.randn(B, T, C * k * k, generator=gen)
            fref = F.fold(t.permute(0, 2, 1), (H, W), k, stride=s).permute(0, 2, 3, 1)
            img = torch.empty(B, H, W, C, device="cuda")
            ops.fold(t.view(B * T, -1).cuda(), C, k, s, img)
            assert rel(img, fref) < 1e-6


@pytest.mark.parametrize("scale", [2, 4, 8])
def test_omnisr_forward_vs_reference_golden(scale):
    """OmniSR (network_omni_sr.py), narrow configuration of g37_omnisr.npz, against the reference's own output: MBConv with
    squeeze-excitation, window and grid attention with relative-position bias, both channel attentions, gated depthwise
    feed-forwards, ESA (stride-2 conv, 7/3 max pooling, bilinear resize); x4: a 13 x 18 input, zero-padded to the window."""
    from dlib.models.network_omni_sr import OmniSR
    g = {k[len(f"x{scale}/"):]: v for k, v in load("g37_omnisr").items() if k.startswith(f"x{scale}/")}
    net = OmniSR(input_shape=1, upscale=scale, num_feat=16, res_num=2, block_num=1)
    sd = O.seeded_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], int(g["seed"]))
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(g["x"].cuda()).cpu()
    assert y.shape == g["y"].shape
    assert (y - g["y"]).abs().mean().item() <= 1e-5 * max(1.0, g["y"].abs().max().item()) and rel(y, g["y"]) < 3e-5, rel(y, g["y"])
    if scale != 4:                          # the training forward (the tape graph; window-multiple inputs) computes the same image
        net.train()
        yt = net(g["x"].cuda()).detach().cpu()
        assert rel(yt, g["y"]) < 3e-5, rel(yt, g["y"])


def test_omnisr_backward_pieces():
    """srhip_mul, srhip_add_periodic / srhip_sum_periodic (adjoint pair), srhip_maxpool2d_bwd against torch."""
    from srhip import ops
    gen = torch.Generator().manual_seed(5)
    a, b = torch.randn(3, 37, 5, generator=gen).cuda(), torch.randn(3, 37, 5, generator=gen).cuda()
    assert torch.equal(ops.mul(a, b), a * b)
    x, v = torch.randn(7, 4, 6, 8, generator=gen).cuda(), torch.randn(4, 6, 8, generator=gen).cuda()
    ref = x + v
    assert torch.equal(ops.add_periodic(x.clone(), v), ref)
    s = ops.sum_periodic(x, torch.empty_like(v))
    assert (s - x.sum(0)).abs().max().item() <= 1e-5 and torch.equal(s, ops.sum_periodic(x, torch.empty_like(v)))
    for (H, W, k, st) in ((15, 31, 7, 3), (7, 7, 7, 3), (9, 12, 3, 2)):
        xi = torch.randn(2, H, W, 5, generator=gen).cuda()
        xr = xi.clone().requires_grad_(True)
        yr = torch.nn.functional.max_pool2d(xr.permute(0, 3, 1, 2), k, st).permute(0, 2, 3, 1)
        gr = torch.randn(yr.shape, generator=gen).cuda()
        (dref,) = torch.autograd.grad(yr, xr, gr)
        assert torch.equal(ops.maxpool2d(xi, k, st), yr.detach().contiguous())
        assert (ops.maxpool2d_bwd(xi, gr.contiguous(), k, st) - dref).abs().max().item() <= 1e-6


def test_omnisr_training_step_gradients_vs_reference_golden():
    """OmniSR trains: forward in training mode, L1 loss, every parameter gradient of the narrow x2
    configuration against the REFERENCE's own autograd (g47_omnisr_grad.npz, oracle/make_goldens.py::g_omnisr_grad) -- window
    and grid attention as batched GEMMs around the row softmax with the relative-position bias table's gradient, both channel
    attentions on L2-normalised rows with their temperatures, the depthwise convs' weight gradient as the block diagonal of
    one GEMM, squeeze-excitation, the gated feed-forwards, ESA's strided conv / max pooling / bilinear resize.  Gate 2e-5 of a
    tensor's largest entry or 3x the fp32 oracle's own distance from fp64.  The same wiring with torch stand-ins for the
    kernels: tests/test_cpu_tape_logic.py."""
    from dlib.models.network_omni_sr import OmniSR
    from srhip.train import TrainStep, Optimizer
    scale = 2
    g = {k[len(f"x{scale}/"):]: v for k, v in load("g47_omnisr_grad").items() if k.startswith(f"x{scale}/")}
    net = OmniSR(input_shape=1, upscale=scale, num_feat=16, res_num=2, block_num=1)
    sd = O.seeded_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], int(g["seed"]))
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "sgd", lr=0.0, momentum=0.0, nesterov=False, wd=0.0)
    x, tgt = g["x"], g["tgt"]
    ts.step(x.cuda(), tgt.cuda())
    assert abs(ts.loss_values()[0] - float(g["loss"])) <= 2e-5 * max(1.0, float(g["loss"]))
    trainable = {k for k, p in net.named_parameters() if p.requires_grad}
    sd64 = {k: (v.double().requires_grad_(True) if k in trainable else v) for k, v in sd.items()}
    (O.omnisr_forward(sd64, x.double(), scale, res_num=2, block_num=1) - tgt.double()).abs().mean().backward()
    worst, n = ("", 0.0), 0
    for k in ts.fp.names:
        got, ref, r64 = ts.fp.gviews[k].double().cpu(), g["grad/" + k].double(), sd64[k].grad
        den = ref.abs().max().clamp_min(1e-30)
        e, e32 = ((got - ref).abs().max() / den).item(), ((ref - r64).abs().max() / den).item()
        worst, n = max(worst, (k, e), key=lambda t: t[1]), n + 1
        assert e <= max(2e-5, 3.0 * e32), (k, e, e32)
    assert n == int(g["n_grads"])
    print(f"OmniSR x{scale} training step: loss {ts.loss_values()[0]:.6f}, worst gradient error {worst[1]:.2e} ({worst[0]})")


def test_main_cli_trains_omnisr(tmp_path):
    """`main.py --net_type OmniSR --max_iters 20`: the registry net through ModelPlain's step, loss finite and falling."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "sr-caco-2_amd", "main.py"), "--net_type", "OmniSR", "--method", "OmniSR",
                        "--task", "super-resolution", "--scale", "4", "--n_channels", "1", "--h_size", "128", "--batch_size", "2",
                        "--max_iters", "20", "--G_optimizer_lr", "1e-4", "--outd", str(tmp_path)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    losses = [float(l.split("G_loss")[1].split()[0]) for l in p.stdout.splitlines() if "G_loss" in l]
    assert len(losses) == 2 and all(np.isfinite(losses)) and losses[1] < losses[0], losses


def test_omnisr_registry_default_width_vs_oracle():
    """The registry's net (64 features, 5 groups of 4 omni blocks) at x2 on a 24 x 32 input."""
    from dlib.models.network_omni_sr import OmniSR
    net = OmniSR(input_shape=1, upscale=2)
    sd = O.seeded_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], 8)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    gen = torch.Generator().manual_seed(13)
    x = torch.rand(1, 1, 24, 32, generator=gen)
    with torch.no_grad():
        yo = O.omnisr_forward(sd, x, 2)
        y = net(x.cuda()).cpu()
    assert (y - yo).abs().mean().item() <= 1e-5 * max(1.0, yo.abs().max().item()) and rel(y, yo) < 5e-5, rel(y, yo)


GRL_KW = dict(in_chans=1, window_size=8, mlp_ratio=2, qkv_proj_type="linear", anchor_proj_type="avgpool",
              anchor_window_down_factor=2, out_proj_type="linear", conv_type="1conv", upsampler="pixelshuffle",
              local_connection=True)


def test_cosine_window_attention_vs_float64():
    """srhip_cosine_window_attention + srhip_cpb_bias against a float64 statement of Attention.attn under AffineTransform
    (network_grl.py:296-319,338-355): shifted 8x8 windows (roll, partition, region mask, reverse, roll back) and the two
    passes of the anchored stripe attention (4x4 anchor windows against 8x8 stripes), heads of 30 and of 6 channels."""
    from srhip import ops
    gen = torch.Generator().manual_seed(77)
    for B, H, W, heads, d in ((2, 16, 24, 3, 30), (1, 24, 16, 3, 6), (1, 8, 8, 2, 45)):
        C = heads * d
        qkv = torch.randn(B, H, W, 3 * C + 8, generator=gen)
        table = torch.randn(225, heads, generator=gen)
        ls = torch.tensor([2.3, 5.0, 1.0][:heads])
        buf = O.grl_buffers((H, W))

        def ref(q, k, v, index, tab, mask):
            at = torch.nn.functional.normalize(q, dim=-1) @ torch.nn.functional.normalize(k, dim=-1).transpose(-2, -1)
            at = at * ls.double().clamp(max=np.log(100.0)).exp().view(-1, 1, 1)
            N1, N2 = index.shape
            at = at + 16 * torch.sigmoid(tab.double()[index.view(-1)].view(N1, N2, -1).permute(2, 0, 1))
            if mask is not None:
                nW = mask.shape[0]
                at = (at.view(-1, nW, heads, N1, N2) + mask.double()[None, :, None]).view(-1, heads, N1, N2)
            return at.softmax(-1) @ v
        for shift in (4, 0):
            t = qkv.double()[..., :3 * C]
            if shift:
                t = torch.roll(t, (-shift, -shift), (1, 2))
            w = O._grl_partition(t, [8, 8]).view(-1, 64, 3, heads, d).permute(2, 0, 3, 1, 4)
            o = ref(w[0], w[1], w[2], buf["index_w"], table, buf["mask_w"] if shift else None)
            o = O._grl_reverse(o.transpose(1, 2).reshape(-1, 8, 8, C), [8, 8], (H, W))
            if shift:
                o = torch.roll(o, (shift, shift), (1, 2))
            qd = qkv.cuda()
            out = torch.full((B, H, W, C + 4), float("nan"), device="cuda")
            bT = ops.cpb_bias(table.cuda(), buf["index_w"].cuda(), heads)
            ops.cosine_window_attention(qd[..., :C], (8, 8), qd[..., C:2 * C], qd[..., 2 * C:3 * C], (8, 8), ls.cuda(), bT,
                                        out[..., :C], heads, d, shift)
            assert rel(out[..., :C], o) < 5e-6, (B, H, W, d, shift, rel(out[..., :C], o))
            assert torch.isnan(out[..., C:]).all()
        # anchors
        an = torch.randn(B, H // 2, W // 2, C, generator=gen)
        tab_s = torch.randn(121, heads, generator=gen)
        ad = an.double()
        a_w = O._grl_partition(ad, [4, 4]).view(-1, 16, heads, d).permute(0, 2, 1, 3)
        w = O._grl_partition(qkv.double()[..., :3 * C], [8, 8]).view(-1, 64, 3, heads, d).permute(2, 0, 3, 1, 4)
        x1 = ref(a_w, w[1], w[2], buf["index_sh_a2w"], tab_s, None)                  # [nW, heads, 16, d]
        x2 = ref(w[0], a_w, x1, buf["index_sh_w2a"], tab_s, None)
        x2 = O._grl_reverse(x2.transpose(1, 2).reshape(-1, 8, 8, C), [8, 8], (H, W))
        qd, and_ = qkv.cuda(), an.cuda()
        xa = torch.empty_like(and_)
        b1 = ops.cpb_bias(tab_s.cuda(), buf["index_sh_a2w"].cuda(), heads)
        b2 = ops.cpb_bias(tab_s.cuda(), buf["index_sh_w2a"].cuda(), heads)
        ops.cosine_window_attention(and_, (4, 4), qd[..., C:2 * C], qd[..., 2 * C:3 * C], (8, 8), ls.cuda(), b1, xa, heads, d)
        x1_img = O._grl_reverse(x1.transpose(1, 2).reshape(-1, 4, 4, C), [4, 4], (H // 2, W // 2))
        assert rel(xa, x1_img) < 5e-6
        out = torch.empty(B, H, W, C, device="cuda")
        ops.cosine_window_attention(qd[..., :C], (8, 8), and_, xa, (4, 4), ls.cuda(), b2, out, heads, d)
        assert rel(out, x2) < 5e-6
    p = ops.avgpool2d(qkv.cuda().contiguous(), 2)
    assert rel(p, torch.nn.functional.avg_pool2d(qkv.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)) < 1e-6


@pytest.mark.parametrize("scale", [2, 4, 8])
def test_grl_forward_vs_reference_golden(scale):
    """GRL (network_grl.py), narrow configuration of g38_grl.npz, against the reference's own output: shifted / plain cosine
    window attention with the CPB-MLP bias, anchored 'H' / 'W' stripe attention, the conv + channel-attention branch (9
    channels, zero-padded to 12), post-norm residuals, the pixel-shuffle tail; x4: a 13 x 18 input, reflect-padded; x2: a
    16 x 24 input against masks registered for 16 x 16.  Block outputs are checked one by one against the oracle's."""
    from dlib.models.network_grl import GRL
    g = {k[len(f"x{scale}/"):]: v for k, v in load("g38_grl").items() if k.startswith(f"x{scale}/")}
    net = GRL(upscale=scale, img_size=16, depths=[2, 2], embed_dim=36, num_heads_window=[3, 3], num_heads_stripe=[3, 3],
              drop_path_rate=0.0, **GRL_KW)          # (0: the training-mode forward below is compared with the evaluation image)
    sd = O.grl_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], int(g["seed"]), 16)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    taps_o = {}
    O.grl_forward(sd, g["x"], scale, depths=(2, 2), taps=taps_o)
    net.engine.taps = {}
    with torch.no_grad():
        y = net(g["x"].cuda()).cpu()
    taps = net.engine.taps
    net.engine.taps = None
    for k, v in taps.items():
        if k in taps_o and taps_o[k].dim() == 3:
            assert rel(v, taps_o[k]) < 2e-5, (k, rel(v, taps_o[k]))
    assert y.shape == g["y"].shape
    assert (y - g["y"]).abs().mean().item() <= 1e-5 * max(1.0, g["y"].abs().max().item()) and rel(y, g["y"]) < 3e-5, rel(y, g["y"])
    if scale != 4:                          # the training forward (the tape graph; window-multiple inputs) computes the same image
        net.train()
        yt = net(g["x"].cuda()).detach().cpu()
        assert rel(yt, g["y"]) < 3e-5, rel(yt, g["y"])


def test_grl_training_step_gradients_vs_reference_golden():
    """GRL trains: forward in training mode, L1 loss, every parameter gradient of the narrow x2
    configuration against the REFERENCE's own autograd (g48_grl_grad.npz, oracle/make_goldens.py::g_grl_grad) -- cosine window
    attention under the shift (bias + mask per window as a periodic addend), the anchored stripe attention both ways, the
    logit scales (one over the clamp: zero gradient), the CPB MLPs by hand, the C/4-channel convs as im2col + GEMM, the
    post-norm residuals.  Gate 2e-5 of a tensor's largest entry or 3x the fp32 oracle's own distance from fp64 (the logit
    scales: sums of ~1e-6 terms, 2e-3).  The same wiring with torch stand-ins for the kernels: tests/test_cpu_tape_logic.py."""
    from dlib.models.network_grl import GRL
    from srhip.train import TrainStep, Optimizer
    scale = 2
    g = {k[len(f"x{scale}/"):]: v for k, v in load("g48_grl_grad").items() if k.startswith(f"x{scale}/")}
    net = GRL(upscale=scale, img_size=16, in_chans=1, window_size=8, mlp_ratio=2, qkv_proj_type="linear",
              anchor_proj_type="avgpool", anchor_window_down_factor=2, out_proj_type="linear", conv_type="1conv",
              upsampler="pixelshuffle", local_connection=True, depths=[2, 2], embed_dim=36, num_heads_window=[3, 3],
              num_heads_stripe=[3, 3], drop_path_rate=0.0)
    sd = O.grl_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], int(g["seed"]), 16)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "sgd", lr=0.0, momentum=0.0, nesterov=False, wd=0.0)
    x, tgt = g["x"], g["tgt"]
    ts.step(x.cuda(), tgt.cuda())
    assert abs(ts.loss_values()[0] - float(g["loss"])) <= 2e-5 * max(1.0, float(g["loss"]))
    trainable = {k for k, p in net.named_parameters() if p.requires_grad}
    sd64 = {k: (v.double().requires_grad_(True) if k in trainable else v) for k, v in sd.items()}
    (O.grl_forward(sd64, x.double(), scale, depths=(2, 2)) - tgt.double()).abs().mean().backward()
    worst, n = ("", 0.0), 0
    for k in ts.fp.names:
        got, r64 = ts.fp.gviews[k].double().cpu(), sd64[k].grad
        if "grad/" + k in g:
            ref = g["grad/" + k].double()
            den = ref.abs().max().clamp_min(1e-30)
            e, e32 = ((got - ref).abs().max() / den).item(), ((ref - r64).abs().max() / den).item()
        else:
            ref, sums = g["gslice/" + k].double(), g["gsum/" + k].double()
            den = sums[2].clamp_min(1e-30)
            e, e32 = ((got[:2] - ref).abs().max() / den).item(), ((ref - r64[:2]).abs().max() / den).item()
            e = max(e, ((got - r64).abs().max() / den).item() - e32)
        worst, n = max(worst, (k, e), key=lambda t: t[1]), n + 1
        assert e <= max(2e-3 if k.endswith("logit_scale") else 2e-5, 3.0 * e32), (k, e, e32)
    assert n == int(g["n_grads"])
    print(f"GRL x{scale} training step: loss {ts.loss_values()[0]:.6f}, worst gradient error {worst[1]:.2e} ({worst[0]})")


def test_tape_training_steps_replay_from_a_hipgraph():
    """TrainStep.step_graph on the tape graphs written this round (OmniSR: window / grid / channel attention, index_add_ of the
    bias table; GRL: shift masks cached per input size): five steps from the same weights, eager against replayed -- the same
    loss trajectory (the bias tables' gradients go through atomics: 1e-5 relative)."""
    from dlib.models.network_omni_sr import OmniSR
    from dlib.models.network_grl import GRL
    from srhip.train import TrainStep, Optimizer
    gen = torch.Generator().manual_seed(3)
    x, tg = torch.rand(2, 1, 16, 32, generator=gen).cuda(), torch.rand(2, 1, 32, 64, generator=gen).cuda()

    def make(kind):
        torch.manual_seed(11)
        if kind == "omnisr":
            return OmniSR(input_shape=1, upscale=2, num_feat=16, res_num=2, block_num=1)
        return GRL(upscale=2, img_size=16, in_chans=1, window_size=8, mlp_ratio=2, qkv_proj_type="linear", anchor_proj_type="avgpool",
                   anchor_window_down_factor=2, out_proj_type="linear", conv_type="1conv", upsampler="pixelshuffle",
                   local_connection=True, depths=[2, 2], embed_dim=36, num_heads_window=[3, 3], num_heads_stripe=[3, 3],
              drop_path_rate=0.0)
    for kind in ("omnisr", "grl"):
        runs = []
        for mode in ("eager", "graph"):
            net = make(kind).cuda().train()
            ts = TrainStep(net, [("l1", 1.0)])
            ts.opt = Optimizer(ts.fp, "sgd", lr=0.01, momentum=0.9, nesterov=True, wd=0.0)
            losses = []
            for _ in range(5):
                (ts.step if mode == "eager" else ts.step_graph)(x, tg)
                losses.append(ts.loss_values()[0])
            runs.append(losses)
        assert runs[0][-1] < runs[0][0], (kind, runs[0])
        for a, b in zip(*runs):
            assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), (kind, runs)


def test_main_cli_trains_grl(tmp_path):
    """`main.py --net_type GRL --max_iters 20`: the registry net (40 blocks) through ModelPlain's step, loss finite and falling."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "sr-caco-2_amd", "main.py"), "--net_type", "GRL", "--method", "GRL",
                        "--task", "super-resolution", "--scale", "4", "--n_channels", "1", "--h_size", "128", "--batch_size", "2",
                        "--max_iters", "20", "--G_optimizer_lr", "1e-4", "--outd", str(tmp_path)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    losses = [float(l.split("G_loss")[1].split()[0]) for l in p.stdout.splitlines() if "G_loss" in l]
    assert len(losses) == 2 and all(np.isfinite(losses)) and losses[1] < losses[0], losses


def test_grl_registry_default_width_vs_oracle():
    """The registry's width (180 channels, heads of 30, the 45 -> 48 padded local branch), two stages of two blocks, at x2 on a
    24 x 32 input."""
    from dlib.models.network_grl import GRL
    net = GRL(upscale=2, img_size=64, depths=[2, 2], embed_dim=180, num_heads_window=[3, 3], num_heads_stripe=[3, 3], **GRL_KW)
    sd = O.grl_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], 9, 64)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    gen = torch.Generator().manual_seed(14)
    x = torch.rand(2, 1, 24, 32, generator=gen)
    with torch.no_grad():
        yo = O.grl_forward(sd, x, 2, depths=(2, 2))
        y = net(x.cuda()).cpu()
    assert (y - yo).abs().mean().item() <= 1e-5 * max(1.0, yo.abs().max().item()) and rel(y, yo) < 5e-5, rel(y, yo)


def test_grl_registry_net_vs_oracle_and_amp():
    """The registry's net itself (utils_init_default_args.py:166-189: 180 channels, depths 4+4+8+8+8+4+4 = 40 blocks, 3 + 3
    heads, anchors pooled by 2) built through define_G at x8, on 40 x 32 inputs (other than the 64 x 64 its masks were
    registered for) against the oracle; and --amp (one bf16 product in its Linears and convs) within the PSNR gate."""
    from types import SimpleNamespace
    from dlib.models.select_network import define_G
    from dlib.utils.utils_init_default_args import init_net_g
    from dlib.utils import constants
    opt = init_net_g({'net_type': constants.GRL}, {'scale': 8, 'n_channels': 1, 'h_size': 512})
    net = define_G(SimpleNamespace(netG=opt))
    assert len(net.layers) == 7 and sum(len(s.blocks) for s in net.layers) == 40 and net.input_resolution == (64, 64)
    sd = O.grl_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], 21, 64)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    gen = torch.Generator().manual_seed(15)
    x = torch.rand(2, 1, 40, 32, generator=gen)
    with torch.no_grad():
        yo = O.grl_forward(sd, x, 8)
        y = net(x.cuda()).cpu()
        net.amp = True
        ya = net(x.cuda()).cpu()
    assert y.shape == (2, 1, 320, 256)
    assert (y - yo).abs().mean().item() <= 1e-5 * max(1.0, yo.abs().max().item()) and rel(y, yo) < 1e-4, rel(y, yo)
    assert not torch.equal(ya, y)                                          # the reduced-precision kernels did run
    mse = ((ya - yo) ** 2).mean().item() / max(1.0, yo.abs().max().item()) ** 2
    assert mse < 1e-5, mse


def test_dfcan_training_step_gradients_vs_reference_golden():
    """DFCAN trains: forward in training mode, L1 loss, every parameter gradient of the registry's net at
    x2 against the REFERENCE's own autograd (g41_dfcan_grad.npz, oracle/make_goldens.py::g_dfcan_grad; the 64 x 64 x 3 x 3
    weights as two output channels in full + the tensor's sum / sum of magnitudes) -- the GELU / sigmoid backward, the channel
    gate's two Linears, and the spectrum magnitude (|FFT2|^0.8 under the quadrant swap) through stock torch.fft.  Gate 2e-5 of
    a tensor's largest entry; ReLU decisions within rounding of zero excused by the 3x-the-fp32-oracle arm."""
    from dlib.models.network_dfcan import DFCAN
    from srhip.train import TrainStep, Optimizer
    scale = 2
    g = {k[len(f"x{scale}/"):]: v for k, v in load("g41_dfcan_grad").items() if k.startswith(f"x{scale}/")}
    sd = O.dfcan_init_state_dict(scale, 1, seed=int(g["seed"]))
    net = DFCAN(input_shape=1, upscale=scale)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    ts = TrainStep(net, [("l1", 1.0)])
    ts.opt = Optimizer(ts.fp, "sgd", lr=0.0, momentum=0.0, nesterov=False, wd=0.0)
    x, tgt = g["x"], g["tgt"]
    ts.step(x.cuda(), tgt.cuda())
    assert abs(ts.loss_values()[0] - float(g["loss"])) <= 2e-6
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    (O.dfcan_forward(sd64, x.double(), scale) - tgt.double()).abs().mean().backward()
    worst, n = 0.0, 0
    for k in ts.fp.names:
        got = ts.fp.gviews[k].double().cpu()
        r64 = sd64[k].grad
        if "grad/" + k in g:
            ref = g["grad/" + k].double()
            den = ref.abs().max().clamp_min(1e-30)
            e = ((got - ref).abs().max() / den).item()
            e32 = ((ref - r64).abs().max() / den).item()
        else:
            ref, sums = g["gslice/" + k].double(), g["gsum/" + k].double()
            den = sums[2].clamp_min(1e-30)
            e = ((got[:2] - ref).abs().max() / den).item()
            e32 = ((ref - r64[:2]).abs().max() / den).item()
            assert abs(got.sum().item() - sums[0].item()) <= 1e-4 * sums[1].item(), k
            assert abs(got.abs().sum().item() - sums[1].item()) <= 1e-4 * sums[1].item(), k
            # and the whole tensor against the fp64 oracle (the slice pins the oracle to the reference)
            e = max(e, ((got - r64).abs().max() / den).item() - e32)
        worst, n = max(worst, e), n + 1
        assert e <= max(2e-5, 3.0 * e32), (k, e, e32)
    assert n == 166
    print(f"DFCAN x{scale} training step: loss {ts.loss_values()[0]:.6f}, worst gradient error {worst:.2e} of a tensor's largest entry")


def test_main_cli_trains_dfcan(tmp_path):
    """`main.py --net_type DFCAN --max_iters 20`: the registry net through ModelPlain's step, loss finite and falling."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "sr-caco-2_amd", "main.py"), "--net_type", "DFCAN", "--method", "DFCAN",
                        "--task", "super-resolution", "--scale", "4", "--n_channels", "1", "--h_size", "128", "--batch_size", "2",
                        "--max_iters", "20", "--G_optimizer_lr", "1e-4", "--outd", str(tmp_path)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    losses = [float(l.split("G_loss")[1].split()[0]) for l in p.stdout.splitlines() if "G_loss" in l]
    assert len(losses) == 2 and all(np.isfinite(losses)) and losses[1] < losses[0], losses
