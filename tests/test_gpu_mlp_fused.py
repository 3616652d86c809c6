"""The MLP half of a Swin block as one kernel per direction (srhip_mlp_fwd_bx3 / srhip_mlp_bwd_bx3,
mlp_fused.hip) against a float64 statement of Mlp.forward + residual and its autograd
(dlib/models/network_swinir.py:28-45,335-337), and against the separate Linear launches it replaces."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

G = torch.Generator().manual_seed(97531)


def rnd(*shape, scale=1.0):
    return torch.randn(*shape, generator=G) * scale


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def ops():
    from srhip import ops as o
    return o


def _problem(ops, M, C, hidden, nsamp):
    x = rnd(M, C) * 1.5 + rnd(M, 1)
    w1, b1 = rnd(hidden, C, scale=0.1), rnd(hidden, scale=0.3)
    w2, b2 = rnd(C, hidden, scale=0.1), rnd(C, scale=0.3)
    gamma, beta = 1 + rnd(C, scale=0.2), rnd(C, scale=0.2)
    s = torch.rand(nsamp, generator=G) + 0.5 if nsamp else None
    dev = {k: v.cuda() for k, v in dict(x=x, w1=w1, b1=b1, w2=w2, b2=b2, gamma=gamma, beta=beta).items()}
    dev["s"] = None if s is None else s.cuda()
    hp = ops.mlp_hidden_padded(hidden)
    P = {k: ops.Bx3(*shape, "cuda") for k, shape in dict(m1=(hp, C), m2=(C, hp), m2T=(hp, C), m1T=(C, hp),
                                                         w1=(hidden, C), w2=(C, hidden), w1T=(C, hidden),
                                                         w2T=(hidden, C)).items()}
    b1f = torch.empty(hidden, device="cuda")
    tb = ops.PrepTable()
    tb.mlp_planes(dev["w1"], P["m1"], hidden, "rows", gamma=dev["gamma"])
    tb.mlp_planes(dev["w2"], P["m2"], hidden, "k")
    tb.mlp_planes(dev["w2"], P["m2T"], hidden, "rowsT")
    tb.mlp_planes(dev["w1"], P["m1T"], hidden, "kT", gamma=dev["gamma"])
    tb.linear(dev["w1"], P["w1"], gamma=dev["gamma"])
    tb.linear(dev["w1"], P["w1T"], gamma=dev["gamma"], transpose=True)
    tb.linear(dev["w2"], P["w2"])
    tb.linear(dev["w2"], P["w2T"], transpose=True)
    tb.fold_bias(dev["w1"], dev["b1"], dev["beta"], b1f)
    tb.build("cuda").run()
    st = torch.empty(M, 2, device="cuda")
    ops.layernorm_fwd(dev["x"], st)
    return dict(x=x, w1=w1, b1=b1, w2=w2, b2=b2, gamma=gamma, beta=beta, s=s), dev, P, b1f, st


def _reference(cpu, M, rows_per_scale, dy=None):
    """float64: out, h, dx, dh, gh"""
    d = {k: (None if v is None else v.double()) for k, v in cpu.items()}
    x = d["x"].clone().requires_grad_(True)
    xn = F.layer_norm(x, (x.shape[1],), d["gamma"], d["beta"], 1e-5)
    h = xn @ d["w1"].t() + d["b1"]
    h.retain_grad()
    gh = F.gelu(h)
    y = gh @ d["w2"].t() + d["b2"]
    if d["s"] is not None:
        y = y * d["s"].repeat_interleave(rows_per_scale)[:M, None]
    out = x + y
    if dy is None:
        return out.detach(), h.detach()
    out.backward(dy.double())
    return out.detach(), h.detach(), x.grad, h.grad, gh.detach()


@pytest.mark.parametrize("M,C,hidden,nsamp", [(4096, 180, 360, 0), (1000, 180, 360, 4), (64, 180, 360, 1),
                                              (333, 96, 256, 3), (2048, 192, 384, 2), (515, 120, 200, 0)])
def test_mlp_fused_forward_and_backward(ops, M, C, hidden, nsamp):
    cpu, dev, P, b1f, st = _problem(ops, M, C, hidden, nsamp)
    rps = -(-M // nsamp) if nsamp else 1
    dy = rnd(M, C)
    out_ref, h_ref, dx_ref, dh_ref, gh_ref = _reference(cpu, M, rps, dy)
    out = torch.full((M, C), float("nan"), device="cuda")
    h = torch.full((M, hidden), float("nan"), device="cuda")
    st_out = torch.empty(M, 2, device="cuda")
    ops.mlp_fwd(dev["x"], st, P["m1"], b1f, P["m2"], dev["b2"], out, h=h, rowscale=dev["s"], rows_per_scale=rps,
                stats_out=st_out)
    assert relerr(h, h_ref) < 2e-6
    assert relerr(out, out_ref) < 2e-6
    mean, var = out_ref.mean(1), out_ref.var(1, unbiased=False)
    assert relerr(st_out[:, 0], mean) < 1e-5 and relerr(st_out[:, 1], (var + 1e-5).rsqrt()) < 1e-5
    # inference form: no h
    out2 = torch.empty_like(out)
    ops.mlp_fwd(dev["x"], st, P["m1"], b1f, P["m2"], dev["b2"], out2, rowscale=dev["s"], rows_per_scale=rps)
    assert torch.equal(out2, out)
    # the launches it replaces
    h_u = torch.empty_like(h)
    out_u = torch.empty_like(out)
    ops.gemm_nt(dev["x"], P["w1"], b1f, out=h_u, a_mode=1, ln_stats=st)
    ops.gemm_nt(h_u, P["w2"], dev["b2"], out=out_u, a_mode=2, epi=2, R=dev["x"], rowscale=dev["s"], rows_per_scale=rps)
    assert relerr(h, h_u) < 1e-6 and relerr(out, out_u) < 1e-6

    # ---- backward
    dyd = dy.cuda()
    dh = torch.full((M, hidden), float("nan"), device="cuda")
    gh = torch.full((M, hidden), float("nan"), device="cuda")
    dx = torch.full((M, C), float("nan"), device="cuda")
    ops.mlp_bwd(dyd, P["m2T"], P["m1T"], h, dh, gh, dev["x"], st, dx, rowscale=dev["s"], rows_per_scale=rps)
    assert relerr(gh, gh_ref) < 2e-6
    assert relerr(dh, dh_ref) < 5e-6
    assert relerr(dx, dx_ref) < 5e-6
    dh_u, gh_u, dx_u = torch.empty_like(dh), torch.empty_like(gh), torch.empty_like(dx)
    ops.gemm_nt(dyd, P["w2T"], None, out=dh_u, epi=3, R=h, rowscale=dev["s"], rows_per_scale=rps, aux=gh_u)
    ops.gemm_nt_lnbwd(dh_u, P["w1T"], dev["x"], st, dyd, dx_u)
    assert relerr(dh, dh_u) < 1e-6 and relerr(gh, gh_u) < 1e-6 and relerr(dx, dx_u) < 2e-6


def test_mlp_fused_rejects_shapes_it_does_not_take(ops):
    assert not ops.mlp_fusable(180, 180) and not ops.mlp_fusable(256, 360) and ops.mlp_fusable(180, 360)
    x = torch.zeros(64, 180, device="cuda")
    st = torch.zeros(64, 2, device="cuda")
    bad = ops.Bx3(ops.mlp_hidden_padded(720), 180, "cuda")
    bad2 = ops.Bx3(180, ops.mlp_hidden_padded(720), "cuda")
    with pytest.raises(ops.SrhipError):
        ops.mlp_fwd(x, st, bad, torch.zeros(720, device="cuda"), bad2, torch.zeros(180, device="cuda"),
                    torch.empty_like(x))
