"""The MLP half of a Swin block as one kernel per direction on the Linear GEMMs' two-plane fp16 operands
(srhip_mlp_fwd_f16x2 / srhip_mlp_bwd_f16x2, mlp_f16.hip) against a float64 statement of Mlp.forward + residual and
its autograd (dlib/models/network_swinir.py:28-45,335-337), and against the separate Linear launches it replaces."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

G = torch.Generator().manual_seed(24680)


def rnd(*shape, scale=1.0):
    return torch.randn(*shape, generator=G) * scale


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def ops():
    from srhip import ops as o
    return o


def _problem(ops, M, C, hidden, nsamp, wscale=0.1):
    x = rnd(M, C) * 1.5 + rnd(M, 1)
    w1, b1 = rnd(hidden, C, scale=wscale), rnd(hidden, scale=0.3)
    w2, b2 = rnd(C, hidden, scale=wscale), rnd(C, scale=0.3)
    gamma, beta = 1 + rnd(C, scale=0.2), rnd(C, scale=0.2)
    s = torch.rand(nsamp, generator=G) + 0.5 if nsamp else None
    dev = {k: v.cuda() for k, v in dict(x=x, w1=w1, b1=b1, w2=w2, b2=b2, gamma=gamma, beta=beta).items()}
    dev["s"] = None if s is None else s.cuda()
    P = {k: ops.Bx3(*shape, "cuda") for k, shape in dict(w1=(hidden, C), w2=(C, hidden), w1T=(C, hidden),
                                                         w2T=(hidden, C)).items()}
    b1f = torch.empty(hidden, device="cuda")
    tb = ops.PrepTable()
    tb.linear(dev["w1"], P["w1"], gamma=dev["gamma"], f16=True)
    tb.linear(dev["w1"], P["w1T"], gamma=dev["gamma"], transpose=True, f16=True)
    tb.linear(dev["w2"], P["w2"], f16=True)
    tb.linear(dev["w2"], P["w2T"], transpose=True, f16=True)
    tb.fold_bias(dev["w1"], dev["b1"], dev["beta"], b1f)
    tb.build("cuda").run()
    st = torch.empty(M, 2, device="cuda")
    ops.layernorm_fwd(dev["x"], st)
    return dict(x=x, w1=w1, b1=b1, w2=w2, b2=b2, gamma=gamma, beta=beta, s=s), dev, P, b1f, st


def _reference(cpu, M, rows_per_scale, dy):
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
    out.backward(dy.double())
    return out.detach(), h.detach(), x.grad, h.grad, gh.detach()


SHAPES = [(4096, 180, 360, 0), (1000, 180, 360, 4), (64, 180, 360, 1), (333, 96, 256, 3), (2048, 192, 384, 2),
          (515, 120, 200, 0), (200, 60, 120, 2), (130, 180, 180, 0), (77, 64, 196, 1)]


@pytest.mark.parametrize("M,C,hidden,nsamp", SHAPES)
def test_mlp_f16_forward_and_backward(ops, M, C, hidden, nsamp):
    cpu, dev, P, b1f, st = _problem(ops, M, C, hidden, nsamp)
    rps = -(-M // nsamp) if nsamp else 1
    dy = rnd(M, C)
    out_ref, h_ref, dx_ref, dh_ref, gh_ref = _reference(cpu, M, rps, dy)
    out = torch.full((M, C), float("nan"), device="cuda")
    h = torch.full((M, hidden), float("nan"), device="cuda")
    st_out = torch.empty(M, 2, device="cuda")
    ops.mlp_fwd_f16(dev["x"], st, P["w1"], b1f, P["w2"], dev["b2"], out, h=h, rowscale=dev["s"], rows_per_scale=rps,
                    stats_out=st_out)
    assert relerr(h, h_ref) < 2e-6
    assert relerr(out, out_ref) < 2e-6
    mean, var = out_ref.mean(1), out_ref.var(1, unbiased=False)
    assert relerr(st_out[:, 0], mean) < 1e-5 and relerr(st_out[:, 1], (var + 1e-5).rsqrt()) < 1e-5
    # inference form: no h, no statistics
    out2 = torch.empty_like(out)
    ops.mlp_fwd_f16(dev["x"], st, P["w1"], b1f, P["w2"], dev["b2"], out2, rowscale=dev["s"], rows_per_scale=rps)
    assert torch.equal(out2, out)
    # the launches it replaces (exact-f32 MFMA: the operand planes of odd shapes are not what srhip_gemm_nt_f16x2 takes)
    h_u = torch.empty_like(h)
    out_u = torch.empty_like(out)
    w1f = (dev["w1"] * dev["gamma"][None, :]).contiguous()
    ops.gemm_nt(dev["x"], w1f, b1f, out=h_u, a_mode=1, ln_stats=st)
    ops.gemm_nt(h_u, dev["w2"], dev["b2"], out=out_u, a_mode=2, epi=2, R=dev["x"], rowscale=dev["s"], rows_per_scale=rps)
    assert relerr(h, h_u) < 2e-6 and relerr(out, out_u) < 2e-6

    # ---- backward
    dyd = dy.cuda()
    dh = torch.full((M, hidden), float("nan"), device="cuda")
    gh = torch.full((M, hidden), float("nan"), device="cuda")
    dx = torch.full((M, C), float("nan"), device="cuda")
    ops.mlp_bwd_f16(dyd, P["w2T"], P["w1T"], h, dh, gh, dev["x"], st, dx, rowscale=dev["s"], rows_per_scale=rps)
    assert relerr(gh, gh_ref) < 2e-6
    assert relerr(dh, dh_ref) < 5e-6
    assert relerr(dx, dx_ref) < 5e-6


@pytest.mark.parametrize("M,C,hidden,nsamp", [(4096, 180, 360, 4), (1000, 180, 360, 0), (333, 96, 256, 3), (77, 64, 196, 1),
                                              (2048, 192, 384, 2)])
def test_mlp_f16_backward_with_chained_product(ops, M, C, hidden, nsamp):
    """srhip_mlp_bwd_chain_f16x2: the proj Linear's data gradient out3 = s3 * (dx @ W3^T) behind the MLP backward in
    the same kernel, against float64 and against the unchained kernel (dx, dh, gh must not change at all)."""
    cpu, dev, P, b1f, st = _problem(ops, M, C, hidden, nsamp)
    rps = -(-M // nsamp) if nsamp else 1
    dy = rnd(M, C)
    _, h_ref, dx_ref, _, _ = _reference(cpu, M, rps, dy)
    w3 = rnd(C, C, scale=0.1)
    s3 = torch.rand(nsamp, generator=G) + 0.5 if nsamp else None
    P3 = ops.Bx3(C, C, "cuda")
    tb = ops.PrepTable()
    tb.linear(w3.cuda(), P3, f16=True)
    tb.build("cuda").run()
    h = h_ref.float().cuda()
    dyd = dy.cuda()
    out = {}
    for chained in (False, True):
        dh = torch.full((M, hidden), float("nan"), device="cuda")
        gh = torch.full((M, hidden), float("nan"), device="cuda")
        dx = torch.full((M, C), float("nan"), device="cuda")
        o3 = torch.full((M, C), float("nan"), device="cuda")
        ops.mlp_bwd_f16(dyd, P["w2T"], P["w1T"], h, dh, gh, dev["x"], st, dx, rowscale=dev["s"], rows_per_scale=rps,
                        chain=(P3, o3, None if s3 is None else s3.cuda()) if chained else None)
        out[chained] = (dh, gh, dx, o3)
    for a, b in zip(out[False][:3], out[True][:3]):
        assert torch.equal(a, b)
    ref3 = dx_ref @ w3.double().t()
    if s3 is not None:
        ref3 = ref3 * s3.double().repeat_interleave(rps)[:M, None]
    assert relerr(out[True][3], ref3) < 5e-6
    # the launch it replaces
    o3u = torch.empty(M, C, device="cuda")
    ops.gemm_nt(out[True][2], w3.cuda(), None, out=o3u, epi=2, rowscale=None if s3 is None else s3.cuda(), rows_per_scale=rps)
    assert relerr(out[True][3], o3u) < 2e-6


@pytest.mark.parametrize("M,C,hidden,nsamp,chain", [(4096, 180, 360, 4, True), (1000, 180, 360, 0, True), (333, 96, 256, 3, False),
                                                    (77, 64, 196, 1, True), (2048, 192, 384, 2, False)])
def test_mlp_f16_backward_with_front_product(ops, M, C, hidden, nsamp, chain):
    """srhip_mlp_bwd_front_chain_f16x2: the incoming gradient dy = res0 + LayerNorm_backward(X0 @ W0^T; x0, stats0) is
    computed in the kernel (the qkv Linear's data gradient of the Swin block behind), against float64 and against the
    two launches it replaces (everything behind dy must then be bit-identical: dy itself is compared to tolerance)."""
    cpu, dev, P, b1f, st = _problem(ops, M, C, hidden, nsamp)
    rps = -(-M // nsamp) if nsamp else 1
    K0 = 3 * C
    X0 = rnd(M, K0) * torch.exp(rnd(M, 1))           # rows decades apart, and the three 192-k passes differ in scale
    X0[:, :C] *= 30.0
    w0 = rnd(C, K0, scale=0.08)
    x0 = rnd(M, C) * 1.3 + rnd(M, 1)
    res0 = rnd(M, C)
    w3 = rnd(C, C, scale=0.1)
    s3 = torch.rand(nsamp, generator=G) + 0.5 if nsamp else None
    P0, P3 = ops.Bx3(C, K0, "cuda"), ops.Bx3(C, C, "cuda")
    tb = ops.PrepTable()
    tb.linear(w0.cuda(), P0, f16=True)
    tb.linear(w3.cuda(), P3, f16=True)
    tb.build("cuda").run()
    st0 = torch.empty(M, 2, device="cuda")
    ops.layernorm_fwd(x0.cuda(), st0)
    # float64 reference of dy
    x0d = x0.double().requires_grad_(True)
    xn = F.layer_norm(x0d, (C,), None, None, 1e-5)
    xn.backward(X0.double() @ w0.double().t())
    dy_ref = res0.double() + x0d.grad
    _, h_ref, dx_ref, dh_ref, gh_ref = _reference(cpu, M, rps, dy_ref)
    h = h_ref.float().cuda()
    ch = (P3, None, None if s3 is None else s3.cuda()) if chain else None

    def run(front):
        dy = torch.full((M, C), float("nan"), device="cuda")
        dh = torch.full((M, hidden), float("nan"), device="cuda")
        gh = torch.full((M, hidden), float("nan"), device="cuda")
        dx = torch.full((M, C), float("nan"), device="cuda")
        o3 = torch.full((M, C), float("nan"), device="cuda")
        if not front:
            ops.gemm_nt_lnbwd(X0.cuda(), P0, x0.cuda(), st0, res0.cuda(), dy)
        ops.mlp_bwd_f16(dy, P["w2T"], P["w1T"], h, dh, gh, dev["x"], st, dx, rowscale=dev["s"], rows_per_scale=rps,
                        chain=(ch[0], o3, ch[2]) if chain else None,
                        front=(X0.cuda(), P0, x0.cuda(), st0, res0.cuda()) if front else None)
        return dy, dh, gh, dx, o3
    fused = run(True)
    assert relerr(fused[0], dy_ref) < 4e-6
    assert relerr(fused[1], dh_ref) < 8e-6 and relerr(fused[2], gh_ref) < 2e-6 and relerr(fused[3], dx_ref) < 8e-6
    if C in (180, 192):          # the stand-alone Linear kernel takes 192-column tiles only
        sep = run(False)
        assert relerr(fused[0], sep[0]) < 2e-6 and relerr(fused[1], sep[1]) < 4e-6 and relerr(fused[3], sep[3]) < 4e-6
    if chain:
        ref3 = dx_ref @ w3.double().t()
        if s3 is not None:
            ref3 = ref3 * s3.double().repeat_interleave(rps)[:M, None]
        assert relerr(fused[4], ref3) < 8e-6


def test_mlp_f16_rows_decades_apart(ops):
    """Block exponents are per token row: rows whose magnitudes differ by 1e8 keep f32-grade accuracy relative to
    THEMSELVES (gradient rows; the forward's LayerNorm output has an a-priori range)."""
    M, C, hidden = 512, 180, 360
    cpu, dev, P, b1f, st = _problem(ops, M, C, hidden, 0)
    h = torch.empty(M, hidden, device="cuda")
    out = torch.empty(M, C, device="cuda")
    ops.mlp_fwd_f16(dev["x"], st, P["w1"], b1f, P["w2"], dev["b2"], out, h=h)
    rowmag = torch.exp(torch.linspace(-9.2, 9.2, M)).unsqueeze(1)          # 1e-4 .. 1e4
    dy = rnd(M, C) * rowmag
    _, _, dx_ref, dh_ref, _ = _reference(cpu, M, 1, dy)
    dh, gh, dx = torch.empty_like(h), torch.empty_like(h), torch.empty_like(out)
    ops.mlp_bwd_f16(dy.cuda(), P["w2T"], P["w1T"], h, dh, gh, dev["x"], st, dx)
    for got, ref in ((dh, dh_ref), (dx, dx_ref)):
        e = ((got.double().cpu() - ref).abs().amax(1) / ref.abs().amax(1)).max().item()
        assert e < 2e-5, e


def test_mlp_f16_rejects_shapes_it_does_not_take(ops):
    assert ops.mlp_f16_fusable(180, 360) and not ops.mlp_f16_fusable(256, 512) and not ops.mlp_f16_fusable(60, 120)
    x = torch.zeros(64, 180, device="cuda")
    st = torch.zeros(64, 2, device="cuda")
    w1, w2 = ops.Bx3(720, 180, "cuda"), ops.Bx3(180, 720, "cuda")
    w1.fmt = w2.fmt = 1
    with pytest.raises(ops.SrhipError):
        ops.mlp_fwd_f16(x, st, w1, torch.zeros(720, device="cuda"), w2, torch.zeros(180, device="cuda"),
                        torch.empty_like(x))
