"""The W-MSA half of a Swin block, forward, as one kernel (srhip_wmsa_fwd_f16x2, wmsa_f16.hip) against a float64
statement of norm1 + roll + window_partition + WindowAttention.forward + window_reverse + roll + residual
(dlib/models/network_swinir.py:288-334,153-176), and against the three launches it replaces."""
import pytest
import torch
import torch.nn.functional as F

from oracle import sr_oracle as O

pytestmark = pytest.mark.gpu

G = torch.Generator().manual_seed(13579)


def rnd(*shape, scale=1.0):
    return torch.randn(*shape, generator=G) * scale


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def ops():
    from srhip import ops as o
    return o


def ref_wmsa(x, gamma, beta, wq, bq, wp, bp, table, s, B, H, W, heads, shift):
    """float64: qkv, att, out"""
    C = x.shape[1]
    d = C // heads
    xn = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    qkv = xn @ wq.t() + bq
    t = qkv.reshape(B, H, W, 3 * C)
    if shift:
        t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
    xw = O.window_partition(t, 8).reshape(-1, 64, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = xw[0] * d ** -0.5, xw[1], xw[2]
    att = q @ k.transpose(-2, -1)
    rpi = O.relative_position_index(8)
    att = att + table[rpi.reshape(-1)].reshape(64, 64, heads).permute(2, 0, 1)[None]
    if shift:
        m = O.shifted_window_mask(H, W, 8, shift).to(att.dtype)
        nw = m.shape[0]
        att = (att.reshape(B, nw, heads, 64, 64) + m[None, :, None]).reshape(-1, heads, 64, 64)
    att = att.softmax(-1)
    o = (att @ v).transpose(1, 2).reshape(-1, 8, 8, C)
    o = O.window_reverse(o, 8, H, W)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    a = o.reshape(B * H * W, C)
    y = a @ wp.t() + bp
    if s is not None:
        y = y * s.repeat_interleave(H * W)[:, None]
    return qkv, a, x + y


SHAPES = [(2, 16, 16, 180, 6, 0, True), (2, 16, 16, 180, 6, 4, True), (1, 64, 64, 180, 6, 4, False),
          (1, 16, 24, 60, 6, 4, True), (3, 24, 16, 60, 6, 0, False), (1, 72, 40, 180, 6, 4, True),
          (1, 16, 24, 96, 6, 4, False), (2, 16, 16, 192, 6, 0, True), (1, 24, 24, 64, 2, 4, True),
          (1, 8, 8, 180, 6, 0, False), (1, 16, 16, 160, 5, 4, True), (2, 16, 16, 128, 8, 4, True)]


@pytest.mark.parametrize("B,H,W,C,heads,shift,drop", SHAPES)
def test_wmsa_f16_forward(ops, B, H, W, C, heads, shift, drop):
    assert ops.wattn_f16_ok(C, heads)         # the planes below are built in the fp16x2 format whatever the shape
    T = B * H * W
    x = rnd(T, C) * 1.5 + rnd(T, 1)
    gamma, beta = 1 + rnd(C, scale=0.2), rnd(C, scale=0.2)
    wq, bq = rnd(3 * C, C, scale=0.12), rnd(3 * C, scale=0.3)
    wp, bp = rnd(C, C, scale=0.1), rnd(C, scale=0.3)
    table = rnd(225, heads, scale=0.5)
    s = torch.rand(B, generator=G) + 0.5 if drop else None
    d = {k: v.cuda() for k, v in dict(x=x, gamma=gamma, beta=beta, wq=wq, bq=bq, wp=wp, bp=bp, table=table).items()}
    sd = None if s is None else s.cuda()
    Pq, Pp = ops.Bx3(3 * C, C, "cuda"), ops.Bx3(C, C, "cuda")
    bqf = torch.empty(3 * C, device="cuda")
    biasF, biasG = torch.empty(heads, 64, 64).cuda(), torch.empty(heads, 64, 64).cuda()
    tb = ops.PrepTable()
    tb.linear(d["wq"], Pq, gamma=d["gamma"], f16=True)
    tb.linear(d["wp"], Pp, f16=True)
    tb.fold_bias(d["wq"], d["bq"], d["beta"], bqf)
    tb.build("cuda").run()
    ops.bias_expand_f16(d["table"], biasF, biasG)
    st = torch.empty(T, 2, device="cuda")
    ops.layernorm_fwd(d["x"], st)

    qkv = torch.full((T, 3 * C), float("nan"), device="cuda")
    att = torch.full((T, C), float("nan"), device="cuda")
    out = torch.full((T, C), float("nan"), device="cuda")
    st_out = torch.full((T, 2), float("nan"), device="cuda")
    ops.wmsa_fwd_f16(d["x"], st, Pq, bqf, Pp, d["bp"], biasF, qkv, att, out, B, H, W, heads, shift, rowscale=sd,
                     stats_out=st_out)
    r_qkv, r_att, r_out = ref_wmsa(x.double(), gamma.double(), beta.double(), wq.double(), bq.double(), wp.double(),
                                   bp.double(), table.double(), None if s is None else s.double(), B, H, W, heads,
                                   shift)
    assert relerr(qkv, r_qkv) < 2e-6
    assert relerr(att, r_att) < 4e-6
    assert relerr(out, r_out) < 2e-6
    mean, var = r_out.mean(1), r_out.var(1, unbiased=False)
    assert relerr(st_out[:, 0], mean) < 1e-5 and relerr(st_out[:, 1], (var + 1e-5).rsqrt()) < 1e-5
    # without the statistics
    out2 = torch.empty_like(out)
    ops.wmsa_fwd_f16(d["x"], st, Pq, bqf, Pp, d["bp"], biasF, qkv, att, out2, B, H, W, heads, shift, rowscale=sd)
    assert torch.equal(out2, out)
    if heads in (5, 6):      # inference form: q, k, v stay in the registers of the head's wave, no qkv is written
        out3, att3 = torch.empty_like(out), torch.empty_like(att)
        ops.wmsa_fwd_f16(d["x"], st, Pq, bqf, Pp, d["bp"], biasF, None, att3, out3, B, H, W, heads, shift, rowscale=sd)
        assert torch.equal(out3, out) and torch.equal(att3, att)

    # the three launches it replaces
    if ops.wattn_f16_ok(C, heads):
        qkv_u, att_u, out_u = torch.empty_like(qkv), torch.empty_like(att), torch.empty_like(out)
        wqf = (d["wq"] * d["gamma"][None, :]).contiguous()
        ops.gemm_nt(d["x"], wqf, bqf, out=qkv_u, a_mode=1, ln_stats=st)
        ops.window_attention_fwd_f16(qkv_u, att_u, biasF, B, H, W, C, heads, shift)
        ops.gemm_nt(att_u, d["wp"], d["bp"], out=out_u, epi=2, R=d["x"], rowscale=sd, rows_per_scale=H * W)
        assert relerr(qkv, qkv_u) < 2e-6 and relerr(att, att_u) < 4e-6 and relerr(out, out_u) < 2e-6


def test_wmsa_f16_rejects_shapes_it_does_not_take(ops):
    from srhip._lib import SrhipError
    x = torch.zeros(64, 200, device="cuda")
    with pytest.raises((SrhipError, AssertionError)):
        ops.wmsa_fwd_f16(x, torch.zeros(64, 2, device="cuda"), ops.Bx3(600, 200, "cuda"), torch.zeros(600, device="cuda"),
                         ops.Bx3(200, 200, "cuda"), torch.zeros(200, device="cuda"), torch.zeros(5, 64, 64, device="cuda"),
                         torch.zeros(64, 600, device="cuda"), torch.zeros(64, 200, device="cuda"),
                         torch.zeros(64, 200, device="cuda"), 1, 8, 8, 5, 0)
