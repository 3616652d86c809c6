"""GPU parity tests, kernel by kernel: every libsrhip entry point (called through
the C-ABI) against a plain PyTorch-fp32 CPU statement of the same op / the
oracle.  Tolerances: bit-exact for index ops and integer-valued metrics;
fp32 contractions within 2e-5 relative to the output scale (different
summation order than aten; exact-f32 MFMA, no reduced precision anywhere)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import sr_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from srhip import ops as _ops
    return _ops


def dev(t):
    return t.cuda().contiguous()


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def check(a, b, tol, what):
    e = relerr(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:g}"


G = torch.Generator().manual_seed(1234)


def rnd(*shape, scale=1.0):
    return torch.randn(*shape, generator=G) * scale


# ------------------------------------------------------------------ gemm_nt
@pytest.mark.parametrize("M,N,K", [(300, 180, 180), (4096, 540, 180), (1000, 60, 60),
                                   (777, 64, 64), (2048, 360, 180), (515, 180, 360),
                                   (64, 120, 60), (130, 256, 64), (33000, 180, 180)])
def test_gemm_nt_bias(ops, M, N, K):
    A, W, b = rnd(M, K), rnd(N, K, scale=0.1), rnd(N)
    out = ops.gemm_nt(dev(A), dev(W), dev(b))
    check(out, F.linear(A, W, b), 2e-5, f"gemm_nt {M}x{N}x{K}")


def test_gemm_nt_batched_slices(ops):
    """srhip_gemm_nt_batched: the (sample, head) products of an attention -- 'bhid,bhjd->bhij' on head slices of row-major
    [B*T][heads*dh] matrices and 'bhij,bhjd->bhid' back into such a matrix (network_act.py:151-227) -- in one launch each,
    against float64; ragged token counts (441 -> pitch 444), 3 samples x 5 heads."""
    gen = torch.Generator().manual_seed(5)
    B, heads, dh, Tq, Tk = 3, 5, 36, 100, 441
    Tk4 = (Tk + 3) & ~3
    q = torch.randn(B * Tq, heads * dh + 8, generator=gen).cuda()
    k = torch.randn(B * Tk, 2 * heads * dh, generator=gen).cuda()
    dots = torch.zeros(B * heads, Tq, Tk4, device="cuda")
    ops.gemm_nt_batched(q[:, :dh], (Tq * q.stride(0), dh), k[:, :dh], (Tk * k.stride(0), dh), dots[0, :, :Tk],
                        (heads * Tq * Tk4, Tq * Tk4), Tq, Tk, dh, B * heads, heads)
    qd = q[:, :heads * dh].double().view(B, Tq, heads, dh).permute(0, 2, 1, 3)
    kd = k[:, :heads * dh].double().view(B, Tk, heads, dh).permute(0, 2, 1, 3)
    ref = qd @ kd.transpose(-1, -2)
    assert relerr(dots.view(B, heads, Tq, Tk4)[..., :Tk], ref) < 1e-6
    assert (dots.view(B, heads, Tq, Tk4)[..., Tk:] == 0).all()
    v = k[:, heads * dh:]
    vt = torch.zeros(B * heads, dh, Tk4, device="cuda")
    vt[:, :, :Tk].copy_(v.reshape(B, Tk, heads, dh).permute(0, 2, 3, 1).reshape(B * heads, dh, Tk))
    out = torch.full((B * Tq, heads * dh + 4), float("nan"), device="cuda")
    ops.gemm_nt_batched(dots[0], (heads * Tq * Tk4, Tq * Tk4), vt[0], (heads * dh * Tk4, dh * Tk4), out[:, :dh],
                        (Tq * out.stride(0), dh), Tq, dh, Tk4, B * heads, heads)
    vd = v.double().view(B, Tk, heads, dh).permute(0, 2, 1, 3)
    ref2 = (dots.double().view(B, heads, Tq, Tk4)[..., :Tk] @ vd).permute(0, 2, 1, 3).reshape(B * Tq, heads * dh)
    assert relerr(out[:, :heads * dh], ref2) < 1e-6
    assert torch.isnan(out[:, heads * dh:]).all()


def test_gemm_nt_prologues_epilogues(ops):
    M, N, K = 1024, 180, 180
    A, W, b = rnd(M, K), rnd(N, K, scale=0.1), rnd(N)
    R = rnd(M, N)
    # LayerNorm prologue: (A-mean)*rstd
    mean = A.mean(1)
    rstd = 1.0 / torch.sqrt(A.var(1, unbiased=False) + 1e-5)
    stats = torch.stack([mean, rstd], 1)
    out = ops.gemm_nt(dev(A), dev(W), dev(b), a_mode=1, ln_stats=dev(stats))
    check(out, F.linear((A - mean[:, None]) * rstd[:, None], W, b), 2e-5, "ln prologue")
    # GELU prologue
    out = ops.gemm_nt(dev(A), dev(W), dev(b), a_mode=2)
    check(out, F.linear(F.gelu(A), W, b), 2e-5, "gelu prologue")
    # relu
    out = ops.gemm_nt(dev(A), dev(W), dev(b), epi=1)
    check(out, F.relu(F.linear(A, W, b)), 2e-5, "relu epilogue")
    # residual + per-sample scale (4 samples of 256 rows)
    rs = torch.tensor([1.0, 0.0, 1.25, 2.0])
    out = ops.gemm_nt(dev(A), dev(W), dev(b), epi=2, R=dev(R), rowscale=dev(rs),
                      rows_per_scale=256, alpha=0.5)
    ref = R + 0.5 * rs.repeat_interleave(256)[:, None] * F.linear(A, W, b)
    check(out, ref, 2e-5, "residual epilogue")
    # per-row scale fallback (rows_per_scale not a multiple of the tile)
    rs2 = torch.rand(M // 8 + 1, generator=G)
    out = ops.gemm_nt(dev(A), dev(W), dev(b), epi=2, R=dev(R), rowscale=dev(rs2), rows_per_scale=8)
    ref = R + rs2.repeat_interleave(8)[:M, None] * F.linear(A, W, b)
    check(out, ref, 2e-5, "residual epilogue per-row scale")
    # dgelu: s*acc*gelu'(R)
    Rg = R.clone().requires_grad_(True)
    F.gelu(Rg).sum().backward()
    out = ops.gemm_nt(dev(A), dev(W), None, epi=3, R=dev(R), rowscale=dev(rs), rows_per_scale=256)
    ref = rs.repeat_interleave(256)[:, None] * F.linear(A, W) * Rg.grad
    check(out, ref, 2e-5, "dgelu epilogue")
    # relu mask
    out = ops.gemm_nt(dev(A), dev(W), None, epi=4, R=dev(R))
    check(out, F.linear(A, W) * (R > 0), 2e-5, "relu-mask epilogue")
    # strided output / input (column slices of wider buffers)
    wide = torch.zeros(M, 540).cuda()
    ops.gemm_nt(dev(A), dev(W), dev(b), out=wide[:, 180:360])
    check(wide[:, 180:360], F.linear(A, W, b), 2e-5, "strided out")
    assert wide[:, :180].abs().max() == 0 and wide[:, 360:].abs().max() == 0


# ------------------------------------------------------------------ conv3x3
@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 16, 16, 180, 180), (1, 24, 40, 60, 60),
                                         (1, 20, 12, 64, 64), (1, 16, 16, 64, 256),
                                         (2, 64, 64, 180, 64), (1, 9, 7, 16, 16),
                                         (8, 64, 64, 180, 180), (8, 128, 128, 64, 64),
                                         (8, 128, 128, 64, 256), (1, 512, 512, 64, 64)])
def test_conv3x3_fwd_bwd(ops, B, H, W, Ci, Co):
    """forward, data gradient and weight gradient at every size, the EDSR training shapes included
    (8x128^2 and 512^2 at 64 features: the weight gradient there runs the bf16x3 slice plans --
    three blocks per CU for 64-wide tiles -- that the small cases never reach)."""
    x = rnd(B, Ci, H, W)
    w = rnd(Co, Ci, 3, 3, scale=0.05)
    b = rnd(Co)
    xh = dev(x.permute(0, 2, 3, 1))
    wd = dev(w)
    wp = torch.empty(9, Co, Ci).cuda()
    wpt = torch.empty(9, Ci, Co).cuda()
    ops.pack_conv_weight(wd, wp, wpt)
    y = ops.conv3x3(xh, wp, dev(b), Co)
    ref = F.conv2d(x, w, b, padding=1)
    check(y.permute(0, 3, 1, 2), ref, 2e-5, "conv fwd")
    # data gradient through the flipped/transposed pack
    dy = rnd(B, Co, H, W)
    dyh = dev(dy.permute(0, 2, 3, 1))
    dx = ops.conv3x3(dyh, wpt, None, Ci)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    F.conv2d(xr, wr, br, padding=1).backward(dy)
    check(dx.permute(0, 3, 1, 2), xr.grad, 2e-5, "conv bwd-data")
    dw = torch.empty_like(wd)
    db = torch.empty(Co).cuda()
    ops.conv3x3_wgrad(dyh, xh, dw, db)
    check(dw, wr.grad, 2e-5, "conv bwd-weight")
    check(db, br.grad, 2e-5, "conv bwd-bias")


def test_conv3x3_epilogues(ops):
    B, H, W, C = 2, 16, 24, 64
    x, w, b, R = rnd(B, C, H, W), rnd(C, C, 3, 3, scale=0.05), rnd(C), rnd(B, C, H, W)
    xh, Rh = dev(x.permute(0, 2, 3, 1)), dev(R.permute(0, 2, 3, 1))
    wp = torch.empty(9, C, C).cuda()
    ops.pack_conv_weight(dev(w), wp, None)
    ref = F.conv2d(x, w, b, padding=1)
    y = ops.conv3x3(xh, wp, dev(b), C, epi=1)
    check(y.permute(0, 3, 1, 2), F.relu(ref), 2e-5, "conv relu")
    y = ops.conv3x3(xh, wp, dev(b), C, epi=2, R=Rh, alpha=0.1)
    check(y.permute(0, 3, 1, 2), R + 0.1 * ref, 2e-5, "conv residual")
    y = ops.conv3x3(xh, wp, None, C, epi=4, R=Rh)
    check(y.permute(0, 3, 1, 2), F.conv2d(x, w, None, padding=1) * (R > 0), 2e-5, "conv relu-mask")


@pytest.mark.parametrize("B,H,W,rs", [(2, 16, 24, 1.0), (1, 9, 21, 0.1), (3, 4, 16, 1.0), (8, 64, 64, 1.0), (1, 130, 70, 0.5)])
def test_resblock64_one_launch_fwd_and_bwd(ops, B, H, W, rs):
    """srhip_resblock64_{fwd,bwd}_f16x2 (one launch per EDSR ResBlock and direction, round 6) against torch on the CPU --
    ResBlock.forward of the reference (network_nlsn.py:72-93) and its autograd: a, out; da, dx.  Sizes that are not multiples
    of the 4 x 16 tile, one-tile images, the x8 training shape, res_scale != 1; bias and ReLU decisions at the image border
    (the ring of mid pixels OUTSIDE the image must be zeros, not conv values: checked by the border rows of `out`)."""
    if not ops.resblock64_fusable(64):
        pytest.skip("fp16x2 conv operands are off")
    C = 64
    x = rnd(B, C, H, W)
    w1, w2 = rnd(C, C, 3, 3, scale=0.05), rnd(C, C, 3, 3, scale=0.05)
    b1, b2 = rnd(C, scale=0.3), rnd(C, scale=0.3)
    planes = {k: ops.Bx3(9 * C, C, "cuda") for k in ("w1", "w2", "w1t", "w2t")}
    w1d, w2d = dev(w1), dev(w2)
    tb = ops.PrepTable()
    tb.conv(w1d, planes["w1"])
    tb.conv(w2d, planes["w2"])
    tb.conv(w1d, planes["w1t"], data_grad=True)
    tb.conv(w2d, planes["w2t"], data_grad=True)
    tb.build("cuda").run()
    assert all(v.fmt == 1 for v in planes.values())
    xh = dev(x.permute(0, 2, 3, 1))
    a, out = torch.empty_like(xh), torch.empty_like(xh)
    ops.resblock64_fwd(xh, planes["w1"], dev(b1), planes["w2"], dev(b2), rs, a, out)
    xr = x.clone().requires_grad_(True)
    ar = F.relu(F.conv2d(xr, w1, b1, padding=1))
    outr = xr + rs * F.conv2d(ar, w2, b2, padding=1)
    check(a.permute(0, 3, 1, 2), ar, 2e-5, "resblock a")
    check(out.permute(0, 3, 1, 2), outr, 2e-5, "resblock out")
    # the unfused launches give the same numbers to rounding (their own halo-tile exponents differ from the mid tile's)
    wp1, wp2 = planes["w1"], planes["w2"]
    a_u = ops.conv3x3(xh, wp1, dev(b1), C, epi=1)
    out_u = ops.conv3x3(a_u, wp2, dev(b2), C, epi=2, R=xh, alpha=rs)
    check(out, out_u, 1e-5, "resblock vs two launches")
    # backward: g -> da (gradient wrt the first conv's output, ReLU mask applied), dx
    g = rnd(B, C, H, W)
    gh = dev(g.permute(0, 2, 3, 1))
    da, dx = torch.empty_like(xh), torch.empty_like(xh)
    # ReLU decisions of the DEVICE activation (pixels within rounding of zero may fall on the other side on the CPU)
    a_dev = a.permute(0, 3, 1, 2).cpu()
    mask = (a_dev > 0).float()
    ops.resblock64_bwd(gh, planes["w2t"], planes["w1t"], a, rs, da, dx)
    dar = rs * F.conv_transpose2d(g, w2, padding=1) * mask
    dxr = g + F.conv_transpose2d(dar, w1, padding=1)
    check(da.permute(0, 3, 1, 2), dar, 2e-5, "resblock da")
    check(dx.permute(0, 3, 1, 2), dxr, 2e-5, "resblock dx")
    with pytest.raises(RuntimeError):
        ops.resblock64_fwd(xh, planes["w1"], dev(b1), planes["w2"], dev(b2), rs, a, xh)      # out aliases x


# ------------------------------------------------------------------ gemm_tn
@pytest.mark.parametrize("M,NI,NJ", [(5000, 180, 180), (4096, 540, 180), (3000, 360, 180),
                                     (2500, 180, 360), (700, 60, 60), (1234, 64, 256),
                                     (32768, 180, 180)])
def test_linear_wgrad(ops, M, NI, NJ):
    dY, X = rnd(M, NI), rnd(M, NJ)
    dW = torch.empty(NI, NJ).cuda()
    db = torch.empty(NI).cuda()
    ops.linear_wgrad(dev(dY), dev(X), dW, db)
    check(dW, dY.double().t() @ X.double(), 2e-5, "dW")
    check(db, dY.double().sum(0), 2e-5, "db")


def test_linear_wgrad_modes(ops):
    M, NI, NJ = 2048, 180, 180
    dY, X = rnd(M, NI), rnd(M, NJ)
    rs = torch.tensor([1.0, 0.0, 1.5, 2.0, 1.0, 1.0, 0.5, 1.0])
    dW = torch.empty(NI, NJ).cuda()
    db = torch.empty(NI).cuda()
    # per-sample scale on dY, GELU on X
    ops.linear_wgrad(dev(dY), dev(X), dW, db, a_rowscale=dev(rs), a_rowscale_rows=256, b_mode=2)
    dYs = dY * rs.repeat_interleave(256)[:, None]
    check(dW, dYs.double().t() @ F.gelu(X).double(), 2e-5, "dW scale+gelu")
    check(db, dYs.double().sum(0), 2e-5, "db scale")
    # LayerNorm-folded Linear: W_f = W*gamma, b_f = b + W.beta
    Wt, gamma, beta = rnd(NI, NJ, scale=0.1), 1 + 0.1 * rnd(NJ), 0.1 * rnd(NJ)
    xr = X.clone()
    Wr, gr, br = (t.clone().requires_grad_(True) for t in (Wt, gamma, beta))
    bias = torch.zeros(NI, requires_grad=True)
    y = F.linear(F.layer_norm(xr, (NJ,), gr, br), Wr, bias)
    y.backward(dY)
    mean = X.mean(1)
    rstd = 1.0 / torch.sqrt(X.var(1, unbiased=False) + 1e-5)
    stats = dev(torch.stack([mean, rstd], 1))
    dg, dbt = torch.empty(NJ).cuda(), torch.empty(NJ).cuda()
    ops.linear_wgrad(dev(dY), dev(X), dW, db, b_mode=1, ln_stats=stats,
                     ln=(dev(Wt), dev(gamma), dev(beta), dg, dbt))
    check(dW, Wr.grad, 3e-5, "ln-folded dW")
    check(db, bias.grad, 3e-5, "ln-folded db")
    check(dg, gr.grad, 3e-5, "ln-folded dgamma")
    check(dbt, br.grad, 3e-5, "ln-folded dbeta")
    # and the fold itself
    Wf, bf = torch.empty(NI, NJ).cuda(), torch.empty(NI).cuda()
    b0 = rnd(NI)
    ops.fold_layernorm(dev(Wt), dev(b0), dev(gamma), dev(beta), Wf, bf)
    check(Wf, Wt * gamma[None], 1e-6, "fold W")
    check(bf, b0 + Wt @ beta, 1e-5, "fold b")
    Tt = torch.empty(NJ, NI).cuda()
    ops.transpose(dev(Wt), Tt)
    assert torch.equal(Tt.cpu(), Wt.t().contiguous())


# ------------------------------------------------------------------ layernorm
@pytest.mark.parametrize("M,C", [(1000, 180), (4096, 60), (513, 64), (77, 256)])
def test_layernorm(ops, M, C):
    x, g, b, dy, res = rnd(M, C), 1 + 0.1 * rnd(C), 0.1 * rnd(C), rnd(M, C), rnd(M, C)
    stats = torch.empty(M, 2).cuda()
    y = torch.empty(M, C).cuda()
    ops.layernorm_fwd(dev(x), stats, y, dev(g), dev(b))
    check(y, F.layer_norm(x, (C,), g, b), 1e-5, "ln fwd")
    check(stats[:, 0], x.mean(1), 1e-5, "ln mean")
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, g, b))
    F.layer_norm(xr, (C,), gr, br).backward(dy)
    out = torch.empty(M, C).cuda()
    dg, dbt = torch.empty(C).cuda(), torch.empty(C).cuda()
    ops.layernorm_bwd(dev(dy), dev(x), stats, out, res=dev(res), gamma=dev(g), dgamma=dg, dbeta=dbt)
    check(out, res + xr.grad, 2e-5, "ln bwd dx")
    check(dg, gr.grad, 2e-5, "ln dgamma")
    check(dbt, br.grad, 2e-5, "ln dbeta")
    # gradient w.r.t. the normalised value (gamma folded elsewhere)
    xr2 = x.clone().requires_grad_(True)
    F.layer_norm(xr2, (C,)).backward(dy)
    ops.layernorm_bwd(dev(dy), dev(x), stats, out)
    check(out, xr2.grad, 2e-5, "ln bwd dxhat")


# ------------------------------------------------------------------ attention
def ref_window_attention(qkv, table, B, H, W, C, heads, shift):
    """oracle statement: roll + partition + attention + reverse + roll."""
    d = C // heads
    x = qkv.reshape(B, H, W, 3 * C)
    if shift:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = O.window_partition(x, 8).reshape(-1, 64, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = xw[0] * d ** -0.5, xw[1], xw[2]
    att = q @ k.transpose(-2, -1)
    rpi = O.relative_position_index(8)
    att = att + table[rpi.reshape(-1)].reshape(64, 64, heads).permute(2, 0, 1)[None]
    if shift:
        m = O.shifted_window_mask(H, W, 8, shift)
        nw = m.shape[0]
        att = (att.reshape(B, nw, heads, 64, 64) + m[None, :, None]).reshape(-1, heads, 64, 64)
    att = att.softmax(-1)
    o = (att @ v).transpose(1, 2).reshape(-1, 8, 8, C)
    o = O.window_reverse(o, 8, H, W)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return o.reshape(B * H * W, C)


@pytest.mark.parametrize("B,H,W,C,heads,shift", [(2, 16, 16, 180, 6, 0), (2, 16, 16, 180, 6, 4),
                                                 (1, 16, 24, 60, 6, 4), (3, 24, 16, 60, 6, 0),
                                                 (1, 64, 64, 180, 6, 4), (1, 72, 72, 180, 6, 4),
                                                 (1, 16, 24, 96, 6, 4), (2, 16, 16, 192, 6, 0),    # head dims 16, 32
                                                 (1, 24, 24, 64, 2, 4)])
def test_window_attention(ops, B, H, W, C, heads, shift):
    T = B * H * W
    qkv = rnd(T, 3 * C)
    table = rnd(225, heads, scale=0.5)
    dout = rnd(T, C)
    biasT = torch.empty(heads, 64, 64).cuda()
    biasN = torch.empty(heads, 64, 64).cuda()
    ops.bias_expand(dev(table), biasT, biasN)
    rpi = O.relative_position_index(8)
    dense = table[rpi.reshape(-1)].reshape(64, 64, heads).permute(2, 0, 1)
    # the images are stored in MFMA-lane order [head][a][b][lane][16] (wattn.hip: wa_img_index):
    #   row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5), col = lane & 31
    #   imgT tile (kb=a, qb=b): bias[query = col + 32 b][key = row + 32 a]
    #   imgN tile (qb=a, kb=b): bias[query = row + 32 a][key = col + 32 b]
    q = torch.arange(16).view(1, 1, 1, 16)
    lane = torch.arange(64).view(1, 1, 64, 1)
    a = torch.arange(2).view(2, 1, 1, 1)
    b = torch.arange(2).view(1, 2, 1, 1)
    row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)
    col = lane & 31
    expT = dense[:, (col + 32 * b).expand(2, 2, 64, 16), (row + 32 * a).expand(2, 2, 64, 16)]
    expN = dense[:, (row + 32 * a).expand(2, 2, 64, 16), (col + 32 * b).expand(2, 2, 64, 16)]
    assert torch.equal(biasT.cpu().reshape(heads, 2, 2, 64, 16), expT)
    assert torch.equal(biasN.cpu().reshape(heads, 2, 2, 64, 16), expN)
    out = torch.empty(T, C).cuda()
    ops.window_attention_fwd(dev(qkv), out, biasT, B, H, W, C, heads, shift)
    qr, tr = qkv.clone().requires_grad_(True), table.clone().requires_grad_(True)
    ref = ref_window_attention(qr, tr, B, H, W, C, heads, shift)
    check(out, ref, 1e-5, "attention fwd")
    ref.backward(dout)
    dqkv = torch.empty(T, 3 * C).cuda()
    dbiasT = torch.full((heads, 64, 64), 123.0).cuda()      # overwritten, not accumulated into
    ops.window_attention_bwd(dev(qkv), dev(dout), dqkv, biasT, biasN, dbiasT, B, H, W, C, heads, shift)
    check(dqkv[:, :C], qr.grad[:, :C], 2e-5, "attention dq")
    check(dqkv[:, C:2 * C], qr.grad[:, C:2 * C], 2e-5, "attention dk")
    check(dqkv[:, 2 * C:], qr.grad[:, 2 * C:], 2e-5, "attention dv")
    dtable = torch.empty(225, heads).cuda()
    ops.bias_grad(dbiasT, dtable)
    check(dtable, tr.grad, 5e-5, "attention dtable")


ATT_SHAPES = [(2, 16, 16, 180, 6, 0), (2, 16, 16, 180, 6, 4), (1, 16, 24, 60, 6, 4), (3, 24, 16, 60, 6, 0),
              (1, 64, 64, 180, 6, 4), (1, 72, 72, 180, 6, 4), (1, 16, 24, 96, 6, 4), (2, 16, 16, 192, 6, 0),
              (1, 24, 24, 64, 2, 4)]


@pytest.mark.parametrize("regime", ["fresh", "saturated"])
@pytest.mark.parametrize("B,H,W,C,heads,shift", ATT_SHAPES)
def test_window_attention_f16x2(ops, B, H, W, C, heads, shift, regime):
    """wattn2.hip: the attention core on two fp16 planes / three products against a float64 statement of the
    reference's roll + partition + attention + reverse; 'saturated' = trained-like logits (q, k rows whose scales
    are decades apart, |logits| up to ~30: one-hot-ish softmax rows next to flat ones)."""
    T = B * H * W
    qkv = rnd(T, 3 * C)
    if regime == "saturated":
        rowmag = torch.exp(rnd(T, 1) * 1.5)
        qkv = torch.cat([qkv[:, :C] * rowmag * 3.0, qkv[:, C:2 * C] * torch.exp(rnd(T, 1)), qkv[:, 2 * C:] * torch.exp(rnd(1, C) * 2)], 1)
    table = rnd(225, heads, scale=0.5 if regime == "fresh" else 2.0)
    biasF, biasG = torch.empty(heads, 64, 64).cuda(), torch.empty(heads, 64, 64).cuda()
    ops.bias_expand_f16(dev(table), biasF, biasG)
    rpi = O.relative_position_index(8)
    dense = table[rpi.reshape(-1)].reshape(64, 64, heads).permute(2, 0, 1)      # [head][query][key]
    # img[head][I][J][lane][e] = bias[query 16 I + (lane & 15)][key 16 J + 4 (lane >> 4) + e]
    I = torch.arange(4).view(4, 1, 1, 1)
    J = torch.arange(4).view(1, 4, 1, 1)
    lane = torch.arange(64).view(1, 1, 64, 1)
    e = torch.arange(4).view(1, 1, 1, 4)
    exp = dense[:, (16 * I + (lane & 15)).expand(4, 4, 64, 4), (16 * J + 4 * (lane >> 4) + e).expand(4, 4, 64, 4)]
    assert torch.equal(biasF.cpu().reshape(heads, 4, 4, 64, 4), exp)
    # imgG[head][J][I][lane][e] = bias[query 16 I + 4 (lane >> 4) + e][key 16 J + (lane & 15)]  (first tile index = key tile)
    expG = dense[:, (16 * J + 4 * (lane >> 4) + e).expand(4, 4, 64, 4), (16 * I + (lane & 15)).expand(4, 4, 64, 4)]
    assert torch.equal(biasG.cpu().reshape(heads, 4, 4, 64, 4), expG)
    out = torch.full((T, C), float("nan")).cuda()
    ops.window_attention_fwd_f16(dev(qkv), out, biasF, B, H, W, C, heads, shift)
    ref64 = ref_window_attention(qkv.double(), table.double(), B, H, W, C, heads, shift)
    ref32 = ref_window_attention(qkv, table, B, H, W, C, heads, shift)
    # f32-grade.  A two-plane fp16 operand carries 22 significant bits against f32's 24, and these contractions are short
    # (head dim <= 32, 64 keys): operand rounding, not accumulation, sets the error -- up to 4x an f32 computation's per
    # operand.  Gate: within 8x of what the reference's own f32 computation is from float64 (floor 2e-6 of the largest entry).
    e32 = (ref32.double() - ref64).abs().max().item()
    e16 = (out.cpu().double() - ref64).abs().max().item()
    assert e16 <= max(8.0 * e32, 2e-6 * ref64.abs().max().item()), (e16, e32)
    check(out, ref32, 1e-5 if regime == "fresh" else 1e-4, "attention fwd f16x2")
    # ---- backward against float64 autograd; "f32-grade" = within 8x of the f32 autograd's own distance from it (see above)
    dout = rnd(T, C)
    q64, t64 = qkv.double().requires_grad_(True), table.double().requires_grad_(True)
    ref_window_attention(q64, t64, B, H, W, C, heads, shift).backward(dout.double())
    q32, t32 = qkv.clone().requires_grad_(True), table.clone().requires_grad_(True)
    ref_window_attention(q32, t32, B, H, W, C, heads, shift).backward(dout)
    dqkv = torch.full((T, 3 * C), float("nan")).cuda()
    dbiasT = torch.full((heads, 64, 64), 123.0).cuda()      # overwritten, not accumulated into
    ops.window_attention_bwd_f16(dev(qkv), dev(dout), dqkv, biasF, biasG, dbiasT, B, H, W, C, heads, shift)
    dtable = torch.empty(225, heads).cuda()
    ops.bias_grad(dbiasT, dtable)
    for name, got, r64, r32 in (("dq", dqkv[:, :C], q64.grad[:, :C], q32.grad[:, :C]),
                                ("dk", dqkv[:, C:2 * C], q64.grad[:, C:2 * C], q32.grad[:, C:2 * C]),
                                ("dv", dqkv[:, 2 * C:], q64.grad[:, 2 * C:], q32.grad[:, 2 * C:]),
                                ("dtable", dtable, t64.grad, t32.grad)):
        e32 = (r32.double() - r64).abs().max().item()
        e16 = (got.cpu().double() - r64).abs().max().item()
        assert e16 <= max(8.0 * e32, 4e-6 * r64.abs().max().item()), (name, e16, e32, r64.abs().max().item())


# ------------------------------------------------------------------ edge convs
def test_conv_cin1_cout1(ops):
    B, H, W, Co = 2, 20, 28, 180
    x, w, b = torch.rand(B, 1, H, W, generator=G), rnd(Co, 1, 3, 3, scale=0.3), rnd(Co)
    y = ops.conv3x3_cin1_fwd(dev(x[:, 0]), dev(w), dev(b), Co)
    check(y.permute(0, 3, 1, 2), F.conv2d(x, w, b, padding=1), 1e-5, "cin1 fwd")
    dy = rnd(B, Co, H, W)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.conv2d(x, wr, br, padding=1).backward(dy)
    dw, db = torch.empty(Co, 1, 3, 3).cuda(), torch.empty(Co).cuda()
    ops.conv3x3_cin1_wgrad(dev(x[:, 0]), dev(dy.permute(0, 2, 3, 1)), dw, db)
    check(dw, wr.grad, 2e-5, "cin1 dW")
    check(db, br.grad, 2e-5, "cin1 db")
    # Cout = 1 tail: forward, data gradient (cin1 kernel with flipped taps) and
    # weight gradient (cin1 wgrad with roles swapped + flipped taps)
    Ci = 64
    xt, wt, bt = rnd(B, Ci, H, W), rnd(1, Ci, 3, 3, scale=0.1), rnd(1)
    xth = dev(xt.permute(0, 2, 3, 1))
    yt = ops.conv3x3_cout1_fwd(xth, dev(wt), dev(bt))
    check(yt, F.conv2d(xt, wt, bt, padding=1)[:, 0], 2e-5, "cout1 fwd")
    dyt = rnd(B, 1, H, W)
    xr, wr2, br2 = (t.clone().requires_grad_(True) for t in (xt, wt, bt))
    F.conv2d(xr, wr2, br2, padding=1).backward(dyt)
    dxt = ops.conv3x3_cin1_fwd(dev(dyt[:, 0]), dev(wt), None, Ci, flip=True)
    check(dxt.permute(0, 3, 1, 2), xr.grad, 2e-5, "cout1 bwd-data")
    dwt = torch.empty(1, Ci, 3, 3).cuda()
    ops.conv3x3_cin1_wgrad(dev(dyt[:, 0]), xth, dwt, None, flip=True)
    check(dwt, wr2.grad, 2e-5, "cout1 dW")


@pytest.mark.parametrize("B,H,W,Ci", [(1, 40, 50, 4), (2, 33, 17, 60), (1, 70, 36, 180), (1, 32, 16, 256)])
def test_conv_cout1_channel_counts(ops, B, H, W, Ci):
    """Cout = 1 tail conv (four output pixels per thread, 16-channel chunks): channel counts that are not multiples of
    the chunk, images that are not multiples of the 16 x 32 tile, against float64."""
    xt, wt, bt = rnd(B, Ci, H, W), rnd(1, Ci, 3, 3, scale=0.1), rnd(1)
    yt = ops.conv3x3_cout1_fwd(dev(xt.permute(0, 2, 3, 1)), dev(wt), dev(bt))
    ref = F.conv2d(xt.double(), wt.double(), bt.double(), padding=1)[:, 0]
    assert (yt.cpu().double() - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item()) * (1 + Ci / 64)


# ------------------------------------------------------------------ index ops
@pytest.mark.parametrize("r,Co,h,w", [(8, 1, 8, 8), (2, 64, 6, 10), (3, 2, 4, 4), (8, 1, 64, 64), (8, 64, 5, 7), (4, 32, 6, 3),
                                      (3, 40, 4, 5), (8, 16, 9, 4)])
def test_pixel_shuffle_bit_exact(ops, r, Co, h, w):
    B = 2
    x = torch.arange(B * Co * r * r * h * w, dtype=torch.float32).reshape(B, Co * r * r, h, w)
    ref = F.pixel_shuffle(x, r)
    xh = dev(x.permute(0, 2, 3, 1))
    assert torch.equal(ops.pixel_shuffle(xh, r).cpu(), ref)
    assert torch.equal(ops.pixel_shuffle(xh, r, nhwc_out=True).cpu(), ref.permute(0, 2, 3, 1))
    assert torch.equal(ops.pixel_shuffle(dev(ref), r, inverse=True).cpu(), xh.cpu())
    assert torch.equal(ops.pixel_shuffle(dev(ref.permute(0, 2, 3, 1)), r, nhwc_out=True,
                                         inverse=True).cpu(), xh.cpu())


# ------------------------------------------------------------------ losses
def test_losses(ops):
    pred = torch.rand(2, 1, 96, 80, generator=G)
    tgt = torch.rand(2, 1, 96, 80, generator=G)
    wgt = torch.rand(2, 1, 96, 80, generator=G) * 2
    for name, mode, w in (("l1", 0, None), ("l2", 1, None), ("l1w", 0, wgt)):
        p = pred.clone().requires_grad_(True)
        ref = O.loss_l1(p, tgt, 2.0, w) if mode == 0 else O.loss_l2(p, tgt, 2.0)
        ref.backward()
        grad = torch.empty_like(pred).cuda()
        val = ops.loss_l1l2(dev(pred), dev(tgt), mode, 2.0, None if w is None else dev(w), grad)
        check(val, ref.detach().reshape(1), 1e-6, name)
        check(grad, p.grad, 1e-6, name + " grad")
    for ws in (11, 19):
        p = pred.clone().requires_grad_(True)
        ref = O.loss_neg_ssim(p, tgt, 5.0, ws)
        ref.backward()
        grad = torch.empty_like(pred).cuda()
        val = ops.ssim_loss(dev(pred), dev(tgt), ws, 5.0, grad)
        check(val, ref.detach().reshape(1), 1e-4, f"ssim{ws}")  # fp32 E[x^2]-mu^2 cancellation
        check(grad, p.grad, 2e-4, f"ssim{ws} grad")
    # accumulate: L2 + 5*SSIM(19) as one MasterLoss (README recipe)
    p = pred.clone().requires_grad_(True)
    tot, _ = O.master_loss(p, tgt, [("l2", 1.0), ("ssim", 5.0, 19)])
    tot.backward()
    grad = torch.empty_like(pred).cuda()
    val = ops.loss_l1l2(dev(pred), dev(tgt), 1, 1.0, None, grad)
    ops.ssim_loss(dev(pred), dev(tgt), 19, 5.0, grad, val, grad_accum=True, loss_accum=True)
    check(val, tot.detach().reshape(1), 1e-4, "master")
    check(grad, p.grad, 2e-4, "master grad")


# ------------------------------------------------------------------ metrics
def test_metrics(ops):
    hr = (torch.rand(3, 1, 96, 112, generator=G) * 255).round() / 255
    hr[2] = hr[2] * 0 + 0.25
    pr = hr + 0.05 * rnd(3, 1, 96, 112)
    pr[1] = hr[1]
    ths = (4, 5, 6, 7, 8, 9, 10, 300)
    border = 8
    out = ops.metrics_psnr_family(dev(pr), dev(hr), border, ths).cpu()
    ssim = ops.metrics_ssim(dev(pr), dev(hr), border, ths).cpu()
    a, b = O.tensor2uint82float(pr), O.tensor2uint82float(hr)
    for k, th in enumerate((None,) + ths):
        roi = None if th is None else (b >= th).float()
        psnr = O.metric_psnr(a, b, border, roi)
        mse = O.metric_mse(a, b, border, roi)
        nrmse = O.metric_nrmse(a, b, border, roi)
        assert torch.equal(out[:, k, 2], mse), f"mse th={th}"      # integer sums: exact
        assert (out[:, k, 0] - psnr).abs().max() < 1e-9, f"psnr th={th}"
        assert (out[:, k, 3] - nrmse).abs().max() < 1e-12, f"nrmse th={th}"
        ya = O.gray_to_y(a / 255.0) * 255.0
        yb = O.gray_to_y(b / 255.0) * 255.0
        psnr_y = O.metric_psnr(ya, yb, border, roi)
        assert (out[:, k, 1] - psnr_y).abs().max() < 1e-4, f"psnr_y th={th}"
        s = O.metric_ssim(a, b, border, roi)
        assert (ssim[:, k] - s).abs().max() < 5e-5, f"ssim th={th}"  # fp32 cancellation
    assert abs(out[1, 0, 0].item() - 498.1308) < 1e-3
    # already-u8 inputs take the same path
    out2 = ops.metrics_psnr_family(dev(a), dev(b), border, ths, inputs_are_u8=True).cpu()
    assert torch.equal(out2[:, :, 2], out[:, :, 2])


# ------------------------------------------------------------------ optimizers
def test_optimizers(ops):
    n = 100003
    p0, gs = rnd(n), rnd(3, n)
    for name in ("adam", "adam_wd", "sgd"):
        p, m, v = dev(p0), torch.zeros(n).cuda(), torch.zeros(n).cuda()
        po, mo, vo = p0.clone(), torch.zeros(n), torch.zeros(n)
        for i in range(3):
            if name == "sgd":
                ops.sgd_step(p, dev(gs[i]), m, 0.01, first=(i == 0))
                O.sgd_nesterov_step(po, gs[i], mo, i == 0, 0.01)
            else:
                wd = 1e-4 if name == "adam_wd" else 0.0
                ops.adam_step(p, dev(gs[i]), m, v, i + 1, 2e-4, wd=wd)
                O.adam_step(po, gs[i], mo, vo, i + 1, 2e-4, wd=wd)
            assert (p.cpu() - po).abs().max() < 5e-7, (name, i)
    flag = torch.zeros(1, dtype=torch.int32).cuda()
    ops.nonfinite_flag(dev(p0), flag)
    assert flag.item() == 0
    bad = p0.clone()
    bad[777] = float("nan")
    ops.nonfinite_flag(dev(bad), flag)
    assert flag.item() == 1


def test_grad_norm_clip_and_ema_kernels(ops):
    """srhip_grad_norm_clip / srhip_ema_update against the reference's own calls (tests/golden/g50_clip_ema.npz: torch's
    clip_grad_norm_ -> optimizer.step() -> ModelBase.update_E, model_plain.py:350-361,393-394) on the flat layout of the
    training step (tensors padded to 16 bytes), then at the README net's size against the oracle; NaN norms propagate as
    torch's clamp does; a set skip flag leaves netE alone; world-size scaling of the norm."""
    import os
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g50_clip_ema.npz"))
    max_norm, decay = float(g["max_norm"]), float(g["decay"])
    shapes = [g[f"p0/{j}"].shape for j in range(3)]
    offs, off = [], 0
    for sh in shapes:
        offs.append(off)
        off += (int(np.prod(sh)) + 3) // 4 * 4

    def flat(prefix):
        f = torch.zeros(off)
        for j, sh in enumerate(shapes):
            f[offs[j]:offs[j] + int(np.prod(sh))] = torch.from_numpy(g[f"{prefix}/{j}"]).reshape(-1)
        return f
    for name in ("adam", "sgd"):
        p, e = dev(flat("p0")), dev(flat("p0"))
        m, v = torch.zeros(off).cuda(), torch.zeros(off).cuda()
        nc = torch.zeros(2).cuda()
        for step in range(3):
            gr = dev(flat(f"g/{step}"))
            ops.grad_norm_clip(gr, 1.0, max_norm, nc)
            ref_norm = float(g[f"{name}/norms"][step])
            assert abs(nc[0].item() - ref_norm) <= 2e-6 * ref_norm
            assert (nc[1].item() == 1.0) == (ref_norm + 1e-6 <= max_norm)
            if name == "adam":
                ops.adam_step(p, gr, m, v, step + 1, 2e-4, wd=1e-4)
            else:
                ops.sgd_step(p, gr, m, 0.01, first=(step == 0))
            ops.ema_update(e, p, decay)
            assert (p.cpu() - flat(f"{name}/p/{step}")).abs().max() < 5e-7, (name, step)
            assert (e.cpu() - flat(f"{name}/e/{step}")).abs().max() < 5e-7, (name, step)
    n = 7865884 + 36                                         # the README SwinIR's flat gradient
    gr = rnd(n) * 1e-3
    go = [gr.clone()]
    total, coef = O.clip_grad_norm(go, 0.5)
    d = dev(gr)
    nc = torch.zeros(2).cuda()
    ops.grad_norm_clip(d, 1.0, 0.5, nc)
    # the kernel sums the squares in double; torch's own f32 vector_norm over ONE 7.9 M-element tensor carries ~2e-4 of
    # accumulation error (the reference clips 330 smaller tensors: norm of norms).  Gate against the f64 norm; the f32
    # restatement only has to agree to ITS accuracy
    total64 = float(gr.double().norm())
    coef64 = min(1.0, 0.5 / (total64 + 1e-6))
    assert abs(nc[0].item() - total64) <= 2e-6 * total64 and abs(nc[1].item() - coef64) <= 1e-6
    assert abs(nc[0].item() - float(total)) <= 1e-3 * float(total)
    assert (d.cpu() - gr * coef64).abs().max() <= 1e-6 * gr.abs().max()
    d2 = dev(gr * 4)                                          # the SUM of four ranks' gradients: norm of g / 4
    ops.grad_norm_clip(d2, 0.25, 0.5, nc)
    assert abs(nc[0].item() - total64) <= 2e-6 * total64
    assert (d2.cpu() * 0.25 - gr * coef64).abs().max() <= 1e-6 * gr.abs().max()
    a1, a2 = dev(gr), dev(gr)                                 # bit-identical from run to run (fixed-order reduction)
    n1, n2 = torch.zeros(2).cuda(), torch.zeros(2).cuda()
    ops.grad_norm_clip(a1, 1.0, 0.5, n1)
    ops.grad_norm_clip(a2, 1.0, 0.5, n2)
    assert torch.equal(a1, a2) and torch.equal(n1, n2)
    bad = gr.clone()
    bad[12345] = float("nan")
    db = dev(bad)
    ops.grad_norm_clip(db, 1.0, 0.5, nc)
    assert math.isnan(nc[0].item()) and math.isnan(nc[1].item()) and torch.isnan(db).all()     # clip_grad_norm_'s behaviour
    e0 = rnd(1000)
    e, p = dev(e0), dev(rnd(1000))
    flag = torch.ones(1, dtype=torch.int32).cuda()
    ops.ema_update(e, p, 0.999, flag)
    assert torch.equal(e.cpu(), e0)
    flag.zero_()
    ops.ema_update(e, p, 0.999, flag)
    eo = [e0.clone()]
    O.ema_update(eo, [p.cpu()], 0.999)
    assert (e.cpu() - eo[0]).abs().max() <= 5e-7          # values O(1): two roundings each side
    ops.ema_update(e, p, 0.0)                                 # update_E(0): a copy (model_plain.py:82-84)
    assert torch.equal(e, p)
    with pytest.raises(RuntimeError):
        ops.grad_norm_clip(dev(gr), 1.0, 0.0, nc)             # max_norm must be positive


def test_errors_are_loud(ops):
    with pytest.raises(RuntimeError):
        ops.gemm_nt(torch.zeros(4, 6).cuda(), torch.zeros(4, 6).cuda())       # K % 4 != 0
    with pytest.raises(RuntimeError):
        ops.gemm_nt(torch.zeros(4, 8), torch.zeros(4, 8))                      # CPU tensors
    with pytest.raises(RuntimeError):
        ops.window_attention_fwd(torch.zeros(64, 90).cuda(), torch.zeros(64, 30).cuda(),
                                 torch.zeros(6, 64, 64).cuda(), 1, 8, 8, 30, 6, 4)  # shift on 8x8
