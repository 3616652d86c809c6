"""bf16x3 split-MFMA contractions (srhip_gemm_nt_bx3 / srhip_conv3x3_nhwc_bx3) against
float64 aten and against the exact-f32 MFMA kernels: same prologues / epilogues,
f32-level accuracy (the split keeps 24 significant bits per operand)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from srhip import ops as o
    return o


G = torch.Generator().manual_seed(4321)


def rnd(*shape, scale=1.0):
    return torch.randn(*shape, generator=G) * scale


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def test_split_is_exact(ops):
    """h + m + l reproduces the f32 value to <= 1 ulp of the 24-bit mantissa."""
    W = rnd(180, 180) * torch.exp(rnd(180, 180) * 3)
    bx = ops.split_bf16x3(W.cuda())
    kp = bx.planes.shape[-1]
    assert kp == 192 and bx.planes.shape == (3, 180, 192)
    parts = (bx.planes.cpu().to(torch.int32) << 16).view(torch.float32).double()  # bf16 bits -> f32
    # storage is 16-k sub-chunk major [3][Kp/16][rows][16]: back to [3][rows][Kp]
    parts = parts.reshape(3, kp // 16, 180, 16).permute(0, 2, 1, 3).reshape(3, 180, kp)
    rec = parts.sum(0)
    assert rec[:, 180:].abs().max() == 0
    err = ((rec[:, :180] - W.double()).abs() / W.double().abs().clamp_min(1e-30)).max().item()
    assert err <= 2.0 ** -23, err


@pytest.mark.parametrize("M,N,K", [(300, 180, 180), (4096, 540, 180), (1000, 60, 60), (777, 64, 64),
                                   (2048, 360, 180), (515, 180, 360), (64, 120, 60), (130, 256, 64),
                                   (33000, 180, 180), (999, 180, 540), (100, 48, 20),
                                   (140000, 64, 64),    # tall + narrow: the 128-row tiles of gemm_ntb.hip
                                   # register-resident-W kernel (gemm_ntr.hip: K <= 192, M >= 4096): ragged rows,
                                   # K not a multiple of 32, widths 180 / 360 / 540 and a single <= 192 block
                                   (4133, 540, 180), (16384, 360, 180), (8200, 180, 168), (5000, 128, 180),
                                   (4096, 192, 192), (6000, 100, 100),
                                   # k_ntw (gemm_ntw.hip, 192-column tiles): ragged last tile, one / two / odd stage counts,
                                   # K tails inside an 8-k octet, fewer rows than a tile
                                   (1000, 200, 64), (515, 448, 64), (300, 192, 32), (257, 196, 100), (40, 180, 84),
                                   (2049, 724, 212)])
def test_gemm_bx3_matches_f32(ops, M, N, K):
    A, W, b = rnd(M, K), rnd(N, K, scale=0.1), rnd(N)
    ref = F.linear(A.double(), W.double(), b.double())
    f32 = ops.gemm_nt(A.cuda(), W.cuda(), b.cuda())
    bx = ops.gemm_nt(A.cuda(), ops.split_bf16x3(W.cuda()), b.cuda())
    e32, ebx = relerr(f32, ref), relerr(bx, ref)
    assert ebx <= max(2.0 * e32, 1e-6), f"bx3 {ebx:.3e} vs f32 kernel {e32:.3e}"


@pytest.mark.parametrize("M", [1024, 8192])       # 8192 rows: the register-resident-W kernel (gemm_ntr.hip)
def test_gemm_bx3_prologues_epilogues(ops, M):
    N, K = 180, 180
    A, W, b, R = rnd(M, K), rnd(N, K, scale=0.1), rnd(N), rnd(M, N)
    Wb = ops.split_bf16x3(W.cuda())
    d = lambda t: t.cuda()
    mean = A.mean(1)
    rstd = 1.0 / torch.sqrt(A.var(1, unbiased=False) + 1e-5)
    stats = torch.stack([mean, rstd], 1)
    tol = 2e-6
    out = ops.gemm_nt(d(A), Wb, d(b), a_mode=1, ln_stats=d(stats))
    assert relerr(out, F.linear(((A - mean[:, None]) * rstd[:, None]).double(), W.double(), b.double())) < tol
    out = ops.gemm_nt(d(A), Wb, d(b), a_mode=2)
    assert relerr(out, F.linear(F.gelu(A.double()), W.double(), b.double())) < tol
    out = ops.gemm_nt(d(A), Wb, d(b), epi=1)
    assert relerr(out, F.relu(F.linear(A.double(), W.double(), b.double()))) < tol
    rs = torch.tensor([1.0, 0.0, 1.25, 2.0])
    out = ops.gemm_nt(d(A), Wb, d(b), epi=2, R=d(R), rowscale=d(rs), rows_per_scale=M // 4, alpha=0.5)
    ref = R.double() + 0.5 * rs.double().repeat_interleave(M // 4)[:, None] * F.linear(A.double(), W.double(), b.double())
    assert relerr(out, ref) < tol
    Rg = R.double().clone().requires_grad_(True)
    F.gelu(Rg).sum().backward()
    out = ops.gemm_nt(d(A), Wb, None, epi=3, R=d(R), rowscale=d(rs), rows_per_scale=M // 4)
    ref = rs.double().repeat_interleave(M // 4)[:, None] * F.linear(A.double(), W.double()) * Rg.grad
    assert relerr(out, ref) < tol
    aux = torch.empty(M, N).cuda()
    out2 = ops.gemm_nt(d(A), Wb, None, epi=3, R=d(R), rowscale=d(rs), rows_per_scale=M // 4, aux=aux)
    assert torch.equal(out2, out) and relerr(aux, F.gelu(R.double())) < tol
    out = ops.gemm_nt(d(A), Wb, None, epi=4, R=d(R))
    assert relerr(out, F.linear(A.double(), W.double()) * (R > 0)) < tol
    wide = torch.zeros(M, 540).cuda()
    ops.gemm_nt(d(A), Wb, d(b), out=wide[:, 180:360])
    assert relerr(wide[:, 180:360], F.linear(A.double(), W.double(), b.double())) < tol
    assert wide[:, :180].abs().max() == 0 and wide[:, 360:].abs().max() == 0
    # A as a column slice of a wider buffer (qkv -> q)
    Aw = rnd(M, 540)
    out = ops.gemm_nt(d(Aw)[:, 180:360], Wb, d(b))
    assert relerr(out, F.linear(Aw[:, 180:360].double(), W.double(), b.double())) < tol


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 16, 16, 180, 180), (1, 24, 40, 60, 60), (1, 20, 12, 64, 64),
                                         (1, 16, 16, 64, 256), (2, 64, 64, 180, 64), (1, 9, 7, 16, 16),
                                         (8, 64, 64, 180, 180), (8, 128, 128, 64, 64), (8, 128, 128, 64, 256)])
def test_conv_bx3_matches_f32(ops, B, H, W, Ci, Co):
    x, w, b = rnd(B, Ci, H, W), rnd(Co, Ci, 3, 3, scale=0.05), rnd(Co)
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    wp = torch.empty(9, Co, Ci).cuda()
    wpt = torch.empty(9, Ci, Co).cuda()
    ops.pack_conv_weight(w.cuda(), wp, wpt)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    y32 = ops.conv3x3(xh, wp, b.cuda(), Co)
    ybx = ops.conv3x3(xh, ops.split_bf16x3(wp), b.cuda(), Co)
    e32, ebx = relerr(y32.permute(0, 3, 1, 2), ref), relerr(ybx.permute(0, 3, 1, 2), ref)
    assert ebx <= max(2.0 * e32, 2e-6), f"conv bx3 {ebx:.3e} vs f32 kernel {e32:.3e}"      # (1e-6 failed on some draws: the operands come from one shared generator)
    dy = rnd(B, Co, H, W)
    dyh = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dx = ops.conv3x3(dyh, ops.split_bf16x3(wpt), None, Ci)
    xr = x.double().clone().requires_grad_(True)
    F.conv2d(xr, w.double(), b.double(), padding=1).backward(dy.double())
    # f32 accumulation over up to 9 x 256 products; 2e-6 failed on one draw of the shared generator at Co = 256
    assert relerr(dx.permute(0, 3, 1, 2), xr.grad) < 4e-6


def test_conv_bx3_epilogues(ops):
    B, H, W, C = 2, 16, 24, 64
    x, w, b, R = rnd(B, C, H, W), rnd(C, C, 3, 3, scale=0.05), rnd(C), rnd(B, C, H, W)
    xh, Rh = x.permute(0, 2, 3, 1).contiguous().cuda(), R.permute(0, 2, 3, 1).contiguous().cuda()
    wp = torch.empty(9, C, C).cuda()
    ops.pack_conv_weight(w.cuda(), wp, None)
    wb = ops.split_bf16x3(wp)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    y = ops.conv3x3(xh, wb, b.cuda(), C, epi=1)
    assert relerr(y.permute(0, 3, 1, 2), F.relu(ref)) < 2e-6
    y = ops.conv3x3(xh, wb, b.cuda(), C, epi=2, R=Rh, alpha=0.1)
    assert relerr(y.permute(0, 3, 1, 2), R.double() + 0.1 * ref) < 2e-6
    y = ops.conv3x3(xh, wb, None, C, epi=4, R=Rh)
    assert relerr(y.permute(0, 3, 1, 2), F.conv2d(x.double(), w.double(), None, padding=1) * (R > 0)) < 2e-6


def test_prep_table_matches_single_ops(ops, monkeypatch):
    """One-launch weight preparation == fold_layernorm / transpose / pack_conv_weight /
    bias_expand / split_bf16x3 applied weight by weight (bit-exact).  The bf16x3 form of the Linear planes (the fp16x2
    form the table emits by default for 192-column GEMMs is checked in tests/test_gpu_fallback_kernels.py)."""
    monkeypatch.setattr(ops, "F16X2", False)
    monkeypatch.setattr(ops, "F16X2_CONV", False)
    N, K, heads = 540, 180, 6
    W, b, g, be = rnd(N, K, scale=0.1).cuda(), rnd(N).cuda(), (1 + 0.1 * rnd(K)).cuda(), rnd(K, scale=0.1).cuda()
    cw = rnd(64, 180, 3, 3, scale=0.05).cuda()
    tab = rnd(225, heads).cuda()
    tb = ops.PrepTable()
    o = {k: ops.Bx3(*s, "cuda") for k, s in dict(w=(N, K), wT=(K, N), f=(N, K), fT=(K, N), cp=(9 * 64, 180),
                                                 cpt=(9 * 180, 64)).items()}
    bq = torch.empty(N).cuda()
    bT, bN = torch.empty(heads, 64, 64).cuda(), torch.empty(heads, 64, 64).cuda()
    tb.linear(W, o["w"]); tb.linear(W, o["wT"], transpose=True)
    tb.linear(W, o["f"], gamma=g); tb.linear(W, o["fT"], gamma=g, transpose=True)
    tb.conv(cw, o["cp"]); tb.conv(cw, o["cpt"], data_grad=True)
    tb.fold_bias(W, b, be, bq); tb.bias_expand(tab, bT, bN, heads)
    tb.build("cuda").run()
    Wf, bf = torch.empty(N, K).cuda(), torch.empty(N).cuda()
    ops.fold_layernorm(W, b, g, be, Wf, bf)
    WT, WfT = torch.empty(K, N).cuda(), torch.empty(K, N).cuda()
    ops.transpose(W, WT); ops.transpose(Wf, WfT)
    wp, wpt = torch.empty(9, 64, 180).cuda(), torch.empty(9, 180, 64).cuda()
    ops.pack_conv_weight(cw, wp, wpt)
    rT, rN = torch.empty(heads, 64, 64).cuda(), torch.empty(heads, 64, 64).cuda()
    ops.bias_expand(tab, rT, rN)
    for k, ref in dict(w=W, wT=WT, f=Wf, fT=WfT, cp=wp, cpt=wpt).items():
        assert torch.equal(o[k].planes, ops.split_bf16x3(ref).planes), k
    assert (bq - bf).abs().max().item() < 1e-6
    assert torch.equal(bT, rT) and torch.equal(bN, rN)


@pytest.mark.parametrize("M,N,K", [(1024, 180, 360), (777, 180, 540), (300, 64, 64), (2048, 120, 180), (32768, 180, 360)])
def test_gemm_lnbwd_fused(ops, M, N, K):
    """GEMM + LayerNorm backward in one kernel == aten autograd of LayerNorm (no affine) applied to x,
    upstream gradient dxh = A @ W^T, plus the residual gradient."""
    A, W = rnd(M, K), rnd(N, K, scale=0.1)
    x, res = rnd(M, N) * 2 + 0.5, rnd(M, N)
    xd = x.double().requires_grad_(True)
    y = F.layer_norm(xd, (N,), eps=1e-5)
    dxh = F.linear(A.double(), W.double())
    y.backward(dxh)
    ref = res.double() + xd.grad
    stats = torch.empty(M, 2).cuda()
    ops.layernorm_fwd(x.cuda(), stats)
    out = torch.empty(M, N).cuda()
    ops.gemm_nt_lnbwd(A.cuda(), ops.split_bf16x3(W.cuda()), x.cuda(), stats, res.cuda(), out)
    assert relerr(out, ref) < 5e-6
    out2 = torch.full((M, N), float("nan")).cuda()
    ops.gemm_nt_lnbwd(A.cuda(), ops.split_bf16x3(W.cuda()), x.cuda(), stats, None, out2)
    assert relerr(out2, xd.grad) < 5e-6


@pytest.mark.parametrize("M,N,K", [(1024, 180, 180), (700, 180, 360), (300, 64, 64), (555, 120, 60),
                                   (8192, 180, 180), (4101, 180, 168), (4500, 120, 180)])   # the last three: gemm_ntr.hip
def test_gemm_row_stats(ops, M, N, K):
    """stats_out of the GEMM epilogue == layernorm_fwd statistics of the GEMM output."""
    A, W, b, R = rnd(M, K), rnd(N, K, scale=0.1), rnd(N), rnd(M, N) * 3 + 1
    Wb = ops.split_bf16x3(W.cuda())
    st = torch.full((M, 2), float("nan")).cuda()
    out = ops.gemm_nt(A.cuda(), Wb, b.cuda(), epi=2, R=R.cuda(), alpha=0.7, stats_out=st)
    ref = R.double() + 0.7 * F.linear(A.double(), W.double(), b.double())
    assert relerr(out, ref) < 2e-6
    mean = ref.mean(1)
    rstd = 1.0 / torch.sqrt(ref.var(1, unbiased=False) + 1e-5)
    assert relerr(st[:, 0], mean) < 5e-6 and relerr(st[:, 1], rstd) < 5e-6
    st2 = torch.empty(M, 2).cuda()
    ops.layernorm_fwd(out, st2)
    assert (st - st2).abs().max().item() < 1e-5


@pytest.mark.parametrize("B,H,W,Ci,Co", [(1, 12, 20, 128, 128), (2, 9, 7, 128, 192), (1, 16, 16, 256, 128), (3, 8, 8, 180, 180),
                                         # 192-column tiles on two fp16 planes (k_tnb_hc): ragged column tuples (184, 200 are not
                                         # multiples of 3), rows that are no multiple of the chunk, several slices and tiles
                                         (2, 9, 7, 180, 184), (1, 13, 11, 200, 192), (1, 40, 56, 180, 180), (2, 24, 24, 384, 180)])
def test_conv_wgrad_bx3_borders_and_tails(ops, B, H, W, Ci, Co):
    """conv weight gradient (bf16x3 / fp16x2 bodies) on images whose rows are not multiples of the 32-token chunk
    (border taps, partial last chunk, ragged slices) against float64 autograd."""
    x, w, dy = rnd(B, Ci, H, W), rnd(Co, Ci, 3, 3, scale=0.05), rnd(B, Co, H, W)
    wr = w.double().clone().requires_grad_(True)
    br = torch.zeros(Co, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wr, br, padding=1).backward(dy.double())
    dW, db = torch.empty(Co, Ci, 3, 3).cuda(), torch.empty(Co).cuda()
    ops.conv3x3_wgrad(dy.permute(0, 2, 3, 1).contiguous().cuda(), x.permute(0, 2, 3, 1).contiguous().cuda(), dW, db)
    assert relerr(dW, wr.grad) < 2e-6 and relerr(db, br.grad) < 2e-6


@pytest.mark.parametrize("M", [31, 100, 4097, 33000])
def test_linear_wgrad_bx3_ragged_rows(ops, M):
    """grouped bf16x3 weight gradient with row counts that are not multiples of the chunk / slice sizes."""
    NI, NJ = 180, 360
    dY, X = rnd(M, NI), rnd(M, NJ)
    rs = torch.rand(M // 16 + 1, generator=G) + 0.5
    st = torch.stack([X.mean(1), 1 / torch.sqrt(X.var(1, unbiased=False) + 1e-5)], 1).contiguous()
    dW, db = torch.empty(NI, NJ).cuda(), torch.empty(NI).cuda()
    dW2, db2 = torch.empty(NI, NJ).cuda(), torch.empty(NI).cuda()
    ops.linear_wgrad_grouped([
        dict(dY=dY.cuda(), X=X.cuda(), dW=dW, db=db, a_rowscale=rs.cuda(), a_rowscale_rows=16),
        dict(dY=dY.cuda(), X=X.cuda(), dW=dW2, db=db2, b_mode=1, ln_stats=st.cuda())])
    dYs = dY.double() * rs.double().repeat_interleave(16)[:M, None]
    assert relerr(dW, dYs.t() @ X.double()) < 2e-6 and relerr(db, dYs.sum(0)) < 2e-6
    Xn = (X.double() - st[:, :1].double()) * st[:, 1:].double()
    assert relerr(dW2, dY.double().t() @ Xn) < 2e-6 and relerr(db2, dY.double().sum(0)) < 2e-6


@pytest.mark.parametrize("n,B,H,W,C", [(3, 2, 16, 16, 64), (5, 1, 20, 12, 64), (33, 1, 24, 40, 64), (2, 1, 9, 7, 128)])
def test_conv_wgrad_batched_matches_float64(ops, n, B, H, W, C):
    """srhip_conv3x3_wgrad_batched_bx3: n same-shape problems in one launch (EDSR body, deferred weight
    gradients) against float64 autograd, ragged borders and odd sizes included; outputs overwritten."""
    items, refs = [], []
    for k in range(n):
        x, dy = rnd(B, C, H, W), rnd(B, C, H, W)
        xr = x.double()
        wr = torch.zeros(C, C, 3, 3, dtype=torch.float64, requires_grad=True)
        br = torch.zeros(C, dtype=torch.float64, requires_grad=True)
        F.conv2d(xr, wr, br, padding=1).backward(dy.double())
        refs.append((wr.grad, br.grad))
        items.append((dy.permute(0, 2, 3, 1).contiguous().cuda(), x.permute(0, 2, 3, 1).contiguous().cuda(),
                      torch.full((C, C, 3, 3), 7.0).cuda(), torch.full((C,), 7.0).cuda()))
    ops.conv3x3_wgrad_batched(items)
    for (_, _, dw, db), (rw, rb) in zip(items, refs):
        assert relerr(dw, rw) < 2e-6 and relerr(db, rb) < 2e-6


def _conv_wgrad_ref(x, dy):
    Co, Ci = dy.shape[1], x.shape[1]
    wr = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    br = torch.zeros(Co, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wr, br, padding=1).backward(dy.double())
    return wr.grad, br.grad


def _row_relerr(dW, ref):
    """largest error of a dW row (one output channel) relative to that row's largest entry"""
    a, b = dW.detach().double().cpu().flatten(1), ref.flatten(1)
    return ((a - b).abs().amax(1) / b.abs().amax(1).clamp_min(1e-300)).max().item()


@pytest.mark.parametrize("B,H,W,Ci,Co,ps2", [(1, 5, 64, 64, 64, False), (2, 7, 128, 64, 64, False), (3, 4, 64, 64, 256, True),
                                              (1, 9, 192, 64, 256, True), (2, 33, 64, 64, 128, False), (1, 3, 256, 128, 64, False),
                                              # channel counts that are no multiple of 64 (SwinIR's convs): the last 64-column
                                              # tile is partly empty
                                              (2, 6, 64, 180, 180, False), (1, 5, 128, 180, 64, False), (2, 4, 64, 64, 180, False),
                                              (1, 8, 64, 128, 96, False), (1, 4, 64, 100, 256, True), (2, 5, 128, 68, 72, False)])
def test_conv_wgrad_strip_form(ops, B, H, W, Ci, Co, ps2):
    """The strip form of the nine-tap conv weight gradient (k_tnb9s: image width a multiple of 64; dY staged once per row with
    a one-pixel halo and shifted in registers, X rows kept in a ring): several strips (the halo comes from the neighbour
    strip), image borders inside a slice's row range (batch > 1), slices of one or two rows, the PixelShuffle form -- every
    row of dW against float64 autograd."""
    x, dy = rnd(B, Ci, H, W), rnd(B, Co, H, W)
    rw, rb = _conv_wgrad_ref(x, dy)
    dW, db = torch.full((Co, Ci, 3, 3), 7.0).cuda(), torch.full((Co,), 7.0).cuda()
    dyn = dy.permute(0, 2, 3, 1).contiguous()
    if ps2:       # the gradient of the PixelShuffle(2) output: [B, 2H, 2W, Co/4], channel c*4 + (2i + j) at pixel (2y+i, 2x+j)
        dyn = F.pixel_shuffle(dy, 2).permute(0, 2, 3, 1).contiguous()
    ops.conv3x3_wgrad(dyn.cuda(), x.permute(0, 2, 3, 1).contiguous().cuda(), dW, db, ps2=ps2)
    assert _row_relerr(dW, rw) < 2e-6 and relerr(db, rb) < 2e-6


def test_conv_wgrad_strip_form_exponents_are_checked(ops):
    """The block fixes a column's power-of-two scale from its first row and checks the column's true maximum at the end:
    (a) a channel that is zero in the first rows and 1e-4 of the others later, (b) pixels 1e6 times larger than the first
    row's further down, (c) a channel 1e-6 of the others throughout, (d) all-zero channels -- every row of dW, relative to
    its own largest entry, stays at f32 grade (the block runs a second time with exact scales where the first guess fails)."""
    B, H, W, C = 1, 40, 64, 64
    x, dy = rnd(B, C, H, W), rnd(B, C, H, W)
    x[:, 3, :6] = 0
    x[:, 3, 6:] *= 1e-4
    dy[:, 5, :6] = 0
    dy[:, 5, 6:] *= 1e-4
    dy[:, :, 25:] *= 1e6
    x[:, 9] *= 1e-6
    dy[:, 11] *= 1e-6
    x[:, 20] = 0
    dy[:, 21] = 0
    rw, rb = _conv_wgrad_ref(x, dy)
    dW, db = torch.empty(C, C, 3, 3).cuda(), torch.empty(C).cuda()
    ops.conv3x3_wgrad(dy.permute(0, 2, 3, 1).contiguous().cuda(), x.permute(0, 2, 3, 1).contiguous().cuda(), dW, db)
    # rows: relative to the row's largest entry; the all-zero dY channel's row is exactly zero
    assert dW[21].abs().max().item() == 0 and dW[:, 20].abs().max().item() == 0
    keep = [i for i in range(C) if i != 21]
    assert _row_relerr(dW[keep], rw[keep]) < 3e-6
    # columns (input channels) too: the small X channels against their own scale
    a, b = dW.detach().double().cpu().permute(1, 0, 2, 3).flatten(1), rw.permute(1, 0, 2, 3).flatten(1)
    keepc = [j for j in range(C) if j != 20]
    assert ((a[keepc] - b[keepc]).abs().amax(1) / b[keepc].abs().amax(1)).max().item() < 3e-6
    assert relerr(db, rb) < 2e-6


def test_conv_wgrad_batched_strip_form_second_pass(ops):
    """The batched launch on 64-wide images (strip form) with items whose exponents do not hold in the first pass (pixels that
    grow by 1e6 down the image, a channel far below the others): the flagged blocks run again in the second pass, every item's
    partial sums stay its own (the per-block words sit behind ALL items' partial sums), every row of dW at f32 grade."""
    n, B, H, W, C = 5, 2, 24, 64, 64
    items, refs = [], []
    for k in range(n):
        x, dy = rnd(B, C, H, W), rnd(B, C, H, W)
        if k % 2 == 1:
            dy[:, :, 10:] *= 1e6
            x[:, 7, :4] = 0
            x[:, 7, 4:] *= 1e-5
        refs.append(_conv_wgrad_ref(x, dy))
        items.append((dy.permute(0, 2, 3, 1).contiguous().cuda(), x.permute(0, 2, 3, 1).contiguous().cuda(),
                      torch.full((C, C, 3, 3), 7.0).cuda(), torch.full((C,), 7.0).cuda()))
    ops.conv3x3_wgrad_batched(items)
    for (_, _, dw, db), (rw, rb) in zip(items, refs):
        assert _row_relerr(dw, rw) < 3e-6 and relerr(db, rb) < 2e-6
        a, b = dw.detach().double().cpu().permute(1, 0, 2, 3).flatten(1), rw.permute(1, 0, 2, 3).flatten(1)
        assert ((a - b).abs().amax(1) / b.abs().amax(1)).max().item() < 3e-6
