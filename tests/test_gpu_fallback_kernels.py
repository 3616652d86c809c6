"""The LDS-staged kernels that the W-direct ones replaced by default (k_ntp<3>, k_ntb<1,3>, k_ntb<2,1>, k_ntb<1,1>) stay
selectable (SRHIP_NTW=0, SRHIP_NTCW=0, SRHIP_NTCW2=0, SRHIP_NTCW2_SMALL=0: same-box A/Bs, tools/ab_ntw.sh / ab_ntcw.sh) --
the switches are read once per process, so each arm runs in a child process and checks a GEMM with every epilogue family
and three conv shapes against float64."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Switches read by the C side (common.h: sr_getenv) are live only in an experiments build (make EXPERIMENTS=1 ->
# sr-caco-2_amd/lib/libsrhip_exp.so, built by __graft_entry__.build()); the shipped library ignores them.  An arm that sets
# one runs on that build (SRHIP_LIB) and its child asserts that the build really reads them; Python-side switches
# (srhip/ops.py, the engines) work with either library.
C_SIDE = {"SRHIP_NTW", "SRHIP_NTW_GRID", "SRHIP_NTW_ROT", "SRHIP_TN_F16X2", "SRHIP_TN_F16X2_LINEAR", "SRHIP_TN_GROUP_XCD",
          "SRHIP_TN_T3", "SRHIP_TN_XCD", "SRHIP_WA_XCD", "SRHIP_TN_BLOCKS", "SRHIP_WMSA_NW"}
EXP_LIB = os.path.join(ROOT, "sr-caco-2_amd", "lib", "libsrhip_exp.so")
ASSERT_EXP = "from srhip import ops as _o\nassert _o.lib.srhip_experiments_enabled() == 1, 'the loaded library ignores C-side switches'\n"


def child_env(env):
    """(environment, code prefix) of an arm's child process."""
    if not (set(env) & C_SIDE):
        return dict(os.environ, **env), ""
    if not os.path.isfile(EXP_LIB):
        pytest.skip(f"{sorted(set(env) & C_SIDE)} are read by an experiments build only and {EXP_LIB} is absent "
                    "(make -C sr-caco-2_amd/csrc EXPERIMENTS=1 OUT=../lib/libsrhip_exp.so OBJDIR=../lib/obj_exp)")
    return dict(os.environ, SRHIP_LIB=EXP_LIB, **env), "import sys\nsys.path.insert(0, 'sr-caco-2_amd')\n" + ASSERT_EXP

CHILD = textwrap.dedent('''
    import sys
    sys.path.insert(0, "sr-caco-2_amd")
    import torch, torch.nn.functional as F
    from srhip import ops
    g = torch.Generator().manual_seed(5)
    rnd = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
    rel = lambda a, b: ((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max()).item()
    # GEMM 180 x 360 with residual + row statistics, and 540 x 180 with the LayerNorm prologue
    M = 4133
    A, W, b, R = rnd(M, 360), rnd(180, 360, scale=0.1), rnd(180), rnd(M, 180)
    st = torch.empty(M, 2, device="cuda")
    y = ops.gemm_nt(A.cuda(), ops.split_bf16x3(W.cuda()), b.cuda(), epi=2, R=R.cuda(), stats_out=st)
    ref = R.double() + F.linear(A.double(), W.double(), b.double())
    assert rel(y, ref) < 2e-6, rel(y, ref)
    assert rel(st[:, 0], ref.mean(1)) < 1e-5
    A2, W2 = rnd(M, 180) * 2 + 0.3, rnd(540, 180, scale=0.1)
    s2 = torch.stack([A2.mean(1), 1 / torch.sqrt(A2.var(1, unbiased=False) + 1e-5)], 1).contiguous()
    y2 = ops.gemm_nt(A2.cuda(), ops.split_bf16x3(W2.cuda()), None, a_mode=1, ln_stats=s2.cuda())
    ref2 = F.linear((A2.double() - s2[:, :1].double()) * s2[:, 1:].double(), W2.double())
    assert rel(y2, ref2) < 2e-6, rel(y2, ref2)
    # convs: 180 -> 180 at 8 x 64 x 64 (64-pixel x 192-column tiles), 64 -> 64 at 8 x 128 x 128 and 2 x 24 x 40
    for B, H, Wd, Ci, Co in ((8, 64, 64, 180, 180), (8, 128, 128, 64, 64), (2, 24, 40, 64, 64)):
        x, w, bb = rnd(B, Ci, H, Wd), rnd(Co, Ci, 3, 3, scale=0.05), rnd(Co)
        wp = torch.empty(9, Co, Ci, device="cuda"); wpt = torch.empty(9, Ci, Co, device="cuda")
        ops.pack_conv_weight(w.cuda(), wp, wpt)
        yc = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().cuda(), ops.split_bf16x3(wp), bb.cuda(), Co, epi=1)
        refc = F.relu(F.conv2d(x.double(), w.double(), bb.double(), padding=1))
        assert rel(yc.permute(0, 3, 1, 2), refc) < 3e-6, (B, H, Wd, Ci, Co, rel(yc.permute(0, 3, 1, 2), refc))
    print("ok")
''')


@pytest.mark.parametrize("env", [{}, {"SRHIP_NTW": "0"}, {"SRHIP_F16X2": "0", "SRHIP_F16X2_CONV": "0"}, {"SRHIP_NTCW": "0", "SRHIP_NTCW2": "0", "SRHIP_NTCW2_SMALL": "0"},
                                 {"SRHIP_NTW_GRID": "0", "SRHIP_NTW_ROT": "0", "SRHIP_NTCW2_WIDE": "0"}])
def test_switchable_kernels_match_float64(env):
    cenv, pre = child_env(env)
    r = subprocess.run([sys.executable, "-c", pre + CHILD], cwd=ROOT, env=cenv, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (env, r.stdout[-500:], r.stderr[-1500:])


F16_CHILD = textwrap.dedent('''
    import sys
    sys.path.insert(0, "sr-caco-2_amd")
    import torch, torch.nn.functional as F
    from srhip import ops
    assert ops.F16X2
    torch.manual_seed(0)
    for (M, N, K, a_mode) in ((256, 180, 64, 0), (300, 180, 180, 0), (4096, 540, 180, 1), (515, 180, 360, 2), (4133, 360, 180, 0)):
        A = (torch.randn(M, K) * 0.7 + 0.1) * torch.exp(torch.randn(M, 1) * 3)        # rows 1e-4 .. 1e4 apart
        W = torch.randn(N, K) * 0.05 * torch.exp(torch.randn(N, 1))
        b = torch.randn(N) * 0.0
        out = ops.Bx3(N, K, "cuda")
        tb = ops.PrepTable(); Wc = W.cuda().contiguous(); tb.linear(Wc, out); tb.build("cuda").run()
        st = torch.stack([A.mean(1), 1 / torch.sqrt(A.var(1, unbiased=False) + 1e-5)], 1).contiguous()
        y = ops.gemm_nt(A.cuda(), out, b.cuda(), a_mode=a_mode, ln_stats=st.cuda() if a_mode == 1 else None)
        Ad = A.double()
        if a_mode == 1: Ad = (Ad - st[:, :1].double()) * st[:, 1:].double()
        if a_mode == 2: Ad = F.gelu(Ad)
        ref = F.linear(Ad, W.double(), b.double())
        f32 = F.linear(Ad.float(), W, b).double()
        # every ROW against itself: the per-row block exponent keeps small rows as accurate as large ones
        e = ((y.double().cpu() - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-300)).max().item()
        e32 = ((f32 - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-300)).max().item()
        assert e <= max(3.0 * e32, 1e-6), (M, N, K, a_mode, e, e32)
    print("ok")
''')


def test_fp16x2_three_product_gemm_is_f32_grade_per_row():
    """The default Linear path (k_nth2: two fp16 planes, per-row power-of-two scales, three products) and its pre-pass form
    (k_nth, SRHIP_F16X2_PASSES=0): the worst ROW, relative to itself, is within 3x of what a plain f32 matmul gives, on
    operands whose rows are 8 decades apart."""
    for extra in ({}, {"SRHIP_F16X2_PASSES": "0"}):
        r = subprocess.run([sys.executable, "-c", F16_CHILD], cwd=ROOT, env=dict(os.environ, SRHIP_F16X2="1", **extra),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (extra, r.stdout[-500:], r.stderr[-1500:])


CONV16_CHILD = textwrap.dedent('''
    import sys
    sys.path.insert(0, "sr-caco-2_amd")
    import torch, torch.nn.functional as F
    from srhip import ops
    assert ops.F16X2_CONV
    torch.manual_seed(0)
    for (B, H, W, Ci, Co, kind) in ((2, 24, 40, 64, 64, "feat"), (1, 9, 7, 64, 64, "feat"), (8, 128, 128, 64, 64, "grad"),
                                    (2, 64, 64, 128, 64, "feat"), (8, 64, 64, 64, 64, "grad"),
                                    (2, 64, 64, 180, 180, "feat"), (1, 9, 7, 180, 180, "grad"), (2, 40, 24, 180, 180, "grad")):
        if kind == "feat":
            x = F.relu(torch.randn(B, Ci, H, W)) * torch.exp(torch.randn(B, 1, 1, 1))
        else:       # gradient-like: tiny, and every pixel at its own scale (e^2.5N apart inside a halo tile)
            x = torch.randn(B, Ci, H, W) * 1e-7 * torch.exp(torch.randn(B, 1, H, W) * 2.5)
        w = torch.randn(Co, Ci, 3, 3) * (2.0 / (9 * Ci)) ** 0.5 * torch.exp(torch.randn(Co, 1, 1, 1))
        b = torch.randn(Co) * (0.1 if kind == "feat" else 0.0)
        wp, wpt = ops.Bx3(9 * Co, Ci, "cuda"), ops.Bx3(9 * Ci, Co, "cuda")
        tb = ops.PrepTable(); wc = w.cuda().contiguous(); tb.conv(wc, wp); tb.conv(wc, wpt, data_grad=True); tb.build("cuda").run()
        assert wp.fmt == 1 and wpt.fmt == 1
        pix = lambda t, ref: ((t - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-300)).max().item()
        y = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().cuda(), wp, b.cuda(), Co).permute(0, 3, 1, 2).double().cpu()
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        e, e32 = pix(y, ref), pix(F.conv2d(x, w, b, padding=1).double(), ref)
        assert e <= max(3.0 * e32, 2e-6), ("fwd", B, H, W, Ci, Co, e, e32)
        if True:
            dy = torch.randn(B, Co, H, W) * torch.exp(torch.randn(B, 1, H, W))
            dx = ops.conv3x3(dy.permute(0, 2, 3, 1).contiguous().cuda(), wpt, None, Ci).permute(0, 3, 1, 2).double().cpu()
            refd = F.conv_transpose2d(dy.double(), w.double(), padding=1)
            e, e32 = pix(dx, refd), pix(F.conv_transpose2d(dy, w, padding=1).double(), refd)
            assert e <= max(3.0 * e32, 2e-6), ("dgrad", B, H, W, Ci, Co, e, e32)
    # conv 64 -> 256 with the PixelShuffle(2) fused into the store, and its data gradient read from the shuffled gradient
    for (B, H, W, Ci, Fo) in ((2, 24, 40, 64, 64), (1, 9, 7, 64, 64), (4, 128, 128, 64, 64)):
        Co = 4 * Fo
        x = F.relu(torch.randn(B, Ci, H, W)) * torch.exp(torch.randn(B, 1, 1, 1))
        w = torch.randn(Co, Ci, 3, 3) * (2.0 / (9 * Ci)) ** 0.5 * torch.exp(torch.randn(Co, 1, 1, 1))
        b = torch.randn(Co) * 0.1
        wp, wpt = ops.Bx3(9 * Co, Ci, "cuda"), ops.Bx3(9 * Ci, Co, "cuda")
        tb = ops.PrepTable(); wc = w.cuda().contiguous(); tb.conv(wc, wp, ps2=True); tb.conv(wc, wpt, data_grad=True, ps2=True)
        tb.build("cuda").run()
        assert wp.fmt == 1 and wpt.fmt == 1
        up = torch.empty(B, 2 * H, 2 * W, Fo, device="cuda")
        ops.conv3x3_ps2(x.permute(0, 2, 3, 1).contiguous().cuda(), wp, b.cuda(), up)
        ref = F.pixel_shuffle(F.conv2d(x.double(), w.double(), b.double(), padding=1), 2)
        e = pix(up.permute(0, 3, 1, 2).double().cpu(), ref)
        e32 = pix(F.pixel_shuffle(F.conv2d(x, w, b, padding=1), 2).double(), ref)
        assert e <= max(3.0 * e32, 2e-6), ("ps2 fwd", B, H, W, e, e32)
        dyu = torch.randn(B, Fo, 2 * H, 2 * W) * 1e-6 * torch.exp(torch.randn(B, 1, 2 * H, 2 * W) * 2.0)
        dx = torch.empty(B, H, W, Ci, device="cuda")
        ops.conv3x3_ps2_bwd_data(dyu.permute(0, 2, 3, 1).contiguous().cuda(), wpt, dx)
        refd = F.conv_transpose2d(F.pixel_unshuffle(dyu.double(), 2), w.double(), padding=1)
        e = pix(dx.permute(0, 3, 1, 2).double().cpu(), refd)
        e32 = pix(F.conv_transpose2d(F.pixel_unshuffle(dyu, 2), w, padding=1).double(), refd)
        assert e <= max(3.0 * e32, 2e-6), ("ps2 dgrad", B, H, W, e, e32)
    print("ok")
''')


def test_fp16x2_three_product_conv_is_f32_grade_per_pixel():
    """The default 64-column conv path (k_nhcw2: two fp16 planes, one power-of-two scale per weight output channel and per
    activation halo tile as a running scale over the channel chunks, three products): the worst output PIXEL, relative to
    itself, is within 3x of an f32 conv -- forward and data gradient, on feature-like and on gradient-like inputs whose
    pixels are decades apart inside a tile."""
    r = subprocess.run([sys.executable, "-c", CONV16_CHILD], cwd=ROOT, env=dict(os.environ, SRHIP_F16X2_CONV="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-1500:])


TN16_CHILD = textwrap.dedent('''
    import sys
    sys.path.insert(0, "sr-caco-2_amd")
    import torch, torch.nn.functional as F
    from srhip import ops
    torch.manual_seed(0)
    relmax = lambda t, ref: ((t.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    for (B, H, W, Ci, Co, ps2, kind) in ((2, 24, 40, 64, 64, False, "flat"), (1, 9, 7, 64, 64, False, "flat"),
                                         (4, 96, 96, 64, 64, False, "grad"), (2, 64, 64, 64, 256, True, "grad"),
                                         (2, 48, 48, 128, 64, False, "rise"), (1, 40, 40, 64, 64, False, "zero")):
        x = F.relu(torch.randn(B, Ci, H, W)) * torch.exp(torch.randn(B, 1, 1, 1))
        dy = torch.randn(B, Co, H, W)
        if kind == "grad":      # tiny, every pixel at its own scale, channels decades apart
            dy = dy * 1e-7 * torch.exp(torch.randn(B, 1, H, W) * 2.5) * torch.exp(torch.randn(1, Co, 1, 1) * 3.0)
        if kind == "rise":      # magnitudes that grow by 2^40 along the token order: the running scales must follow
            ramp = torch.exp2(torch.linspace(-30, 10, B * H * W)).reshape(B, 1, H, W)
            dy, x = dy * ramp, x * ramp
        if kind == "zero":      # channels that are zero everywhere / until late
            dy[:, :5] = 0; x[:, 7] = 0; dy[:, 9, : H // 2] = 0
        wr = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
        br = torch.zeros(Co, dtype=torch.float64, requires_grad=True)
        F.conv2d(x.double(), wr, br, padding=1).backward(dy.double())
        w32 = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
        F.conv2d(x, w32, None, padding=1).backward(dy)
        dW, db = torch.empty(Co, Ci, 3, 3).cuda(), torch.empty(Co).cuda()
        xn = x.permute(0, 2, 3, 1).contiguous().cuda()
        if ps2:
            dyn = F.pixel_shuffle(dy, 2).permute(0, 2, 3, 1).contiguous().cuda()
            ops.conv3x3_wgrad(dyn, xn, dW, db, ps2=True)
        else:
            ops.conv3x3_wgrad(dy.permute(0, 2, 3, 1).contiguous().cuda(), xn, dW, db)
        # per output channel (a row of dW is one dY column: its own scale), relative to the row's largest entry
        ref = wr.grad.reshape(Co, -1)
        den = ref.abs().max(1, keepdim=True).values.clamp_min(1e-300)
        e = ((dW.double().cpu().reshape(Co, -1) - ref).abs() / den).max().item()
        e32 = ((w32.grad.double().reshape(Co, -1) - ref).abs() / den).max().item()
        assert e <= max(3.0 * e32, 2e-6), (B, H, W, Ci, Co, ps2, kind, e, e32)
        assert relmax(db, br.grad) < 2e-6
        if kind == "zero":
            assert (dW[:5] == 0).all() and (dW[:, 7] == 0).all()
    # grouped Linear weight gradients (192-column tiles, tnb_body_h): DropPath row scale on dY, plain / LayerNorm / GELU on X
    gen = torch.Generator().manual_seed(5)
    for (M, kind) in ((4097, "grad"), (33000, "rise"), (100, "zero")):
        NI, NJ = 180, 360
        dY, X = torch.randn(M, NI, generator=gen), torch.randn(M, NJ, generator=gen) * 1.5 + 0.3
        if kind == "grad":
            dY = dY * 1e-7 * torch.exp(torch.randn(M, 1, generator=gen) * 2.5) * torch.exp(torch.randn(1, NI, generator=gen) * 3.0)
        if kind == "rise":
            ramp = torch.exp2(torch.linspace(-30, 10, M)).reshape(M, 1)
            dY, X = dY * ramp, X * ramp
        if kind == "zero":
            dY[:, :5] = 0; X[:, 7] = 0; dY[: M // 2, 9] = 0
        rs = torch.rand(M // 16 + 1, generator=gen) + 0.5
        st = torch.stack([X.mean(1), 1 / torch.sqrt(X.var(1, unbiased=False) + 1e-5)], 1).contiguous()
        outs = [(torch.empty(NI, NJ).cuda(), torch.empty(NI).cuda()) for _ in range(3)]
        ops.linear_wgrad_grouped([
            dict(dY=dY.cuda(), X=X.cuda(), dW=outs[0][0], db=outs[0][1], a_rowscale=rs.cuda(), a_rowscale_rows=16),
            dict(dY=dY.cuda(), X=X.cuda(), dW=outs[1][0], db=outs[1][1], b_mode=1, ln_stats=st.cuda()),
            dict(dY=dY.cuda(), X=X.cuda(), dW=outs[2][0], db=outs[2][1], b_mode=2)])
        dYs = dY.double() * rs.double().repeat_interleave(16)[:M, None]
        Xn = (X.double() - st[:, :1].double()) * st[:, 1:].double()
        refs = [(dYs, X.double()), (dY.double(), Xn), (dY.double(), F.gelu(X.double()))]
        refs32 = [((dY * rs.repeat_interleave(16)[:M, None]), X), (dY, (X - st[:, :1]) * st[:, 1:]), (dY, F.gelu(X))]
        for (dw, db), (a, b), (a32, b32) in zip(outs, refs, refs32):
            ref = a.t() @ b
            den = ref.abs().max(1, keepdim=True).values.clamp_min(1e-300)
            e = ((dw.double().cpu() - ref).abs() / den).max().item()
            e32 = (((a32.t() @ b32).double() - ref).abs() / den).max().item()
            assert e <= max(3.0 * e32, 2e-6), ("linear", M, kind, e, e32)
            assert relmax(db, a.sum(0)) < 2e-6
            if kind == "zero":
                assert (dw[:5] == 0).all()
    print("ok")
''')


@pytest.mark.parametrize("env", [{}, {"SRHIP_TN_F16X2": "0", "SRHIP_TN_F16X2_LINEAR": "0", "SRHIP_TN_GROUP_XCD": "0"}])
def test_fp16x2_three_product_conv_weight_gradient_is_f32_grade(env):
    """The three-tap conv weight-gradient kernels (k_tnb3 / k_tnb3_conv_batched) with two fp16 planes, one running
    power-of-two scale per operand COLUMN (channel) and three products: every row of dW, relative to its largest entry,
    is within 3x of an f32 autograd -- on gradient-like inputs (pixels and channels decades apart), on magnitudes that
    rise by 2^40 along the tokens (the scales must follow and the sums be rescaled) and with all-zero channels;
    the same for the grouped Linear weight gradients (tnb_body_h: DropPath row scale, LayerNorm and GELU prologues).
    Second arm: the bf16x3 / six-product forms and the old block order."""
    cenv, pre = child_env(env)
    r = subprocess.run([sys.executable, "-c", pre + TN16_CHILD], cwd=ROOT, env=cenv,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (env, r.stdout[-500:], r.stderr[-1500:])
