"""conv 3x3 + PixelShuffle(2) as one kernel per direction (srhip_conv3x3_ps2_bx3, _bwd_data_bx3, _wgrad_bx3):
the Upsampler stage of EDSR (dlib/models/network_nlsn.py:89-93,103-108) against float64 aten and against the
conv + index-kernel pair it replaces."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

G = torch.Generator().manual_seed(2468)


def rnd(*shape, scale=1.0):
    return torch.randn(*shape, generator=G) * scale


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def ops():
    from srhip import ops as o
    return o


@pytest.mark.parametrize("B,H,W,Cin,Fo", [(2, 24, 40, 64, 64), (1, 17, 33, 64, 64), (3, 64, 64, 64, 64),
                                          (1, 8, 8, 128, 64)])
def test_conv_pixelshuffle_fused_forward_backward(ops, B, H, W, Cin, Fo):
    Co = 4 * Fo
    assert ops.ps2_fusable(Cin, Co)
    x = rnd(B, Cin, H, W)
    w = rnd(Co, Cin, 3, 3, scale=0.05)
    b = rnd(Co, scale=0.2)
    # float64 reference
    xd = x.double().requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yd = F.pixel_shuffle(F.conv2d(xd, wd, bd, padding=1), 2)
    dy = rnd(*yd.shape)
    yd.backward(dy.double())
    # device operands (NHWC)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    wc, bc = w.cuda().contiguous(), b.cuda()
    wp, wpt = ops.Bx3(9 * Co, Cin, "cuda"), ops.Bx3(9 * Cin, Co, "cuda")
    wp0, wpt0 = ops.Bx3(9 * Co, Cin, "cuda"), ops.Bx3(9 * Cin, Co, "cuda")
    tb = ops.PrepTable()
    tb.conv(wc, wp, ps2=True)
    tb.conv(wc, wpt, data_grad=True, ps2=True)
    tb.conv(wc, wp0)
    tb.conv(wc, wpt0, data_grad=True)
    tb.build("cuda").run()
    # ---- forward
    y = torch.full((B, 2 * H, 2 * W, Fo), float("nan"), device="cuda")
    ops.conv3x3_ps2(xn, wp, bc, y)
    assert relerr(y.permute(0, 3, 1, 2), yd) < 2e-6
    c = ops.conv3x3(xn, wp0, bc, Co)
    y0 = torch.empty_like(y)
    ops.pixel_shuffle(c, 2, nhwc_out=True, out=y0)
    # the same products, stored elsewhere -- in the same order until round 5; since then a block starts its nine-tap walk at a
    # tap that depends on its index in the launch (weight lines spread over the L2, gemm_ntw.hip), and the two launches number
    # their blocks differently: rounding-level difference
    assert (y - y0).abs().max().item() <= 2e-6 * y0.abs().max().item()
    # relu epilogue rides along
    yr = torch.empty_like(y)
    ops.conv3x3_ps2(xn, wp, bc, yr, epi=1)
    assert torch.equal(yr, y.clamp_min(0))
    # ---- data gradient
    dyn = dy.permute(0, 2, 3, 1).contiguous().cuda()      # gradient of the shuffled image, NHWC
    dx = torch.full((B, H, W, Cin), float("nan"), device="cuda")
    ops.conv3x3_ps2_bwd_data(dyn, wpt, dx)
    assert relerr(dx.permute(0, 3, 1, 2), xd.grad) < 2e-6
    dc = torch.empty(B, H, W, Co, device="cuda")
    ops.pixel_shuffle(dyn, 2, nhwc_out=True, inverse=True, out=dc)
    dx0 = ops.conv3x3(dc, wpt0, None, Cin)
    assert relerr(dx, dx0) < 3e-6                 # the reduce order over the 4F channels differs
    # ---- weight / bias gradient
    dW = torch.full((Co, Cin, 3, 3), float("nan"), device="cuda")
    db = torch.full((Co,), float("nan"), device="cuda")
    ops.conv3x3_wgrad(dyn, xn, dW, db, ps2=True)
    assert relerr(dW, wd.grad) < 3e-6 and relerr(db, bd.grad) < 3e-6
    dW0, db0 = torch.empty_like(dW), torch.empty_like(db)
    ops.conv3x3_wgrad(dc, xn, dW0, db0)
    assert relerr(dW, dW0) < 1e-6 and relerr(db, db0) < 1e-6


def test_edsr_engine_uses_the_fused_upsampler(ops, monkeypatch):
    """EDSR-baseline (64 features): the fused and the unfused upsampler give the same image and gradients."""
    from dlib.models.network_edsr_liif import EDSR_LIIF
    outs = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("SRHIP_FUSE_PS", fuse)
        torch.manual_seed(5)
        net = EDSR_LIIF(in_chans=1, n_resblocks=2, n_feats=64, scale=4, rgb_range=1., res_scale=1., local_ensemble=True,
                        feat_unfold=True, cell_decode=True).cuda().train()
        assert net.engine.fuse_ps == (fuse == "1")
        x = torch.rand(2, 1, 24, 24, generator=torch.Generator().manual_seed(1)).cuda()
        y = net(x)
        (y * torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)).sum().backward()
        outs[fuse] = (y.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()})
    assert relerr(outs["1"][0], outs["0"][0]) < 1e-6
    for k, g in outs["1"][1].items():
        assert relerr(g, outs["0"][1][k]) < 1e-5, k        # f32 rounding through different reduce orders
