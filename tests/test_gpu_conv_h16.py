"""fp16-storage convs of the evaluation path (conv_h16.hip, include/srhip.h srhip_conv3x3_nhwc_h16 / _cin1_h16 / _cout1_h16)
against float64 aten on the SAME rounded operands: the fp16 activations as they are, the weight as the leading fp16 plane under
its per-output-channel power-of-two scale (emulated here), so that what remains is the f32 accumulation order and the final
rounding of the result to fp16 (2^-11 relative)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _w_hi(w):
    """the leading fp16 plane of srhip_prep_table job kind 4: one power-of-two scale per output channel"""
    mx = w.abs().flatten(1).max(1)[0].clamp_min(1e-30)
    sc = torch.exp2(torch.floor(torch.log2(16384.0 / mx))).view(-1, 1, 1, 1)
    return ((w * sc).half().double() / sc.double())


def _planes(w, ps2=False):
    from srhip import ops
    Co, Ci = w.shape[:2]
    wp = ops.Bx3(9 * Co, Ci, "cuda")
    tb = ops.PrepTable()
    tb.conv(w, wp, ps2=ps2)
    tb.build("cuda").run()
    assert wp.fmt == 1
    return wp


def _check(y, ref):
    y, ref = y.double(), ref.double()
    tol = 2.0 ** -10 * ref.abs() + 2e-4 * ref.abs().max()
    bad = ((y - ref).abs() > tol).sum().item()
    assert bad == 0, (bad, ((y - ref).abs() / ref.abs().max()).max().item())


@pytest.mark.parametrize("Cin,Cout,hw,B", [(64, 64, (21, 19), 2), (64, 64, (64, 64), 8), (128, 128, (9, 33), 1), (64, 256, (16, 16), 3),
                                           (256, 64, (12, 20), 1)])
def test_conv3x3_h16_epilogues(Cin, Cout, hw, B):
    from srhip import ops
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, hw[0], hw[1], Cin, generator=g).cuda().half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).cuda()
    b = (torch.randn(Cout, generator=g) * 0.1).cuda()
    r = torch.randn(B, hw[0], hw[1], Cout, generator=g).cuda().half()
    wp = _planes(w)
    pre = F.conv2d(x.double().permute(0, 3, 1, 2), _w_hi(w), None, padding=1).permute(0, 2, 3, 1)
    _check(ops.conv3x3_h16(x, wp, None, Cout), pre)
    _check(ops.conv3x3_h16(x, wp, b, Cout, epi=1), torch.relu(pre + b.double()))
    _check(ops.conv3x3_h16(x, wp, b, Cout, epi=2, R=r, alpha=0.1), r.double() + 0.1 * (pre + b.double()))
    _check(ops.conv3x3_h16(x, wp, None, Cout, epi=8, R=r), torch.relu(r.double() + pre))


@pytest.mark.parametrize("hw,B", [((16, 16), 2), ((13, 21), 1)])
def test_conv3x3_h16_pixelshuffle2(hw, B):
    from srhip import ops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, hw[0], hw[1], 64, generator=g).cuda().half()
    w = (torch.randn(256, 64, 3, 3, generator=g) / 24.0).cuda()
    b = (torch.randn(256, generator=g) * 0.1).cuda()
    wp = _planes(w, ps2=True)
    pre = F.conv2d(x.double().permute(0, 3, 1, 2), _w_hi(w), b.double(), padding=1)
    ref = F.pixel_shuffle(pre, 2).permute(0, 2, 3, 1)
    _check(ops.conv3x3_h16(x, wp, b, 256, ps2=True), ref)


def test_edge_convs_h16():
    from srhip import ops
    g = torch.Generator().manual_seed(9)
    x = torch.rand(2, 19, 23, generator=g).cuda()
    w1 = (torch.randn(64, 1, 3, 3, generator=g) / 3.0).cuda()
    b1 = (torch.randn(64, generator=g) * 0.1).cuda()
    f = ops.conv3x3_cin1_h16(x, w1, b1, 64, relu=True)
    ref = torch.relu(F.conv2d(x.double()[:, None], w1.double(), b1.double(), padding=1)).permute(0, 2, 3, 1)
    _check(f, ref)
    w2 = (torch.randn(1, 64, 3, 3, generator=g) / 24.0).cuda()
    b2 = torch.tensor([0.05], device="cuda")
    y = ops.conv3x3_cout1_h16(f, w2, b2, add=x)
    # (the tail conv multiplies fp16 pairs -- v_dot2_f32_f16, f32 accumulate --: its weights are rounded to fp16 too)
    ref2 = F.conv2d(f.double().permute(0, 3, 1, 2), w2.half().double(), b2.double(), padding=1)[:, 0] + x.double()
    assert ((y.double() - ref2).abs().max() / ref2.abs().max()).item() < 1e-5


def test_conv3x3_h16_bn_prologue_leaky_and_1x1():
    from srhip import ops
    g = torch.Generator().manual_seed(21)
    B, H, W, Cin, Cout = 2, 18, 20, 128, 64
    x = torch.randn(B, H, W, Cin, generator=g).cuda().half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).cuda()
    b = (torch.randn(Cout, generator=g) * 0.1).cuda()
    wp = _planes(w)
    mean, var = torch.randn(Cin, generator=g) * 0.3, torch.rand(Cin, generator=g) + 0.5
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2
    rstd = torch.rsqrt(var + 1e-5)
    coef = torch.stack([mean, rstd, gamma * rstd, beta]).cuda().contiguous()
    # the prologue's activation is rounded to fp16 before the product, as a separate fp16 BatchNorm-ReLU pass would
    act = torch.relu((x.float() - coef[0]) * coef[2] + coef[3]).half()
    pre = F.conv2d(act.double().permute(0, 3, 1, 2), _w_hi(w), b.double(), padding=1).permute(0, 2, 3, 1)
    _check(ops.conv3x3_h16(x, wp, b, Cout, in_bn=coef), pre)
    # LeakyReLU epilogue
    pre0 = F.conv2d(x.double().permute(0, 3, 1, 2), _w_hi(w), b.double(), padding=1).permute(0, 2, 3, 1)
    _check(ops.conv3x3_h16(x, wp, b, Cout, epi=6, alpha=0.2), F.leaky_relu(pre0, 0.2))
    # a 1x1 conv held in the centre tap
    w1 = torch.zeros(Cout, Cin, 3, 3, device="cuda")
    w1[:, :, 1, 1] = torch.randn(Cout, Cin, generator=g).cuda() / Cin ** 0.5
    wp1 = _planes(w1)
    ref1 = F.conv2d(act.double().permute(0, 3, 1, 2), _w_hi(w1), None, padding=1).permute(0, 2, 3, 1)
    _check(ops.conv3x3_h16(x, wp1, None, Cout, in_bn=coef, center_only=True), ref1)
    # tail conv with the prologue; head conv with LeakyReLU
    w2 = (torch.randn(1, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).cuda()
    y = ops.conv3x3_cout1_h16(x, w2, None, in_bn=coef)
    act32 = torch.relu((x.double() - coef[0].double()) * coef[2].double() + coef[3].double())
    ref2 = F.conv2d(act32.permute(0, 3, 1, 2), w2.double(), None, padding=1)[:, 0]
    assert ((y.double() - ref2).abs().max() / ref2.abs().max()).item() < 1e-5
    img = torch.rand(2, 11, 13, generator=g).cuda()
    wh = (torch.randn(64, 1, 3, 3, generator=g) / 3.0).cuda()
    f = ops.conv3x3_cin1_h16(img, wh, None, 64, leaky=0.2)
    _check(f, F.leaky_relu(F.conv2d(img.double()[:, None], wh.double(), None, padding=1), 0.2).permute(0, 2, 3, 1))


def test_srcnn_fused_forward_h16():
    """srhip_srcnn_fwd_h16 against float64 on the same rounded operands (fp16 patches, leading fp16 weight planes, the
    1024-channel map rounded to fp16 behind its ReLU as the kernel does)."""
    from srhip import ops
    g = torch.Generator().manual_seed(31)
    T = 128 * 37 + 5
    a = torch.zeros(T, 32)
    a[:, :25] = torch.rand(T, 25, generator=g)
    a16 = a.cuda().half()
    w1 = torch.zeros(1024, 32, 3, 3, device="cuda")
    w1[:, :25, 1, 1] = (torch.randn(1024, 25, generator=g) * 0.2).cuda()
    w2 = torch.zeros(128, 1024, 3, 3, device="cuda")
    w2[:, :, 1, 1] = (torch.randn(128, 1024, generator=g) / 32.0).cuda()
    b1, b2 = (torch.randn(1024, generator=g) * 0.1).cuda(), (torch.randn(128, generator=g) * 0.1).cuda()
    w3, b3 = (torch.randn(128, generator=g) * 0.1).cuda(), torch.tensor([0.03], device="cuda")
    p1, p2 = ops.Bx3(9 * 1024, 32, "cuda"), ops.Bx3(9 * 128, 1024, "cuda")
    tb = ops.PrepTable()
    tb.conv(w1, p1, force_f16=True)
    tb.conv(w2, p2, force_f16=True)
    tb.build("cuda").run()
    y = torch.empty(T, device="cuda")
    ops.srcnn_fwd_h16(a16, p1, b1, p2, b2, w3, b3, y)
    W1, W2 = _w_hi(w1)[:, :, 1, 1], _w_hi(w2)[:, :, 1, 1]
    h1 = torch.relu(a16.double() @ W1.t() + b1.double()).half().double()
    h2 = torch.relu(h1 @ W2.t() + b2.double())
    ref = h2 @ w3.double() + b3.double()
    err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-4, err
    # the patch matrix built inside the kernel from the image == srhip_im2col_c1's (rounded to fp16)
    img = torch.rand(3, 21, 37, generator=g).cuda()
    a0 = ops.im2col_c1(img, 5, 28)
    a16b = torch.zeros(a0.shape[0], 32, device="cuda", dtype=torch.float16)
    a16b[:, :28] = a0
    y1, y2 = torch.empty(a0.shape[0], device="cuda"), torch.empty(a0.shape[0], device="cuda")
    ops.srcnn_fwd_h16(a16b, p1, b1, p2, b2, w3, b3, y1)
    ops.srcnn_fwd_h16(None, p1, b1, p2, b2, w3, b3, y2, image=img)
    assert torch.equal(y1, y2)
