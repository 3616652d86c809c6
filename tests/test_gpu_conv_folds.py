"""srhip_conv3x3_nhwc_split_ex (include/srhip.h): the folds of the plain CNNs' element-wise neighbours into the 3x3 conv --
evaluation-mode BatchNorm + ReLU on the input (MemNet, network_memnet.py:27-34), residual + ReLU (DRRN, network_drrn.py:58-62),
PReLU and PReLU + addend (DBPN's projection units, network_dbpn.py:93-99) -- against (1) the unfused launches of this
library, bit for bit (the folds reorder no arithmetic), and (2) float64 aten."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _setup(Cin, Cout, B=2, H=21, W=19, seed=0):
    from srhip import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).cuda()
    b = (torch.randn(Cout, generator=g) * 0.1).cuda()
    r = torch.randn(B, H, W, Cout, generator=g).cuda()
    wp = ops.Bx3(9 * Cout, Cin, "cuda")
    tb = ops.PrepTable()
    tb.conv(w, wp)
    tb.build("cuda").run()
    return ops, x, w, b, r, wp


def _ref_conv(x, w, b):
    return F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)


def _close(y, ref, tol=2e-5):
    e = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    assert e < tol, e


@pytest.mark.parametrize("Cin,Cout", [(64, 64), (128, 128), (64, 256), (32, 48)])
def test_residual_relu_and_prelu_epilogues(Cin, Cout):
    ops, x, w, b, r, wp = _setup(Cin, Cout)
    B, H, W, _ = x.shape
    pre = _ref_conv(x, w, b)
    # epi 8 = epi 2 + ReLU
    y2 = ops.conv3x3(x, wp, b, Cout, epi=2, R=r)
    ops.relu_mask(y2, y2)
    y8 = ops.conv3x3(x, wp, b, Cout, epi=8, R=r)
    assert torch.equal(y8, y2)
    _close(y8, torch.relu(pre + r.double()))
    # epi 9 = conv, then srhip_prelu_fwd
    slope = torch.tensor([0.25], device="cuda")
    y0 = ops.conv3x3(x, wp, b, Cout)
    yp = torch.empty_like(y0)
    ops.call("srhip_prelu_fwd", y0.data_ptr(), slope.data_ptr(), yp.data_ptr(), y0.numel(), ops._st())
    y9 = ops.conv3x3(x, wp, b, Cout, epi=9, slope=slope)
    assert torch.equal(y9, yp)
    _close(y9, F.prelu(pre, slope.double()))
    # epi 10 = ... then + alpha * R
    for alpha in (-1.0, 1.0):
        y10 = ops.conv3x3(x, wp, b, Cout, epi=10, R=r, alpha=alpha, slope=slope)
        _close(y10, F.prelu(pre, slope.double()) + alpha * r.double())
        z = yp.clone()
        ops.axpby(z, r, alpha, 1.0)
        assert (y10 - z).abs().max().item() <= 1e-6 * z.abs().max().item()      # fma vs mul + add


@pytest.mark.parametrize("Cin,Cout,hw", [(64, 64, (21, 19)), (64, 64, (40, 48)), (128, 128, (9, 17)), (64, 256, (16, 16))])
def test_batchnorm_relu_input_prologue(Cin, Cout, hw):
    ops, x, w, b, r, wp = _setup(Cin, Cout, H=hw[0], W=hw[1], seed=3)
    if wp.fmt != 1:
        pytest.skip("the prologue runs on the fp16x2 conv kernel")
    B, H, W, _ = x.shape
    g = torch.Generator().manual_seed(4)
    mean, var = torch.randn(Cin, generator=g) * 0.3, torch.rand(Cin, generator=g) + 0.5
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2
    rstd = torch.rsqrt(var + 1e-5)
    coef = torch.stack([mean, rstd, gamma * rstd, beta]).cuda().contiguous()
    a = torch.empty_like(x)
    ops.bn_apply(x.view(-1, Cin), coef, a.view(-1, Cin), relu=True)
    y_two = ops.conv3x3(a, wp, b, Cout, epi=2, R=r)
    y_one = ops.conv3x3(x, wp, b, Cout, epi=2, R=r, in_bn=coef)
    assert torch.equal(y_one, y_two)
    act = torch.relu((x.double() - coef[0].double()) * coef[2].double() + coef[3].double())
    _close(y_one, _ref_conv(act, w, b) + r.double())


def test_entry_point_rejects_what_it_cannot_run():
    ops, x, w, b, r, wp = _setup(64, 64)
    slope = torch.tensor([0.25], device="cuda")
    with pytest.raises(RuntimeError):
        ops.conv3x3(x, wp, b, 64, epi=9)                       # no slope
    with pytest.raises(RuntimeError):
        ops.conv3x3(x, wp, b, 64, epi=8)                       # no addend
    with pytest.raises(RuntimeError):
        ops.conv3x3(x, wp, b, 64, epi=10, slope=slope)         # no addend


@pytest.mark.parametrize("Co,r", [(64, 8), (64, 4), (64, 2), (16, 8)])
def test_pixel_shuffle_with_addend(Co, r):
    from srhip import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 5, 7, Co * r * r, generator=g).cuda()
    a = torch.randn(2, 5 * r, 7 * r, Co, generator=g).cuda()
    ref = F.pixel_shuffle(x.permute(0, 3, 1, 2), r).permute(0, 2, 3, 1)
    if not ops.pixel_shuffle_add_ok(Co, r):
        with pytest.raises(RuntimeError):
            ops.pixel_shuffle(x, r, nhwc_out=True, add=a, fac=-1.0)
        return
    for fac in (1.0, -1.0):
        y = ops.pixel_shuffle(x, r, nhwc_out=True, add=a, fac=fac)
        assert torch.equal(y, ref + fac * a) or (y - (ref + fac * a)).abs().max().item() <= 1e-6
