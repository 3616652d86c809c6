"""Test infrastructure: torch (CPU) stand-ins for the libsrhip entry points the training tapes of srhip/act_engine.py,
omnisr_engine.py and grl_engine.py call, so that the
HOST logic of the tape -- which op feeds which, what each backward closure accumulates where, parameter names -- can be checked
on a machine without a GPU against the reference's gradients (tests/test_cpu_tape_logic.py).  Nothing here is the product: the
kernels themselves are tested on the GPU (tests/test_gpu_*.py), and the package has no CPU path (srhip.ops raises on CPU
tensors).  install(monkeypatch) swaps the functions in for one test."""
import torch
import torch.nn.functional as F


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


def _w_of(wp):           # [9, Co, Ci] -> [Co, Ci, 3, 3]
    return wp.reshape(3, 3, wp.shape[1], wp.shape[2]).permute(2, 3, 0, 1)


def pack_conv_weight(w, wp=None, wpt=None):
    if wp is not None:
        wp.copy_(w.permute(2, 3, 0, 1).reshape(9, w.shape[0], w.shape[1]))
    if wpt is not None:     # the data-gradient conv: taps flipped, channels swapped
        wpt.copy_(w.flip(2, 3).permute(2, 3, 1, 0).reshape(9, w.shape[1], w.shape[0]))


def conv3x3(X, Wp, bias, Cout, out=None, epi=0, R=None, rowscale=None, alpha=1.0, in_bn=None, slope=None):
    y = _nhwc(F.conv2d(_nchw(X), _w_of(Wp), bias, padding=1))
    if epi == 1:
        y = torch.relu(y)
    elif epi == 2:
        y = R + alpha * y
    else:
        assert epi == 0, epi
    if out is None:
        return y.contiguous()
    out.copy_(y)
    return out


def conv3x3_wgrad(dY, X, dW, db, ps2=False):
    w = torch.zeros_like(dW, requires_grad=True)
    b = torch.zeros(dW.shape[0], requires_grad=True)
    y = F.conv2d(_nchw(X), w, b, padding=1)
    gw, gb = torch.autograd.grad(y, (w, b), _nchw(dY))
    dW.copy_(gw)
    if db is not None:
        db.copy_(gb)


def gemm_nt(A, W, bias=None, out=None, **kw):
    assert not kw, kw
    y = A @ W.t()
    if bias is not None:
        y = y + bias
    if out is None:
        return y
    out.copy_(y)
    return out


def linear_wgrad(dY, X, dW, db, **kw):
    assert not kw, kw
    dW.copy_(dY.t() @ X)
    if db is not None:
        db.copy_(dY.sum(0))


def gemm_nt_batched(A, a_z, W, w_z, C, c_z, M, N, K, zcount, zdiv):
    assert zdiv == 1 and a_z[1] == 0 and w_z[1] == 0 and c_z[1] == 0
    for z in range(zcount):
        a = torch.as_strided(A, (M, K), (A.stride(-2), 1), A.storage_offset() + z * a_z[0])
        w = torch.as_strided(W, (N, K), (W.stride(-2), 1), W.storage_offset() + z * w_z[0])
        c = torch.as_strided(C, (M, N), (C.stride(-2), 1), C.storage_offset() + z * c_z[0])
        c.copy_(a @ w.t())
    return C


def softmax_rows_(x, scale=1.0):
    x.copy_(torch.softmax(scale * x, 1))
    return x


def softmax_rows_bwd_(P, dP, dlse=None):
    assert dlse is None
    dP.copy_(P * (dP - (P * dP).sum(1, keepdim=True)))
    return dP


def layernorm_rows(x, gamma, beta, out, eps=1e-5):
    out.copy_(F.layer_norm(x, (x.shape[1],), gamma, beta, eps))
    return out


def layernorm_rows_bwd(dy, x, gamma, dx, dgamma, dbeta, eps=1e-5):
    xx = x.detach().clone().requires_grad_(True)
    g = gamma.detach().clone().requires_grad_(True)
    b = torch.zeros_like(gamma, requires_grad=True)
    a, c, d = torch.autograd.grad(F.layer_norm(xx, (x.shape[1],), g, b, eps), (xx, g, b), dy)
    dx.copy_(a)
    dgamma.copy_(c)
    dbeta.copy_(d)
    return dx


def unfold(x, C, k, s, pad, out, ldx=None):
    B, H, W = x.shape[:3]
    t = F.unfold(_nchw(x[..., :C]), k, stride=s, padding=pad)            # [B, C k k, nT]
    out[:, :C * k * k].copy_(t.transpose(1, 2).reshape(-1, C * k * k))
    return out


def fold(tok, C, k, s, out):
    B, H, W = out.shape[:3]
    nT = tok.shape[0] // B
    t = tok[:, :C * k * k].reshape(B, nT, C * k * k).transpose(1, 2)
    out[..., :C].copy_(_nhwc(F.fold(t, (H, W), k, stride=s)))
    return out


def axpby(y, x, a, b):
    y.copy_(a * x + b * y)


def leaky_relu_(x, slope):
    x.copy_(F.leaky_relu(x, slope))
    return x


def relu_mask(g, y):
    g.mul_((y > 0).to(g.dtype))


def leaky_relu_mask(g, y, slope):
    g.mul_(torch.where(y > 0, torch.ones_like(y), torch.full_like(y, slope)))


def nearest_up2(lo, hi, adjoint=False):
    B, h, w, C = lo.shape
    if adjoint:
        lo.copy_(hi.reshape(B, h, 2, w, 2, C).sum((2, 4)))
        return lo
    hi.copy_(lo.repeat_interleave(2, 1).repeat_interleave(2, 2))
    return hi


def unary(x, y, kind):
    y.copy_(F.gelu(x) if kind == "gelu" else torch.sigmoid(x))
    return y


def unary_bwd(ref, g, out, kind):
    if kind == "gelu":
        xx = ref.detach().clone().requires_grad_(True)
        out.copy_(torch.autograd.grad(F.gelu(xx), xx, g)[0])
    else:
        out.copy_(g * ref * (1 - ref))
    return out


class _Scratch:
    def __init__(self):
        self.bufs = {}

    def get(self, name, n, dtype=torch.float32, device="cpu"):
        t = self.bufs.get(name)
        if t is None or t.numel() < n:
            t = self.bufs[name] = torch.zeros(n, dtype=dtype)
        return t


SCRATCH = _Scratch()


def channel_gate(feat, w1, b1, w2, b2, x0, x1, out, mid_act="relu"):
    B, H, W, C = feat.shape
    pool = feat.mean((1, 2))
    z = pool @ w1.t() + (0 if b1 is None else b1)
    z = torch.relu(z) if mid_act == "relu" else F.silu(z)
    gate = torch.sigmoid(z @ w2.t() + (0 if b2 is None else b2))
    SCRATCH.get("gate_vec", B * C)[:B * C].copy_(gate.reshape(-1))
    out.copy_((0 if x0 is None else x0) + x1 * gate.view(B, 1, 1, C))
    return out


def _k1(w, flip):        # [C][9] kernels of a 1-channel conv as [C, 3, 3]
    k = w.reshape(-1, 3, 3)
    return k.flip(1, 2) if flip else k


def conv3x3_cin1_fwd(x, w, bias, Co, out=None, flip=False):
    y = _nhwc(F.conv2d(x[:, None], _k1(w, flip)[:, None], bias, padding=1))
    if out is None:
        return y.contiguous()
    out.copy_(y)
    return out


def conv3x3_cin1_wgrad(a, b, dW, db, flip=False):
    """a [B,H,W] the 1-channel image, b [B,H,W,C]: not flipped dW[c] = sum b[p, c] a[p + tap]; flipped the correlation the
    other way round (the weight gradient of the C -> 1 conv: a = dy, b = its input)."""
    C = b.shape[3]
    w = torch.zeros(C, 1, 3, 3, requires_grad=True)
    if not flip:
        gw = torch.autograd.grad(F.conv2d(a[:, None], w, None, padding=1), w, _nchw(b))[0]
    else:
        wt = torch.zeros(1, C, 3, 3, requires_grad=True)
        gw = torch.autograd.grad(F.conv2d(_nchw(b), wt, None, padding=1), wt, a[:, None])[0]
    dW.copy_(gw.reshape(dW.shape))
    if db is not None:
        db.copy_(b.sum((0, 1, 2)))


def conv3x3_cout1_fwd(x, w, bias, out=None):
    y = F.conv2d(_nchw(x), w.reshape(1, -1, 3, 3), bias, padding=1)[:, 0]
    if out is None:
        return y.contiguous()
    out.copy_(y)
    return out


def sum_into(x, out):
    out.copy_(x.sum().reshape(1))


def pixel_shuffle(x, r, nhwc_out=False, inverse=False, out=None, add=None, fac=1.0):
    assert nhwc_out and add is None
    y = _nhwc(F.pixel_unshuffle(_nchw(x), r) if inverse else F.pixel_shuffle(_nchw(x), r))
    if out is None:
        return y.contiguous()
    out.copy_(y)
    return out


def dwconv3x3(x, w, bias, out):
    C = out.shape[3]
    out.copy_(_nhwc(F.conv2d(_nchw(x[..., :C]), w.reshape(C, 1, 3, 3), bias, padding=1, groups=C)))
    return out


def gelu_gate(x, out):
    C = out.shape[1]
    out.copy_(F.gelu(x[:, :C]) * x[:, C:])
    return out


def mul_sigmoid(x, g, out):
    out.copy_((x * torch.sigmoid(g)).reshape(out.shape))
    return out


def mul(a, b, out=None):
    if out is None:
        return a * b
    out.copy_(a * b)
    return out


def rowdot(a, b, out=None):
    r = (a * b).sum(1)
    if out is None:
        return r
    out.copy_(r)
    return out


def add_periodic(x, v):
    x.view(-1, v.numel()).add_(v.reshape(1, -1))
    return x


def sum_periodic(x, out):
    out.copy_(x.reshape(-1, out.numel()).sum(0).reshape(out.shape))
    return out


def maxpool2d(x, k, s):
    return _nhwc(F.max_pool2d(_nchw(x), k, s)).contiguous()


def maxpool2d_bwd(x, g, k, s):
    xx = x.detach().clone().requires_grad_(True)
    return torch.autograd.grad(_nhwc(F.max_pool2d(_nchw(xx), k, s)), xx, g)[0].contiguous()


def bilinear_resize(x, Ho, Wo):
    return _nhwc(F.interpolate(_nchw(x), (Ho, Wo), mode="bilinear", align_corners=False)).contiguous()


NAMES = ["nearest_up2", "leaky_relu_mask", "dwconv3x3", "gelu_gate", "mul_sigmoid", "mul", "rowdot", "add_periodic", "sum_periodic", "maxpool2d", "maxpool2d_bwd",
         "bilinear_resize",
         "pack_conv_weight", "conv3x3", "conv3x3_wgrad", "gemm_nt", "linear_wgrad", "gemm_nt_batched", "softmax_rows_",
         "softmax_rows_bwd_", "layernorm_rows", "layernorm_rows_bwd", "unfold", "fold", "axpby", "leaky_relu_", "relu_mask",
         "unary", "unary_bwd", "channel_gate", "conv3x3_cin1_fwd", "conv3x3_cin1_wgrad", "conv3x3_cout1_fwd", "sum_into",
         "pixel_shuffle", "SCRATCH"]


def install(monkeypatch):
    from srhip import ops
    for n in NAMES:
        monkeypatch.setattr(ops, n, globals()[n])
    monkeypatch.setattr(ops, "bx3_nt_for", lambda *c: False)
