"""EDSR-baseline on libsrhip, registered as ``EDSR_LIIF``.

The reference's ``network_edsr_liif.py`` is absent from its tree
(select_network.py:39-50 imports a missing file; SURVEY.md section "Five facts",
item 2), so this network follows the EDSR building blocks the reference does
ship (default_conv / ResBlock / Upsampler, network_nlsn.py:38-128) wired as
NLSN.forward without the attention modules (:355-369), with the EDSR-baseline
sizes of utils_init_default_args.py:37-50 (64 feats, 16 blocks, res_scale 1).
The LIIF implicit decoder has no reference source and no test: it is NOT
reproduced (parity unpinned); the tail is the pixel-shuffle Upsampler.
state_dict keys: head.0, body.{k}.body.{0,2}, body.{N}, tail.0.{0,2,..}, tail.1.
GPU only; CPU tensors raise.
"""
import math

import torch
import torch.nn as nn

from srhip import ops
from srhip.module_path import refresh_if_params_changed

__all__ = ['EDSR_LIIF', 'EDSR']


class _Box(nn.Module):
    pass


def _conv3(co, ci):
    m = _Box()
    bound = 1.0 / math.sqrt(ci * 9)
    m.weight = nn.Parameter((torch.rand(co, ci, 3, 3) * 2 - 1) * bound)
    m.bias = nn.Parameter((torch.rand(co) * 2 - 1) * bound)
    return m


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, need_grad, *params):
        ctx.net = net
        ctx.need_dx = x.requires_grad
        # reduced-precision matmuls only for inference (net.amp, set by --amp): training stays f32-accurate
        with ops.amp_inference(getattr(net, "amp", False) and not need_grad):
            y = net.engine.forward(x, None, save=need_grad)
        return y.clone() if need_grad else y

    @staticmethod
    def backward(ctx, dy):
        net = ctx.net
        names = [k for k, _ in net.named_parameters()]
        grads = {k: torch.empty_like(p) for k, p in net.named_parameters()}
        dx = net.engine.backward(dy.contiguous(), grads, need_dx=ctx.need_dx)
        return (dx, None, None) + tuple(grads[k] for k in names)


class EDSR_LIIF(nn.Module):
    def __init__(self, in_chans=1, n_resblocks=16, n_feats=64, scale=2, rgb_range=1.,
                 local_ensemble=True, feat_unfold=True, cell_decode=True, res_scale=1., **kwargs):
        super().__init__()
        if in_chans != 1:
            raise NotImplementedError("EDSR on libsrhip: 1-channel microscopy patches only")
        if scale & (scale - 1) or scale < 2:
            raise NotImplementedError("EDSR on libsrhip: scale must be a power of two")
        if n_feats % 4 or n_feats > 256:
            raise NotImplementedError("EDSR on libsrhip: n_feats must be a multiple of 4, <= 256")
        self.in_chans, self.n_resblocks, self.n_feats = in_chans, n_resblocks, n_feats
        self.scale, self.upscale, self.res_scale, self.img_range = scale, scale, res_scale, rgb_range
        self.head = nn.ModuleList([_conv3(n_feats, in_chans)])
        body = []
        for _ in range(n_resblocks):
            rb = _Box()
            rb.body = nn.ModuleList([_conv3(n_feats, n_feats), nn.Identity(), _conv3(n_feats, n_feats)])
            body.append(rb)
        body.append(_conv3(n_feats, n_feats))
        self.body = nn.ModuleList(body)
        up = []
        for _ in range(int(math.log2(scale))):
            up += [_conv3(4 * n_feats, n_feats), nn.Identity()]
        self.tail = nn.ModuleList([nn.ModuleList(up), _conv3(in_chans, n_feats)])
        self._engine = None

    @property
    def engine(self):
        if self._engine is None:
            from srhip.edsr_engine import EDSREngine
            self._engine = EDSREngine(self)
        return self._engine

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._engine = None
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        if self._engine is not None:
            self._engine.invalidate()
        return out

    def weights_changed(self):
        if self._engine is not None:
            self._engine.invalidate()

    def sample_drop_path(self, batch, device):
        return None

    def prepare_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("EDSR (libsrhip) runs on the GPU only: move the model and the input "
                               "to cuda; there is no CPU fallback")
        assert x.dim() == 4 and x.shape[1] == 1, x.shape
        return x.float().contiguous()[:, 0], x.shape[2], x.shape[3]

    def forward(self, x):
        xi, h, w = self.prepare_input(x)
        params = [p for _, p in self.named_parameters()]
        need_grad = torch.is_grad_enabled() and (xi.requires_grad or any(p.requires_grad for p in params))
        refresh_if_params_changed(self, params)   # stock torch.optim wrote the weights?
        return _NetFn.apply(xi, self, need_grad, *params)


EDSR = EDSR_LIIF
