"""MSLapSRN on libsrhip (reference dlib/models/network_mslapsr.py:53-188; registry select_network.py:103-108):
same constructor, ``forward((B,1,h,w)) -> (B,1,s*h,s*w)`` with the images of the earlier octaves in
``intermediate_outs`` (the trainer's multi-scale loss reads them, model_plain.py:277-314), the reference's
state_dict keys (``conv1.0.*``, ``laplacian_pyramid_conv{1,4,7}.{0..9}.cl.0.*`` and ``.10.*`` for the transposed
conv, ``laplacian_pyramid_conv{2,5,8}.*``, ``laplacian_pyramid_conv{3,6,9}.*``) and initialisation (PyTorch's
defaults: the reference defines ``_initialize_weights`` but never calls it); the compute is
``srhip.mslapsrn_engine.MSLapSRNEngine``.  1-channel inputs; GPU only (CPU tensors raise)."""
import math

import torch
import torch.nn as nn

from srhip.module_path import refresh_if_params_changed

__all__ = ['MSLapSRN']


def _default_init_(weight, bias):
    """nn.Conv2d / nn.ConvTranspose2d.reset_parameters"""
    nn.init.kaiming_uniform_(weight, a=math.sqrt(5))
    fan_in, _ = nn.init._calculate_fan_in_and_fan_out(weight)
    bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
    nn.init.uniform_(bias, -bound, bound)


class _Conv(nn.Module):            # nn.Conv2d(ci, co, 3, 1, 1): parameters only
    def __init__(self, co, ci):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(co, ci, 3, 3))
        self.bias = nn.Parameter(torch.empty(co))
        _default_init_(self.weight, self.bias)


class _ConvT(nn.Module):           # nn.ConvTranspose2d(ci, co, 4, 2, 1): parameters only
    def __init__(self, ci, co):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(ci, co, 4, 4))
        self.bias = nn.Parameter(torch.empty(co))
        _default_init_(self.weight, self.bias)


class ConvLayer(nn.Module):        # network_mslapsr.py:53-64
    def __init__(self, channels):
        super().__init__()
        self.cl = nn.ModuleList([_Conv(channels, channels)])


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, need_grad, *params):
        ctx.net = net
        # --amp (inference only; training stays f32-accurate): the convs run ONE product of the operands' leading fp16 planes
        # (11 significant bits under the block exponents) -- PSNR within 0.003 dB of the f32-accurate forward on the
        # seeded-weights check (gate 0.01 dB, tests/test_gpu_amp.py; with one bf16 product, round 2, VDSR was 0.023 dB off)
        from srhip import ops
        with ops.amp_inference(getattr(net, "amp", False) and not need_grad):
            y = net.engine.forward(x, None, save=need_grad)
        inter = [t.clone() for t in net.engine.intermediate_outs]
        ctx.n_inter = len(inter)
        return (y.clone() if need_grad else y, *inter)

    @staticmethod
    def backward(ctx, dy, *d_inter):
        net = ctx.net
        names = [k for k, _ in net.named_parameters()]
        grads = {k: torch.empty_like(p) for k, p in net.named_parameters()}
        net.engine.backward(dy.contiguous(), grads, d_inter=[None if g is None else g.contiguous() for g in d_inter])
        return (None, None, None) + tuple(grads[k] for k in names)


class MSLapSRN(nn.Module):
    def __init__(self, upscale: int = 2, in_chans: int = 3) -> None:
        super().__init__()
        assert upscale in [2, 4, 8], upscale
        if in_chans != 1:
            raise NotImplementedError("MSLapSRN on libsrhip: 1-channel microscopy patches only")
        self.upscale, self.scale, self.in_chans = upscale, upscale, in_chans
        self.conv1 = nn.ModuleList([_Conv(64, in_chans)])                       # + LeakyReLU(0.2)
        for o in range(int(math.log2(upscale))):
            layers = [ConvLayer(64) for _ in range(10)] + [_ConvT(64, 64)]      # + LeakyReLU(0.2) (index 11, no parameters)
            setattr(self, f"laplacian_pyramid_conv{3 * o + 1}", nn.ModuleList(layers))
            setattr(self, f"laplacian_pyramid_conv{3 * o + 2}", _ConvT(1, 1))
            setattr(self, f"laplacian_pyramid_conv{3 * o + 3}", _Conv(in_chans, 64))
        self.intermediate_outs = []     # hold only intermediate predictions
        self._engine = None

    def flush(self):
        self.intermediate_outs = []

    def _initialize_weights(self) -> None:                                      # network_mslapsr.py:176-188
        for m in self.modules():
            if isinstance(m, _Conv):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                nn.init.constant_(m.bias, 0)
            if isinstance(m, _ConvT):
                c1, c2, h, w = m.weight.shape
                f = (h + 1) // 2
                center = f - 1 if h % 2 == 1 else f - 0.5
                r = 1 - (torch.arange(h, dtype=torch.float64) - center).abs() / f
                m.weight.data.copy_((r[:, None] * r[None, :]).float().view(1, 1, h, w).repeat(c1, c2, 1, 1))
                m.bias.data.zero_()
        self.weights_changed()

    @property
    def engine(self):
        if self._engine is None:
            from srhip.mslapsrn_engine import MSLapSRNEngine
            self._engine = MSLapSRNEngine(self)
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
            raise RuntimeError("MSLapSRN (libsrhip) runs on the GPU only: move the model and the input to cuda; "
                               "there is no CPU fallback")
        assert x.dim() == 4 and x.shape[1] == self.in_chans, f'c: {x.shape}, img-nc: {self.in_chans}'
        return x.float().contiguous()[:, 0], x.shape[2], x.shape[3]

    def forward(self, x):
        self.intermediate_outs = []
        xi, h, w = self.prepare_input(x)
        params = [p for _, p in self.named_parameters()]
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        refresh_if_params_changed(self, params)   # stock torch.optim wrote the weights?
        y, *inter = _NetFn.apply(xi, self, need_grad, *params)
        self.intermediate_outs = list(inter)
        return y
