"""VDSR on libsrhip (reference dlib/models/network_vdsr.py:24-126; registry select_network.py:200-205):
same constructor, ``forward((B,1,h,w) in [0,1]) -> (B,1,s*h,s*w)``, state_dict keys ``conv1.0.weight``,
``trunk.{k}.conv.weight`` (18), ``conv2.weight`` and the reference's weight initialisation; the compute is
``srhip.vdsr_engine.VDSREngine``.  1-channel inputs; GPU only (CPU tensors raise)."""
from math import sqrt

import torch
import torch.nn as nn

from srhip.module_path import refresh_if_params_changed

__all__ = ['VDSR']


class _Conv(nn.Module):
    def __init__(self, co, ci):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(co, ci, 3, 3))


class ConvReLU(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = _Conv(channels, channels)


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
        return y.clone() if need_grad else y

    @staticmethod
    def backward(ctx, dy):
        net = ctx.net
        names = [k for k, _ in net.named_parameters()]
        grads = {k: torch.empty_like(p) for k, p in net.named_parameters()}
        net.engine.backward(dy.contiguous(), grads)
        return (None, None, None) + tuple(grads[k] for k in names)


class VDSR(nn.Module):
    def __init__(self, in_chans: int, upscale: int) -> None:
        super().__init__()
        assert isinstance(upscale, int) and upscale > 0, upscale
        assert isinstance(in_chans, int) and in_chans > 0, in_chans
        if in_chans != 1:
            raise NotImplementedError("VDSR on libsrhip: 1-channel microscopy patches only")
        self.upscale, self.scale, self.in_chans = upscale, upscale, in_chans
        self.global_residual = None
        self.x_interp = None
        self.conv1 = nn.ModuleList([_Conv(64, in_chans)])            # + ReLU (network_vdsr.py:57-60)
        self.trunk = nn.ModuleList([ConvReLU(64) for _ in range(18)])
        self.conv2 = _Conv(in_chans, 64)
        self._engine = None
        self._initialize_weights()

    def _initialize_weights(self):                                    # network_vdsr.py:121-126
        for m in self.modules():
            if isinstance(m, _Conv):
                m.weight.data.normal_(0.0, sqrt(2 / (3 * 3 * m.weight.shape[0])))

    def flush(self):
        self.global_residual = None
        self.x_interp = None

    @property
    def engine(self):
        if self._engine is None:
            from srhip.vdsr_engine import VDSREngine
            self._engine = VDSREngine(self)
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
            raise RuntimeError("VDSR (libsrhip) runs on the GPU only: move the model and the input to cuda; "
                               "there is no CPU fallback")
        assert x.dim() == 4 and x.shape[1] == self.in_chans, f'c: {x.shape}, img-nc: {self.in_chans}'
        return x.float().contiguous()[:, 0], x.shape[2], x.shape[3]

    def forward(self, x):
        self.flush()
        xi, h, w = self.prepare_input(x)
        params = [p for _, p in self.named_parameters()]
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        refresh_if_params_changed(self, params)   # stock torch.optim wrote the weights?
        return _NetFn.apply(xi, self, need_grad, *params)
