"""SRCNN on libsrhip.

Drop-in for the reference's registered ``dlib.models.network_srcnn.SRCNN`` (network_srcnn.py:23-69,
select_network.py:207-210): ctor ``(in_chans)``, ``forward(x)`` on an input that is already at the
target resolution, ``state_dict`` keys ``features.0 / map.0 / reconstruction`` (+ ``.weight/.bias``), the
reference's initialisation law.  The module holds parameters only; ``forward`` runs
srhip/srcnn_engine.py (GEMMs on the bf16x3 kernels); CPU tensors raise."""
import math

import torch
import torch.nn as nn

from srhip.module_path import refresh_if_params_changed

__all__ = ['SRCNN']


class _Conv(nn.Module):
    def __init__(self, co, ci, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(co, ci, k, k))
        self.bias = nn.Parameter(torch.zeros(co))


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, need_grad, *params):
        ctx.net = net
        # --amp (inference only; training stays f32-accurate): fp16 storage of the 1024-channel feature map, one product
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


class SRCNN(nn.Module):
    def __init__(self, in_chans: int) -> None:
        super().__init__()
        assert isinstance(in_chans, int) and in_chans > 0, in_chans
        if in_chans != 1:
            raise NotImplementedError("SRCNN on libsrhip: 1-channel microscopy patches only")
        self.in_chans, self.scale, self.upscale, self.img_range = in_chans, 1, 1, 1.
        self.features = nn.ModuleList([_Conv(1024, in_chans, 5)])        # + ReLU
        self.map = nn.ModuleList([_Conv(128, 1024, 1)])                  # + ReLU
        self.reconstruction = _Conv(in_chans, 128, 1)
        self._engine = None
        self._initialize_weights()

    def _initialize_weights(self):                                       # network_srcnn.py:63-72
        for m in (self.features[0], self.map[0], self.reconstruction):
            co, k2 = m.weight.shape[0], m.weight.shape[2] * m.weight.shape[3]
            nn.init.normal_(m.weight.data, 0.0, math.sqrt(2 / (co * k2)))
            nn.init.zeros_(m.bias.data)
        nn.init.normal_(self.reconstruction.weight.data, 0.0, 0.001)
        nn.init.zeros_(self.reconstruction.bias.data)

    @property
    def engine(self):
        if self._engine is None:
            from srhip.srcnn_engine import SRCNNEngine
            self._engine = SRCNNEngine(self)
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
            raise RuntimeError("SRCNN (libsrhip) runs on the GPU only: move the model and the input to cuda; "
                               "there is no CPU fallback")
        assert x.dim() == 4 and x.shape[1] == self.in_chans, f'c: {x.shape}, img-nc: {self.in_chans}'
        return x.float().contiguous()[:, 0], x.shape[2], x.shape[3]

    def forward(self, x):
        xi, h, w = self.prepare_input(x)
        params = [p for _, p in self.named_parameters()]
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        refresh_if_params_changed(self, params)   # stock torch.optim wrote the weights?
        return _NetFn.apply(xi, self, need_grad, *params)
