"""DRRN on libsrhip (reference dlib/models/network_drrn.py:22-126; registry select_network.py:192-198):
same constructor, ``forward((B,1,h,w) in [0,1]) -> (B,1,s*h,s*w)``, state_dict keys ``conv1.1.weight``,
``trunk.residual_unit.{1,3}.weight``, ``conv2.1.weight`` and the reference's Kaiming (fan_out, relu)
initialisation; the compute is ``srhip.drrn_engine.DRRNEngine``.  1-channel inputs; GPU only."""
import torch
import torch.nn as nn

from srhip.module_path import refresh_if_params_changed

__all__ = ['DRRN']


class _Conv(nn.Module):
    def __init__(self, co, ci):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(co, ci, 3, 3))


class RecursiveBlock(nn.Module):
    def __init__(self, num_channels, num_residual_unit):
        super().__init__()
        self.num_residual_unit = num_residual_unit
        # indices as in the reference's nn.Sequential(ReLU, Conv, ReLU, Conv)
        self.residual_unit = nn.ModuleList([nn.Identity(), _Conv(num_channels, num_channels), nn.Identity(),
                                            _Conv(num_channels, num_channels)])


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


class DRRN(nn.Module):
    def __init__(self, in_chans: int, upscale: int, num_residual_units: int):
        super().__init__()
        assert isinstance(upscale, int) and upscale > 0, upscale
        assert isinstance(in_chans, int) and in_chans > 0, in_chans
        if in_chans != 1:
            raise NotImplementedError("DRRN on libsrhip: 1-channel microscopy patches only")
        self.upscale, self.scale, self.in_chans = upscale, upscale, in_chans
        self.global_residual = None
        self.x_interp = None
        self.conv1 = nn.ModuleList([nn.Identity(), _Conv(128, in_chans)])
        self.trunk = RecursiveBlock(128, num_residual_units)
        self.conv2 = nn.ModuleList([nn.Identity(), _Conv(in_chans, 128)])
        self._engine = None
        self._initialize_weights()

    def _initialize_weights(self):                                    # network_drrn.py:128-136
        for m in self.modules():
            if isinstance(m, _Conv):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def flush(self):
        self.global_residual = None
        self.x_interp = None

    @property
    def engine(self):
        if self._engine is None:
            from srhip.drrn_engine import DRRNEngine
            self._engine = DRRNEngine(self)
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
            raise RuntimeError("DRRN (libsrhip) runs on the GPU only: move the model and the input to cuda; "
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
