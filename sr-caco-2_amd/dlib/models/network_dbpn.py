"""DBPN on libsrhip (reference dlib/models/network_dbpn.py:445-577; registry select_network.py:139-147): same
constructor, ``forward((B,1,h,w) in [0,1]) -> (B,1,s*h,s*w)``, the reference's state_dict keys (``feat0.conv.*``,
``feat0.act.weight``, ``up1.up_conv1.deconv.*`` ... ``output_conv.conv.*``) and its initialisation (kaiming-normal conv /
deconv weights, zero biases, PReLU 0.25); the compute is ``srhip.dbpn_engine.DBPNEngine`` (a tape graph over the
libsrhip kernels).  1-channel inputs; GPU only (CPU tensors raise)."""
import torch
import torch.nn as nn

from srhip.module_path import refresh_if_params_changed

__all__ = ['DBPN']


class _PReLU(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.full((1,), 0.25))       # nn.PReLU() default


class _Conv(nn.Module):
    def __init__(self, ci, co, k, transposed=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(ci, co, k, k) if transposed else torch.empty(co, ci, k, k))
        self.bias = nn.Parameter(torch.zeros(co))
        nn.init.kaiming_normal_(self.weight)                      # network_dbpn.py:521-530


class ConvBlock(nn.Module):                                       # network_dbpn.py:71-105 (norm None)
    def __init__(self, ci, co, k, stride, padding, activation='prelu'):
        super().__init__()
        self.conv = _Conv(ci, co, k)
        self.k, self.stride, self.padding = k, stride, padding
        if activation == 'prelu':
            self.act = _PReLU()
        self.activation = activation


class DeconvBlock(nn.Module):                                     # network_dbpn.py:108-143
    def __init__(self, ci, co, k, stride, padding):
        super().__init__()
        self.deconv = _Conv(ci, co, k, transposed=True)
        self.k, self.stride, self.padding = k, stride, padding
        self.act = _PReLU()


class UpBlock(nn.Module):                                         # network_dbpn.py:190-205
    def __init__(self, nf, k, s, p, num_stages=0):
        super().__init__()
        if num_stages:                                            # D_UpBlock :225-246
            self.conv = ConvBlock(nf * num_stages, nf, 1, 1, 0)
        self.up_conv1 = DeconvBlock(nf, nf, k, s, p)
        self.up_conv2 = ConvBlock(nf, nf, k, s, p)
        self.up_conv3 = DeconvBlock(nf, nf, k, s, p)


class DownBlock(nn.Module):                                       # network_dbpn.py:272-287, D_DownBlock :306-327
    def __init__(self, nf, k, s, p, num_stages=0):
        super().__init__()
        if num_stages:
            self.conv = ConvBlock(nf * num_stages, nf, 1, 1, 0)
        self.down_conv1 = ConvBlock(nf, nf, k, s, p)
        self.down_conv2 = DeconvBlock(nf, nf, k, s, p)
        self.down_conv3 = ConvBlock(nf, nf, k, s, p)


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, need_grad, *params):
        ctx.net = net
        from srhip import ops
        with ops.amp_inference(getattr(net, "amp", False) and not need_grad):       # --amp: inference only
            y = net.engine.forward(x, None, save=need_grad)
        return y.clone() if need_grad else y

    @staticmethod
    def backward(ctx, dy):
        net = ctx.net
        names = [k for k, _ in net.named_parameters()]
        grads = {k: torch.empty_like(p) for k, p in net.named_parameters()}
        net.engine.backward(dy.contiguous(), grads)
        return (None, None, None) + tuple(grads[k] for k in names)


class TapeNet(nn.Module):
    """What every tape-engine mirror shares: lazy engine, invalidation hooks, the TrainStep protocol."""
    engine_cls = None

    def _init_protocol(self, upscale, in_chans):
        if in_chans != 1:
            raise NotImplementedError(f"{type(self).__name__} on libsrhip: 1-channel microscopy patches only")
        self.upscale, self.scale, self.in_chans = upscale, upscale, in_chans
        self._engine = None

    @property
    def engine(self):
        if self._engine is None:
            self._engine = self._make_engine()
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

    def flush(self):
        pass

    def prepare_input(self, x):
        if not x.is_cuda:
            raise RuntimeError(f"{type(self).__name__} (libsrhip) runs on the GPU only: move the model and the input to "
                               f"cuda; there is no CPU fallback")
        assert x.dim() == 4 and x.shape[1] == self.in_chans, f'c: {x.shape}, img-nc: {self.in_chans}'
        return x.float().contiguous()[:, 0], x.shape[2], x.shape[3]

    def forward(self, x):
        self.flush()
        xi, h, w = self.prepare_input(x)
        params = [p for _, p in self.named_parameters()]
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        refresh_if_params_changed(self, params)   # stock torch.optim wrote the weights?
        return _NetFn.apply(xi, self, need_grad, *params)


class DBPN(TapeNet):
    def __init__(self, upscale: int = 2, in_chans: int = 3, base_filter: int = 64, feat: int = 256,
                 num_stages: int = 3):
        super().__init__()
        if upscale == 2:
            k, s, p = 6, 2, 2
        elif upscale == 4:
            k, s, p = 8, 4, 2
        elif upscale == 8:
            k, s, p = 12, 8, 2
        else:
            raise NotImplementedError(upscale)
        self._init_protocol(upscale, in_chans)
        self.num_stages = num_stages
        self.kernel, self.stride, self.padding = k, s, p
        self.feat0 = ConvBlock(in_chans, feat, 3, 1, 1)
        self.feat1 = ConvBlock(feat, base_filter, 1, 1, 0)
        self.up1 = UpBlock(base_filter, k, s, p)
        self.down1 = DownBlock(base_filter, k, s, p)
        self.up2 = UpBlock(base_filter, k, s, p)
        for i in range(2, 7):
            setattr(self, f"down{i}", DownBlock(base_filter, k, s, p, i))
            setattr(self, f"up{i + 1}", UpBlock(base_filter, k, s, p, i))
        self.output_conv = ConvBlock(num_stages * base_filter, in_chans, 3, 1, 1, activation=None)

    def _make_engine(self):
        from srhip.dbpn_engine import DBPNEngine
        return DBPNEngine(self)
