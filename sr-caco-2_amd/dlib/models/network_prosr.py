"""ProSR on libsrhip (reference dlib/models/network_prosr.py:256-470; registry select_network.py:110-128): same
constructor, ``forward((B,1,h,w) in [0,1]) -> (B,1,s*h,s*w)`` with the coarser pyramid levels' predictions in
``intermediate_outs`` (the trainer's multi-scale loss, model_plain.py:234-275), the reference's state_dict keys
(``init_conv_{s}.conv.1.*``, ``pyramid_residual_{i}.residual_denseblock_{b}.dense_block.denselayer{l}.conv_1.*`` ...)
and its initialisation (torch defaults, zero conv biases); the compute is ``srhip.prosr_engine.ProSREngine``.  The
residual-dense-block configuration of the registry (residual_denseblock=True, ps_woReLU=False); 1-channel inputs; GPU only."""
from collections import OrderedDict
from math import log2

import torch
import torch.nn as nn

from dlib.models.network_dbpn import TapeNet

__all__ = ['ProSR']


class _C(nn.Module):
    """a conv's parameters (nn.Conv2d default initialisation, bias zeroed: init_weights, network_prosr.py:226-231)"""
    def __init__(self, ci, co, k, bias=True):
        super().__init__()
        ref = nn.Conv2d(ci, co, k, bias=bias)
        self.weight = ref.weight
        if bias:
            self.bias = nn.Parameter(torch.zeros(co))
        else:
            self.register_parameter('bias', None)


class Conv2d(nn.Module):
    """reference Conv2d (network_prosr.py:38-86): Sequential(ReflectionPad2d, nn.Conv2d) -> parameters under `conv.1.`"""
    def __init__(self, ci, co, k=3):
        super().__init__()
        self.conv = nn.Sequential(OrderedDict([("0", nn.Identity()), ("1", _C(ci, co, k))]))


class _DenseLayer(nn.Module):
    def __init__(self, ci, growth, bn_size):
        super().__init__()
        self.conv_1 = _C(ci, bn_size * growth, 1)
        self.conv_2 = Conv2d(bn_size * growth, growth, 3)


class _Comp(nn.Module):
    def __init__(self, ci, co):
        super().__init__()
        self.conv1 = _C(ci, co, 1, bias=False)


class DenseResidualBlock(nn.Module):
    def __init__(self, num_layers, ci, bn_size, growth):
        super().__init__()
        self.dense_block = nn.Sequential(OrderedDict(
            (f"denselayer{l + 1}", _DenseLayer(ci + l * growth, growth, bn_size)) for l in range(num_layers)))
        self.comp = _Comp(ci + num_layers * growth, ci)


class _Upsampler(nn.Module):
    def __init__(self, planes):
        super().__init__()
        self.m = nn.Sequential(OrderedDict([("0", Conv2d(planes, 4 * planes, 3))]))      # + PixelShuffle(2) + ReLU


class ProSR(TapeNet):
    def __init__(self, upscale: int = 8, in_chans: int = 3, residual_denseblock: bool = True,
                 num_init_features: int = 160, bn_size: int = 4, growth_rate: int = 40, ps_woReLU: bool = False,
                 level_config: list = [[8, 8, 8, 8, 8, 8, 8, 8, 8], [8, 8, 8], [8]], level_compression: int = -1,
                 res_factor: float = 0.2, max_num_feature: int = 312, block_compression: float = 0.4):
        super().__init__()
        if not residual_denseblock:
            raise NotImplementedError("ProSR on libsrhip: the residual-dense-block form (the registry's configuration)")
        self._init_protocol(upscale, in_chans)
        self.max_scale = upscale
        self.n_pyramids = int(log2(upscale))
        assert 2 ** self.n_pyramids == upscale and len(level_config) >= self.n_pyramids
        self.res_factor, self.ps_woReLU = res_factor, ps_woReLU
        self.level_config = [list(c) for c in level_config[:self.n_pyramids]]
        nf = num_init_features
        for s in range(1, self.n_pyramids + 1):
            setattr(self, f"init_conv_{s}", Conv2d(in_chans, num_init_features, 3))
        self.level_planes = []
        for i in range(self.n_pyramids):
            mods = OrderedDict()
            if i != 0:
                out_planes = num_init_features if level_compression <= 0 else int(level_compression * nf)
                mods[f"compression_{i}"] = _Comp(nf, out_planes)
                nf = out_planes
            for b, nl in enumerate(self.level_config[i]):
                mods[f"residual_denseblock_{b + 1}"] = DenseResidualBlock(nl, nf, bn_size, growth_rate)
            fin = OrderedDict()
            if nf > max_num_feature:
                fin["final_comp"] = _Comp(nf, max_num_feature)
                nf = max_num_feature
            fin["final_conv"] = Conv2d(nf, nf, 3)
            mods["final_conv"] = nn.Sequential(fin)
            setattr(self, f"pyramid_residual_{i + 1}", nn.Sequential(mods))
            setattr(self, f"pyramid_residual_{i + 1}_residual_upsampler", _Upsampler(nf))
            setattr(self, f"reconst_{i + 1}", nn.Sequential(OrderedDict([("final_conv", Conv2d(nf, in_chans, 3))])))
            self.level_planes.append(nf)
        self.intermediate_outs = []

    def flush(self):
        self.intermediate_outs = []

    def _make_engine(self):
        from srhip.prosr_engine import ProSREngine
        return ProSREngine(self)

    def forward(self, x):
        y = super().forward(x)
        self.intermediate_outs = list(self.engine.intermediate_outs or [])
        return y
