"""OmniSR on libsrhip (reference dlib/models/network_omni_sr.py:527-591; registry select_network.py:169-181): same
constructor, ``forward((B,1,h,w) in [0,1]) -> (B,1,s*h,s*w)`` (inputs zero-padded to multiples of the window, the output
cropped) and the reference's state_dict keys, shapes and order (``residual_layer.{g}.residual_layer.{b}.layer.{0,2,4,5,6,8,
10,11,12}.*``, ``residual_layer.{g}.residual_layer.{n}``, ``residual_layer.{g}.esa.*``, ``input``, ``output``, ``up.0``).
The compute is ``srhip.omnisr_engine.OmniSREngine``.  Training through the tape graph of the engine (window-multiple patches); 1-channel inputs; GPU only."""
import torch
import torch.nn as nn

from dlib.models.network_dbpn import TapeNet

__all__ = ['OmniSR']


class _LayerNorm2d(nn.Module):                                    # :55-65
    def __init__(self, channels):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))


class _SE(nn.Module):                                             # :133-148 (Reduce / Rearrange hold no parameters)
    def __init__(self, dim, shrinkage_rate=0.25):
        super().__init__()
        hidden = int(dim * shrinkage_rate)
        self.gate = nn.Sequential(nn.Identity(), nn.Linear(dim, hidden, bias=False), nn.SiLU(), nn.Linear(hidden, dim, bias=False),
                                  nn.Sigmoid(), nn.Identity())


class _MBConvResidual(nn.Module):                                 # :151-189, expansion 1
    def __init__(self, dim):
        super().__init__()
        self.fn = nn.Sequential(nn.Conv2d(dim, dim, 1), nn.GELU(), nn.Conv2d(dim, dim, 3, padding=1, groups=dim), nn.GELU(),
                                _SE(dim), nn.Conv2d(dim, dim, 1))


class _Attention(nn.Module):                                      # :212-256
    def __init__(self, dim, dim_head, window_size, with_pe):
        super().__init__()
        self.heads = dim // dim_head
        self.with_pe = with_pe
        self.to_qkv = nn.Linear(dim, dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(dim, dim, bias=False), nn.Dropout(0.0))
        if with_pe:
            self.rel_pos_bias = nn.Embedding((2 * window_size - 1) ** 2, self.heads)
            pos = torch.arange(window_size)
            grid = torch.stack(torch.meshgrid(pos, pos, indexing="ij")).reshape(2, -1).t()
            rel = grid[:, None, :] - grid[None, :, :] + (window_size - 1)
            self.register_buffer('rel_pos_indices', (rel * torch.tensor([2 * window_size - 1, 1])).sum(dim=-1), persistent=False)


class _PreNormResidual(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class _ConvPreNormResidual(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = _LayerNorm2d(dim)
        self.fn = fn


class _GatedFFN(nn.Module):                                       # :308-322, no biases
    def __init__(self, dim):
        super().__init__()
        self.project_in = nn.Conv2d(dim, dim * 2, 1, bias=False)
        self.dwconv = nn.Conv2d(dim * 2, dim * 2, 3, padding=1, groups=dim * 2, bias=False)
        self.project_out = nn.Conv2d(dim, dim, 1, bias=False)


class _ChannelAttention(nn.Module):                               # :332-351 / :381-400
    def __init__(self, dim, heads, window_size):
        super().__init__()
        self.heads, self.ps = heads, window_size
        self.temperature = nn.Parameter(torch.ones(heads, 1, 1))
        self.qkv = nn.Conv2d(dim, dim * 3, 1, bias=False)
        self.qkv_dwconv = nn.Conv2d(dim * 3, dim * 3, 3, padding=1, groups=dim * 3, bias=False)
        self.project_out = nn.Conv2d(dim, dim, 1, bias=False)


class _OSABlock(nn.Module):                                       # :430-492 (the Rearrange entries keep the indices)
    def __init__(self, c, window_size, with_pe):
        super().__init__()
        def att():
            return _PreNormResidual(c, _Attention(c, c // 4, window_size, with_pe))
        def ffn():
            return _ConvPreNormResidual(c, _GatedFFN(c))
        def ca():
            return _ConvPreNormResidual(c, _ChannelAttention(c, 4, window_size))
        self.layer = nn.Sequential(_MBConvResidual(c), nn.Identity(), att(), nn.Identity(), ffn(), ca(), ffn(), nn.Identity(),
                                   att(), nn.Identity(), ffn(), ca(), ffn())


class _ESA(nn.Module):                                            # :85-102
    def __init__(self, f, n_feats):
        super().__init__()
        self.conv1 = nn.Conv2d(n_feats, f, 1)
        self.conv_f = nn.Conv2d(f, f, 1)
        self.conv2 = nn.Conv2d(f, f, 3, stride=2, padding=0)
        self.conv3 = nn.Conv2d(f, f, 3, padding=1)
        self.conv4 = nn.Conv2d(f, n_feats, 1)


class _OSAG(nn.Module):                                           # :495-524
    def __init__(self, c, bias, block_num, window_size, pe):
        super().__init__()
        self.residual_layer = nn.Sequential(*([_OSABlock(c, window_size, pe) for _ in range(block_num)]
                                              + [nn.Conv2d(c, c, 1, 1, 0, bias=bias)]))
        self.esa = _ESA(max(c // 4, 16), c)


class OmniSR(TapeNet):
    amp_takes_effect = False      # every contraction of this net runs on the exact-f32 kernels: --amp changes nothing
    eval_graph_default = True     # ModelPlain.test() replays the evaluation forward from a hipGraph (SRHIP_EVAL_GRAPH=0: eager)
    def __init__(self, input_shape: int = 3, upscale: int = 2, num_feat: int = 64, res_num: int = 5, bias: bool = True,
                 window_size: int = 8, block_num: int = 4, pe: bool = True, ffn_bias: bool = True):
        super().__init__()
        if window_size != 8 or num_feat % 4 or num_feat // 4 > 16:
            raise NotImplementedError("OmniSR on libsrhip: window 8, num_feat a multiple of 4, <= 64")
        self._init_protocol(upscale, input_shape)
        self.num_feat, self.window_size = num_feat, window_size
        self.residual_layer = nn.Sequential(*[_OSAG(num_feat, bias, block_num, window_size, pe) for _ in range(res_num)])
        self.input = nn.Conv2d(input_shape, num_feat, 3, 1, 1, bias=bias)
        self.output = nn.Conv2d(num_feat, num_feat, 3, 1, 1, bias=bias)
        self.up = nn.Sequential(nn.Conv2d(num_feat, input_shape * upscale ** 2, 3, padding=1, bias=bias), nn.PixelShuffle(upscale))

    def _make_engine(self):
        from srhip.omnisr_engine import OmniSREngine
        return OmniSREngine(self)
