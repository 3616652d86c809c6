"""ACT on libsrhip (reference dlib/models/network_act.py:321-541; registry select_network.py): same constructor,
``forward((B,1,h,w) in [0,1]) -> (B,1,s*h,s*w)`` and the reference's state_dict keys, shapes and order (``sub_mean / add_mean``,
``head.{0,1,2}``, ``linear_encoding``, ``mhsa_block.{i}.{0,1}``, ``csta_block.{i}.{0..4}``, ``cnn_branch.{g}.body.{r}`` and the
unused ``cnn_branch.4`` conv, ``fusion_block / fusion_mlp / fusion_cnn``, ``conv_last``, ``tail``): released weights load with
strict=True.  The compute is ``srhip.act_engine.ACTEngine``.  Training through the tape graph of the engine; 1-channel inputs; images of
at least 6 x 6 pixels; GPU only."""
import math

import torch
import torch.nn as nn

from dlib.models.network_dbpn import TapeNet

__all__ = ['ACT']


def _conv(ci, co, k, bias=True):
    return nn.Conv2d(ci, co, k, padding=k // 2, bias=bias)


class _MeanShift(nn.Conv2d):                                      # :36-47
    def __init__(self, sign):
        super().__init__(3, 3, kernel_size=1)
        self.weight.data = torch.eye(3).view(3, 3, 1, 1)
        self.bias.data = sign * torch.tensor([0.4488, 0.4371, 0.4040])
        for p in self.parameters():
            p.requires_grad = False


class _ResBlock(nn.Module):                                       # :50-76
    def __init__(self, nf, k):
        super().__init__()
        self.body = nn.Sequential(_conv(nf, nf, k), nn.ReLU(True), _conv(nf, nf, k))


class _PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class _PreNorm2(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.fn = fn


class _FeedForward(nn.Module):                                    # :136-148
    def __init__(self, dim, mlp_dim):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, mlp_dim), nn.GELU(), nn.Dropout(0.0), nn.Linear(mlp_dim, dim), nn.Dropout(0.0))


class _SelfAttention(nn.Module):                                  # :151-183
    def __init__(self, dim, heads, dim_head):
        super().__init__()
        self.to_qkv = nn.Linear(dim, dim_head * heads * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(dim_head * heads, dim), nn.Dropout(0.0))


class _CrossAttention(nn.Module):                                 # :186-227
    def __init__(self, dim, heads, dim_head):
        super().__init__()
        self.to_q = nn.Linear(dim, dim_head * heads, bias=False)
        self.to_kv = nn.Linear(dim, dim_head * heads * 2, bias=False)
        self.to_out = nn.Sequential(nn.Linear(dim_head * heads, dim), nn.Dropout(0.0))


class _CALayer(nn.Module):                                        # :230-247
    def __init__(self, channel, reduction):
        super().__init__()
        self.conv_du = nn.Sequential(nn.Conv2d(channel, channel // reduction, 1), nn.ReLU(True),
                                     nn.Conv2d(channel // reduction, channel, 1), nn.Sigmoid())


class _RCAB(nn.Module):                                           # :250-277
    def __init__(self, nf, reduction):
        super().__init__()
        self.body = nn.Sequential(_conv(nf, nf, 3), nn.ReLU(True), _conv(nf, nf, 3), _CALayer(nf, reduction))


class _ResidualGroup(nn.Module):                                  # :280-301
    def __init__(self, nf, reduction, n_resblocks):
        super().__init__()
        self.body = nn.Sequential(*([_RCAB(nf, reduction) for _ in range(n_resblocks)] + [_conv(nf, nf, 3)]))


class _FB(nn.Module):                                             # :304-318
    def __init__(self, nf):
        super().__init__()
        self.body = nn.Sequential(_conv(nf, nf, 1, bias=False), nn.ReLU(True), _conv(nf, nf, 1, bias=False))


class ACT(TapeNet):
    def __init__(self, upscale: int = 2, in_chans: int = 3, img_range: float = 1.0, n_feats: int = 64, n_resgroups: int = 4,
                 n_resblocks: int = 12, reduction: int = 16, n_heads: int = 8, n_layers: int = 8, dropout_rate: float = 0.0,
                 n_fusionblocks: int = 4, token_size: int = 3, expansion_ratio: int = 4):
        super().__init__()
        if upscale & (upscale - 1) or upscale < 2:
            raise NotImplementedError(f"ACT on libsrhip: power-of-two scales (got {upscale})")
        if dropout_rate != 0.0:
            raise NotImplementedError("ACT on libsrhip: dropout_rate 0 (evaluation)")
        if n_resgroups < n_fusionblocks or n_layers // 2 < n_fusionblocks:
            raise ValueError("ACT: n_resgroups and n_layers // 2 must cover n_fusionblocks (network_act.py:477-516)")
        self._init_protocol(upscale, in_chans)
        self.n_feats, self.n_resblocks, self.n_heads = n_feats, n_resblocks, n_heads
        self.token_size, self.n_fusionblocks = token_size, n_fusionblocks
        emb = n_feats * token_size ** 2
        self.embedding_dim = emb
        hidden = emb * expansion_ratio
        dh = emb // n_heads
        self.dim_head = dh
        self.sub_mean = _MeanShift(-1.0)
        self.add_mean = _MeanShift(1.0)
        self.head = nn.Sequential(_conv(in_chans, n_feats, 3), _ResBlock(n_feats, 5), _ResBlock(n_feats, 5))
        self.linear_encoding = nn.Linear(emb, emb)
        self.mhsa_block = nn.ModuleList([
            nn.ModuleList([_PreNorm(emb, _SelfAttention(emb, n_heads, dh)), _PreNorm(emb, _FeedForward(emb, hidden))])
            for _ in range(n_layers // 2)])
        self.csta_block = nn.ModuleList([
            nn.ModuleList([
                nn.Sequential(nn.LayerNorm(emb * 2), nn.Linear(emb * 2, emb // 2), nn.GELU(), nn.Linear(emb // 2, emb // 2)),
                _PreNorm2(emb // 2, _CrossAttention(emb // 2, n_heads // 2, dh)),
                _PreNorm2(emb // 2, _CrossAttention(emb // 2, n_heads // 2, dh)),
                nn.Sequential(nn.LayerNorm(emb // 2), nn.Linear(emb // 2, emb // 2), nn.GELU(), nn.Linear(emb // 2, emb * 2)),
                _PreNorm(emb, _FeedForward(emb, hidden)),
            ]) for _ in range(n_layers // 2)])
        self.cnn_branch = nn.Sequential(*([_ResidualGroup(n_feats, reduction, n_resblocks) for _ in range(n_resgroups)]
                                          + [_conv(n_feats, n_feats, 3)]))
        self.fusion_block = nn.ModuleList([nn.Sequential(*[_FB(n_feats * 2) for _ in range(4)]) for _ in range(n_fusionblocks)])
        self.fusion_mlp = nn.ModuleList([
            nn.Sequential(nn.LayerNorm(emb), nn.Linear(emb, hidden), nn.GELU(), nn.Linear(hidden, emb))
            for _ in range(n_fusionblocks - 1)])
        self.fusion_cnn = nn.ModuleList([
            nn.Sequential(_conv(n_feats, n_feats, 3), nn.ReLU(True), _conv(n_feats, n_feats, 3))
            for _ in range(n_fusionblocks - 1)])
        self.conv_last = _conv(n_feats * 2, n_feats, 3)
        up = []
        for _ in range(int(math.log2(upscale))):
            up += [_conv(n_feats, 4 * n_feats, 3), nn.PixelShuffle(2)]
        self.tail = nn.Sequential(nn.Sequential(*up), _conv(n_feats, in_chans, 3))

    def _make_engine(self):
        from srhip.act_engine import ACTEngine
        return ACTEngine(self)
