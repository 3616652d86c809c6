"""ENLCN on libsrhip (reference dlib/models/network_enlcn.py:369-448; registry select_network.py:92-101): same
constructor, ``forward((B,1,h,w) in [0,1]) -> (B,1,s*h,s*w)`` and the reference's state_dict keys, shapes and order
(``sub_mean.* / add_mean.*`` -- the frozen MeanShift convs, built but not applied :423,441 --, ``head.0.*``,
``body.{i}.conv_match1.0.* ... body.{i}.attn_fn.projection_matrix`` for the ENLCA blocks, ``body.{i}.body.{0,2}.*`` for
the ResBlocks, ``tail.0.{0,2,4}.*``, ``tail.1.*``): released weights load with strict=True.  The compute is
``srhip.enlcn_engine.ENLCNEngine`` (tape graph over the libsrhip kernels): evaluation and training (ENLCA's autograd:
srhip.tape.Tape.enlca; gradients pinned against the reference's, tests/golden/g39_enlcn_grad.npz); 1-channel inputs; GPU only."""
import math

import torch
import torch.nn as nn

from dlib.models.network_dbpn import TapeNet

__all__ = ['ENLCN']


class _Conv(nn.Module):
    def __init__(self, ci, co, k):
        super().__init__()
        ref = nn.Conv2d(ci, co, k, padding=k // 2)                # default_conv :129-136: nn.Conv2d's own initialisation
        self.weight, self.bias = ref.weight, ref.bias


class _MeanShift(nn.Module):                                      # :26-37, 3-channel whatever in_chans is
    def __init__(self, sign):
        super().__init__()
        self.weight = nn.Parameter(torch.eye(3).view(3, 3, 1, 1), requires_grad=False)
        self.bias = nn.Parameter(sign * torch.tensor([0.4488, 0.4371, 0.4040]), requires_grad=False)


def _orthogonal_chunk(cols):                                      # :43-50
    q, _ = torch.linalg.qr(torch.randn(cols, cols))
    return q.t()


def _projection_matrix(rows, cols):                               # gaussian_orthogonal_random_matrix :52-81, scaling 0
    blocks = [_orthogonal_chunk(cols) for _ in range(rows // cols)]
    rem = rows - (rows // cols) * cols
    if rem:
        blocks.append(_orthogonal_chunk(cols)[:rem])
    mult = torch.randn(rows, cols).norm(dim=1)
    return torch.diag(mult) @ torch.cat(blocks)


class _ENLA(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.register_buffer('projection_matrix', _projection_matrix(128, dim))


class _ENLCA(nn.Module):                                          # :303-366
    def __init__(self, channel, reduction=4):
        super().__init__()
        self.conv_match1 = nn.Sequential(_Conv(channel, channel // reduction, 1))
        self.conv_match2 = nn.Sequential(_Conv(channel, channel // reduction, 1))
        self.conv_assembly = nn.Sequential(_Conv(channel, channel, 1))
        self.attn_fn = _ENLA(channel // reduction)


class _ResBlock(nn.Module):                                       # :139-160
    def __init__(self, nf):
        super().__init__()
        self.body = nn.Sequential(_Conv(nf, nf, 3), nn.ReLU(True), _Conv(nf, nf, 3))


class ENLCN(TapeNet):
    def __init__(self, upscale: int = 2, n_resblock: int = 32, n_feats: int = 256, res_scale: float = 0.1,
                 img_range: float = 1., in_chans: int = 3):
        super().__init__()
        if upscale & (upscale - 1) or upscale < 2:
            raise NotImplementedError(f"ENLCN on libsrhip: power-of-two scales (got {upscale})")
        if n_feats % 16 or n_feats > 256:
            raise NotImplementedError(f"ENLCN on libsrhip: n_feats a multiple of 16, <= 256 (got {n_feats})")
        self._init_protocol(upscale, in_chans)
        self.n_resblock, self.n_feats, self.res_scale, self.img_range = n_resblock, n_feats, res_scale, img_range
        self.sub_mean = _MeanShift(-1.0)
        self.add_mean = _MeanShift(1.0)
        self.head = nn.Sequential(_Conv(in_chans, n_feats, 3))
        body = [_ENLCA(n_feats)]
        for i in range(n_resblock):
            body.append(_ResBlock(n_feats))
            if (i + 1) % 8 == 0:
                body.append(_ENLCA(n_feats))
        body.append(_Conv(n_feats, n_feats, 3))
        self.body = nn.ModuleList(body)
        up = []
        for _ in range(int(math.log2(upscale))):
            up += [_Conv(n_feats, 4 * n_feats, 3), nn.PixelShuffle(2)]
        self.tail = nn.Sequential(nn.Sequential(*up), _Conv(n_feats, in_chans, 3))

    def _make_engine(self):
        from srhip.enlcn_engine import ENLCNEngine
        return ENLCNEngine(self)
