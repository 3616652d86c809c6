"""NLSN on libsrhip (reference dlib/models/network_nlsn.py:296-369; registry select_network.py:149-160): same constructor,
``forward((B,1,h,w) in [0,1]) -> (B,1,s*h,s*w)`` and the reference's state_dict keys, shapes and order (``sub_mean.* /
add_mean.*`` -- frozen MeanShift convs, built but not applied :356,363 --, ``head.0.*``, ``body.{i}.conv_match.0.* /
conv_assembly.0.*`` for the attention blocks, ``body.{i}.body.{0,2}.*`` for the ResBlocks, ``tail.0.{0,2,4}.*``,
``tail.1.*``): released weights load with strict=True.  The compute is ``srhip.nlsn_engine.NLSNEngine``.  As in the
reference, the LSH rotations are drawn anew at every forward (:152-155), so two forwards of the same input differ at the
level the hashing decides; the token order inside a hash bucket is by token index here (the reference leaves it to
torch.sort).  Trains (srhip.tape.Tape.nlsa: the attention core differentiated in chunk-major dense form); 1-channel inputs; at least chunk_size pixels per image; GPU only."""
import math

import torch
import torch.nn as nn

from dlib.models.network_dbpn import TapeNet
from dlib.models.network_enlcn import _Conv, _MeanShift, _ResBlock

__all__ = ['NLSN']


class _NLSA(nn.Module):                                           # :131-143
    def __init__(self, channels, reduction=4):
        super().__init__()
        self.conv_match = nn.Sequential(_Conv(channels, channels // reduction, 3))
        self.conv_assembly = nn.Sequential(_Conv(channels, channels, 1))


class NLSN(TapeNet):
    def __init__(self, upscale: int = 2, n_resblocks: int = 32, n_feats: int = 256, n_hashes: int = 4,
                 res_scale: float = 0.1, img_range: float = 1., in_chans: int = 3, chunk_size: int = 144):
        super().__init__()
        if upscale & (upscale - 1) or upscale < 2:
            raise NotImplementedError(f"NLSN on libsrhip: power-of-two scales (got {upscale})")
        if n_feats % 16 or n_feats > 256:
            raise NotImplementedError(f"NLSN on libsrhip: n_feats a multiple of 16, <= 256 (got {n_feats})")
        self._init_protocol(upscale, in_chans)
        self.n_resblocks, self.n_feats, self.n_hashes = n_resblocks, n_feats, n_hashes
        self.res_scale, self.img_range, self.chunk_size = res_scale, img_range, chunk_size
        self.sub_mean = _MeanShift(-1.0)
        self.add_mean = _MeanShift(1.0)
        self.head = nn.Sequential(_Conv(in_chans, n_feats, 3))
        body = [_NLSA(n_feats)]
        for i in range(n_resblocks):
            body.append(_ResBlock(n_feats))
            if (i + 1) % 8 == 0:
                body.append(_NLSA(n_feats))
        body.append(_Conv(n_feats, n_feats, 3))
        self.body = nn.Sequential(*body)
        up = []
        for _ in range(int(math.log2(upscale))):
            up += [_Conv(n_feats, 4 * n_feats, 3), nn.PixelShuffle(2)]
        self.tail = nn.Sequential(nn.Sequential(*up), _Conv(n_feats, in_chans, 3))

    def _make_engine(self):
        from srhip.nlsn_engine import NLSNEngine
        return NLSNEngine(self)
