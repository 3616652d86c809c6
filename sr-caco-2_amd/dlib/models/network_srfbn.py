"""SRFBN on libsrhip (reference dlib/models/network_srfbn.py:586-721; registry select_network.py:130-137): same
constructor, ``forward((B,1,h,w) in [0,1]) -> (B,1,s*h,s*w)`` (the last of the ``num_steps`` predictions; all of them
in ``intermediate_outs`` for the trainer's curriculum loss, model_plain.py:202-232), the reference's state_dict keys
(``conv_in.0.*``, ``conv_in.1.weight``, ``block.upBlocks.{i}.0.*`` ..., the frozen ``sub_mean`` / ``add_mean`` 1x1
convs included) and PReLU(init 0.2); the compute is ``srhip.srfbn_engine.SRFBNEngine``.  1-channel inputs; GPU only."""
import torch
import torch.nn as nn

from dlib.models.network_dbpn import TapeNet

__all__ = ['SRFBN']


class _Conv(nn.Module):
    def __init__(self, ci, co, k, transposed=False):
        super().__init__()
        ref = nn.ConvTranspose2d(ci, co, k) if transposed else nn.Conv2d(ci, co, k)     # torch's default initialisation
        self.weight, self.bias = ref.weight, ref.bias


class _PReLU(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.full((1,), 0.2))        # activation('prelu'): init = slope 0.2 (network_srfbn.py:38-46)


def _block(ci, co, k, transposed=False, act=True):
    """ConvBlock / DeconvBlock in mode 'CNA' without a norm layer (network_srfbn.py:88-121): Sequential(conv[, PReLU])."""
    mods = [_Conv(ci, co, k, transposed)]
    if act:
        mods.append(_PReLU())
    return nn.Sequential(*mods)


class _MeanShift(nn.Module):
    """network_srfbn.py:123-133: a frozen 3-channel 1x1 conv that forward() never applies (the calls are commented out,
    :655,:671); kept for the state_dict."""
    def __init__(self, sign):
        super().__init__()
        self.weight = nn.Parameter(torch.eye(3).view(3, 3, 1, 1), requires_grad=False)
        self.bias = nn.Parameter(sign * 255. * torch.tensor([0.4488, 0.4371, 0.4040]), requires_grad=False)


class FeedbackBlock(nn.Module):
    def __init__(self, nf, groups, k):
        super().__init__()
        self.num_groups = groups
        self.compress_in = _block(2 * nf, nf, 1)
        self.upBlocks = nn.ModuleList([_block(nf, nf, k, transposed=True) for _ in range(groups)])
        self.downBlocks = nn.ModuleList([_block(nf, nf, k) for _ in range(groups)])
        self.uptranBlocks = nn.ModuleList([_block(nf * (i + 1), nf, 1) for i in range(1, groups)])
        self.downtranBlocks = nn.ModuleList([_block(nf * (i + 1), nf, 1) for i in range(1, groups)])
        self.compress_out = _block(groups * nf, nf, 1)


class SRFBN(TapeNet):
    def __init__(self, upscale: int = 2, in_chans: int = 3, num_features: int = 64, num_steps: int = 4,
                 num_groups: int = 6, act_type='prelu', norm_type=None):
        super().__init__()
        ksp = {2: (6, 2, 2), 3: (7, 3, 2), 4: (8, 4, 2), 8: (12, 8, 2)}
        if upscale not in ksp:
            raise NotImplementedError(upscale)
        if act_type != 'prelu' or norm_type is not None:
            raise NotImplementedError("SRFBN on libsrhip: act_type 'prelu', no norm layer (the registry's configuration)")
        self._init_protocol(upscale, in_chans)
        self.kernel, self.stride, self.padding = ksp[upscale]
        self.num_steps, self.num_features, self.upscale_factor = num_steps, num_features, upscale
        nf, k = num_features, self.kernel
        self.sub_mean = _MeanShift(-1)
        self.conv_in = _block(in_chans, 4 * nf, 3)
        self.feat_in = _block(4 * nf, nf, 1)
        self.block = FeedbackBlock(nf, num_groups, k)
        self.out = _block(nf, nf, k, transposed=True)
        self.conv_out = _block(nf, in_chans, 3, act=False)
        self.add_mean = _MeanShift(1)
        self.intermediate_outs = []

    def flush(self):
        self.intermediate_outs = []

    def _make_engine(self):
        from srhip.srfbn_engine import SRFBNEngine
        return SRFBNEngine(self)

    def forward(self, x):
        y = super().forward(x)
        self.intermediate_outs = list(self.engine.all_outs)       # num_steps predictions, the last one = the output
        return y
