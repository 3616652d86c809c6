"""DFCAN on libsrhip (reference dlib/models/network_dfcan.py:86-116; registry select_network.py:162-167): same constructor
(``input_shape`` = input channels, ``upscale``), ``forward((B,1,h,w) in [0,1]) -> (B,1,s*h,s*w)`` (the last op is a
sigmoid) and the reference's state_dict keys, shapes and order (``input.0.*``, ``RGs.{g}.RCABs.{r}.conv_gelu1.0.* ...
conv_sigmoid.0.*``, ``conv_gelu.0.*``, ``conv_sigmoid.0.*``).  The compute is ``srhip.dfcan_engine.DFCANEngine``: the
spectrum's magnitude as a separable DFT with f64 accumulation.  Trains (srhip.tape.Tape.fourier_gate: the spectrum magnitude differentiated through stock torch.fft); 1-channel inputs;
LR images up to 256 x 256; GPU only."""
import torch.nn as nn

from dlib.models.network_dbpn import TapeNet

__all__ = ['DFCAN']


def _conv(ci, co, k):
    return nn.Sequential(nn.Conv2d(ci, co, kernel_size=k, stride=1, padding=k // 2))


class _RCAB(nn.Module):                                           # :39-57 (activations live in the engine)
    def __init__(self):
        super().__init__()
        self.conv_gelu1 = _conv(64, 64, 3)
        self.conv_gelu2 = _conv(64, 64, 3)
        self.conv_relu1 = _conv(64, 64, 3)
        self.conv_relu2 = _conv(64, 4, 1)
        self.conv_sigmoid = _conv(4, 64, 1)


class _ResGroup(nn.Module):                                       # :73-84
    def __init__(self, n_RCAB=4):
        super().__init__()
        self.RCABs = nn.Sequential(*[_RCAB() for _ in range(n_RCAB)])


class DFCAN(TapeNet):
    def __init__(self, input_shape, upscale=2):
        super().__init__()
        self._init_protocol(upscale, input_shape)
        self.input = _conv(input_shape, 64, 3)
        self.RGs = nn.Sequential(*[_ResGroup(4) for _ in range(4)])
        self.conv_gelu = _conv(64, 64 * upscale ** 2, 3)
        self.pixel_shuffle = nn.PixelShuffle(upscale)
        self.conv_sigmoid = _conv(64, input_shape, 3)

    def _make_engine(self):
        from srhip.dfcan_engine import DFCANEngine
        return DFCANEngine(self)
