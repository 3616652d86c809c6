"""Import shim for the *real* reference (``/root/reference``) -- test tooling.

Only ``oracle/make_goldens.py`` uses this, and only in the build container:
the GPU box has no ``/root/reference``.  It registers empty namespace packages
for ``dlib`` / ``dlib.losses`` (skipping their heavy ``__init__``) and stubs the
third-party modules the image lacks, so that the reference's SR modules import
and run on CPU (recipe: SURVEY.md section 8c).
"""
import collections.abc
import sys
import types

import torch.nn as nn

REF = "/root/reference"


def _pkg(name, path=None):
    m = types.ModuleType(name)
    if path:
        m.__path__ = [path]
    sys.modules[name] = m
    return m


def install():
    if "dlib" in sys.modules and getattr(sys.modules["dlib"], "_ref_shim", False):
        return
    sys.dont_write_bytecode = True
    d = _pkg("dlib", REF + "/dlib")
    d._ref_shim = True
    _pkg("dlib.losses", REF + "/dlib/losses")
    for n in ["cv2", "torchvision", "torchvision.utils", "munch", "tifffile",
              "more_itertools", "texttable", "kornia", "omegaconf", "fairscale"]:
        _pkg(n)
    sys.modules["torchvision.utils"].make_grid = None
    p = _pkg("pynvml", "/x")
    ps = _pkg("pynvml.smi")
    ps.nvidia_smi = None
    p.smi = ps
    _pkg("timm", "/x")
    _pkg("timm.models", "/x")
    tl = _pkg("timm.models.layers")
    tl.to_2tuple = lambda x: tuple(x) if isinstance(
        x, collections.abc.Iterable) and not isinstance(x, str) else (x, x)
    tl.trunc_normal_ = lambda t, mean=0., std=1., a=-2., b=2.: \
        nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    class DropPath(nn.Module):  # timm semantics: per-sample mask / keep_prob
        def __init__(self, drop_prob=0., scale_by_keep=True):
            super().__init__()
            self.p = drop_prob
            self.sk = scale_by_keep

        def forward(self, x):
            if self.p == 0. or not self.training:
                return x
            k = 1 - self.p
            r = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(k)
            if k > 0 and self.sk:
                r.div_(k)
            return x * r
    tl.DropPath = DropPath
    # network_grl.py:13-14 reads two names at import: OmegaConf.create (a dict with attribute access around three
    # constructor flags) and fairscale's checkpoint_wrapper (activation checkpointing, off in the registry's config)
    sys.modules["omegaconf"].OmegaConf = type("OmegaConf", (), {"create": staticmethod(lambda d: types.SimpleNamespace(**d))})
    fn = _pkg("fairscale.nn")
    fn.checkpoint_wrapper = lambda m, **k: m
    sys.modules["fairscale"].nn = fn
