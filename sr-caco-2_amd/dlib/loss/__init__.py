"""SR losses on libsrhip: MasterLoss + L1 / L2 / NegativeSsim, and the optional
Charbonnier / L2Sum / local-variation terms.

Mirrors the reference's ``dlib.loss`` surface on the hot path
(dlib/loss/master.py:19-56, dlib/loss/core.py:17-127, dlib/loss/main.py:45-99,
154-186): same class names, constructor keywords, ``forward(epoch=, y_pred=,
y_target=, trg_per_pixel_weight=, model=)`` call, ``l_holder`` / ``n_holder``
bookkeeping and ``update_t``.  Each term runs ONE fused HIP kernel sequence that
produces the value and d loss / d y_pred together; autograd only scales that
stored gradient.  Of the reference's 13 optional terms (off by default,
utils_config.py:279-374) Charbonnier, L2Sum, ImageGradientLoss, LaplacianFilterLoss,
LocalVariationLoss and their three Norm* variants are built the same way
(dlib/loss/main.py:102-151,328-674), as are BoundedPrediction (extended log barrier,
:189-237 + dlib/losses/elb.py), WeightsSparsityLoss (:938-959) and LocalMoments (:240-325);
HistogramMatch and KDEMatch with all their metrics (NORM1 / NORM2 / KL / BHATTACHARYYA, :677-898);
CrossEntropyL (it needs a segmentation head) is not (NotImplementedError on use).
"""
import re

import torch
import torch.nn as nn

from srhip import ops

__all__ = ['MasterLoss', 'ElementaryLoss', 'L1', 'L2', 'NegativeSsim', 'L2Sum', 'Charbonnier',
           'ImageGradientLoss', 'LaplacianFilterLoss', 'LocalVariationLoss', 'NormImageGradientLoss',
           'NormLaplacianFilterLoss', 'NormLocalVariationLoss', 'BoundedPrediction', 'WeightsSparsityLoss',
           'LocalMoments', 'HistogramMatch', 'KDEMatch']

KL, BH = 'KL', 'BHATTACHARYYA'      # dlib/utils/constants.py:699-700

NORM1, NORM2 = '1', '2'      # dlib/utils/constants.py:696-697


def _snake(name):
    s1 = re.sub('(.)([A-Z][a-z]+)', r'\1_\2', name)
    return re.sub('([a-z0-9])([A-Z])', r'\1_\2', s1).lower()


class _FusedLoss(torch.autograd.Function):
    """value + gradient computed together by ``kernel(pred, grad_out, value_out)``."""

    @staticmethod
    def forward(ctx, pred, kernel):
        if not pred.is_cuda:
            raise RuntimeError("dlib.loss (libsrhip) runs on the GPU only; there is no CPU fallback")
        p = pred.detach().float().contiguous()
        grad = torch.empty_like(p) if pred.requires_grad else None
        val = torch.empty(1, device=p.device, dtype=torch.float32)
        kernel(p, grad, val)
        ctx.save_for_backward(grad)
        return val.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


class ElementaryLoss(nn.Module):
    def __init__(self, cuda_id=0, name=None, lambda_=1., elb=nn.Identity(), start_epoch=None,
                 end_epoch=None, restore_range=False, color_min=0, color_max=255,
                 use_residuals=False):
        super().__init__()
        self._name, self.lambda_, self.elb = name, lambda_, elb
        self.start_epoch = start_epoch
        self.end_epoch = None if end_epoch == -1 else end_epoch
        self.c_epoch = 0
        self._device = torch.device(cuda_id) if not isinstance(cuda_id, torch.device) else cuda_id
        assert not use_residuals, "use_residuals is not on the hot path"
        self.use_residuals = use_residuals

    @property
    def _zero(self):   # value of a switched-off term (lazy: no device touch at construction)
        return torch.zeros(1, device=self._device)

    def is_on(self, _epoch=None):
        e = self.c_epoch if _epoch is None else _epoch
        s, t = self.start_epoch, self.end_epoch
        if s is None and t is None:
            return True
        if s is not None and t is not None:
            return s <= e <= t
        return e <= t if s is None else e >= s

    def update_t(self):
        # core.py:80-82 tests for dlib.loss.elb.ELB -- NOT the dlib.losses.elb.ELB that define_loss
        # builds (see dlib/loss/elb.py): kept, so the barrier schedule behaves as in the reference
        from dlib.loss.elb import ELB as _CoreELB
        if isinstance(self.elb, _CoreELB):
            self.elb.update_t()

    @property
    def __name__(self):
        return _snake(self.__class__.__name__) if self._name is None else self._name

    @staticmethod
    def sanity_check_trg_per_pixel_weight(w, y):
        assert y.ndim == 4 and w.ndim == 4 and w.shape[1] == 1
        assert (w.shape[0], w.shape[2], w.shape[3]) == (y.shape[0], y.shape[2], y.shape[3])

    def forward(self, epoch, y_pred=None, y_target=None, trg_per_pixel_weight=None, model=None):
        self.c_epoch = epoch


class L1(ElementaryLoss):
    """lambda * mean(|y_pred - y_target| (* weight)); dlib/loss/main.py:45-76."""

    def forward(self, epoch, y_pred=None, y_target=None, trg_per_pixel_weight=None, model=None):
        super().forward(epoch=epoch)
        if not self.is_on():
            return self._zero
        w = trg_per_pixel_weight
        if w is not None:
            self.sanity_check_trg_per_pixel_weight(w, y_target)
            assert y_pred.shape[1] == 1, "per-pixel weights: 1-channel predictions"
            w = w.float().contiguous()
        t = y_target.float().contiguous()
        return _FusedLoss.apply(y_pred, lambda p, g, v: ops.loss_l1l2(p, t, 0, self.lambda_, w, g, v))


class L2(ElementaryLoss):
    """lambda * mean((y_pred - y_target)^2); dlib/loss/main.py:79-99."""

    def forward(self, epoch, y_pred=None, y_target=None, trg_per_pixel_weight=None, model=None):
        super().forward(epoch=epoch)
        if not self.is_on():
            return self._zero
        t = y_target.float().contiguous()
        return _FusedLoss.apply(y_pred, lambda p, g, v: ops.loss_l1l2(p, t, 1, self.lambda_, None, g, v))


class NegativeSsim(ElementaryLoss):
    """-lambda * mean_b(mean_hw(ssim_map)), Gaussian sigma-1.5 window (default 11,
    README recipe 19), zero padding; dlib/loss/main.py:154-186, dlib/loss/ssim.py."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.window_size = 11

    def set_window_size(self, window_size):
        assert isinstance(window_size, int) and window_size > 0
        self.window_size = window_size

    def forward(self, epoch, y_pred=None, y_target=None, trg_per_pixel_weight=None, model=None):
        super().forward(epoch=epoch)
        if not self.is_on():
            return self._zero
        assert y_pred.shape[1] == 1, "SSIM loss kernel: 1-channel images"
        t = y_target.float().contiguous()
        ws = self.window_size
        return _FusedLoss.apply(y_pred, lambda p, g, v: ops.ssim_loss(p, t, ws, self.lambda_, g, v))


class L2Sum(ElementaryLoss):
    """lambda * sum((y_pred - y_target)^2); dlib/loss/main.py:102-122."""

    def forward(self, epoch, y_pred=None, y_target=None, trg_per_pixel_weight=None, model=None):
        super().forward(epoch=epoch)
        if not self.is_on():
            return self._zero
        t = y_target.float().contiguous()
        return _FusedLoss.apply(y_pred, lambda p, g, v: ops.loss_pointwise(p, t, 3, self.lambda_, grad=g, loss_out=v))


class Charbonnier(ElementaryLoss):
    """lambda * mean(sqrt((y_target - y_pred)^2 + eps)), eps 1e-9; dlib/loss/main.py:125-151."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.eps = 1e-9

    def set_eps(self, eps):
        assert eps > 0
        self.eps = eps

    def forward(self, epoch, y_pred=None, y_target=None, trg_per_pixel_weight=None, model=None):
        super().forward(epoch=epoch)
        if not self.is_on():
            return self._zero
        t = y_target.float().contiguous()
        return _FusedLoss.apply(
            y_pred, lambda p, g, v: ops.loss_pointwise(p, t, 2, self.lambda_, self.eps, grad=g, loss_out=v))


class BoundedPrediction(ElementaryLoss):
    """y - eps <= y_hat <= y + eps as two extended-log-barrier terms: lambda * (ELB(y_hat - y - eps) +
    ELB(y - eps - y_hat)) / 2, on [0, color_max] when restore_range; dlib/loss/main.py:189-237."""

    def __init__(self, restore_range=False, color_min=0, color_max=255, **kwargs):
        super().__init__(**kwargs)
        from dlib.losses.elb import ELB
        assert isinstance(self.elb, ELB)            # main.py:193 (the dlib.losses copy)
        assert isinstance(color_max, int) and color_min < color_max
        self.restore_range, self.color_min, self.color_max = restore_range, color_min, color_max
        self.eps = 0.0
        self.eps_already_set = False

    def set_eps(self, eps):
        assert eps >= 0, eps
        assert isinstance(eps, float), type(eps)
        self.eps = eps
        self.eps_already_set = True

    def forward(self, epoch, y_pred=None, y_target=None, trg_per_pixel_weight=None, model=None):
        super().forward(epoch=epoch)
        if not self.is_on():
            return self._zero
        assert y_target.shape == y_pred.shape, f'{y_target.shape}, {y_pred.shape}'
        t = y_target.float().contiguous()
        tb = float(self.elb.get_t())
        sc = float(self.color_max) if self.restore_range else 1.0
        return _FusedLoss.apply(y_pred, lambda p, g, v: ops.loss_bounded(
            p, t, self.lambda_, self.eps, tb, sc, grad=g, loss_out=v))


class LocalMoments(ElementaryLoss):
    """lambda * mean(KL(N(target patch) || N(pred patch)) * [target patch variance == 0]) over 3x3 patches
    (reflect padding, unbiased variance, both variances + 1); dlib/loss/main.py:240-325.  As in the reference,
    ``set_ksz`` records the list but the patch operators stay the ones built for ksz = [3] (:244-263)."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.ksz = [3]
        self.ksz_already_set = False
        self.eps = 1.

    def set_ksz(self, ksz):
        assert len(ksz) > 0, len(ksz)
        for k in ksz:
            assert isinstance(k, int) and k > 1, k
        ksz.sort(reverse=False)
        self.ksz = ksz
        self.ksz_already_set = True

    def forward(self, epoch, y_pred=None, y_target=None, trg_per_pixel_weight=None, model=None):
        super().forward(epoch=epoch)
        if not self.is_on():
            return self._zero
        assert y_target.shape == y_pred.shape, f'{y_target.shape}, {y_pred.shape}'
        assert y_pred.ndim == 4 and y_pred.shape[1] == 1, "supports only one channel (local_terms.py:36)"
        t = y_target.float().contiguous()
        return _FusedLoss.apply(y_pred, lambda p, g, v: ops.loss_local_moments(p, t, self.lambda_, grad=g, loss_out=v))


_HIST_NORM = {NORM1: 1, NORM2: 2, KL: 3, BH: 4}      # metric codes of srhip_loss_hist / srhip_loss_kde


class HistogramMatch(ElementaryLoss):
    """p = (soft histogram + 1) normalised, 256 bins over [0, 1], sigmoid sharpness sigma (1e5); dlib/loss/main.py:690-782.
    NORM1 / NORM2: lambda * mean_{b,bin} nrm(p_pred - p_target); KL: lambda * KLDivLoss(batchmean)(log p_pred, p_target);
    BHATTACHARYYA: lambda * elb(-sum_bin sqrt(p_pred p_target))."""

    def __init__(self, color_min=0, color_max=255, **kwargs):
        super().__init__(**kwargs)
        self.color_min, self.color_max = color_min, color_max
        self.norm_str = NORM2
        self.sigma = 1e5
        self.already_set = False
        self.nbins = len(list(range(color_min, color_max))) + 1

    def set_it(self, norm_str, sigma):
        assert isinstance(sigma, float) and sigma > 0., sigma
        assert isinstance(norm_str, str) and norm_str in (NORM2, NORM1, KL, BH), norm_str
        if norm_str == BH:
            from dlib.losses.elb import ELB
            assert isinstance(self.elb, ELB)            # main.py:732 (the dlib.losses copy)
        self.sigma, self.norm_str, self.already_set = sigma, norm_str, True
        self.nbins = len(list(range(self.color_min, self.color_max))) + 1

    def forward(self, epoch, y_pred=None, y_target=None, trg_per_pixel_weight=None, model=None):
        super().forward(epoch=epoch)
        if not self.is_on():
            return self._zero
        assert y_target.shape == y_pred.shape, f'{y_target.shape}, {y_pred.shape}'
        t = y_target.float().contiguous()
        norm = _HIST_NORM[self.norm_str]
        tb = float(self.elb.get_t()) if norm == 4 else 1.0
        return _FusedLoss.apply(y_pred, lambda p, g, v: ops.loss_hist(
            p, t, self.lambda_, norm, self.sigma, self.nbins, grad=g, loss_out=v, elb_t=tb))


class KDEMatch(ElementaryLoss):
    """lambda * mean_{b,bin} nrm((kde_pred + 1e-4) - (kde_target + 1e-4)) / bins over a Gaussian KDE (bandwidth
    1/255^2, 256 points on [0, 1]) of each 1-channel image; dlib/loss/main.py:785-898.  NORM1 / NORM2 metrics."""

    def __init__(self, color_min=0, color_max=1, **kwargs):
        super().__init__(**kwargs)
        assert color_min == 0, color_min              # main.py:793-794
        assert color_max == 1, color_max
        self.color_min, self.color_max = color_min, color_max
        self.kde_bw = 1. / (255. ** 2)
        self.ndim = 1
        self.norm_str = NORM2
        self.already_set = False
        self.nbins = 256                              # len(np.arange(0, 1, 1/255)) + 1

    def set_it(self, norm_str, kde_bw, ndim, nbins):
        assert isinstance(kde_bw, float) and kde_bw > 0., kde_bw
        assert isinstance(norm_str, str) and norm_str in (NORM2, NORM1, BH), norm_str
        assert isinstance(ndim, int) and ndim == 1, ndim
        assert isinstance(nbins, int) and nbins > 0, nbins
        if norm_str == BH:
            from dlib.losses.elb import ELB
            assert isinstance(self.elb, ELB)            # main.py:839 (the dlib.losses copy)
        self.kde_bw, self.norm_str, self.ndim, self.already_set = kde_bw, norm_str, ndim, True
        self.nbins = 256                              # the reference recomputes it from the colour range (:825)

    def forward(self, epoch, y_pred=None, y_target=None, trg_per_pixel_weight=None, model=None):
        super().forward(epoch=epoch)
        if not self.is_on():
            return self._zero
        assert y_target.shape == y_pred.shape, f'{y_target.shape}, {y_pred.shape}'
        assert y_pred.ndim == 4 and y_pred.shape[1] == self.ndim
        t = y_target.float().contiguous()
        norm = _HIST_NORM[self.norm_str]
        tb = float(self.elb.get_t()) if norm == 4 else 1.0
        return _FusedLoss.apply(y_pred, lambda p, g, v: ops.loss_kde(
            p, t, self.lambda_, norm, self.kde_bw, self.nbins, grad=g, loss_out=v, elb_t=tb))


class _SparsityFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lam, *params):
        val = torch.zeros(1, device=params[0].device, dtype=torch.float32)
        for p in params:
            if not p.is_cuda:
                raise RuntimeError("dlib.loss (libsrhip) runs on the GPU only; there is no CPU fallback")
            ops.l1_sparsity(p.detach().contiguous().view(-1), lam, None, val, loss_accum=True)
        ctx.save_for_backward(*params)
        ctx.lam = lam
        return val.reshape(())

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(torch.sign(p) * (ctx.lam * g) for p in ctx.saved_tensors)


class WeightsSparsityLoss(ElementaryLoss):
    """lambda * sum over model.parameters() of ||w||_1; dlib/loss/main.py:938-959.  (In the fused
    training step the term is one kernel over the flat parameter buffer: srhip.train.TrainStep.)"""

    def forward(self, epoch, y_pred=None, y_target=None, trg_per_pixel_weight=None, model=None):
        super().forward(epoch=epoch)
        assert model is not None
        if not self.is_on():
            return self._zero
        return _SparsityFn.apply(self.lambda_, *list(model.parameters()))


class _LocalVariationTerm(ElementaryLoss):
    """lambda * mean(nrm(op(y_pred) - op(y_target))) (or, Norm* variants, of the 2-norms over the
    operator's channels) for a replicate-padded stencil operator; dlib/loss/main.py:328-674,
    dlib/loss/local_variations.py:18-141.  One fused kernel: value + gradient."""
    kind, channel_norm, has_ksz = None, False, False

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.ksz = 3
        self.norm_str = NORM2
        self.already_set = False

    def _set(self, norm_str, ksz=3):
        assert isinstance(norm_str, str) and norm_str in (NORM1, NORM2), norm_str
        assert isinstance(ksz, int) and ksz % 2 == 1 and ksz > 2, ksz
        self.norm_str, self.ksz, self.already_set = norm_str, ksz, True

    def forward(self, epoch, y_pred=None, y_target=None, trg_per_pixel_weight=None, model=None):
        super().forward(epoch=epoch)
        if not self.is_on():
            return self._zero
        assert y_target.shape == y_pred.shape, f'{y_target.shape}, {y_pred.shape}'
        assert y_pred.ndim == 4 and y_pred.shape[1] == 1, "supports only grey (local_variations.py:45)"
        t = y_target.float().contiguous()
        norm = 1 if self.norm_str == NORM1 else 2
        return _FusedLoss.apply(y_pred, lambda p, g, v: ops.loss_stencil(
            p, t, self.kind, self.lambda_, norm, self.ksz, self.channel_norm, grad=g, loss_out=v))


class ImageGradientLoss(_LocalVariationTerm):
    kind = "grad"

    def set_it(self, norm_str):
        self._set(norm_str)


class LaplacianFilterLoss(_LocalVariationTerm):
    kind = "laplace"

    def set_it(self, norm_str):
        self._set(norm_str)


class LocalVariationLoss(_LocalVariationTerm):
    kind, has_ksz = "lv", True

    def set_it(self, ksz, norm_str):
        self._set(norm_str, ksz)


class NormImageGradientLoss(ImageGradientLoss):
    channel_norm = True


class NormLaplacianFilterLoss(LaplacianFilterLoss):
    channel_norm = True


class NormLocalVariationLoss(LocalVariationLoss):
    channel_norm = True


class MasterLoss(nn.Module):
    """Sum of elementary losses; l_holder = [total, term1, ...] (master.py:46-56)."""

    def __init__(self, cuda_id=0, name=None):
        super().__init__()
        self._name = name
        self.losses, self.l_holder = [], []
        self.n_holder = [self.__name__]
        self._device = torch.device(cuda_id) if not isinstance(cuda_id, torch.device) else cuda_id

    def add(self, loss_):
        self.losses.append(loss_)
        self.n_holder.append(loss_.__name__)

    def update_t(self):
        for loss in self.losses:
            loss.update_t()

    @property
    def __name__(self):
        return _snake(self.__class__.__name__) if self._name is None else self._name

    def terms(self):
        """('l1', lam) | ('l2', lam) | ('ssim', lam, window) | ('charbonnier', lam, eps) |
        ('l2sum', lam) | ('grad'|'laplace'|'lv'|'norm_*', lam, norm, ksz) for the fused
        training step (srhip.train.TrainStep)."""
        out = []
        for l in self.losses:
            if isinstance(l, L1):
                out.append(("l1", l.lambda_))
            elif isinstance(l, L2):
                out.append(("l2", l.lambda_))
            elif isinstance(l, NegativeSsim):
                out.append(("ssim", l.lambda_, l.window_size))
            elif isinstance(l, Charbonnier):
                out.append(("charbonnier", l.lambda_, l.eps))
            elif isinstance(l, L2Sum):
                out.append(("l2sum", l.lambda_))
            elif isinstance(l, BoundedPrediction):
                # the ELB module itself rides along: its t can change between steps
                out.append(("boundpred", l.lambda_, l.eps, l.elb, l.restore_range, l.color_max))
            elif isinstance(l, WeightsSparsityLoss):
                out.append(("w_sparsity", l.lambda_))
            elif isinstance(l, LocalMoments):
                out.append(("local_moments", l.lambda_))
            elif isinstance(l, KDEMatch):
                out.append(("kde", l.lambda_, _HIST_NORM[l.norm_str], l.kde_bw, l.nbins, l.elb))
            elif isinstance(l, HistogramMatch):
                out.append(("hist", l.lambda_, _HIST_NORM[l.norm_str], l.sigma, l.nbins, l.elb))
            elif isinstance(l, _LocalVariationTerm):
                out.append((("norm_" if l.channel_norm else "") + l.kind, l.lambda_,
                            1 if l.norm_str == NORM1 else 2, l.ksz))
            else:
                raise NotImplementedError(type(l).__name__)
        return out

    def forward(self, **kwargs):
        assert self.losses != []
        self.l_holder = [loss(**kwargs).reshape(()) for loss in self.losses]
        total = sum(self.l_holder)
        self.l_holder = [total] + self.l_holder
        return total

    def to_device(self):
        return self

    def __str__(self):
        return "{}(): {}".format(self.__class__.__name__, ", ".join(self.n_holder))
