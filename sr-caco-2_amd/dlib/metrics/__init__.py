"""SR evaluation metrics on libsrhip.

The reference keeps them in ``dlib/utils/utils_image.py`` (843-1198, 369-372,
618-653) and calls them from ``utils_trainer._compute_metrics`` (:961-1032);
``north_star`` names the surface ``dlib.metrics``.  Same function names,
arguments ``(img1, img2, border=0, roi=None)`` and ``(B,)`` results.  The ROI is
given as the reference builds it, ``roi = (H_img >= th).float()``; because the
kernels take thresholds, a mask is mapped back to its threshold when it has
that form and rejected otherwise (``sweep`` is the fast fused entry point: all
five metrics x all thresholds in two passes over the images).
"""
import torch

from srhip import ops

__all__ = ['tensor2uint82float', 'mbatch_gpu_calculate_psnr', 'mbatch_gpu_calculate_mse',
           'mbatch_gpu_calculate_nrmse', 'mbatch_gpu_calculate_ssim', 'sweep']


def _need_cuda(t):
    if not t.is_cuda:
        raise RuntimeError("dlib.metrics (libsrhip) runs on the GPU only; there is no CPU fallback")


def tensor2uint82float(img):
    """(img.clamp(0,1)*255).round().clamp(0,255); utils_image.py:369-372."""
    _need_cuda(img)
    return ops.tensor2uint82float(img)


def _threshold_of(roi, img2):
    """roi == (img2 >= th) for an integer th in [0,256]?  Returns th or raises."""
    if roi is None:
        return None
    assert roi.ndim == 4 and roi.shape[1] == 1 and roi.shape[0] == img2.shape[0]
    on = roi > 0
    if not bool(on.any()):
        return 256                      # empty ROI: nothing reaches the threshold
    th = int(img2[on].min().item())
    if not torch.equal(on, img2 >= th):
        raise NotImplementedError("libsrhip metrics take ROIs of the form (H >= threshold) "
                                  "(utils_trainer.py:983-988)")
    return th


def sweep(E, H, border=0, thresholds=(), inputs_are_u8=False):
    """All metrics in two fused passes.  E, H: (B,1,h,w) in [0,1] (or already
    u8-valued floats).  Returns dict name -> (B, 1+len(thresholds)) tensors; column 0
    is 'no ROI', column k the ROI H_u8 >= thresholds[k-1]."""
    _need_cuda(E)
    assert E.shape == H.shape and E.ndim == 4 and E.shape[1] == 1, "1-channel images"
    E, H = E.float().contiguous(), H.float().contiguous()
    fam = ops.metrics_psnr_family(E, H, border, tuple(thresholds), inputs_are_u8)
    ssim = ops.metrics_ssim(E, H, border, tuple(thresholds), inputs_are_u8)
    return {"psnr": fam[:, :, 0], "psnr_y": fam[:, :, 1], "mse": fam[:, :, 2],
            "nrmse": fam[:, :, 3], "ssim": ssim}


def _one(name, img1, img2, border, roi):
    _need_cuda(img1)
    assert img1.ndim == 4 and img1.shape == img2.shape
    th = _threshold_of(roi, img2)
    out = sweep(img1, img2, border, () if th is None else (th,), inputs_are_u8=True)[name]
    return out[:, 0 if th is None else 1].contiguous()


def mbatch_gpu_calculate_psnr(img1, img2, border=0, roi=None):
    """utils_image.py:843-891; inputs in [0,255]; fp64 result."""
    return _one("psnr", img1, img2, border, roi)


def mbatch_gpu_calculate_mse(img1, img2, border=0, roi=None):
    """utils_image.py:894-934."""
    return _one("mse", img1, img2, border, roi)


def mbatch_gpu_calculate_nrmse(img, y, border=0, roi=None):
    """utils_image.py:937-1007."""
    return _one("nrmse", img, y, border, roi)


def mbatch_gpu_calculate_ssim(x, y, border=0, roi=None):
    """utils_image.py:1120-1198; fp32 result."""
    return _one("ssim", x, y, border, roi)
