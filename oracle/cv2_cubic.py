"""Test infrastructure (imported by tests/ only): cv2.resize(..., interpolation=cv2.INTER_CUBIC) on 1-channel images,
restated in numpy from OpenCV's published algorithm (imgproc/resize.cpp) -- cv2 is absent from this image and OpenCV's
source is not under /root/reference (the reference calls it at dlib/datasets/dataset_dpsr.py:659-683).

PARITY UNPINNED against cv2 itself: no golden vector of cv2 output exists here.  What is restated: pixel-centre
mapping fx = (dx + 0.5) * scale - 0.5 (scale from the double inv_scale = dsize / ssize), interpolateCubic with A = -0.75 in
float32, replicate borders; uint8: coefficients rounded to 1/2048 as shorts, horizontal pass in int32, vertical pass
(sum + 2^21) >> 22 saturated to uint8 (the scalar reference path; OpenCV's vectorised vertical pass rounds a float sum
and may differ by one grey level on rare near-ties); float32: plain float32 sums in tap order."""
import numpy as np


def _coeffs(x):
    x = x.astype(np.float32)
    A = np.float32(-0.75)
    one = np.float32(1)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], -1).astype(np.float32)


def _axis(n_src, n_dst):
    scale = 1.0 / (float(n_dst) / float(n_src))
    f = ((np.arange(n_dst) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    idx = np.clip(s[:, None] - 1 + np.arange(4)[None, :], 0, n_src - 1)
    return idx, _coeffs(f)


def resize_cubic(img, size):
    """img: [H, W] uint8 or float32; size = (width, height) as cv2's dsize.  Returns [height, width] of the same dtype."""
    Wo, Ho = int(size[0]), int(size[1])
    H, W = img.shape
    xi, cx = _axis(W, Wo)
    yi, cy = _axis(H, Ho)
    if img.dtype == np.uint8:
        ax = np.rint(cx * np.float32(2048)).astype(np.int64)
        ay = np.rint(cy * np.float32(2048)).astype(np.int64)
        rows = (img.astype(np.int64)[:, xi] * ax[None]).sum(-1)              # [H, Wo]
        out = (rows[yi] * ay[:, :, None]).sum(1)                             # [Ho, Wo]
        return np.clip((out + (1 << 21)) >> 22, 0, 255).astype(np.uint8)
    assert img.dtype == np.float32, img.dtype
    g = img[:, xi]                                                           # [H, Wo, 4]
    rows = ((g[..., 0] * cx[None, :, 0] + g[..., 1] * cx[None, :, 1]) + g[..., 2] * cx[None, :, 2]) + g[..., 3] * cx[None, :, 3]
    r = rows[yi]                                                             # [Ho, 4, Wo]
    return (((r[:, 0] * cy[:, 0:1] + r[:, 1] * cy[:, 1:2]) + r[:, 2] * cy[:, 2:3]) + r[:, 3] * cy[:, 3:4]).astype(np.float32)
