"""Training-crop sampling of the reference's dataset (dlib/datasets/dataset_dpsr.py:293-507):
where in a high-resolution tile the next training patch is cut.

``PatchSampler`` keeps the reference's constructor and call protocol -- ``sampler(img_u8_hxw,
return_roi) -> (row0, col0, roi_uint8 | None)`` -- for the sampling styles that need nothing but
numpy: 'uniform' (Python's ``random.randint``, :319-328) and 'roi' with a fixed threshold (:330-369:
one ``np.random.multinomial`` draw from probabilities proportional to ``exp(5 * roi) + 1`` over the
(H-P) x (W-P) candidate origins).  With the same ``random`` / ``np.random`` seeds it returns the
reference's origins (golden g21).  The Otsu threshold ('automatic_threshold', skimage) and the
distance-transform styles ('edt', 'edt*roi') are not part of this build.

``DeviceRoiSampler`` is the MI355X form of the 'roi' style: the tiles stay resident in HBM as
uint8, one launch draws the origins of a whole batch from device-side uniforms by the inverse CDF of
the SAME probabilities (srhip_roi_sample), and they go straight into srhip_patch_gather -- no
host round trip per sample, so 8 GPUs x 8 patches per step are not host-bound."""
import math
import random

import numpy as np

SAMPLE_UNIF, SAMPLE_ROI, SAMPLE_EDT, SAMPLE_EDTXROI = 'uniform', 'roi', 'edt', 'edt*roi'
SAMPLE_PATCHES = [SAMPLE_UNIF, SAMPLE_ROI, SAMPLE_EDT, SAMPLE_EDTXROI]
TH_AUTO, TH_FIX = 'automatic_threshold', 'fix_threshold'
ROI_STYLE_TH = [TH_AUTO, TH_FIX]


def roi_probabilities(img: np.ndarray, threshold: float, psize: int) -> np.ndarray:
    """Probability of every candidate origin, (H-P) x (W-P), as the reference builds it (:343-347)."""
    h, w = img.shape
    lo, hi = int(psize / 2), math.ceil(psize / 2)
    roi = (img >= threshold).astype(np.float64)[lo:h - hi, lo:w - hi]
    wgt = np.exp(roi * 5.)
    return ((wgt.flatten() + 1.) / (wgt + 1.).sum()).reshape(roi.shape)


class PatchSampler(object):
    def __init__(self, sample_type: str, psize: int, nbr_colors: int, threshold_style: str, threshold: float):
        assert sample_type in SAMPLE_PATCHES, sample_type
        assert isinstance(psize, int) and psize > 0, psize
        assert isinstance(nbr_colors, int) and nbr_colors > 0, nbr_colors
        assert threshold_style in ROI_STYLE_TH, f"{threshold_style} not in {ROI_STYLE_TH}"
        if sample_type in (SAMPLE_EDT, SAMPLE_EDTXROI):
            raise NotImplementedError(f"sample_type {sample_type!r}: the distance-transform samplers are not built")
        self.sample_type, self.psize, self.nbr_colors = sample_type, psize, nbr_colors
        self.threshold_style, self.threshold = threshold_style, threshold

    def _threshold(self):
        if self.threshold_style != TH_FIX:
            raise NotImplementedError("'automatic_threshold' (skimage.filters.threshold_otsu) is not part of this "
                                      "build: configure sample_tr_patch_th_style='fix_threshold'")
        return self.threshold

    def _uniform(self, img: np.ndarray):
        h, w = img.shape
        return random.randint(0, max(0, h - self.psize)), random.randint(0, max(0, w - self.psize))

    def _roi(self, img: np.ndarray):
        p = roi_probabilities(img, self._threshold(), self.psize)
        hit = np.random.multinomial(1, p.flatten(), size=1).reshape(p.shape).nonzero()
        r0, c0 = int(hit[0][0]), int(hit[1][0])
        assert 0 <= r0 <= img.shape[0] - self.psize and 0 <= c0 <= img.shape[1] - self.psize
        return r0, c0

    def __call__(self, img: np.ndarray, return_roi: bool):
        assert img.ndim == 2, img.ndim
        roi_u8 = None
        if return_roi:
            roi_u8 = (img >= self._threshold()).astype(np.uint8)
        if self.sample_type == SAMPLE_UNIF:
            r0, c0 = self._uniform(img)
        else:
            r0, c0 = self._roi(img)
        return r0, c0, roi_u8


class DeviceRoiSampler:
    """'roi' sampling for a batch on the GPU.  ``tiles``: resident uint8 CUDA tensors [H, W] (the
    images the reference thresholds: the low-resolution tile interpolated to high resolution,
    dataset_dpsr.py:852-857); ``sample(ids)`` -> int32 [B, 2] origins (row, col) on the device."""

    def __init__(self, tiles, psize: int, threshold: int, seed: int = 0):
        import torch
        self.tiles, self.psize, self.threshold = tiles, int(psize), int(threshold)
        self.gen = torch.Generator(device=tiles[0].device)
        self.gen.manual_seed(int(seed))

    def sample(self, ids):
        import torch
        from srhip import ops
        u = torch.rand(len(ids), dtype=torch.float64, device=self.tiles[0].device, generator=self.gen)
        return ops.roi_sample(self.tiles, ids, self.psize, self.threshold, u), u
