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


class ResidentTrainSet:
    """Training batches assembled ON the GPU from tiles that stay resident in HBM as uint8 (a CACO-2
    split is a few hundred 8-bit tiles: tens of MB against 288 GB) -- the MI355X form of the TRAIN phase of
    DatasetDPSR.__getitem__ + DataLoader (dataset_dpsr.py:746-757,826-947, utils_dataloaders.py:138-158) for
    sets that ship true low-resolution tiles:

      per epoch   ShardedSampler(shuffle, seed + epoch, drop_last) -> this rank's minibatch index lists
      per batch   crop origins on the device ('uniform': torch.randint; 'roi': srhip_roi_sample on the
                  low-resolution tile interpolated to the high-resolution size and thresholded, as
                  :852-857), LR origin = HR origin // scale (:866-867), one augmentation mode 0..7 per
                  sample (:890), srhip_patch_gather for h_im and l_im (bit-exact crop / flip / rotate /
                  uint8 -> float32, g12)

    and returns the batch dict the trainer feeds ModelPlain: l_im, h_im, h_id, l_id.  Not produced: the
    cv2-bicubic 'l_to_h_img' tensors (SRCNN-style nets), per-pixel weights, the LR-only blur / noise
    augmentations (flags da_blur / da_dot_bin_noise / da_add_gaus_noise must be off).  The ROI image
    uses torch's bicubic kernel where the reference uses cv2's (both round to uint8): same regions up
    to boundary pixels of the thresholded mask."""

    def __init__(self, args, pairs_h: dict, pairs_l: dict, device, rank: int = 0, world: int = 1):
        import torch
        import torch.nn.functional as F
        from dlib.utils.utils_dataloaders import imread_gray_uint8, ShardedSampler
        for flag in ('da_blur', 'da_dot_bin_noise', 'da_add_gaus_noise', 'ppiw', 'augment'):
            if getattr(args, flag, False):
                raise NotImplementedError(f"ResidentTrainSet: --{flag} is not part of the device pipeline")
        self.args, self.device, self.sf = args, torch.device(device), int(args.scale)
        self.psize, self.batch = int(args.h_size), int(args.batch_size)
        self.ids_h = list(pairs_h.keys())
        self.ids_l = [pairs_h[k]['low_path_key'] for k in self.ids_h]
        self.hr, self.lr, self.roi_src = [], [], []
        style = getattr(args, 'sample_tr_patch', SAMPLE_UNIF)
        if style not in (SAMPLE_UNIF, SAMPLE_ROI):
            raise NotImplementedError(f"sample_tr_patch={style!r}")
        self.style = style
        for hk, lk in zip(self.ids_h, self.ids_l):
            h = torch.from_numpy(imread_gray_uint8(pairs_h[hk]['abs_path'])[:, :, 0].copy())
            l = torch.from_numpy(imread_gray_uint8(pairs_l[lk]['abs_path'])[:, :, 0].copy())
            hh, ww = h.shape[0] - h.shape[0] % self.sf, h.shape[1] - h.shape[1] % self.sf      # modcrop
            h = h[:hh, :ww].contiguous()
            assert l.shape == (hh // self.sf, ww // self.sf), (hk, tuple(h.shape), tuple(l.shape))
            assert hh >= self.psize and ww >= self.psize, f"{hk}: tile {hh}x{ww} < patch {self.psize}"
            self.hr.append(h.to(self.device))
            self.lr.append(l.to(self.device))
            if style == SAMPLE_ROI:
                up = F.interpolate(l[None, None].float(), size=(hh, ww), mode='bicubic', align_corners=False)
                self.roi_src.append(up.round().clamp(0, 255).to(torch.uint8)[0, 0].contiguous().to(self.device))
        if style == SAMPLE_ROI:
            if getattr(args, 'sample_tr_patch_th_style', TH_FIX) != TH_FIX:
                raise NotImplementedError("ROI sampling needs sample_tr_patch_th_style='fix_threshold' in this build")
            self.roi = DeviceRoiSampler(self.roi_src, self.psize, int(args.sample_tr_patch_th), seed=int(args.myseed or 0) + rank)
        self.sampler = ShardedSampler(len(self.ids_h), world, rank, shuffle=True, seed=int(args.myseed or 0), drop_last=True)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(args.myseed or 0) * 1000003 + rank)

    def __len__(self):
        return len(self.sampler) // self.batch

    def epoch(self, epoch: int):
        """Iterator over this rank's batch dicts of one epoch (set_epoch semantics of utils_trainer.py:325-326)."""
        import torch
        from srhip import ops
        self.sampler.set_epoch(epoch)
        P, sf = self.psize, self.sf
        for ids in self.sampler.batches(self.batch, drop_last=True):
            B = len(ids)
            if self.style == SAMPLE_ROI:
                org, _ = self.roi.sample(ids)
                org = org.cpu().tolist()                      # B x 2 ints: the only host round trip of the batch
            else:
                u = torch.rand(B, 2, device=self.device, generator=self.gen).cpu()
                org = [[int(u[b, 0] * (self.hr[i].shape[0] - P + 1)), int(u[b, 1] * (self.hr[i].shape[1] - P + 1))]
                       for b, i in enumerate(ids)]
            modes = torch.randint(0, 8, (B,), device=self.device, generator=self.gen).cpu().tolist()
            y0, x0 = [o[0] for o in org], [o[1] for o in org]
            batch = ops.train_batch(self.hr, self.lr, ids, y0, x0, modes, P, sf)
            batch['h_id'] = [self.ids_h[i] for i in ids]
            batch['l_id'] = [self.ids_l[i] for i in ids]
            batch['origin'], batch['mode'] = org, modes
            yield batch
