"""Training-crop sampling of the reference's dataset (dlib/datasets/dataset_dpsr.py:293-507):
where in a high-resolution tile the next training patch is cut.

``PatchSampler`` keeps the reference's constructor and call protocol -- ``sampler(img_u8_hxw,
return_roi) -> (row0, col0, roi_uint8 | None)`` -- for the sampling styles that need nothing but
numpy: 'uniform' (Python's ``random.randint``, :319-328) and 'roi' with a fixed threshold (:330-369:
one ``np.random.multinomial`` draw from probabilities proportional to ``exp(5 * roi) + 1`` over the
(H-P) x (W-P) candidate origins), 'edt' / 'edt*roi' (:371-457: scipy's Euclidean distance transform of the ROI, as
the reference) and both threshold styles ('fix_threshold'; 'automatic_threshold' = Otsu, restated from skimage's
published algorithm in dlib/datasets/lowres.py since skimage is not in this image).  With the same ``random`` /
``np.random`` seeds it returns the reference's origins (golden g21, g32).

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
        self.sample_type, self.psize, self.nbr_colors = sample_type, psize, nbr_colors
        self.threshold_style, self.threshold = threshold_style, threshold

    def _threshold(self, img=None):
        if self.threshold_style == TH_FIX:
            return self.threshold
        from dlib.datasets.lowres import otsu_threshold          # threshold_otsu(image=img, nbins=self.nbr_colors) (:479-480)
        return otsu_threshold(img, self.nbr_colors)

    def _uniform(self, img: np.ndarray):
        h, w = img.shape
        return random.randint(0, max(0, h - self.psize)), random.randint(0, max(0, w - self.psize))

    def _draw(self, p: np.ndarray, img: np.ndarray):
        hit = np.random.multinomial(1, p.flatten(), size=1).reshape(p.shape).nonzero()
        r0, c0 = int(hit[0][0]), int(hit[1][0])
        assert 0 <= r0 <= img.shape[0] - self.psize and 0 <= c0 <= img.shape[1] - self.psize
        return r0, c0

    def _roi(self, img: np.ndarray, th):
        return self._draw(roi_probabilities(img, th, self.psize), img)

    def origin_probabilities(self, img: np.ndarray, th) -> np.ndarray:
        """probability of every candidate origin for the 'roi' / 'edt' / 'edt*roi' styles (:343-347,:389-392,:431-439)"""
        if self.sample_type == SAMPLE_ROI:
            return roi_probabilities(img, th, self.psize)
        from scipy.ndimage import distance_transform_edt as edt_fn
        h, w = img.shape
        lo, hi = int(self.psize / 2), math.ceil(self.psize / 2)
        roi = (img >= th).astype(np.float64)
        edt = edt_fn(input=roi, return_distances=True, return_indices=False)[lo:h - hi, lo:w - hi]
        if self.sample_type == SAMPLE_EDT:
            return ((edt.flatten() + 1.) / (edt + 1.).sum()).reshape(edt.shape)
        croi = roi[lo:h - hi, lo:w - hi]
        t1 = np.exp(croi * 5.)
        p_roi = (t1.flatten() + 1.) / (t1 + 1.).sum()
        t2 = np.exp(edt)
        p_edt = (t2.flatten() + 1.) / (t2 + 1.).sum()
        prob = p_roi * p_edt
        return (prob / prob.sum()).reshape(edt.shape)

    def __call__(self, img: np.ndarray, return_roi: bool):
        assert img.ndim == 2, img.ndim
        roi_u8 = None
        th = None
        if return_roi or self.sample_type != SAMPLE_UNIF:
            th = self._threshold(img)
            if return_roi:
                roi_u8 = (img >= th).astype(np.uint8)
        if self.sample_type == SAMPLE_UNIF:
            r0, c0 = self._uniform(img)
        else:
            r0, c0 = self._draw(self.origin_probabilities(img, th), img)
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


def EvalPairsLike(args, pairs_h, pairs_l):
    """the LR resolution logic (true tile or synthesis) is EvalPairs' (utils_dataloaders.py)"""
    from dlib.utils.utils_dataloaders import EvalPairs
    return EvalPairs(args, pairs_h, pairs_l)


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

    and returns the batch dict the trainer feeds ModelPlain: l_im, h_im, l_to_h_img (+ _aug), h_id, l_id and, with
    --ppiw, h_per_pixel_weight.  Also from the reference's item (round 3):
      * pairs without a true LR tile (or --use_interpolated_low): the LR tile is synthesised once at construction as the
        reference does per item (:776-804; dlib/datasets/lowres.py, seeded by the item index);
      * the ROI image of the samplers is the LR tile brought to the HR size by srhip_resize_cubic (cv2.resize, :813-821);
        'edt' / 'edt*roi' and the Otsu threshold draw on the host through PatchSampler (scipy), 'roi' + fixed threshold
        on the device;
      * the LR-only augmentations (--da_blur / --da_dot_bin_noise / --da_add_gaus_noise, :899-905) run on the host on
        the B low-resolution patches (a few hundred pixels each) with the reference's numpy stream, then l_to_h_img is
        resized from the augmented patch on the device and clipped (:905-906);
      * --ppiw: per-colour weights from the split's HR histogram (:592-645), per-pixel lookup on the device (:1037-1056)."""

    def __init__(self, args, pairs_h: dict, pairs_l: dict, device, rank: int = 0, world: int = 1):
        import torch
        import torch.nn.functional as F
        from dlib.utils.utils_dataloaders import imread_gray_uint8, ShardedSampler
        from dlib.datasets import lowres
        if getattr(args, 'augment', False):
            raise NotImplementedError("ResidentTrainSet: --augment (the reference drops it too: dataset_dpsr.py:860-862 'passed')")
        self.args, self.device, self.sf = args, torch.device(device), int(args.scale)
        self.psize, self.batch = int(args.h_size), int(args.batch_size)
        self.ids_h = list(pairs_h.keys())
        self.ids_l = [pairs_h[k]['low_path_key'] for k in self.ids_h]
        self.hr, self.lr, self.roi_src = [], [], []
        style = getattr(args, 'sample_tr_patch', SAMPLE_UNIF)
        assert style in SAMPLE_PATCHES, style
        self.style = style
        th_style = getattr(args, 'sample_tr_patch_th_style', TH_FIX)
        self.device_roi = style == SAMPLE_ROI and th_style == TH_FIX
        self.host_sampler = None
        if style != SAMPLE_UNIF and not self.device_roi:
            self.host_sampler = PatchSampler(style, self.psize, int(getattr(args, 'color_max', 255)) + 1, th_style,
                                             float(getattr(args, 'sample_tr_patch_th', 0) or 0))
        ev = EvalPairsLike(args, pairs_h, pairs_l)
        hr_host = []
        for idx, (hk, lk) in enumerate(zip(self.ids_h, self.ids_l)):
            h_path = pairs_h[hk]['abs_path']
            h_full = imread_gray_uint8(h_path)
            l_np, _ = ev.low_res_u8(idx, h_full, h_path)                  # true tile, or the reference's synthesis
            h = torch.from_numpy(h_full[:, :, 0].copy())
            l = torch.from_numpy(np.array(l_np[:, :, 0], copy=True))
            hh, ww = h.shape[0] - h.shape[0] % self.sf, h.shape[1] - h.shape[1] % self.sf      # modcrop
            h = h[:hh, :ww].contiguous()
            l = l[:hh // self.sf, :ww // self.sf].contiguous()
            assert l.shape == (hh // self.sf, ww // self.sf), (hk, tuple(h.shape), tuple(l.shape))
            assert hh >= self.psize and ww >= self.psize, f"{hk}: tile {hh}x{ww} < patch {self.psize}"
            self.hr.append(h.to(self.device))
            self.lr.append(l.to(self.device))
            hr_host.append(h.numpy())
            if style != SAMPLE_UNIF:      # the image the reference thresholds: cv2.resize(LR, HR size, INTER_CUBIC) (:813-821,852-857)
                self.roi_src.append(lowres.l_to_h(self.lr[-1][None], (hh, ww))[0].contiguous())
        self.roi_host = [t.cpu().numpy() for t in self.roi_src] if self.host_sampler is not None else None
        if self.device_roi:
            self.roi = DeviceRoiSampler(self.roi_src, self.psize, int(args.sample_tr_patch_th), seed=int(args.myseed or 0) + rank)
        self.da = any(getattr(args, f, False) for f in ('da_blur', 'da_dot_bin_noise', 'da_add_gaus_noise'))
        self.ppiw_table = None
        if getattr(args, 'ppiw', False):
            self.ppiw_table = torch.from_numpy(lowres.per_color_weights(
                hr_host, int(getattr(args, 'color_min', 0)), int(getattr(args, 'color_max', 255)),
                float(getattr(args, 'ppiw_min_per_col_w', 1e-3)))).float().to(self.device)
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
            if self.device_roi:
                org, _ = self.roi.sample(ids)
                org = org.cpu().tolist()                      # B x 2 ints: the only host round trip of the batch
            elif self.host_sampler is not None:               # 'edt' / 'edt*roi' / Otsu: the reference's sampler on the host
                org = [list(self.host_sampler(self.roi_host[i], False)[:2]) for i in ids]
            else:
                u = torch.rand(B, 2, device=self.device, generator=self.gen).cpu()
                org = [[int(u[b, 0] * (self.hr[i].shape[0] - P + 1)), int(u[b, 1] * (self.hr[i].shape[1] - P + 1))]
                       for b, i in enumerate(ids)]
            modes = torch.randint(0, 8, (B,), device=self.device, generator=self.gen).cpu().tolist()
            y0, x0 = [o[0] for o in org], [o[1] for o in org]
            batch = ops.train_batch(self.hr, self.lr, ids, y0, x0, modes, P, sf)
            if self.da:       # LR-only augmentations (:899-905): B patches of (P / s)^2 pixels through the reference's numpy code
                from dlib.datasets import lowres
                host = batch['l_im'].permute(0, 2, 3, 1).cpu().numpy()
                host = np.stack([lowres.apply_lr_augmentations(np.ascontiguousarray(host[b]), self.args) for b in range(B)])
                batch['l_im'] = torch.from_numpy(host).permute(0, 3, 1, 2).contiguous().to(self.device)
            # the LR patch at the HR size: cv2.resize(img_l, (P, P), INTER_CUBIC) clipped to [0, 1] (:905-907)
            up = ops.clip01_(ops.resize_cubic(batch['l_im'][:, 0].contiguous(), (P, P)))
            batch['l_to_h_img'] = up[:, None]
            batch['l_to_h_img_aug'] = batch['l_to_h_img']
            if self.ppiw_table is not None:
                from dlib.datasets import lowres
                batch['h_per_pixel_weight'] = lowres.per_pixel_weight(batch['h_im'], self.ppiw_table)
            batch['h_id'] = [self.ids_h[i] for i in ids]
            batch['l_id'] = [self.ids_l[i] for i in ids]
            batch['origin'], batch['mode'] = org, modes
            yield batch
