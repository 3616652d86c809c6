"""The low-resolution side of the reference's dataset items (dlib/datasets/dataset_dpsr.py): what surrounds the
patch-level hot path on the data side.

Host functions -- numpy / torch on the CPU, run once per tile at dataset construction or per sample on patches of a few
hundred pixels, with the reference's own random-number streams so that a seeded run draws what the reference draws:
  interpolate_torch       :684-710   LR = uint8(clamp(F.interpolate(HR, 1/s, bicubic)))  (the synthesised LR of sets
                                      without true LR tiles)
  simulate_low_res        :713-744   + seeded Gaussian noise inside the cells' region (CACO-2)
  per_color_weights       :592-645   --ppiw: one loss weight per grey level from the HR histogram of the split
  da_blur / da_dot_bin_noise / da_add_gaus_noise   :1071-1180   the LR-only augmentations on a random block
  otsu_threshold                     skimage.filters.threshold_otsu on a uint8 image (the 'automatic_threshold' ROI style;
                                      skimage is not in this image: restated, parity unpinned)
Device functions (libsrhip):
  l_to_h                  :659-683   cv2.resize(INTER_CUBIC) of the LR image to the HR size: srhip_resize_cubic
  per_pixel_weight        :1037-1056 weight image of an HR patch: a lookup of the per-colour table
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def is_caco2(path: str) -> bool:
    """utils_image.py:202-208"""
    return ('caco2' in path) and any(c in path for c in ('CELL0', 'CELL1', 'CELL2'))


def interpolate_torch(x: np.ndarray, scale: float, mode: str = 'bicubic', min_v: float = 0.0, max_v: float = 255.) -> np.ndarray:
    """dataset_dpsr.py:684-710: x uint8 HWC -> uint8 (int(H*scale), int(W*scale), C); the float result is clamped and
    TRUNCATED by astype (not rounded)."""
    assert isinstance(x, np.ndarray) and x.dtype == np.uint8 and x.ndim == 3 and x.shape[-1] in (1, 3), (x.dtype, x.shape)
    v = torch.from_numpy(x).float().permute(2, 0, 1)
    c, h, w = v.shape
    out = F.interpolate(v.unsqueeze(0), size=(int(h * scale), int(w * scale)), mode=mode).squeeze(0)
    out = torch.clamp(out, min=min_v, max=max_v)
    return out.permute(1, 2, 0).numpy().astype(x.dtype)


def simulate_low_res(x: np.ndarray, seed: int, th: float, sigma: float, min_v: float = 0.0, max_v: float = 255.) -> np.ndarray:
    """dataset_dpsr.py:713-744: Gaussian noise N(v, sigma) on the pixels >= th (the cells), seeded per image index with the
    reference's set_seed (utils_reproducibility.py:89-115)."""
    assert sigma >= 0.0 and isinstance(sigma, float) and th >= 0 and isinstance(th, float)
    assert isinstance(x, np.ndarray) and x.ndim == 3 and x.shape[-1] in (1, 3)
    v = torch.from_numpy(x).float()
    roi = (v >= th).float()
    # The reference runs this inside a DataLoader worker and re-seeds the worker's GLOBAL generators (torch, numpy, random)
    # with the item index; here the call happens in the main process (resident sets are built up front), where that would
    # leave every rank's numpy / random streams at the last tile's index and perturb whatever is drawn next (patch
    # samplers, LR augmentations, DropPath).  The only draw below is torch.normal: a private generator seeded with the
    # index yields the very stream the re-seeded default generator would (golden g31), and touches nothing global.
    gen = torch.Generator()
    gen.manual_seed(seed)
    new_low = torch.normal(mean=v, std=sigma, generator=gen)
    new_low = torch.clamp(new_low, 0.0, 255.)
    new_low = new_low * roi + (1 - roi) * v
    new_low = torch.clamp(new_low, min=min_v, max=max_v)
    return new_low.numpy().astype(x.dtype)


def per_color_weights(hr_images, color_min: int, color_max: int, min_w: float) -> np.ndarray:
    """dataset_dpsr.py:592-645: hr_images = iterable of uint8 [H, W] tiles of the training split -> (nbr_colors,) weights in
    [min_w, 1], rarest grey level heaviest."""
    nbr = len(list(range(color_min, color_max))) + 1
    full = 1.
    for img in hr_images:
        full = np.histogram(img, bins=nbr, range=(color_min, color_max), density=False)[0] + full
    full = nbr * full / float(full.sum())
    w = 1. / full
    w = w / w.sum()
    w += 1e-8
    lo, hi = w.min(), w.max()
    return ((1. - min_w) * (w - lo) / (hi - lo) + min_w).flatten()


def per_pixel_weight(h_patch: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """dataset_dpsr.py:1037-1056 on the device: h_patch float [B, 1, P, P] in [0, 1] (= uint8 / 255) -> the weight of every
    pixel's grey level, float32 of the same shape (a table lookup: index plumbing, no arithmetic)."""
    idx = (h_patch * 255.).to(torch.uint8).long()        # (img_h * 255.).type(torch.uint8): truncation, as the reference
    return table.to(h_patch.device, torch.float32)[idx]


def random_block(h: int, w: int, area: float):
    """get_random_coordinates_block (:1060-1070): np.random draws in the reference's order."""
    assert 0. <= area <= 1., area
    ratio = np.random.randn() * 0.01 + area
    bh, bw = np.int64(h * ratio), np.int64(w * ratio)
    ch = np.random.randint(0, h - bh + 1)
    cw = np.random.randint(0, w - bw + 1)
    return ch, cw, bh, bw


def da_blur(img: np.ndarray, prob: float, area: float, sigma: float) -> np.ndarray:
    """np_blur (:1071-1108): a Gaussian blur (scipy.ndimage.gaussian_filter over all three axes of the HWC array) inside
    or -- with probability 0.98 -- outside a random block."""
    from scipy.ndimage import gaussian_filter
    assert img.ndim == 3
    if area == 0 or np.random.rand(1) >= prob:
        return img
    h, w, c = img.shape
    ch, cw, bh, bw = random_block(h, w, area)
    blurred = gaussian_filter(input=img, sigma=sigma)
    if np.random.rand(1) >= 0.98:
        img[ch:ch + bh, cw:cw + bw, :] = blurred[ch:ch + bh, cw:cw + bw, :]
    else:
        im = np.copy(blurred)
        im[ch:ch + bh, cw:cw + bw, :] = img[ch:ch + bh, cw:cw + bw, :]
        img = im
    return img


def da_dot_bin_noise(img: np.ndarray, prob: float, area: float, p: float) -> np.ndarray:
    """np_prod_binary_noise (:1111-1144): a Bernoulli(1 - p) mask on a random block."""
    assert img.ndim == 3
    if area == 0 or np.random.rand(1) >= prob:
        return img
    h, w, c = img.shape
    ch, cw, bh, bw = random_block(h, w, area)
    mask = np.random.binomial(n=1, p=1. - p, size=(bh, bw, 1)).astype(np.float32)
    img[ch:ch + bh, cw:cw + bw, :] = img[ch:ch + bh, cw:cw + bw, :] * mask
    return img


def da_add_gaus_noise(img: np.ndarray, prob: float, area: float, std: float) -> np.ndarray:
    """np_add_gaussian_noise (:1147-1180): N(0, std) on a random block."""
    assert img.ndim == 3
    if area == 0 or np.random.rand(1) >= prob:
        return img
    h, w, c = img.shape
    ch, cw, bh, bw = random_block(h, w, area)
    noise = np.random.normal(loc=0.0, scale=std, size=(bh, bw, c))
    img[ch:ch + bh, cw:cw + bw, :] = img[ch:ch + bh, cw:cw + bw, :] + noise
    return img


def apply_lr_augmentations(img_l: np.ndarray, args) -> np.ndarray:
    """The LR-only block of DatasetDPSR.__getitem__ (:899-905) in its order: blur, dot-binary noise, additive Gaussian."""
    if getattr(args, 'da_blur', False):
        img_l = da_blur(img_l, args.da_blur_prob, args.da_blur_area, args.da_blur_sigma)
    if getattr(args, 'da_dot_bin_noise', False):
        img_l = da_dot_bin_noise(img_l, args.da_dot_bin_noise_prob, args.da_dot_bin_noise_area, args.da_dot_bin_noise_p)
    if getattr(args, 'da_add_gaus_noise', False):
        img_l = da_add_gaus_noise(img_l, args.da_add_gaus_noise_prob, args.da_add_gaus_noise_area, args.da_add_gaus_noise_std)
    return img_l


def otsu_threshold(img: np.ndarray, nbins: int = 256):
    """skimage.filters.threshold_otsu(image, nbins) for an integer image (the sampler's 'automatic_threshold' style,
    dataset_dpsr.py:479-480).  skimage is not in this image: restated from its published algorithm -- for integer images
    the histogram has one bin per grey level between the image's min and max (nbins is ignored), and the threshold is the
    bin centre that maximises the between-class variance.  PARITY UNPINNED against skimage."""
    assert np.issubdtype(img.dtype, np.integer), img.dtype
    flat = img.reshape(-1)
    lo, hi = int(flat.min()), int(flat.max())
    if lo == hi:
        return lo
    counts = np.bincount(flat.astype(np.int64) - lo, minlength=hi - lo + 1).astype(np.float64)
    centers = np.arange(lo, hi + 1).astype(np.float64)
    w1 = np.cumsum(counts)
    w2 = np.cumsum(counts[::-1])[::-1]
    m1 = np.cumsum(counts * centers) / w1
    m2 = (np.cumsum((counts * centers)[::-1]) / w2[::-1])[::-1]
    var12 = w1[:-1] * w2[1:] * (m1[:-1] - m2[1:]) ** 2
    return centers[int(np.argmax(var12))]


def l_to_h(img_l: torch.Tensor, size_hw) -> torch.Tensor:
    """_resize_low_to_scale (:659-683) on the device: img_l [B, h, w] uint8 or float32 CUDA -> [B, H, W] of the same dtype."""
    from srhip import ops
    return ops.resize_cubic(img_l.contiguous(), size_hw)
