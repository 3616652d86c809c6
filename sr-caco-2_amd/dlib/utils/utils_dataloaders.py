"""Rank-sharded index streams of the data-parallel input pipeline.

The reference shards training samples with ``DistributedSampler(train_set,
shuffle=True, seed=args.myseed, drop_last=True)`` and re-seeds it every epoch
with ``set_epoch`` (dlib/utils/utils_dataloaders.py:138-148,
dlib/utils/utils_trainer.py:325-326); evaluation with ``shuffle=False,
drop_last=False`` when ``eval_bsize > 1`` (utils_trainer.py:382-386).  This is
the same index law (torch.utils.data.distributed.DistributedSampler, torch
2.x) without the Dataset / DataLoader machinery: the device-side patch
assembly (srhip_patch_gather) consumes plain index lists.  tests/test_cpu_host.py
checks the lists against torch's own sampler for every (n, world, rank, epoch)
tried."""
import math

import torch


class ShardedSampler:
    def __init__(self, n_samples: int, num_replicas: int, rank: int, shuffle: bool = True, seed: int = 0,
                 drop_last: bool = False):
        if not 0 <= rank < num_replicas:
            raise ValueError(f"invalid rank {rank} for {num_replicas} replicas")
        self.n, self.num_replicas, self.rank = int(n_samples), int(num_replicas), int(rank)
        self.shuffle, self.seed, self.drop_last, self.epoch = shuffle, int(seed), drop_last, 0
        if drop_last and self.n % self.num_replicas != 0:
            # the tail that does not divide evenly is dropped
            self.num_samples = math.ceil((self.n - self.num_replicas) / self.num_replicas)
        else:
            self.num_samples = math.ceil(self.n / self.num_replicas)
        self.total_size = self.num_samples * self.num_replicas

    def set_epoch(self, epoch: int):
        """Every rank draws the SAME permutation from seed + epoch, then takes its stride."""
        self.epoch = int(epoch)

    def indices(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        if not self.drop_last:
            pad = self.total_size - len(idx)
            if pad <= len(idx):
                idx += idx[:pad]
            else:
                idx += (idx * math.ceil(pad / len(idx)))[:pad]
        else:
            idx = idx[:self.total_size]
        assert len(idx) == self.total_size
        return idx[self.rank:self.total_size:self.num_replicas]

    def __iter__(self):
        return iter(self.indices())

    def __len__(self):
        return self.num_samples

    def batches(self, batch_size: int, drop_last: bool = True):
        """Index lists of one epoch's minibatches on this rank (DataLoader(batch_size,
        drop_last=True, sampler=...), utils_dataloaders.py:141-148)."""
        idx = self.indices()
        stop = len(idx) - (len(idx) % batch_size if drop_last else 0)
        return [idx[i:i + batch_size] for i in range(0, stop, batch_size)]


def train_sampler(n_samples, world, rank, seed):
    """The reference's training sampler (utils_dataloaders.py:138-139)."""
    return ShardedSampler(n_samples, world, rank, shuffle=True, seed=seed, drop_last=True)


def eval_sampler(n_samples, world, rank):
    """The reference's evaluation sampler (DistributedSampler(shuffle=False), default drop_last=False)."""
    return ShardedSampler(n_samples, world, rank, shuffle=False, drop_last=False)


# ----------------------------------------------------------------------------------------------
# Evaluation sets from the reference's fold files (dlib/utils/utils_dataloaders.py:27-53,196-304,
# dlib/datasets/dataset_dpsr.py:746-757,826-838,930-947,981-1005).
# ----------------------------------------------------------------------------------------------
import os
from os.path import join

import numpy as np

from dlib.utils import constants


def get_pairs(path_file: str) -> dict:
    """l_h.txt / h_l.txt: one '<key_1>,<key_2>' per line, keys = paths relative to the dataset
    directory (unique ids), in file order."""
    assert os.path.isfile(path_file), path_file
    pairs = {}
    with open(path_file, 'r') as f:
        for line in f.readlines():
            a, b = line.strip('\n').split(',')
            assert a not in pairs, a
            pairs[a] = b
    return pairs


def dataset_dir(ds_name: str) -> str:
    """constants.DS_DIR of the reference (constants.py:472-520): every CACO2 set lives under
    'caco2', every BioSR set under 'biosr'."""
    low = ds_name.lower()
    if low.startswith('caco2'):
        return 'caco2'
    if low.startswith('biosr'):
        return 'biosr'
    raise ValueError(f'unknown dataset {ds_name!r}')


def imread_gray_uint8(path: str) -> np.ndarray:
    """cv2.imread(path, 0) of utils_image.py:237-243 for the 8-bit single-channel TIFF / PNG tiles of
    the SR-CACO-2 folds: HxWx1 uint8.  (PIL: cv2 is not part of this build.)"""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ('L', 'P', '1'):
            if im.mode in ('I;16', 'I', 'F'):
                raise NotImplementedError(f'{path}: {im.mode} image; the reference reads 8-bit tiles')
            im = im.convert('L')      # cv2 IMREAD_GRAYSCALE semantics for colour files (not used by the folds)
        a = np.asarray(im, dtype=np.uint8)
    return a[:, :, None]


class EvalPairs:
    """EVAL-phase items of DatasetDPSR: per index the dict the evaluation loop consumes (dataset_dpsr.py:746-838,948-1005)
    with l_im / h_im / l_to_h_img float32 CHW in [0,1] = uint8 / 255 (utils_image.py:322-323,381-382).

    True low-resolution tiles are read; where a pair has none (or --use_interpolated_low) the LR input is synthesised as
    the reference does for CACO-2 tiles (:776-804: bicubic down-scaling of the HR tile + seeded noise in the cells'
    region; dlib/datasets/lowres.py).  'l_to_h_img' -- the LR image brought to the HR size by cv2.resize(INTER_CUBIC),
    what the SRCNN-style nets consume (model_plain.py:184-195) -- comes from srhip_resize_cubic on the device."""

    def __init__(self, args, pairs_h: dict, pairs_l: dict):
        self.args, self.pairs_h, self.pairs_l = args, pairs_h, pairs_l
        self.im_h_ids = list(pairs_h.keys())
        self.sf = args.scale
        # ids <-> floats (dataset_dpsr.py:583-590): per-image details travel through all_gather as floats
        self.im_h_ids_to_float = {k: float(i) for i, k in enumerate(self.im_h_ids)}
        self.float_to_im_h_ids = {v: k for k, v in self.im_h_ids_to_float.items()}

    def __len__(self):
        return len(self.im_h_ids)

    def low_res_u8(self, index: int, img_h: np.ndarray, h_path: str):
        """(LR tile uint8 HWC, its path): the true tile, or the reference's synthesis (:776-804)."""
        from dlib.datasets import lowres
        h_id = self.im_h_ids[index]
        l_id = self.pairs_h[h_id]['low_path_key']
        l_path = self.pairs_l[l_id]['abs_path'] if (self.pairs_l and l_id in self.pairs_l) else ''
        synth = (not os.path.isfile(l_path)) or bool(getattr(self.args, 'use_interpolated_low', False))
        if not synth:
            return imread_gray_uint8(l_path), l_path
        if not lowres.is_caco2(h_path):
            raise NotImplementedError(f'{h_id}: synthesised low-resolution inputs are built for CACO-2 tiles (dataset_dpsr.py:'
                                      f'789-799; other sets go through utils_image.imresize_np, outside this build)')
        cmin, cmax = getattr(self.args, 'color_min', 0), getattr(self.args, 'color_max', 255)
        lo = lowres.interpolate_torch(img_h, 1. / self.sf, getattr(self.args, 'basic_interpolation', 'bicubic'), cmin, cmax)
        lo = np.clip(lo, a_min=cmin, a_max=cmax)
        lo = lowres.simulate_low_res(np.copy(lo), seed=index, th=float(getattr(self.args, 'inter_low_th', 7.)),
                                     sigma=float(getattr(self.args, 'inter_low_sigma', 6.)))
        return lo, h_path

    def __getitem__(self, index: int) -> dict:
        import torch
        from dlib.datasets import lowres
        h_id = self.im_h_ids[index]
        l_id = self.pairs_h[h_id]['low_path_key']
        h_path = self.pairs_h[h_id]['abs_path']
        img_h_full = imread_gray_uint8(h_path)
        img_l, l_path = self.low_res_u8(index, img_h_full, h_path)      # the synthesis sees the un-cropped tile (:752-760)
        hh, ww = img_h_full.shape[:2]
        img_h = img_h_full[:hh - hh % self.sf, :ww - ww % self.sf]            # modcrop (utils_image.py:295-306)
        to_t = lambda a: torch.from_numpy(np.ascontiguousarray(np.float32(a / 255.))).permute(2, 0, 1).float()
        out = {'l_im': to_t(img_l), 'l_id': l_id, 'l_path': l_path, 'h_im': to_t(img_h), 'h_id': h_id, 'h_path': h_path}
        if torch.cuda.is_available():
            from srhip import ops
            lu8 = torch.from_numpy(np.array(img_l[:, :, 0], copy=True))[None].cuda()
            up = lowres.l_to_h(lu8, img_h.shape[:2])                     # cv2.resize(img_l, (W, H), INTER_CUBIC) (:813-821,836)
            out['l_to_h_img'] = ops.u8_to_unit(up)                       # [1, H, W] float32 on the device
            out['l_to_h_img_aug'] = out['l_to_h_img']
        return out


class EvalLoader:
    """DataLoader(eval_set, batch_size=eval_bsize, shuffle=False, drop_last=False[, sampler]) without
    worker processes: batches of stacked tensors + lists of ids (utils_dataloaders.py:262-285)."""

    def __init__(self, dataset, batch_size: int, sampler=None):
        self.dataset, self.batch_size, self.sampler = dataset, int(batch_size), sampler

    def _indices(self):
        return list(self.sampler) if self.sampler is not None else list(range(len(self.dataset)))

    def __len__(self):
        n = len(self._indices())
        return (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        import torch
        idx = self._indices()
        for i in range(0, len(idx), self.batch_size):
            items = [self.dataset[j] for j in idx[i:i + self.batch_size]]
            out = {}
            for k in items[0]:
                vals = [it[k] for it in items]
                out[k] = torch.stack(vals) if torch.is_tensor(vals[0]) else vals
            yield out


def get_eval_loader(args, ds_name: str, n: int = -1):
    assert isinstance(n, int) and (n == -1 or n > 0), n
    assert f'X_{args.scale}' in ds_name or f'X-{args.scale}' in ds_name, (ds_name, args.scale)
    fold = join(args.splits_root, ds_name)
    pairs_l_h = get_pairs(join(fold, 'l_h.txt'))
    pairs_h_l = get_pairs(join(fold, 'h_l.txt'))
    if n != -1:
        keep = list(pairs_l_h.keys())[:n]
        pairs_l_h = {k: pairs_l_h[k] for k in keep}
        pairs_h_l = {k: v for k, v in pairs_h_l.items() if v in keep}
    base = join(args.data_root, dataset_dir(ds_name))
    strip = lambda k: k.split(constants.CODE_IDENTIFIER)[0]
    pairs_h = {k: {'low_path_key': v, 'abs_path': join(base, strip(k))} for k, v in pairs_h_l.items()}
    pairs_l = {k: {'high_path_key': v, 'abs_path': k if k.startswith('None_') else join(base, strip(k))}
               for k, v in pairs_l_h.items()}
    ds = EvalPairs(args, pairs_h, pairs_l)
    sampler = None
    if getattr(args, 'distributed', False) and args.eval_bsize > 1:
        import torch.distributed as dist
        sampler = eval_sampler(len(ds), dist.get_world_size(), dist.get_rank())
    return EvalLoader(ds, args.eval_bsize, sampler)


def get_train_set(args, device, rank: int = 0, world: int = 1):
    """The training sets of ``args.train_dsets`` as ONE resident device-side set (get_train_loader,
    utils_dataloaders.py:56-158: pairs of all listed datasets merged)."""
    from dlib.datasets.dataset_dpsr import ResidentTrainSet
    names = [x for x in args.train_dsets.split(constants.SEP) if x != '']
    assert names, 'no train sets'
    pairs_h, pairs_l = {}, {}
    strip = lambda k: k.split(constants.CODE_IDENTIFIER)[0]
    for ds in names:
        assert f'X_{args.scale}' in ds or f'X-{args.scale}' in ds, (ds, args.scale)
        fold = join(args.splits_root, ds)
        base = join(args.data_root, dataset_dir(ds))
        for k, v in get_pairs(join(fold, 'h_l.txt')).items():
            assert k not in pairs_h, k
            pairs_h[k] = {'low_path_key': v, 'abs_path': join(base, strip(k))}
        for k, v in get_pairs(join(fold, 'l_h.txt')).items():
            assert k not in pairs_l, k
            pairs_l[k] = {'high_path_key': v, 'abs_path': k if k.startswith('None_') else join(base, strip(k))}
    return ResidentTrainSet(args, pairs_h, pairs_l, device, rank, world)


def get_all_eval_loaders(args, ds_names: str, n: int = -1) -> dict:
    names = [x for x in ds_names.split(constants.SEP) if x != '']
    assert names, f'no eval sets in: {ds_names}'
    return {ds: get_eval_loader(args, ds, n) for ds in names}
