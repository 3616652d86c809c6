"""The one model-like object of the reference's trainer module that sits on the evaluation path:
``Interpolate`` -- the Bicubic baseline row of the results tables (dlib/utils/utils_trainer.py:89-167,
used at :293,1265).  SURVEY section 8 row a18 keeps it on stock PyTorch-ROCm (``F.interpolate`` with
``antialias=True`` is not a kernel target); the class follows the ModelPlain-style protocol the
evaluation loop consumes (feed_data / test / current_visuals).  The trainer loop itself is a caller of
the hot path and out of scope."""
import torch
import torch.nn.functional as F

from dlib.utils import constants

__all__ = ['Interpolate']


# task -> (batch key of the input, batch key of the target, does the scale factor apply?)
_TASK_KEYS = {constants.SUPER_RES: ('l_im', 'h_im', True),
              constants.RECONSTRUCT: ('in_reconstruct', 'trg_reconstruct', False)}


class Interpolate(torch.nn.Module):
    """Bicubic (antialiased) interpolation behind the evaluation protocol; parameter-free."""

    def __init__(self, task: str, scale: int, scale_mode: str):
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError("Interpolate (libsrhip build) runs on the GPU, like the reference "
                               "(utils_trainer.py:93); there is no CPU path")
        if task not in _TASK_KEYS:
            raise AssertionError(f"{task} | {constants.TASKS}")
        if scale_mode not in constants.INTERPOLATION_MODES:
            raise AssertionError(scale_mode)
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.task, self.scale, self.scale_mode = task, scale, scale_mode
        self.L = self.E = self.H = None

    def feed_data(self, data, need_H=True):
        key_in, key_trg, _ = _TASK_KEYS[self.task]
        self.L = data[key_in].to(self.device)
        self.H = data[key_trg].to(self.device) if need_H else self.H

    def forward(self):
        assert self.L.ndim == 4, self.L.ndim
        factor = self.scale if _TASK_KEYS[self.task][2] else 1
        up = F.interpolate(self.L, scale_factor=factor, mode=self.scale_mode, antialias=True)
        self.E = up.clamp_(0.0, 1.0)              # images live in [0, 1] (utils_trainer.py:146-147)

    def test(self):
        self.eval()
        with torch.no_grad():
            self.forward()

    def set_eval_mode(self):
        self.eval()

    def set_train_mode(self):                     # nothing to train
        pass

    def current_visuals(self, need_H=True):
        keys = ('L', 'E', 'H') if need_H else ('L', 'E')
        return {k: getattr(self, k).detach().float() for k in keys}


# ==============================================================================================
# Evaluation over the test folds (reference dlib/utils/utils_trainer.py:533-862,1102-1320): what
# eval.py drives.  Metrics come from the fused sweep (dlib.metrics.sweep: PSNR / PSNR_Y / MSE /
# NRMSE / SSIM x {no ROI + every ROI threshold} in two passes over the images instead of the
# reference's 8 x 5 separate calls, utils_trainer.py:874-930,961-1032); everything else -- flipped-strip
# padding, per-image details, tracker updates, the yaml / pkl files -- is host logic with the
# reference's semantics and file formats.
# ==============================================================================================
import datetime as _dt
import os
from os.path import join

import yaml

import dlib.dllogger as DLLogger
from dlib.utils.utils_tracker import (update_tracker_eval, reset_tracker_eval, write_current_perf_eval,
                                      current_perf_to_str)

__all__ += ['fast_eval', 'evaluate_single_ds', 'evaluate']

_MTRS = (constants.PSNR_MTR, constants.MSE_MTR, constants.NRMSE_MTR, constants.SSIM_MTR, constants.PSNR_Y_MTR)


def ddp_barrier(distributed: bool):
    if distributed:
        import torch.distributed as dist
        dist.barrier()


def reformat_id(img_id: str) -> str:
    """file-system-safe image id (dlib/utils/shared.py: '/' -> '_')."""
    return img_id.replace('/', '_')


def _forward_with_padding(test_data: dict, model, args):
    """SwinIR evaluation wrapper (utils_trainer.py:829-862): the low-resolution batch is extended
    by FLIPPED STRIPS of itself to the next multiple of the window -- a whole extra window when it
    already is one: 64 -> 72 -- then output and input are cropped back.  Other nets: plain test()."""
    if args.netG['net_type'] != constants.SWINIR:
        model.feed_data(test_data)
        model.test()
        return model
    model.feed_data(dict(test_data))
    _, _, h_old, w_old = test_data['l_im'].shape
    wsz = args.netG[f"{args.netG['net_type']}_window_size"]
    h_pad = (h_old // wsz + 1) * wsz - h_old
    w_pad = (w_old // wsz + 1) * wsz - w_old
    im = model.L
    im = torch.cat([im, torch.flip(im[:, :, h_old - h_pad:, :], [2])], 2)      # the pad only is flipped
    im = torch.cat([im, torch.flip(im[:, :, :, w_old - w_pad:], [3])], 3)
    model.L = im
    model.test()
    model.E = model.E[..., :h_old * args.scale, :w_old * args.scale]
    model.L = model.L[..., :h_old, :w_old]
    return model


def _check_finite_nonneg(vals: torch.Tensor, name: str):
    """utils_trainer.py:933-958: the reference logs and exits on inf / nan / negative metric values."""
    bad = (~torch.isfinite(vals)).sum().item() + (vals < 0).sum().item()
    if bad:
        DLLogger.log(f'Terminated due to error: {bad} non-finite or negative values in {name}')
        raise SystemExit(1)


def _save_prediction_png(e_u8: torch.Tensor, path: str):
    from PIL import Image
    import numpy as np
    Image.fromarray(e_u8.squeeze().clamp(0, 255).to(torch.uint8).cpu().numpy().astype(np.uint8)).save(path)


def _fast_update_tracker(args, tracker, mtr_val, split, ds_name, idx_best=None):
    """master metric first (it fixes the index of the best evaluation), the others follow
    (utils_trainer.py:764-798)."""
    master = args.model_select_mtr
    tracker, found = update_tracker_eval(tracker, split, ds_name, master, mtr_val[master], idx_best)
    best = idx_best if idx_best is not None else found
    for k in mtr_val:
        if k != master:
            tracker, _ = update_tracker_eval(tracker, split, ds_name, k, mtr_val[k], best)
    return tracker, best


def fast_eval(model, data_loader, ds_name: str, split: str, tracker: dict, roi_tracker: dict, args,
              current_step: int, epoch: int, nbr_to_plot: int = 2, save_img_dir: str = ''):
    """One dataset: returns (tracker, details, roi_tracker, roi_details) with
    details[h_id] = {psnr, mse, nrmse, ssim, psnr_y} per image (ROI: averaged over the thresholds
    eval_over_roi_also_ths, utils_trainer.py:874-930) and the dataset means appended to the trackers."""
    from dlib import metrics
    if split == constants.TESTSET:
        reset_tracker_eval(tracker, split, ds_name)
        reset_tracker_eval(roi_tracker, split, ds_name)
    model.set_eval_mode()
    border = args.scale
    with_roi = bool(args.eval_over_roi_also)
    ths = tuple(int(t) for t in args.eval_over_roi_also_ths) if with_roi else ()
    if with_roi:
        assert len(ths) > 0
    sums = {m: 0.0 for m in _MTRS}
    roi_sums = {m: 0.0 for m in _MTRS}
    details, roi_details = {}, {}
    seen = 0
    t0 = _dt.datetime.now()
    DLLogger.log(f'Eval: {ds_name} (split: {split})')
    for test_data in data_loader:
        if isinstance(model, Interpolate):
            model.feed_data(test_data)
            model.test()
        else:
            model = _forward_with_padding(test_data, model, args)
        vis = model.current_visuals()
        with torch.no_grad():
            sw = metrics.sweep(vis['E'], vis['H'], border=border, thresholds=ths)
        host = {m: sw[m].double().cpu() for m in _MTRS}            # (B, 1 + nth)
        for m in _MTRS:
            _check_finite_nonneg(host[m], m)
            sums[m] += host[m][:, 0].sum().item()
            if with_roi:
                roi_sums[m] += host[m][:, 1:].mean(dim=1).sum().item()
        for i, img_id in enumerate(test_data['h_id']):
            assert img_id not in details, img_id
            details[img_id] = {m: host[m][i, 0].item() for m in _MTRS}
            if with_roi:
                roi_details[img_id] = {m: host[m][i, 1:].mean().item() for m in _MTRS}
            if seen + i < nbr_to_plot and getattr(args, 'is_master', True) and save_img_dir:
                e_u8 = metrics.tensor2uint82float(vis['E'][i:i + 1].float().contiguous())
                _save_prediction_png(e_u8, join(save_img_dir, f'{reformat_id(img_id)}.png'))
        seen += test_data['l_im'].shape[0]
    n = float(seen)
    if getattr(args, 'distributed', False) and args.eval_bsize > 1:      # rank-sharded evaluation
        from dlib.utils.utils_parallel import sync_metric_sums, sync_dict_across_gpus
        dev = model.device if hasattr(model, 'device') else torch.device('cuda')
        packed = {f'a/{m}': torch.tensor(sums[m], dtype=torch.float64, device=dev) for m in _MTRS}
        packed.update({f'r/{m}': torch.tensor(roi_sums[m], dtype=torch.float64, device=dev) for m in _MTRS})
        tot, n = sync_metric_sums(packed, n)                              # ONE collective for all sums
        sums = {m: tot[f'a/{m}'].item() for m in _MTRS}
        roi_sums = {m: tot[f'r/{m}'].item() for m in _MTRS}
        ids = data_loader.dataset.im_h_ids_to_float

        def gather(det):
            out = {}
            for m in _MTRS:
                g = sync_dict_across_gpus({ids[k]: torch.tensor(v[m], dtype=torch.float64, device=dev)
                                           for k, v in det.items()}, move_sync_vals_to_cpu=True)
                for fid, val in g.items():
                    out.setdefault(data_loader.dataset.float_to_im_h_ids[fid], {})[m] = val.item()
            return out
        details = gather(details)
        if with_roi:
            roi_details = gather(roi_details)
    mtr_val = {m: sums[m] / n for m in _MTRS}
    roi_mtr_val = {m: roi_sums[m] / n for m in _MTRS}
    DLLogger.log(f'Eval time Split: {split}, dataset: {ds_name}:  {_dt.datetime.now() - t0}')
    if not with_roi:
        tracker, _ = _fast_update_tracker(args, tracker, mtr_val, split, ds_name)
    elif args.eval_over_roi_also_model_select:
        roi_tracker, best = _fast_update_tracker(args, roi_tracker, roi_mtr_val, split, ds_name)
        tracker, _ = _fast_update_tracker(args, tracker, mtr_val, split, ds_name, best)
    else:
        tracker, best = _fast_update_tracker(args, tracker, mtr_val, split, ds_name)
        roi_tracker, _ = _fast_update_tracker(args, roi_tracker, roi_mtr_val, split, ds_name, best)
    model.set_train_mode()
    return tracker, details, roi_tracker, roi_details


def evaluate_single_ds(args, model, loader, ds_name: str, tracker: dict, roi_tracker: dict, current_step: int,
                       epoch: int, split: str, nbr_to_plot: int = 10, save_img_dir: str = ''):
    """fast_eval + the files of utils_trainer.py:1102-1181 under <outd_backup>/best-models:
    details_<ds>.yml, roi_details_<ds>.yml, <ds>.yaml, roi-<ds>.yaml."""
    if not os.path.isdir(save_img_dir):
        save_img_dir = join(args.outd, args.save_dir_imgs, split, ds_name)
        os.makedirs(save_img_dir, exist_ok=True)
    tracker, details, roi_tracker, roi_details = fast_eval(
        model, loader, ds_name, split, tracker, roi_tracker, args, current_step, epoch, nbr_to_plot, save_img_dir)
    if getattr(args, 'is_master', True):
        d = join(args.outd_backup, 'best-models')
        os.makedirs(d, exist_ok=True)
        with open(join(d, f'details_{ds_name}.yml'), 'w') as f:
            yaml.dump(details, f)
        status = write_current_perf_eval(tracker, split, ds_name, d, f'{ds_name}.yaml', current_step, epoch)
        roi_status = None
        if args.eval_over_roi_also:
            with open(join(d, f'roi_details_{ds_name}.yml'), 'w') as f:
                yaml.dump(roi_details, f)
            roi_status = write_current_perf_eval(roi_tracker, split, ds_name, d, f'roi-{ds_name}.yaml',
                                                 current_step, epoch)
        DLLogger.log(current_perf_to_str(status, roi_status, args.model_select_mtr,
                                         bool(args.eval_over_roi_also_model_select)))
    return tracker, roi_tracker


def evaluate(args, model, loaders: dict, tracker: dict, roi_tracker: dict, current_step: int, epoch: int,
             split: str, use_best_models: bool = True, nbr_to_plot: int = 10):
    """Test-split evaluation with the best checkpoint(s) (utils_trainer.py:1184-1320): the current
    weights are parked in best-models/G-current_model.pth, every dataset is evaluated with
    best-models/G-model.pth (multi_valid: G-<validset>.pth), followed by the bicubic baseline row
    '<ds>_<basic_interpolation>', then the current weights come back."""
    assert split == constants.TESTSET, split
    distributed = bool(getattr(args, 'distributed', False))
    master = getattr(args, 'is_master', True)
    d = join(args.outd_backup, 'best-models')
    ddp_barrier(distributed)
    DLLogger.log(f'Eval {split}: {constants.SEP.join(loaders.keys())}')
    if master:
        os.makedirs(d, exist_ok=True)
        model.save_current(save_dir=d)
    ddp_barrier(distributed)
    for ds_name in loaders:
        if use_best_models:
            fname = f'G-{ds_name}.pth'.replace(split, constants.VALIDSET) if args.multi_valid else 'G-model.pth'
            path = join(d, fname)
            if not os.path.isfile(path):
                DLLogger.log(f'No best model/checkpoint found for eval over {split} @ {ds_name}: Skipping. '
                             f'Model not found @: {path}')
                continue
            model.load_network(path, model.netG, strict=True, param_key='params')
        everyone = distributed and args.eval_bsize > 1
        ddp_barrier(distributed)
        if everyone or master:
            tracker, roi_tracker = evaluate_single_ds(args, model, loaders[ds_name], ds_name, tracker, roi_tracker,
                                                      current_step, epoch, split, nbr_to_plot)
            base = Interpolate(task=args.task, scale=args.scale, scale_mode=args.basic_interpolation)
            tracker, roi_tracker = evaluate_single_ds(args, base, loaders[ds_name],
                                                      f'{ds_name}_{args.basic_interpolation}', tracker, roi_tracker,
                                                      current_step, epoch, split, nbr_to_plot)
        ddp_barrier(distributed)
    model.load_current(save_dir=d)
    ddp_barrier(distributed)
    return tracker, roi_tracker
