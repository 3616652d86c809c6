"""The one model-like object of the reference's trainer module that sits on the evaluation path:
``Interpolate`` -- the Bicubic baseline row of the results tables (dlib/utils/utils_trainer.py:89-167,
used at :293,1265).  SURVEY section 8 row a18 keeps it on stock PyTorch-ROCm (``F.interpolate`` with
``antialias=True`` is not a kernel target); the class follows the ModelPlain-style protocol the
evaluation loop consumes (feed_data / test / current_visuals).  Below it: the evaluation over folds
(fast_eval / evaluate) and the training loop around the step (train_valid)."""
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
    _sync_replica_buffers(model)        # a collective: here every rank arrives; below the master may evaluate alone
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


# ==============================================================================================
# The training loop around the step (reference dlib/utils/utils_trainer.py:170-530): epochs over the
# rank's minibatches, validation every `checkpoint_eval` iterations with best-model selection,
# `<iter>_G.pth` + `<iter>_optimizerG.pth` every `checkpoint_save` iterations (older ones deleted),
# trackers pickled beside them, the test split scored with the best model at the end.  It leaves the
# experiment folder eval.py reads.  Not reproduced: the matplotlib tracker plots and the
# compute-cluster scratch copies (callers' tooling).
#
# Differences on purpose: per-iteration loss values stay on the device (one small copy per step) and
# reach the tracker in one transfer at every validation / checkpoint / epoch end -- the reference's
# `.item()` per term per step is a host sync per step; the three barriers per iteration
# (:366,388,401) are dropped: the gradient all-reduce is the synchronisation point.
# ==============================================================================================
import math

from dlib.utils.utils_tracker import update_tracker_train, is_last_perf_best_perf, save_tracker
from dlib.utils.utils_config import clean_previous_checkpoints_except_last

__all__ += ['train_valid']


def _period(v, n_mbatchs: int, name: str) -> int:
    """`checkpoint_eval` / `checkpoint_save`: iterations (int) or a fraction of an epoch (float < 1)
    (utils_trainer.py:331-351)."""
    assert v > 0, (name, v)
    if v < 1:
        assert isinstance(v, float), (name, type(v))
        return max(int(v * n_mbatchs), 1)
    assert isinstance(v, int), (name, type(v))
    return v


def _sync_replica_buffers(model):
    """Rank 0's BatchNorm running statistics to every rank (what DDP's broadcast_buffers does at each forward,
    model_base.py:139) -- a collective, so it is issued here, at points EVERY rank reaches, and never from
    model.test() (evaluation may be master-only: utils_trainer.py:382-386)."""
    fn = getattr(model, 'sync_replica_buffers', None)
    if fn is not None:
        fn()


def _validate(args, model, valid_loaders: dict, split: str, tracker: dict, roi_tracker: dict, current_step: int,
              current_epoch: int):
    """utils_trainer.py:170-273: every validation set through fast_eval; when the last evaluation is the best one
    (args.model_select_mtr, ROI-based if eval_over_roi_also_model_select) the weights go to
    best-models/G-model.pth (multi_valid: G-<ds>.pth) with the per-image details beside them."""
    model_is_interp = isinstance(model, Interpolate)
    master = getattr(args, 'is_master', True)
    for ds_name, loader in valid_loaders.items():
        if model_is_interp:
            ds_name = f'{ds_name}_{args.basic_interpolation}'
        save_img_dir = join(args.outd, args.save_dir_imgs, split, ds_name)
        os.makedirs(save_img_dir, exist_ok=True)
        tracker, details, roi_tracker, roi_details = fast_eval(
            model, loader, ds_name, split, tracker, roi_tracker, args, current_step, current_epoch,
            nbr_to_plot=4, save_img_dir=save_img_dir)
        is_last_best = is_last_perf_best_perf(tracker, roi_tracker, args.eval_over_roi_also,
                                              args.eval_over_roi_also_model_select, split=split, ds_name=ds_name,
                                              metric=args.model_select_mtr)
        if not master:
            continue
        d = join(args.outd_backup, 'best-models')
        os.makedirs(d, exist_ok=True)
        if is_last_best:
            if not model_is_interp:
                model.save_best(d, p_name_file=f'{ds_name}.pth' if args.multi_valid else 'model.pth')
            with open(join(d, f'details_{ds_name}.yml'), 'w') as f:
                yaml.dump(details, f)
            if args.eval_over_roi_also:
                with open(join(d, f'roi_details_{ds_name}.yml'), 'w') as f:
                    yaml.dump(roi_details, f)
        status = write_current_perf_eval(tracker, split, ds_name, d if is_last_best else None,
                                         f'{ds_name}.yaml' if is_last_best else None, current_step, current_epoch)
        roi_status = None
        if args.eval_over_roi_also:
            roi_status = write_current_perf_eval(roi_tracker, split, ds_name, d if is_last_best else None,
                                                 f'roi-{ds_name}.yaml' if is_last_best else None, current_step,
                                                 current_epoch)
        DLLogger.log(current_perf_to_str(status, roi_status, args.model_select_mtr,
                                         bool(args.eval_over_roi_also_model_select)))
    return tracker, roi_tracker


class _LossLog:
    """[total, term1, ...] of every iteration, kept on the device until someone needs numbers."""

    def __init__(self):
        self.pending, self.epoch_sum, self.n_epoch = [], None, 0

    def push(self, loss_buf: torch.Tensor):
        self.pending.append(loss_buf.detach().clone())

    def drain(self, tracker: dict, names: list) -> dict:
        if not self.pending:
            return tracker
        vals = torch.stack(self.pending).double().cpu()            # the one transfer
        self.pending = []
        vals[:, 0] = vals[:, 1:].sum(1)                            # slot 0 = MasterLoss total (master.py:46-56)
        for row in vals.tolist():
            tracker = update_tracker_train(tracker, n_losses=names, v_losses=row, period=constants.PR_ITER)
        s = vals.sum(0)
        self.epoch_sum = s if self.epoch_sum is None else self.epoch_sum + s
        self.n_epoch += vals.shape[0]
        return tracker

    def close_epoch(self):
        out = (self.epoch_sum / max(self.n_epoch, 1)).tolist() if self.epoch_sum is not None else None
        self.epoch_sum, self.n_epoch = None, 0
        return out


def train_valid(args, model, train_loader, train_sampler, valid_loaders: dict, test_loaders: dict, tracker: dict,
                roi_tracker: dict, current_step: int):
    """utils_trainer.py:276-530.  ``train_loader``: a dlib.datasets.dataset_dpsr.ResidentTrainSet (``len()`` =
    this rank's minibatches per epoch, ``.epoch(e)`` = their batch dicts under set_epoch(e) semantics; it carries its
    own sampler, so ``train_sampler`` may be None).  ``current_step``: iterations already done (main.py: the newest
    checkpoint's label); the loop resumes at epoch floor(current_step / len) like the reference (:287-290)."""
    distributed = bool(getattr(args, 'distributed', False))
    master = getattr(args, 'is_master', True)
    n_mbatchs = len(train_loader)
    assert n_mbatchs > 0, "the training split is smaller than one batch per rank"
    current_epoch = math.floor(current_step / float(n_mbatchs))
    if current_step == 0 and valid_loaders:        # the Bicubic row of the validation tracker (:292-310)
        tracker, roi_tracker = _validate(args, Interpolate(task=args.task, scale=args.scale,
                                                           scale_mode=args.basic_interpolation),
                                         valid_loaders, constants.VALIDSET, tracker, roi_tracker, 0, 0)
    max_seed = (2 ** 32) - 1
    tr = args.train
    n_check_eval = _period(tr['checkpoint_eval'], n_mbatchs, 'checkpoint_eval')
    n_checkpoint_save = _period(tr['checkpoint_save'], n_mbatchs, 'checkpoint_save')
    max_iters = getattr(args, 'max_iters', None)           # (not a reference option) stop after this many iterations
    losses = _LossLog()
    names = list(model.loss_fn.n_holder)
    stop = False
    for epoch in range(current_epoch, args.max_epochs):
        if stop:
            break
        t0 = _dt.datetime.now()
        if train_sampler is not None and distributed:
            train_sampler.set_epoch(epoch)
        for train_data in train_loader.epoch(epoch):
            current_step += 1
            # every rank re-seeds with myseed + current_step (:359-361): DropPath masks are a function of (seed, step)
            torch.manual_seed(int((args.myseed + current_step) % max_seed))
            model.set_train_mode()
            model.feed_data(train_data)
            model.optimize_parameters(epoch, current_step)
            model.update_learning_rate()
            losses.push(model.step_fn.loss_buf)
            do_eval = bool(valid_loaders) and current_step % n_check_eval == 0
            do_save = current_step % n_checkpoint_save == 0
            if do_eval or do_save:
                if not model.check_finite():
                    DLLogger.log('Terminated due to error: non-finite loss')        # tools.py:55-63
                    raise SystemExit(1)
                tracker = losses.drain(tracker, names)
            if do_eval:
                _sync_replica_buffers(model)          # every rank is here; the evaluation below may be master-only
                if (distributed and args.eval_bsize > 1) or master:
                    tracker, roi_tracker = _validate(args, model, valid_loaders, constants.VALIDSET, tracker,
                                                     roi_tracker, current_step, epoch)
                ddp_barrier(distributed)
            if do_save and master:
                model.save(current_step)
                clean_previous_checkpoints_except_last(model.save_dir, ['G', 'optimizerG'] + (['E'] if model.E_decay > 0 else []))
                save_tracker(args.outd_backup, tracker=tracker, roi_tracker=roi_tracker)
            if max_iters and current_step >= max_iters:
                stop = True
                break
        # (one host sync per epoch) a non-finite loss since the last checkpoint boundary must not reach the tracker / the
        # epoch mean: the reference terminates at the first one (tools.py:55-63)
        if not model.check_finite():
            DLLogger.log('Terminated due to error: non-finite loss')
            raise SystemExit(1)
        tracker = losses.drain(tracker, names)
        epoch_loss = losses.close_epoch()
        if epoch_loss is not None:
            tracker = update_tracker_train(tracker, n_losses=names, v_losses=epoch_loss, period=constants.PR_EPOCH)
            DLLogger.log(f'Epoch {epoch}. Total TR loss: {epoch_loss[0]:.5f}')
        freq = tr.get('test_epoch_freq', 50)
        if epoch > 0 and epoch % freq == 0 and test_loaders and not stop:
            model.flush()
            _sync_replica_buffers(model)
            tracker, roi_tracker = evaluate(args, model, test_loaders, tracker, roi_tracker, -1, -1,
                                            constants.TESTSET, use_best_models=True, nbr_to_plot=30)
        model.loss_fn.update_t()
        DLLogger.log(f'Train epoch runtime: {_dt.datetime.now() - t0}')
    # end of training: the test split with the best model(s) (:470-483), trackers beside the run
    model.flush()
    if test_loaders:
        _sync_replica_buffers(model)
        tracker, roi_tracker = evaluate(args, model, test_loaders, tracker, roi_tracker, -1, -1, constants.TESTSET,
                                        use_best_models=True, nbr_to_plot=30)
    if master:
        save_tracker(args.outd, tracker=tracker, roi_tracker=roi_tracker)
    return tracker, roi_tracker, current_step
