"""Performance tracker of the reference's train / eval loops (dlib/utils/utils_tracker.py:42-108,
110-336): the nested dict that is pickled to ``tracker.pkl`` / ``roi_tracker.pkl`` and summarised in
``best-models/<ds>.yaml``.  Same layout, keys and update rules (so a tracker written by either side
loads in the other); the matplotlib plotting helpers of the reference module are callers' tooling and
not reproduced.

Layout (init_tracker):
  tracker['train'] = {'period_epoch': {loss: {'vals', 'best_val'}}, 'period_iter': {...}}
  tracker['val' | 'test'][<dataset> | <dataset>_<basic_interpolation>][<metric>] = {'vals': [...], 'best_val': v}
"""
import os
import pickle as pkl
from os.path import join
from typing import Tuple

import numpy as np
import torch
import yaml

from dlib.utils import constants

__all__ = ['init_tracker', 'find_last_tracker', 'save_tracker', 'update_tracker_train',
           'update_tracker_eval', 'reset_tracker_eval', 'is_last_perf_best_perf',
           'write_current_perf_eval', 'current_perf_to_str']


def _fresh_metrics() -> dict:
    return {m: {'vals': [], 'best_val': 0.0} for m in constants.METRICS}


def _scalar(v):
    if torch.is_tensor(v):
        return v.detach().item()
    if isinstance(v, np.ndarray):
        return v.item()
    return v


def init_tracker(args) -> dict:
    out = {constants.TRAINSET: {constants.PR_EPOCH: dict(), constants.PR_ITER: dict()}}
    for split, names in ((constants.VALIDSET, args.valid_dsets), (constants.TESTSET, args.test_dsets)):
        out[split] = {}
        subsets = names.split(constants.SEP)
        for s in subsets:                                        # the model's rows first ...
            out[split][s] = _fresh_metrics()
        for s in subsets:                                        # ... then the interpolation baseline's
            out[split][f'{s}_{args.basic_interpolation}'] = _fresh_metrics()
    return out


def find_last_tracker(save_dir: str, args) -> Tuple[dict, dict]:
    """(tracker, roi_tracker) from <save_dir>/{tracker,roi_tracker}.pkl, fresh ones if absent / unreadable."""
    paths = [join(save_dir, 'tracker.pkl'), join(save_dir, 'roi_tracker.pkl')]
    if os.path.isfile(paths[0]):
        try:
            loaded = []
            for p in paths:
                with open(p, 'rb') as f:
                    loaded.append(pkl.load(f))
            return loaded[0], loaded[1]
        except Exception:          # same policy as the reference: start over
            pass
    return init_tracker(args), init_tracker(args)


def save_tracker(save_dir: str, tracker: dict, roi_tracker: dict):
    for name, obj in (('tracker.pkl', tracker), ('roi_tracker.pkl', roi_tracker)):
        with open(join(save_dir, name), 'wb') as f:
            pkl.dump(obj, f, protocol=pkl.HIGHEST_PROTOCOL)


def _last_index_of(vals: list, v) -> int:
    return len(vals) - 1 - vals[::-1].index(v)


def update_tracker_eval(tracker: dict, split: str, ds_name: str, metric: str, value,
                        idx_best: int = None) -> Tuple[dict, int]:
    """Append ``value``.  idx_best None: this metric decides -- best_val = BEST_MTR(metric)(value, best)
    and the index of its LAST occurrence is returned; else best_val = vals[idx_best] (the slave metrics
    follow the master metric's best evaluation)."""
    assert split in (constants.VALIDSET, constants.TESTSET) and metric in constants.METRICS
    v = _scalar(value)
    node = tracker[split][ds_name]
    if metric not in node:
        node[metric] = {'vals': [v], 'best_val': v}
        return tracker, (0 if idx_best is None else None)
    rec = node[metric]
    rec['vals'].append(v)
    if idx_best is None:
        rec['best_val'] = constants.BEST_MTR[metric](v, rec['best_val'])
        return tracker, _last_index_of(rec['vals'], rec['best_val'])
    rec['best_val'] = rec['vals'][idx_best]
    return tracker, None


def reset_tracker_eval(tracker: dict, split: str, ds_name: str) -> dict:
    assert split == constants.TESTSET, split      # test performance is not tracked over time
    tracker[split][ds_name].update(_fresh_metrics())
    return tracker


def update_tracker_train(tracker: dict, n_losses: list, v_losses: list, period: str) -> dict:
    assert period in constants.PERIODS
    node = tracker[constants.TRAINSET][period]
    for name, v in zip(n_losses, v_losses):
        v = _scalar(v)
        if name in node:
            node[name]['vals'].append(v)
            node[name]['best_val'] = min(v, node[name]['best_val'])
        else:
            node[name] = {'vals': [v], 'best_val': v}
    return tracker


def is_last_perf_best_perf(tracker: dict, roi_tracker: dict, eval_over_roi_also: bool,
                           eval_over_roi_also_model_select: bool, split: str, ds_name: str, metric: str) -> bool:
    t = roi_tracker if (eval_over_roi_also and eval_over_roi_also_model_select) else tracker
    rec = t[split][ds_name][metric]
    return rec['best_val'] == rec['vals'][-1]


def write_current_perf_eval(tracker: dict, split: str, ds_name: str, save_dir, name_f: str,
                            current_step: int, current_epoch: int) -> dict:
    """{'last_<m>', 'best_<m>', 'dataset', 'split', 'current_step', 'current_epoch'} -> <save_dir>/<name_f> (yaml)."""
    out = {}
    for m, rec in tracker[split][ds_name].items():
        if rec['vals']:
            out[f'last_{m}'] = rec['vals'][-1]
            out[f'best_{m}'] = rec['best_val']
    out.update(dataset=ds_name, split=split, current_step=current_step, current_epoch=current_epoch)
    if save_dir is not None:
        os.makedirs(save_dir, exist_ok=True)
        with open(join(save_dir, name_f), 'w') as f:
            yaml.dump(out, f)
    return out


def current_perf_to_str(status: dict, roi_status, master_mtr: str, model_select_roi: bool) -> str:
    lines = [f"CURRENT. EPO: {status['current_epoch']}. STEP: {status['current_step']}.",
             f"Dataset: {status['dataset']}.  Split: {status['split']}"]
    for m in constants.METRICS:
        if f'last_{m}' not in status:
            continue
        tail = ' ---> MASTER' if m == master_mtr else ''
        if roi_status is None:
            lines.append(f"{m}: {status[f'best_{m}']:<.6f} [BEST] | {status[f'last_{m}']:<.4f} [LAST]{tail}")
        else:
            a, r = ('', '*') if model_select_roi else ('*', '')
            lines.append(f"{m}: {status[f'best_{m}']:<.6f}{a} (ROI: {roi_status[f'best_{m}']:<.6f}{r}) [BEST] | "
                         f"{status[f'last_{m}']:<.6f} (ROI: {roi_status[f'last_{m}']:<.6f}) [LAST]{tail}")
    return '\n'.join(lines) + '\n'
