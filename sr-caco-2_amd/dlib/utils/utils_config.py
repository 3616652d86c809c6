"""Checkpoint bookkeeping of an experiment folder (reference dlib/utils/utils_config.py:407-482): which
``<iter>_<label>.pth`` is the last one, deletion of the older ones, and the yaml dump of the run's
arguments that eval.py reads back (``config_model.yml`` / ``config_final.yml``)."""
import glob
import os
import re
from os.path import join
from typing import List, Tuple

import yaml

import dlib.dllogger as DLLogger

__all__ = ['find_last_checkpoint', 'delete_previous_checkpoints_except_last',
           'clean_previous_checkpoints_except_last', 'save_config']


def _iters_on_disk(save_dir: str, net_type: str) -> List[int]:
    pat = re.compile(r'^(\d+)_' + re.escape(net_type) + r'\.pth$')
    out = []
    for p in glob.glob(join(save_dir, f'*_{net_type}.pth')):
        m = pat.match(os.path.basename(p))
        if m:
            out.append(int(m.group(1)))
    return out


def find_last_checkpoint(save_dir: str, net_type: str = 'G', pretrained_path: str = '') -> Tuple[int, str]:
    """(iteration, path) of the newest ``<iter>_<net_type>.pth`` under save_dir; (0, pretrained_path) if none
    (utils_config.py:407-434).  net_type: 'G' | 'E' | 'optimizerG'."""
    its = _iters_on_disk(save_dir, net_type)
    if not its:
        return 0, pretrained_path
    last = max(its)
    return last, join(save_dir, f'{last}_{net_type}.pth')


def delete_previous_checkpoints_except_last(save_dir: str, net_type: str = 'G'):
    """utils_config.py:437-454."""
    its = _iters_on_disk(save_dir, net_type)
    if not its:
        DLLogger.log(f'no checkpoint @{net_type} to delete.')
        return
    last = max(its)
    for it in its:
        if it != last:
            path = join(save_dir, f'{it}_{net_type}.pth')
            os.remove(path)
            DLLogger.log(f'deleted checkpoint @{net_type}: {path}')


def clean_previous_checkpoints_except_last(save_dir: str, net_types: List[str]):
    for net_type in net_types:
        delete_previous_checkpoints_except_last(save_dir, net_type)


def _plain(v):
    """yaml.safe_load must read the file back (eval.py): tuples -> lists, unknown objects -> str."""
    if isinstance(v, dict):
        return {str(k): _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, (str, int, float, bool)) or v is None:
        return v
    return str(v)


def save_config(args, save_dir: str, name: str = 'config_final.yml'):
    """yaml dump of the run's arguments (utils_config.py:460-482; utils_parser.py:1397-1401 writes the same dict as
    config_model.yml before training)."""
    d = dict(args) if isinstance(args, dict) else dict(vars(args))
    os.makedirs(save_dir, exist_ok=True)
    with open(join(save_dir, name), 'w') as f:
        yaml.dump(_plain(d), f)
