#!/usr/bin/env python3
"""Evaluate a trained experiment folder on its test split -- the reference's ``eval.py``
(eval.py:46-146) on the libsrhip path:

    python eval.py --cudaid 0 --exp_path <folder> [--data_root <datasets>] [--splits_root <folds>]

<folder> is a reference-format experiment directory: ``config_model.yml`` (the yaml dump of the
run's args, utils_parser.py:1397-1401) and ``best-models/G-model.pth`` (raw ``state_dict``,
model_base.py:173-181).  Written, as by the reference, under ``<folder>/eval_test_<split>/``:
``log.txt``, ``log.json``, ``tracker.pkl``, ``roi_tracker.pkl``; under ``<folder>/best-models/``:
``details_<ds>.yml``, ``<ds>.yaml`` (+ ``roi_details_<ds>.yml``, ``roi-<ds>.yaml`` with
--eval_over_roi_also) and the same four for the bicubic baseline row ``<ds>_bicubic``; predictions of the
first images under ``<folder>/<save_dir_imgs>/test/<ds>/``.

The reference resolves the dataset root from host-specific environment variables
(utils_config.py:24-56); here it is ``--data_root`` / $SRHIP_DATA_ROOT, and the folds directory
``--splits_root`` (default: the config's ``splits_root`` relative to this file's parent)."""
import argparse
import datetime as dt
import os
import sys
from os.path import join, dirname, abspath, isdir

HERE = dirname(abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import yaml  # noqa: E402
import torch  # noqa: E402

from dlib.utils import constants  # noqa: E402
import dlib.dllogger as DLLogger  # noqa: E402
from dlib.utils.tools import Dict2Obj  # noqa: E402
from dlib.models.select_model import define_model  # noqa: E402
from dlib.utils.utils_dataloaders import get_all_eval_loaders  # noqa: E402
from dlib.utils.utils_trainer import evaluate  # noqa: E402
from dlib.utils.utils_tracker import find_last_tracker, save_tracker  # noqa: E402


def evaluate_pretrained(argv=None):
    t0 = dt.datetime.now()
    ap = argparse.ArgumentParser()
    ap.add_argument("--cudaid", type=str, default=None, help="cuda id.")
    ap.add_argument("--exp_path", type=str, default=None)
    ap.add_argument("--data_root", type=str, default=os.environ.get("SRHIP_DATA_ROOT"))
    ap.add_argument("--splits_root", type=str, default=None)
    ns = ap.parse_args(argv)
    exp_path = ns.exp_path
    assert exp_path and isdir(exp_path), exp_path
    with open(join(exp_path, 'config_model.yml'), 'r') as fy:
        args = Dict2Obj(yaml.safe_load(fy))
    args.distributed = False
    if ns.data_root is None:
        raise SystemExit("eval.py: give --data_root (or SRHIP_DATA_ROOT): the parent of the dataset folders")
    args.data_root = ns.data_root
    if ns.splits_root is not None:
        args.splits_root = ns.splits_root
    elif not os.path.isabs(args.splits_root or ''):
        args.splits_root = join(dirname(HERE), args.splits_root or 'folds')
    args.outd = exp_path
    split = args.test_dsets
    assert len(split.split(constants.SEP)) == 1, split
    outd = join(exp_path, f'eval_test_{split}')
    os.makedirs(outd, exist_ok=True)
    DLLogger.init_arb(log_dir=outd, is_master=True, reset=True)
    DLLogger.log(f"Start time: {t0}")
    DLLogger.log(f'Task: {args.task}. Trainset: {args.train_dsets} \t Method: {args.method}.')
    DLLogger.log(f"Evaluate split {split}")
    torch.manual_seed(int(args.myseed or 0))
    torch.cuda.set_device(int(str(ns.cudaid or '0').split(',')[0]))

    args.is_train = False
    args.netG['checkpoint_path_netG'] = join(exp_path, 'best-models/G-model.pth')
    if args.amp:
        DLLogger.log('config has amp=True: evaluating with the reduced-precision (single bf16 product) kernels')
    model = define_model(args)
    model.load()
    model.netG.eval()
    DLLogger.log(model.info_network())

    args.outd = exp_path
    args.outd_backup = exp_path
    args.is_master = True
    loaders = get_all_eval_loaders(args, args.test_dsets, n=-1)
    tracker, roi_tracker = find_last_tracker(outd, args)
    tracker, roi_tracker = evaluate(args=args, model=model, loaders=loaders, tracker=tracker,
                                    roi_tracker=roi_tracker, current_step=-1, epoch=-1,
                                    split=constants.TESTSET, use_best_models=True, nbr_to_plot=30)
    save_tracker(outd, tracker=tracker, roi_tracker=roi_tracker)
    DLLogger.log(f"Bye. ({dt.datetime.now() - t0})")
    return tracker, roi_tracker


if __name__ == '__main__':
    evaluate_pretrained()
