#!/usr/bin/env python3
"""Entry point with the reference's CLI contract for the hot-path networks.

    python main.py --task super-resolution --scale 8 --method SWINIR --net_type swinir \
        --n_channels 1 --h_size 512 --batch_size 8 --G_optimizer_type sgd --G_optimizer_lr 0.01 \
        --G_scheduler_type MyStepLR --G_scheduler_step_size 30 --G_scheduler_gamma 0.5 \
        --swinir_window_size 8 --swinir_depths 6+6+6+6 --swinir_embed_dim 180 \
        --swinir_num_heads 6+6+6+6 --swinir_mlp_ratio 2 --swinir_upsampler pixelshuffledirect \
        --l1 True --max_iters 100 [--distributed True --dist_backend nccl]

Flag names, the ``--net_type`` / ``--method`` pairing check, '+'-separated lists
and the three-stage config (defaults -> per-net defaults -> CLI overrides of
non-None values) follow dlib/utils/utils_parser.py:291-339,900-967,1142-1143 and
dlib/utils/utils_config.py:64-404 for the options this path uses.

Every flag of the reference's parser is known here (utils_parser.py:33-880) and falls into one of three classes:
implemented; accepted without effect because it only names folders / logging / launcher plumbing (IGNORED_FLAGS);
or -- a flag that changes the numbers of a run and is not implemented on this path -- accepted only at the
reference's default and an error otherwise (DEFAULT_ONLY_FLAGS).  A flag nobody knows is an error, as in the
reference (argparse's ``parse_args``; a key missing from the config raises ValueError, utils_parser.py:900-923).

As the reference's main.py (:27-35) it first looks for the newest ``<iter>_G.pth`` /
``<iter>_optimizerG.pth`` under ``<outd>/<save_dir_models>`` and resumes there (weights, optimizer state,
iteration count).  With ``--train_dsets`` (+ ``--valid_dsets`` / ``--test_dsets``, ``--data_root``,
``--splits_root``) it runs the reference's epoch loop (dlib/utils/utils_trainer.py:276-530: validation
every ``--checkpoint_eval``, best-model selection, checkpoints every ``--checkpoint_save``, test split with
the best model at the end) and leaves an experiment folder eval.py accepts.  Without folds it trains on one
synthetic batch of the configured shape through the same ModelPlain protocol and reports patches/sec plus
the metric sweep of utils_trainer.py:961-1032.
"""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from dlib.utils import constants  # noqa: E402
from dlib.utils.shared import safe_str_var  # noqa: E402
from dlib.utils.utils_init_default_args import init_net_g  # noqa: E402


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ('true', '1', 'yes'):
        return True
    if v.lower() in ('false', '0', 'no'):
        return False
    raise argparse.ArgumentTypeError(f'boolean expected, got {v!r}')


def plus_list(v):
    return [int(z) for z in v.split('+')]


def int_or_float(v):
    """--checkpoint_eval / --checkpoint_save (utils_parser.py:205-208): iterations, or a fraction of an epoch."""
    try:
        return int(v)
    except ValueError:
        return float(v)


class Dict2Obj(dict):
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


def get_config(net_type):
    """Defaults for the options on the path (utils_config.py:64-404)."""
    return {
        'task': constants.SUPER_RES, 'net_type': net_type, 'method': constants.NETTYPE_METHOD[net_type],
        'scale': 2, 'n_channels': 1, 'h_size': 96, 'batch_size': 8, 'eval_bsize': 8, 'myseed': 0,
        'distributed': False, 'dist_backend': constants.GLOO, 'cudaid': '0', 'amp': False,
        'eval_graph': False,     # (not a reference option) evaluation forwards replayed from a hipGraph: ModelPlain.test()
        'train_graph': None,     # (not a reference option) training step replayed from a hipGraph: None = the engine's default
        'max_epochs': 1, 'max_iters': 50, 'eval_over_roi_also': False,
        'train_dsets': '', 'valid_dsets': '', 'test_dsets': '', 'data_root': '', 'splits_root': 'folds',
        'sample_tr_patch': 'uniform', 'sample_tr_patch_th_style': 'fix_threshold', 'sample_tr_patch_th': 7,
        # input pipeline (utils_config.py:130-138,215-233,260-262; dlib/datasets/dataset_dpsr.py, lowres.py)
        'use_interpolated_low': False, 'inter_low_th': 7., 'inter_low_sigma': 6.,
        'ppiw': False, 'ppiw_min_per_col_w': 0.001,
        'da_blur': False, 'da_blur_prob': 0.5, 'da_blur_area': 0.3, 'da_blur_sigma': 1.,
        'da_dot_bin_noise': False, 'da_dot_bin_noise_prob': 0.5, 'da_dot_bin_noise_area': 0.3, 'da_dot_bin_noise_p': 0.5,
        'da_add_gaus_noise': False, 'da_add_gaus_noise_prob': 0.5, 'da_add_gaus_noise_area': 0.3,
        'da_add_gaus_noise_std': 0.03,
        'eval_over_roi_also_ths': [4, 5, 6, 7, 8, 9, 10], 'outd': './out',
        # experiment-folder options (utils_config.py:98-126): where checkpoints / best models / images go and how
        # the best model is chosen
        'save_dir_models': 'models', 'save_dir_imgs': 'imgs', 'basic_interpolation': 'bicubic',
        'model_select_mtr': constants.PSNR_MTR, 'eval_over_roi_also_model_select': False, 'valid_n_samples': -1,
        'train': {'l1': True, 'l1_lambda': 1., 'l2': False, 'l2_lambda': 1., 'ssim': False,
                  'ssim_lambda': 1., 'ssim_window_s': 11,
                  # optional terms (utils_config.py:296-357)
                  'l2sum': False, 'l2sum_lambda': 1., 'charbonnier': False, 'charbonnier_lambda': 1.,
                  'charbonnier_eps': 1e-9,
                  'boundpred': False, 'boundpred_lambda': 1., 'boundpred_eps': 1., 'boundpred_restore_range': True,
                  'elb_init_t': 1., 'elb_max_t': 10., 'elb_mulcoef': 1.01,
                  'w_sparsity': False, 'w_sparsity_lambda': 1.,
                  'local_moments': False, 'local_moments_lambda': 1., 'local_moments_ksz': [3],
                  'hist': False, 'hist_lambda': 1., 'hist_sigma': 1e5, 'hist_metric': constants.NORM2,
                  'kde': False, 'kde_lambda': 1., 'kde_nbins': 256, 'kde_kde_bw': 1. / (255. ** 2),
                  'kde_metric': constants.NORM2,
                  'img_grad': False, 'img_grad_lambda': 1., 'img_grad_norm': constants.NORM2,
                  'norm_img_grad': False, 'norm_img_grad_lambda': 1., 'norm_img_grad_type': constants.NORM2,
                  'laplace': False, 'laplace_lambda': 1., 'laplace_norm': constants.NORM2,
                  'norm_laplace': False, 'norm_laplace_lambda': 1., 'norm_laplace_type': constants.NORM2,
                  'loc_var': False, 'loc_var_ksz': 3, 'loc_var_lambda': 1., 'loc_var_norm': constants.NORM2,
                  'norm_loc_var': False, 'norm_loc_var_ksz': 3, 'norm_loc_var_lambda': 1.,
                  'norm_loc_var_type': constants.NORM2,
                  'G_optimizer_type': constants.ADAM, 'G_optimizer_lr': 2e-4, 'G_optimizer_wd': 1e-4,
                  'G_optimizer_beta1': 0.9, 'G_optimizer_beta2': 0.999, 'G_optimizer_eps_adam': 1e-8,
                  'G_optimizer_momentum': 0.9, 'G_optimizer_nesterov': True,
                  # clip_grad_norm_ in front of the optimizer and the moving-average network netE (utils_config.py:153,159,183)
                  'G_optimizer_clipgrad': 0.0, 'E_decay': 0.0, 'E_param_strict': True,
                  'G_scheduler_type': constants.MYSTEPLR, 'G_scheduler_step_size': 30,
                  'G_scheduler_gamma': 0.5, 'G_scheduler_min_lr': 1e-4,
                  'G_scheduler_milestones': [250000, 400000],
                  # checkpoints (utils_config.py:160,185-193): optimizer state saved / resumed with the weights;
                  # validation and checkpoint periods in iterations (int) or as a fraction of an epoch (float < 1)
                  'G_optimizer_reuse': True, 'G_param_strict': True, 'checkpoint_eval': 5000, 'checkpoint_save': 5000,
                  'test_epoch_freq': 50},
    }


# reference flags that select folders, logging, plotting or launcher plumbing: accepted, no effect on the numbers of a run
IGNORED_FLAGS = {'debug_subfolder': str, 'exp_id': str, 'verbose': str2bool, 'fd_exp': str, 'num_workers': int,
                 'plot_epoch_freq': int, 'synch_scratch_epoch_freq': int, 'local_rank': int, 'local_world_size': int,
                 'init_method': str, 'world_size': int, 'is_train': str2bool,
                 'amp_eval': str2bool,      # consumed by the WSOL inference code only (inference_wsol.py:246), not by this task
                 'test_epoch_freq': int}
# reference flags that change what a run computes and are NOT implemented here: only the reference's default passes
# (utils_config.py:64-404), anything else is an error -- never a silently different run
DEFAULT_ONLY_FLAGS = {
    'G_regularizer_orthstep': (float, 0.0), 'G_regularizer_clipstep': (float, 0.0),        # model_plain.py:365-387
    'G_optimizer_amsgrad': (str2bool, False),
    'reconstruct_type': (str, 'low_res'), 'reconstruct_input': (str, 'fake'),
    'net_task': (str, 'regression'), 'ce': (str2bool, False), 'ce_lambda': (float, None),
    'augment': (str2bool, False), 'augment_nbr_steps': (int, None), 'augment_use_roi': (str2bool, None),
    'train_n': (float, 1.0),
}
_RESIDUAL_TERMS = ('l1', 'l2', 'l2sum', 'charbonnier', 'boundpred', 'local_moments', 'img_grad', 'norm_img_grad', 'laplace',
                   'norm_laplace', 'loc_var', 'norm_loc_var')
for _t in _RESIDUAL_TERMS:                                  # <term>_use_residuals (dlib/loss/core.py): the image path only
    DEFAULT_ONLY_FLAGS[f'{_t}_use_residuals'] = (str2bool, False)


def parse_input(argv=None):
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument('--net_type', type=str, default=constants.SWINIR)
    net_type = pre.parse_known_args(argv)[0].net_type          # parsed first (utils_parser.py:1333)
    if net_type not in constants.MODELS:
        raise NotImplementedError(f'--net_type {net_type}: libsrhip runs {constants.MODELS}')
    cfg = get_config(net_type)
    ap = argparse.ArgumentParser(description='SR-CACO-2 hot path on MI355X (libsrhip)',
                                 usage='main.py --net_type <net> --method <METHOD> [flags of the reference parser; -h lists them]')
    for k in ('task', 'net_type', 'method', 'dist_backend', 'cudaid', 'outd', 'train_dsets', 'valid_dsets',
              'test_dsets', 'data_root', 'splits_root', 'sample_tr_patch', 'sample_tr_patch_th_style',
              'save_dir_models', 'save_dir_imgs', 'basic_interpolation', 'model_select_mtr'):
        ap.add_argument(f'--{k}', type=str, default=None)
    for k in ('scale', 'n_channels', 'h_size', 'batch_size', 'eval_bsize', 'myseed', 'max_epochs', 'max_iters',
              'sample_tr_patch_th', 'valid_n_samples'):
        ap.add_argument(f'--{k}', type=int, default=None)
    for k in ('distributed', 'amp', 'eval_over_roi_also', 'eval_graph', 'train_graph', 'eval_over_roi_also_model_select',
              'use_interpolated_low', 'ppiw', 'da_blur', 'da_dot_bin_noise', 'da_add_gaus_noise'):
        ap.add_argument(f'--{k}', type=str2bool, default=None)
    for k in ('inter_low_th', 'inter_low_sigma', 'ppiw_min_per_col_w', 'da_blur_prob', 'da_blur_area', 'da_blur_sigma',
              'da_dot_bin_noise_prob', 'da_dot_bin_noise_area', 'da_dot_bin_noise_p', 'da_add_gaus_noise_prob',
              'da_add_gaus_noise_area', 'da_add_gaus_noise_std'):
        ap.add_argument(f'--{k}', type=float, default=None)
    ap.add_argument('--init_pretrained_path', type=str, default=None)        # netG['init_pretrained_path'] (utils_config.py:146)
    for k, t in IGNORED_FLAGS.items():
        if k not in cfg['train']:
            ap.add_argument(f'--{k}', type=t, default=None)
    for k, (t, _) in DEFAULT_ONLY_FLAGS.items():
        ap.add_argument(f'--{k}', type=t, default=None)
    for k, v in cfg['train'].items():
        t = str2bool if isinstance(v, bool) else (type(v) if not isinstance(v, list) else plus_list)
        if k in ('checkpoint_eval', 'checkpoint_save'):
            t = int_or_float
        ap.add_argument(f'--{k}', type=t, default=None)
    nt = safe_str_var(net_type)
    net_opts = {constants.SWINIR: {'window_size': int, 'depths': plus_list, 'embed_dim': int,
                                   'num_heads': plus_list, 'mlp_ratio': int, 'upsampler': str,
                                   'resi_connection': str, 'img_range': float},
                constants.VDSR: {},
                constants.SRCNN: {},
                constants.MSLAPSR: {},
                constants.MEMNET: {'num_memory_blocks': int, 'num_residual_blocks': int},
                constants.DRRN: {'num_residual_units': int},
                constants.DBPN: {'base_filter': int, 'feat': int, 'num_stages': int},
                constants.ENLCN: {'n_resblock': int, 'n_feats': int, 'res_scale': float},
                constants.DFCAN: {},
                constants.GRL: {'window_size': int, 'img_range': float, 'embed_dim': int, 'mlp_ratio': int},   # utils_parser.py:376-388
                constants.OMNISR: {'num_feat': int, 'res_num': int, 'window_size': int, 'block_num': int},
                constants.ACT: {'n_feats': int, 'n_resgroups': int, 'n_resblocks': int, 'reduction': int, 'n_heads': int,
                                'n_layers': int, 'n_fusionblocks': int, 'token_size': int, 'expansion_ratio': int},
                constants.NLSN: {'n_resblocks': int, 'n_feats': int, 'n_hashes': int, 'chunk_size': int,
                                 'res_scale': float},
                constants.SRFBN: {'num_features': int, 'num_steps': int, 'num_groups': int},
                constants.PROSR: {'num_init_features': int, 'bn_size': int, 'growth_rate': int, 'res_factor': float,
                                  'max_num_feature': int},
                constants.EDSR_LIIF: {'n_feats': int, 'n_resblocks': int, 'res_scale': float,
                                      'img_range': float}}[net_type]
    for k, t in net_opts.items():
        ap.add_argument(f'--{nt}_{k}', type=t, default=None)
    # <net>_init_type / _init_bn_type / _init_gain (select_network.py:285-289): the registry default 'default' keeps each
    # module's own initialisation, which is what define_G does here; another initialiser is not implemented
    init_defaults = {'init_type': (str, constants.INIT_W_DEFAULT), 'init_bn_type': (str, constants.INIT_BN_CONSTANT),
                     'init_gain': (float, 1.0)}
    for k, (t, _) in init_defaults.items():
        ap.add_argument(f'--{nt}_{k}', type=t, default=None)
    ns, unknown = ap.parse_known_args(argv)
    if unknown:
        # the reference's parser is argparse.parse_args(): an unknown flag ends the run (and a key missing from the config
        # raises ValueError, utils_parser.py:900-923) -- it is never skipped
        ap.error(f"unrecognized arguments: {' '.join(unknown)} (not a flag of the reference's parser, or of a network other "
                 f"than --net_type {net_type})")
    for k, (_, dflt) in list(DEFAULT_ONLY_FLAGS.items()) + [(f'{nt}_{k}', v) for k, v in init_defaults.items()]:
        v = getattr(ns, k)
        if v is not None and dflt is not None and v != dflt:
            ap.error(f"--{k} {v}: this option changes what the run computes and is not implemented on this path "
                     f"(only the reference's default {dflt!r} is accepted)")
    # any non-None CLI value overrides the top-level / nested key (utils_parser.py:900-923)
    skip = set(IGNORED_FLAGS) | set(DEFAULT_ONLY_FLAGS) | {f'{nt}_{k}' for k in init_defaults} | {'init_pretrained_path'}
    for k, v in vars(ns).items():
        if v is None or (k in skip and k not in cfg['train']):
            continue
        if k in cfg['train']:
            cfg['train'][k] = v
        elif not k.startswith(nt + '_'):
            cfg[k] = v
    if cfg['method'] != constants.NETTYPE_METHOD[net_type]:
        raise ValueError(f"--method {cfg['method']} does not match --net_type {net_type} "
                         f"({constants.NETTYPE_METHOD[net_type]})")
    # --amp True: evaluation (model.test / eval.py) runs the reduced-precision kernels; training stays fp32-accurate
    cfg['netG'] = init_net_g({'net_type': net_type}, cfg)
    if ns.init_pretrained_path is not None:
        cfg['netG']['init_pretrained_path'] = ns.init_pretrained_path
    for k in net_opts:
        v = getattr(ns, f'{nt}_{k}')
        if v is not None:
            cfg['netG'][f'{nt}_{k}'] = v
    return Dict2Obj(cfg)


def synth_batch(batch, scale, h_size, device, seed):
    """dataset_dpsr.py:685-710 on synthetic data: H uint8-quantised, L = clamp(bicubic_down(H)); 'l_to_h_img' = the LR
    patch brought to the HR size by cv2.resize(INTER_CUBIC) and clipped (:905-906; srhip_resize_cubic), what the
    SRCNN-style nets consume (model_plain.py:184-195)."""
    g = torch.Generator().manual_seed(seed)
    hr = (torch.rand(batch, 1, h_size, h_size, generator=g) * 255).round() / 255
    lr = F.interpolate(hr, scale_factor=1.0 / scale, mode='bicubic').clamp(0, 1)
    out = {'l_im': lr.to(device), 'h_im': hr.to(device)}
    if torch.device(device).type == 'cuda':
        from srhip import ops
        up = ops.clip01_(ops.resize_cubic(out['l_im'][:, 0].contiguous(), (h_size, h_size)))
        out['l_to_h_img'] = up[:, None]
        out['l_to_h_img_aug'] = out['l_to_h_img']
    return out


def _resume_point(args):
    """main.py:27-35 of the reference: the newest <iter>_G.pth / <iter>_optimizerG.pth under
    <outd_backup>/<save_dir_models> become the run's starting weights / optimizer state, and the larger label its
    iteration count.  Nothing there: iteration 0, weights from netG['init_pretrained_path'] if given."""
    from dlib.utils.utils_config import find_last_checkpoint
    models = os.path.join(args.outd_backup, args.save_dir_models)
    # The reference's parser gives every experiment a folder of its own; here --outd defaults to a shared './out'.  A folder
    # whose saved configuration (config_model.yml, written by a run over folds) names ANOTHER network is not this run's to
    # resume: say so instead of failing in load_state_dict or silently training zero iterations.
    cfg_path = os.path.join(args.outd_backup, 'config_model.yml')
    if os.path.isfile(cfg_path):
        import yaml
        try:
            old = yaml.safe_load(open(cfg_path)) or {}
        except Exception:                       # python-tagged values the safe loader refuses: compare the text
            old = {}
            txt = open(cfg_path).read()
            for key in ('net_type', 'scale'):
                import re
                m = re.search(rf'^{key}:\s*(\S+)', txt, re.M)
                if m:
                    old[key] = m.group(1)
        if 'net_type' not in old and isinstance(old.get('netG'), dict) and 'net_type' in old['netG']:
            old['net_type'] = old['netG']['net_type']           # (the reference's own files keep it under netG)
        for key in ('net_type', 'scale'):
            if key in old and str(old[key]) != str(getattr(args, key)):
                raise SystemExit(f"--outd {args.outd_backup} holds an experiment with {key} = {old[key]} (config_model.yml); this "
                                 f"run asks for {getattr(args, key)}: give it a folder of its own (--outd)")
    it_g, path_g = find_last_checkpoint(models, net_type='G',
                                        pretrained_path=args.netG.get('init_pretrained_path', '') or '')
    args.netG['checkpoint_path_netG'] = path_g
    it_o, path_o = find_last_checkpoint(models, net_type='optimizerG')
    args.netG['checkpoint_path_optimizerG'] = path_o
    if float(args.train.get('E_decay', 0.0) or 0.0) > 0:
        # the reference leaves netE's resume as a todo (main.py:32; checkpoint_path_netE stays ''): a resumed run would restart
        # its moving average from the current weights.  Here the newest <iter>_E.pth continues it.
        _, path_e = find_last_checkpoint(models, net_type='E')
        args.netG['checkpoint_path_netE'] = path_e
    return max(it_g, it_o)


def _train_on_folds(args, model, rank, world, current_step):
    """Real folds: the reference's epoch loop (utils_trainer.train_valid) over tiles resident in HBM."""
    import dlib.dllogger as DLLogger
    from dlib.utils.utils_dataloaders import get_train_set, get_all_eval_loaders
    from dlib.utils.utils_tracker import find_last_tracker
    from dlib.utils.utils_trainer import train_valid
    from dlib.utils.utils_config import save_config
    args.multi_valid = len([x for x in (args.valid_dsets or '').split(constants.SEP) if x]) > 1      # utils_parser.py:950
    DLLogger.init_arb(log_dir=args.outd_backup, is_master=args.is_master, reset=current_step == 0)
    if args.is_master:
        save_config(args, args.outd_backup, 'config_model.yml')                 # what eval.py reads (utils_parser.py:1397-1401)
    train_set = get_train_set(args, model.device, rank, world)
    # host-side draws of the input pipeline (edt / Otsu patch samplers, --da_* augmentations: numpy / random): one stream per
    # rank, as the reference's DataLoader workers have
    import random
    import numpy as np
    np.random.seed((int(args.myseed or 0) + rank) % (2 ** 32 - 1))
    random.seed(int(args.myseed or 0) + rank)
    n_valid = args.valid_n_samples if args.valid_n_samples else -1
    valid_loaders = get_all_eval_loaders(args, args.valid_dsets, n=n_valid) if args.valid_dsets else {}
    test_loaders = get_all_eval_loaders(args, args.test_dsets, n=-1) if args.test_dsets else {}
    tracker, roi_tracker = find_last_tracker(args.outd_backup, args)
    t0 = time.perf_counter()
    tracker, roi_tracker, last = train_valid(args, model, train_set, None, valid_loaders, test_loaders, tracker,
                                             roi_tracker, current_step)
    if args.is_master:
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        print(f'iterations {current_step + 1}..{last}: {(last - current_step) * args.batch_size * world / dt_s:.1f} '
              f'patches/s (validation and checkpoints included)')
        save_config(args, args.outd_backup, 'config_final.yml')


def main(argv=None):
    args = parse_input(argv)
    rank, world = 0, 1
    if args.distributed:
        import torch.distributed as dist
        local = int(os.environ.get('LOCAL_RANK', '0'))          # torchrun contract (utils_parser.py:1087)
        torch.cuda.set_device(local)
        dist.init_process_group(args.dist_backend)
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        torch.cuda.set_device(int(str(args.cudaid).split(',')[0]))
    args.is_master = rank == 0
    args.outd_backup = args.outd                                # utils_parser.py:1042 (no compute-cluster scratch here)
    args.is_train = True
    os.makedirs(os.path.join(args.outd_backup, args.save_dir_models), exist_ok=True)
    current_step = _resume_point(args)
    torch.manual_seed(args.myseed)
    from dlib.models.select_model import define_model
    from dlib import metrics
    model = define_model(args)
    model.init_train()      # loads the checkpoint pair found above (model_plain.py:54-66)
    # weights: one seed, and rank 0's replace the others' anyway (TrainStep, as DDP's constructor)
    if rank == 0:
        print(model.info_network())
        if current_step:
            print(f'[libsrhip] RESUMING at iteration {current_step} from {args.netG["checkpoint_path_netG"]} '
                  f'(newest checkpoint under {os.path.join(args.outd_backup, args.save_dir_models)}; --outd selects the folder)',
                  flush=True)
    if args.train_dsets:
        _train_on_folds(args, model, rank, world, current_step)
        if args.distributed:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return
    # no folds given: the same step on one synthetic batch of the configured shape (throughput / plumbing runs)
    from dlib.utils.utils_config import clean_previous_checkpoints_except_last
    batch = synth_batch(args.batch_size, args.scale, args.h_size, model.device, 1000 + rank)
    n_save = args.train['checkpoint_save']
    t0, seen = time.perf_counter(), 0
    for step in range(current_step + 1, args.max_iters + 1):
        # the reference re-seeds EVERY rank with myseed + current_step at each iteration (utils_trainer.py:359-361):
        # DropPath masks are a function of (seed, step), identical across ranks, and a resumed run repeats them
        torch.manual_seed((args.myseed + step) % (2 ** 32 - 1))
        model.feed_data(batch)
        model.optimize_parameters(epoch=0, current_step=step)
        model.update_learning_rate()
        seen += args.batch_size * world
        if step % 10 == 0 or step == args.max_iters:
            if not model.check_finite():
                print('Terminated due to error: non-finite loss')       # tools.py:55-63 semantics
                sys.exit(1)
            if rank == 0:
                torch.cuda.synchronize()
                log = model.current_log()
                print(f"iter {step:6d}  G_loss {log['G_loss']:.6f}  lr {model.current_learning_rate():.2e}  "
                      f"{seen / (time.perf_counter() - t0):8.1f} patches/s")
        if isinstance(n_save, int) and step % n_save == 0 and rank == 0 and step != args.max_iters:
            model.save(step)                                            # utils_trainer.py:403-411
            clean_previous_checkpoints_except_last(model.save_dir, ['G', 'optimizerG'] + (['E'] if model.E_decay > 0 else []))
    # evaluation sweep (utils_trainer.py:961-1032): PSNR / PSNR_Y / MSE / NRMSE / SSIM, optional ROI thresholds
    from dlib.utils.utils_trainer import _forward_with_padding
    model = _forward_with_padding(batch, model, args)          # SwinIR: flipped-strip padding to the next window multiple
    vis = model.current_visuals()
    ths = tuple(args.eval_over_roi_also_ths) if args.eval_over_roi_also else ()
    sw = metrics.sweep(vis['E'], vis['H'], border=args.scale, thresholds=ths)
    if rank == 0:
        for k in (constants.PSNR_MTR, constants.PSNR_Y_MTR, constants.MSE_MTR, constants.NRMSE_MTR,
                  constants.SSIM_MTR):
            full = sw[k][:, 0].double().mean().item()
            msg = f'{k:8s} {full:.4f}'
            if ths:
                msg += f'   ROI(avg over th {list(ths)}) {sw[k][:, 1:].double().mean().item():.4f}'
            print(msg)
        if args.max_iters > current_step:
            print('saved', model.save(args.max_iters))                  # <iter>_G.pth + <iter>_optimizerG.pth
            clean_previous_checkpoints_except_last(model.save_dir, ['G', 'optimizerG'] + (['E'] if model.E_decay > 0 else []))
    if args.distributed:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
