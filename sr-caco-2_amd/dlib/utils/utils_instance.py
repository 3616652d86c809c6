"""Loss / optimizer / scheduler factories (reference dlib/utils/utils_instance.py:23-290)
for the options that are on by default or in the README recipe."""
from dlib import loss as losses
from dlib.utils import constants


def define_loss(args):
    dev = getattr(args, 'c_cudaid', None)
    dev = 0 if dev is None else dev
    m = losses.MasterLoss(cuda_id=dev)
    tr = args.train
    if tr.get('l1', False):
        m.add(losses.L1(cuda_id=dev, lambda_=tr.get('l1_lambda', 1.)))
    if tr.get('l2', False):
        m.add(losses.L2(cuda_id=dev, lambda_=tr.get('l2_lambda', 1.)))
    # same order of addition as the reference (utils_instance.py:47-165): it is the order of l_holder / n_holder
    if tr.get('l2sum', False):
        m.add(losses.L2Sum(cuda_id=dev, lambda_=tr.get('l2sum_lambda', 1.)))
    if tr.get('ssim', False):
        l = losses.NegativeSsim(cuda_id=dev, lambda_=tr.get('ssim_lambda', 1.))
        l.set_window_size(tr.get('ssim_window_s', 11))
        m.add(l)
    if tr.get('local_moments', False):            # utils_instance.py:77-86
        assert not tr.get('local_moments_use_residuals', False), "use_residuals is not on the hot path"
        l = losses.LocalMoments(cuda_id=dev, lambda_=tr.get('local_moments_lambda', 1.))
        l.set_ksz(list(tr.get('local_moments_ksz', [3])))
        m.add(l)
    # optional terms (keys of utils_config.py:296-357)
    if tr.get('charbonnier', False):
        l = losses.Charbonnier(cuda_id=dev, lambda_=tr.get('charbonnier_lambda', 1.))
        l.set_eps(tr.get('charbonnier_eps', 1e-9))
        m.add(l)
    def new_elb():                                # utils_instance.py:16,44-45 (every term gets its own deepcopy)
        from dlib.losses.elb import ELB
        return ELB(init_t=float(tr.get('elb_init_t', 1.)), max_t=float(tr.get('elb_max_t', 10.)),
                   mulcoef=float(tr.get('elb_mulcoef', 1.01)))
    if tr.get('boundpred', False):
        elb = new_elb()
        assert not tr.get('boundpred_use_residuals', False), "use_residuals is not on the hot path"
        l = losses.BoundedPrediction(cuda_id=dev, lambda_=tr.get('boundpred_lambda', 1.), elb=elb,
                                     restore_range=tr.get('boundpred_restore_range', True),
                                     color_max=int(getattr(args, 'color_max', None) or 255))
        l.set_eps(float(tr.get('boundpred_eps', 1.)))
        m.add(l)
    for key, cls, norm_key in (('img_grad', losses.ImageGradientLoss, 'img_grad_norm'),
                               ('norm_img_grad', losses.NormImageGradientLoss, 'norm_img_grad_type'),
                               ('laplace', losses.LaplacianFilterLoss, 'laplace_norm'),
                               ('norm_laplace', losses.NormLaplacianFilterLoss, 'norm_laplace_type'),
                               ('loc_var', losses.LocalVariationLoss, 'loc_var_norm'),
                               ('norm_loc_var', losses.NormLocalVariationLoss, 'norm_loc_var_type')):
        if tr.get(key, False):
            assert not tr.get(key + '_use_residuals', False), "use_residuals is not on the hot path"
            l = cls(cuda_id=dev, lambda_=tr.get(key + '_lambda', 1.))
            if key.endswith('loc_var'):
                l.set_it(ksz=tr.get(key + '_ksz', 3), norm_str=str(tr.get(norm_key, constants.NORM2)))
            else:
                l.set_it(norm_str=str(tr.get(norm_key, constants.NORM2)))
            m.add(l)
    if tr.get('hist', False):                     # utils_instance.py:168-177
        l = losses.HistogramMatch(cuda_id=dev, lambda_=tr.get('hist_lambda', 1.), elb=new_elb(), color_min=0,
                                  color_max=255)
        l.set_it(norm_str=str(tr.get('hist_metric', constants.NORM2)), sigma=float(tr.get('hist_sigma', 1e5)))
        m.add(l)
    if tr.get('kde', False):                      # utils_instance.py:180-190
        l = losses.KDEMatch(cuda_id=dev, lambda_=tr.get('kde_lambda', 1.), elb=new_elb(), color_min=0, color_max=1)
        l.set_it(norm_str=str(tr.get('kde_metric', constants.NORM2)), kde_bw=float(tr.get('kde_kde_bw', 1. / 255. ** 2)),
                 ndim=int(getattr(args, 'n_channels', None) or 1), nbins=int(tr.get('kde_nbins', 256)))
        m.add(l)
    for k in ('ce',):
        if tr.get(k, False):
            raise NotImplementedError(f"loss term --{k} is outside the libsrhip hot path")
    if tr.get('w_sparsity', False):               # last, as in the reference (utils_instance.py:202-208)
        m.add(losses.WeightsSparsityLoss(cuda_id=dev, lambda_=tr.get('w_sparsity_lambda', 1.)))
    assert len(m.n_holder) > 1, "no loss term enabled"
    return m


def optimizer_config(args):
    """dict for srhip.train.Optimizer from the reference's G_optimizer_* / G_scheduler_* keys."""
    tr = args.train
    kind = tr['G_optimizer_type']
    assert kind in constants.OPTIMIZERS, kind
    cfg = dict(kind=kind, lr=tr['G_optimizer_lr'], wd=tr.get('G_optimizer_wd', 0.0))
    if kind == constants.ADAM:
        assert not tr.get('G_optimizer_amsgrad', False), "amsgrad is not on the hot path"
        cfg.update(betas=(tr.get('G_optimizer_beta1', 0.9), tr.get('G_optimizer_beta2', 0.999)),
                   eps=tr.get('G_optimizer_eps_adam', 1e-8))
    else:
        cfg.update(momentum=tr.get('G_optimizer_momentum', 0.9),
                   nesterov=tr.get('G_optimizer_nesterov', True))
    name = tr['G_scheduler_type']
    assert name in constants.STEPSLR, name
    if name == constants.MYSTEPLR:
        cfg['scheduler'] = dict(type=name, step_size=tr['G_scheduler_step_size'],
                                gamma=tr['G_scheduler_gamma'], min_lr=tr.get('G_scheduler_min_lr', 1e-4))
    else:
        cfg['scheduler'] = dict(type=name, milestones=tr['G_scheduler_milestones'],
                                gamma=tr['G_scheduler_gamma'])
    return cfg
