"""Registry names used on the SR hot path.  Values equal the reference's
(dlib/utils/constants.py:32-51,91-96,112-128,139-185,820-828) so that configs,
CLI flags and checkpoints interchange."""
SUPER_RES = 'super-resolution'

SWINIR = 'swinir'
EDSR_LIIF = 'EDSR_LIIF'
VDSR = 'VDSR'  # https://arxiv.org/pdf/1511.04587.pdf (reference constants.py:27)
DRRN = 'DRRN'  # https://ieeexplore.ieee.org/document/8099781 (reference constants.py:29)
SRCNN = 'SRCNN'  # https://arxiv.org/abs/1501.00092 (reference constants.py)
MSLAPSR = 'MSLapSRN'  # https://arxiv.org/pdf/1710.01992.pdf (reference constants.py:47)
MEMNET = 'MemNet'  # https://arxiv.org/pdf/1708.02209.pdf (reference constants.py:28)
DBPN = 'DBPN'  # https://arxiv.org/pdf/1803.02735.pdf (reference constants.py:49)
SRFBN = 'SRFBN'  # https://arxiv.org/pdf/1903.09814.pdf (reference constants.py:46)
PROSR = 'ProSR'  # https://arxiv.org/pdf/1804.02900.pdf (reference constants.py:48)
ENLCN = 'ENLCN'  # https://arxiv.org/pdf/2201.03794.pdf (reference constants.py:36)
NLSN = 'NLSN'  # Mei et al., CVPR 2021 (reference constants.py:38)
DFCAN = 'DFCAN'  # https://www.nature.com/articles/s41592-020-01048-5 (reference constants.py:33)
ACT = 'ACT'  # https://arxiv.org/pdf/2203.07682.pdf (reference constants.py:37)
OMNISR = 'OmniSR'  # https://arxiv.org/pdf/2304.10244.pdf (reference constants.py:34)
GRL = 'GRL'  # https://arxiv.org/pdf/2303.00748.pdf (reference constants.py:35)
MODELS = [SWINIR, EDSR_LIIF, VDSR, DRRN, SRCNN, MSLAPSR, MEMNET, DBPN, SRFBN, PROSR, ENLCN, NLSN, DFCAN, ACT, OMNISR, GRL]

SWINIR_MTH = 'SWINIR'
EDSR_LIIF_MTH = 'EDSR_LIIF'
VDSR_MTH = 'VDSR'
DRRN_MTH = 'DRRN'
SRCNN_MTH = 'SRCNN'
MSLAPSR_MTH = 'MSLAPSR'
MEMNET_MTH = 'MemNet'
DBPN_MTH = 'DBPN'
SRFBN_MTH = 'SRFBN'
PROSR_MTH = 'PROSR'
ENLCN_MTH = 'ENLCN'
NLSN_MTH = 'NLSN'
DFCAN_MTH = 'DFCAN'
ACT_MTH = 'ACT'
OMNISR_MTH = 'OmniSR'
GRL_MTH = 'GRL'
NETTYPE_METHOD = {SWINIR: SWINIR_MTH, EDSR_LIIF: EDSR_LIIF_MTH, VDSR: VDSR_MTH, DRRN: DRRN_MTH, SRCNN: SRCNN_MTH,
                  MSLAPSR: MSLAPSR_MTH, MEMNET: MEMNET_MTH, DBPN: DBPN_MTH, SRFBN: SRFBN_MTH, PROSR: PROSR_MTH,
                  ENLCN: ENLCN_MTH, NLSN: NLSN_MTH, DFCAN: DFCAN_MTH, ACT: ACT_MTH,
                  OMNISR: OMNISR_MTH, GRL: GRL_MTH}

US_PIXEL_SHUFFLE = 'pixelshuffle'
US_PIXEL_SHUFFLE_DIRECT = 'pixelshuffledirect'
US_NEAREST_CONV = 'nearest_conv'        # reference constants.py:93
R_CONNECTION_1CONV = '1conv'
R_CONNECTION_3CONV = '3conv'

INIT_W_DEFAULT = 'init_w_default'
INIT_BN_CONSTANT = 'init_bn_constant'

PSNR_MTR = 'psnr'
SSIM_MTR = 'ssim'
MSE_MTR = 'mse'
NRMSE_MTR = 'nrmse'
PSNR_Y_MTR = 'psnr_y'
SSIM_Y_MTR = 'ssim_y'      # listed by the reference (constants.py:118-131), never computed by its evaluation
METRICS = [PSNR_MTR, SSIM_MTR, MSE_MTR, NRMSE_MTR, PSNR_Y_MTR, SSIM_Y_MTR]
BEST_MTR = {PSNR_MTR: max, SSIM_MTR: max, MSE_MTR: min, NRMSE_MTR: min, PSNR_Y_MTR: max, SSIM_Y_MTR: max}

# splits, tracker periods, fold files (reference constants.py:100-136,193,713)
TRAIN_PHASE, EVAL_PHASE = 'train', 'eval'
TRAINSET, VALIDSET, TESTSET = 'train', 'val', 'test'
SPLITS = [TRAINSET, VALIDSET, TESTSET]
PR_EPOCH, PR_ITER = 'period_epoch', 'period_iter'
PERIODS = [PR_ITER, PR_EPOCH]
SEP = '+'
CODE_IDENTIFIER = 'CODEXXXXXXXIDENTIFIER'

SGD = 'sgd'
ADAM = 'adam'
OPTIMIZERS = [SGD, ADAM]
MULTISTEPLR = 'MultiStepLR'
MYSTEPLR = 'MyStepLR'
STEPSLR = [MULTISTEPLR, MYSTEPLR]

GLOO = 'gloo'
NCCL = 'nccl'

# norms of the optional local-variation loss terms (reference constants.py:696-703)
NORM1 = '1'
NORM2 = '2'
LPNORMS = [NORM1, NORM2]

# tasks / interpolation modes of the Bicubic baseline row (reference constants.py:1-6,218-220)
RECONSTRUCT = 'reconstruct'
TASKS = [SUPER_RES, RECONSTRUCT]
INTER_BICUBIC = 'bicubic'
INTERPOLATION_MODES = [INTER_BICUBIC]
