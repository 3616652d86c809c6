"""Metric entry points under their reference module path
(dlib/utils/utils_image.py); implementation in dlib.metrics."""
from dlib.metrics import (tensor2uint82float, mbatch_gpu_calculate_psnr,  # noqa: F401
                          mbatch_gpu_calculate_mse, mbatch_gpu_calculate_nrmse,
                          mbatch_gpu_calculate_ssim)
