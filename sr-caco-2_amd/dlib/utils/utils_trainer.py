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


class Interpolate(torch.nn.Module):
    def __init__(self, task: str, scale: int, scale_mode: str):
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError("Interpolate (libsrhip build) runs on the GPU, like the reference "
                               "(utils_trainer.py:93); there is no CPU path")
        self.device = torch.device(f'cuda:{torch.cuda.current_device()}')
        self.scale: int = scale
        assert task in constants.TASKS, f"{task} | {constants.TASKS}"
        self.task = task
        assert scale_mode in [constants.INTER_BICUBIC], scale_mode
        self.scale_mode: str = scale_mode
        self.L = self.E = self.H = None

    def feed_data(self, data, need_H=True):
        if self.task == constants.SUPER_RES:
            lk, hk = 'l_im', 'h_im'
        elif self.task == constants.RECONSTRUCT:
            lk, hk = 'in_reconstruct', 'trg_reconstruct'
        else:
            raise NotImplementedError(self.task)
        self.L = data[lk].to(self.device)
        if need_H:
            self.H = data[hk].to(self.device)

    def forward(self):
        x = self.L
        assert x.ndim == 4, x.ndim
        if self.scale_mode != constants.INTER_BICUBIC:
            raise NotImplementedError(f'Not supported : {self.scale_mode}')
        if self.task == constants.SUPER_RES:
            scale = self.scale
        elif self.task == constants.RECONSTRUCT:
            scale = 1
        else:
            raise NotImplementedError(self.task)
        out = F.interpolate(input=x, scale_factor=scale, mode='bicubic', antialias=True)
        self.E = torch.clamp(out, 0.0, 1.0)       # data in [0, 1] (utils_trainer.py:146-147)

    def set_eval_mode(self):
        self.eval()

    def set_train_mode(self):
        pass

    def test(self):
        self.eval()
        with torch.no_grad():
            self.forward()

    def current_visuals(self, need_H=True):
        out = {'L': self.L.detach().float(), 'E': self.E.detach().float()}
        if need_H:
            out['H'] = self.H.detach().float()
        return out
