"""Learning-rate rules of the reference (dlib/learning/lr_scheduler.py:6-94) as
torch schedulers, for code that drives a torch optimizer; the fused training
step (srhip.train.Optimizer) evaluates the same closed forms directly."""
import math

from torch.optim.lr_scheduler import LRScheduler


class MyStepLR(LRScheduler):
    """lr = max(base_lr * gamma ** (epoch // step_size), min_lr)."""

    def __init__(self, optimizer, step_size, gamma=0.1, last_epoch=-1, min_lr=1e-6):
        self.step_size, self.gamma, self.min_lr = step_size, gamma, min_lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        k = self.last_epoch // self.step_size
        return [max(b * self.gamma ** k, self.min_lr) for b in self.base_lrs]


class MyCosineLR(LRScheduler):
    """lr = max(base_lr * coef * (1 + cos((epoch - 1) * pi / max_epochs)), min_lr)."""

    def __init__(self, optimizer, coef, max_epochs, min_lr=1e-9, last_epoch=-1):
        assert isinstance(coef, float) and coef > 0 and max_epochs > 0
        self.coef, self.max_epochs, self.min_lr = coef, float(max_epochs), min_lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        c = self.coef * (1. + math.cos((self.last_epoch - 1) * math.pi / self.max_epochs))
        return [max(b * c, self.min_lr) for b in self.base_lrs]
