"""Extended log-barrier module (reference dlib/loss/elb.py == dlib/losses/elb.py:15-125): holds the
barrier parameter ``t`` and its schedule ``t <- min(t * mulcoef, max_t)``.  The barrier itself is
evaluated inside the fused loss kernels (``srhip_loss_bounded``), which take ``t`` as an argument.

The reference carries TWO copies of this class, ``dlib.loss.elb.ELB`` (what ``ElementaryLoss.update_t``
tests for, core.py:12,80-82) and ``dlib.losses.elb.ELB`` (what ``utils_instance.py:16,44`` builds and
what ``BoundedPrediction`` asserts, main.py:15,193).  They are distinct classes, so a term built by
``define_loss`` never has its ``t`` advanced by ``MasterLoss.update_t()``.  Both are kept here, distinct,
so that the behaviour carries over unchanged."""
import torch
import torch.nn as nn

__all__ = ['ELB']


class _ELBState(nn.Module):
    def __init__(self, init_t=1., max_t=10., mulcoef=1.01):
        super().__init__()
        assert isinstance(mulcoef, float) and mulcoef > 0.
        assert isinstance(init_t, float) and init_t > 0.
        assert isinstance(max_t, float) and max_t > init_t
        self.init_t = init_t
        # float32 state, as the reference's registered buffers (the schedule rounds in f32)
        self.register_buffer("mulcoef", torch.tensor([mulcoef]).float())
        self.register_buffer("t_lb", torch.tensor([init_t]).float())
        self.register_buffer("max_t", torch.tensor([max_t]).float())

    def set_t(self, val):
        assert isinstance(val, float) or (isinstance(val, torch.Tensor) and val.ndim == 1
                                          and val.dtype == torch.float)
        assert val > 0.
        if isinstance(val, float):
            val = torch.tensor([val])
        self.register_buffer("t_lb", val.float().detach().to(self.t_lb.device))

    def get_t(self):
        return self.t_lb

    def update_t(self):
        self.set_t(torch.min(self.t_lb * self.mulcoef, self.max_t))

    def forward(self, fx):
        raise RuntimeError("ELB is evaluated inside the fused libsrhip loss kernels (dlib.loss.BoundedPrediction); "
                           "there is no stand-alone / CPU path")

    def __str__(self):
        return "{}(): ELB method.".format(self.__class__.__name__)


class ELB(_ELBState):
    """dlib/loss/elb.py:15"""
