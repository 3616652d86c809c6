"""``dlib.losses.elb.ELB`` (reference dlib/losses/elb.py:15-125): the copy of the class that
``utils_instance.py:16`` and ``dlib/loss/main.py:15`` import.  Deliberately NOT the same class object as
``dlib.loss.elb.ELB`` -- see the note there."""
from dlib.loss.elb import _ELBState

__all__ = ['ELB']


class ELB(_ELBState):
    """dlib/losses/elb.py:15"""
