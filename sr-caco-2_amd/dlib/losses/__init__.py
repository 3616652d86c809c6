"""``north_star`` names the SR loss surface ``dlib.losses``; in the reference the
SR losses live in ``dlib.loss`` (singular) and ``dlib.losses`` holds the WSOL
legacy.  Both names resolve to the same classes."""
from dlib.loss import *  # noqa: F401,F403
from dlib.loss import __all__  # noqa: F401
