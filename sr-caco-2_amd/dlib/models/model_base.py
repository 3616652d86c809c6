"""The reference splits the wrapper in ModelBase + ModelPlain
(dlib/models/model_base.py:26-211); here the protocol lives in one class."""
from dlib.models.model_plain import ModelPlain as ModelBase  # noqa: F401
