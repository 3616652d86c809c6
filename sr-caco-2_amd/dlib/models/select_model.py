"""reference dlib/models/select_model.py:16-29."""


def define_model(args):
    from dlib.models.model_plain import ModelPlain
    return ModelPlain(args)
