"""Small host helpers with the reference's names (dlib/utils/tools.py)."""


class Dict2Obj(dict):
    """args container: attribute access over a dict (reference tools.py:Dict2Obj; nested dicts such as
    args.netG / args.train stay plain dicts, as the reference code indexes them)."""
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v
