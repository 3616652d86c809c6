import re


def safe_str_var(x: str) -> str:
    """Option-name prefix of a net type (reference: dlib/utils/shared.py:272)."""
    out = re.sub('[^0-9a-zA-Z_]', '_', x)
    return '_' + out if out[0].isdigit() else out
