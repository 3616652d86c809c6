"""ctypes binding of libsrhip.so (the C-ABI declared in include/srhip.h).

The library is the product: there is no CPU or PyTorch fallback.  If it is
missing or fails to load, importing this module raises."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# SRHIP_LIB: another build of the same library (same-box A/B of two builds, tools/ab_lib.sh)
LIB_PATH = os.environ.get("SRHIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libsrhip.so")

HEADER = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "srhip.h")
_CT = {"p": ctypes.c_void_p, "l": ctypes.c_long, "i": ctypes.c_int,
       "f": ctypes.c_float, "d": ctypes.c_double}


def parse_header(path=HEADER):
    """include/srhip.h is the single source of truth: returns
    {name: (return_code, arg_codes)} with p pointer, l long, i int, f float,
    d double, s const char*."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char\*|int|long)\s+(srhip_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        codes = ""
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    codes += "p"
                else:
                    codes += {"long": "l", "int": "i", "float": "f", "double": "d"}[a.split()[0]]
        protos[name] = ({"const char*": "s", "int": "i", "long": "l"}[ret], codes)
    return protos


def _load():
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"libsrhip.so not found at {LIB_PATH}: build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` or "
            f"`make -C sr-caco-2_amd/csrc` (hipcc, --offload-arch=gfx950). "
            f"There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (ret, codes) in parse_header().items():
        fn = getattr(lib, name)           # AttributeError if a symbol is missing
        fn.restype = {"s": ctypes.c_char_p, "i": ctypes.c_int, "l": ctypes.c_long}[ret]
        fn.argtypes = [_CT[c] for c in codes]
    return lib


lib = _load()


class SrhipError(RuntimeError):
    pass


def call(name, *args):
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise SrhipError(f"{name} failed ({rc}): {lib.srhip_last_error().decode()}")
