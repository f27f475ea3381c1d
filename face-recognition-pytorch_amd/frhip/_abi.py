"""ctypes binding of libfrhip.so.  Prototypes are parsed from include/frhip.h so the header is the single
source of truth.  There is NO fallback: a missing library or symbol raises."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HEADER = os.path.join(ROOT, "include", "frhip.h")
LIB_PATH = os.environ.get("FRHIP_LIB_PATH") or os.path.join(HERE, "libfrhip.so")   # override: kernel-variant experiments

DT_BF16, DT_F32 = 0, 1


class FrhipError(RuntimeError):
    pass


def _ctype(decl):
    d = decl.strip()
    if "*" in d:
        return ctypes.c_void_p
    base = d.split()[0] if d.split() else d
    if d.startswith("frhip_stream_t"):
        return ctypes.c_void_p
    if base == "int":
        return ctypes.c_int
    if d.startswith("long long"):
        return ctypes.c_longlong
    if base == "float":
        return ctypes.c_float
    if base == "double":
        return ctypes.c_double
    if base == "size_t":
        return ctypes.c_size_t
    raise ValueError("frhip.h: cannot map parameter %r" % decl)


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function declared in frhip.h"""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int)\s+(frhip_\w+)\s*\(([^)]*)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = " ".join(args.split())
        argtypes = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        protos[name] = (ctypes.c_char_p if "char" in ret else ctypes.c_int, argtypes)
    return protos


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise FrhipError("libfrhip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                             "there is no CPU fallback for the HIP path" % LIB_PATH)
        # torch first: its wheel bundles the HIP runtime (libamdhip64); loading libfrhip.so before it would bind this
        # library to the system copy and the process would end up with two runtimes, one of which sees no device
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in parse_header().items():
            fn = getattr(handle, name)          # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _LIB = handle
    return _LIB


def check(rc, what=""):
    if rc != 0:
        msg = lib().frhip_last_error()
        raise FrhipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
