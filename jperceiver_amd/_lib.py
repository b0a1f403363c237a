"""ctypes binding of libjperceiver_hip.so (the C ABI in include/jperceiver_hip.h).

There is no fallback: if the shared library is missing or a kernel call fails, an exception is
raised.  Prototypes are parsed from the header so the Python side can never drift from the ABI.
"""
from __future__ import annotations

import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("JP_LIB_PATH") or os.path.join(_HERE, "csrc", "libjperceiver_hip.so")   # override: kernel A/B builds
HEADER_PATH = os.path.join(_ROOT, "include", "jperceiver_hip.h")

_CT = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
    "uint64_t": ctypes.c_uint64,
}


_PTR_DTYPE = {"float*": torch.float32, "double*": torch.float64, "int*": torch.int32, "int64_t*": torch.int64,
              "long*": torch.int64, "uint8_t*": torch.uint8}


class JPerceiverHipError(RuntimeError):
    pass


def parse_header(path: str = HEADER_PATH):
    """-> {name: (restype_str, [(ctype_str, argname), ...])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"^((?:const )?\w+\*?) (jp_\w+)\(([^)]*)\);", src, flags=re.M):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        alist = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a)
                alist.append((mm.group(1).strip(), mm.group(2)))
        protos[name] = (ret, alist)
    return protos


def _ctype(tstr: str):
    if tstr.endswith("*"):
        return ctypes.c_void_p
    return _CT[tstr]


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise JPerceiverHipError(
                f"{LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the product path has no CPU/PyTorch fallback)")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self.fn = {}
        for name, (ret, args) in self.protos.items():
            f = getattr(self.cdll, name)   # AttributeError if the .so lacks a declared symbol
            f.argtypes = [_ctype(t) for t, _ in args]
            f.restype = (ctypes.c_char_p if ret.startswith("const char") else None if ret == "void"
                         else ctypes.c_long if ret == "long" else ctypes.c_int)
            self.fn[name] = f
        self.ptr_args = {name: [t.endswith("*") for t, _ in args] for name, (ret, args) in self.protos.items()}
        # element type every pointer argument must have (None = untyped: void* scratch / stream)
        self.ptr_dtypes = {name: [_PTR_DTYPE.get(t.replace("const ", "").strip()) if t.endswith("*") else None
                                  for t, _ in args] for name, (ret, args) in self.protos.items()}

    def last_error(self) -> str:
        s = self.fn["jp_last_error_string"]()
        return s.decode() if s else ""


_LIB = None


def lib() -> _Lib:
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB


def call(name: str, *args):
    """Invoke a jp_* kernel launcher.  Tensor arguments are passed as raw device pointers (must be contiguous, of the
    element type the ABI declares, and all on ONE device); None -> NULL.  The trailing `stream` argument is appended
    automatically: torch's current stream OF THAT DEVICE (not of the process-wide current device)."""
    L = lib()
    f = L.fn[name]
    flags = L.ptr_args[name]
    dtypes = L.ptr_dtypes[name]
    conv = []
    n_user = len(flags) - 1  # last is the stream
    if len(args) != n_user:
        raise TypeError(f"{name}: expected {n_user} arguments, got {len(args)}")
    dev = None
    for a, is_ptr, dt in zip(args, flags, dtypes):
        if is_ptr:
            if a is None:
                conv.append(None)
            elif isinstance(a, torch.Tensor):
                if not a.is_cuda:
                    raise JPerceiverHipError(f"{name}: tensor argument is not on the GPU")
                if not a.is_contiguous():
                    raise JPerceiverHipError(f"{name}: tensor argument is not contiguous")
                if dt is not None and a.dtype != dt:
                    raise JPerceiverHipError(f"{name}: tensor argument has dtype {a.dtype}, the ABI expects {dt}")
                if dev is None:
                    dev = a.device
                elif a.device != dev:
                    raise JPerceiverHipError(f"{name}: tensor arguments live on different devices ({dev} vs {a.device})")
                conv.append(a.data_ptr())
            else:
                conv.append(int(a))
        else:
            conv.append(a)
    conv.append(torch.cuda.current_stream(dev).cuda_stream)
    rc = f(*conv)
    if rc != 0:
        raise JPerceiverHipError(f"{name} failed (rc={rc}): {L.last_error()}")
