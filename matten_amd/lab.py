"""
ctypes binding of libmatten_lab.so (include/matten_lab.h; `make -C matten_amd/csrc lab`): the calibration kernels of the
benchmark harness.  Not part of the product -- nothing under matten_amd/ but this file loads it, and only bench.py and
the tools call it.  `load()` returns None when the library has not been built.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_int, c_int64, c_void_p

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmatten_lab.so")
P = c_void_p
SIGNATURES = {
    "matten_calib_valu_insts_per_simd": (c_int64, [c_int64]),
    "matten_calib_valu": (c_int, [c_int64, P, P, P]),
    "matten_calib_copy": (c_int, [P, P, c_int64, P]),
    "matten_calib_clock_probe": (c_int, [c_int64, P, P]),
}
_lib = None


def load():
    global _lib
    if _lib is None and os.path.exists(LIB_PATH):
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib
