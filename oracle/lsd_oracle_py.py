"""ctypes binding of oracle/lsd_oracle.cpp (test infrastructure only): the LSD branch of the reference's segment producer."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "liboracle_lsd.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "lsd_oracle.cpp")):
            subprocess.check_call(["make", "-s", "-C", HERE, "liboracle_lsd.so"])
        _lib = C.CDLL(so)
        _lib.lsd_oracle_detect.restype = C.c_int
        _lib.lsd_oracle_detect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int]
    return _lib


def detect_filter_lines(gray, length_thres=15.0, cap=20000):
    """line_lbd_detect::detect_filter_lines with use_LSD = true on an 8-bit gray image: (n, 4) float32 x1 y1 x2 y2."""
    g = np.ascontiguousarray(gray, np.uint8)
    out = np.zeros((cap, 4), np.float32)
    n = lib().lsd_oracle_detect(g.ctypes.data, g.shape[1], g.shape[0], float(length_thres), out.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("lsd oracle: more than %d segments" % cap)
    return out[:n].copy()
