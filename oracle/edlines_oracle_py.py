"""ctypes binding of oracle/edlines_oracle.cpp (test infrastructure only): the reference's EDLines segment producer."""
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
        so = os.path.join(HERE, "liboracle_edlines.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "edlines_oracle.cpp")):
            subprocess.check_call(["make", "-s", "-C", HERE, "liboracle_edlines.so"])
        _lib = C.CDLL(so)
    return _lib


def blur(gray):
    g = np.ascontiguousarray(gray, np.uint8)
    out = np.zeros_like(g)
    lib().oracle_edlines_blur(g.ctypes.data_as(C.POINTER(C.c_ubyte)), g.shape[1], g.shape[0], out.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return out


def detect_filter_lines(gray, length_thres=15.0, cap=20000):
    """line_lbd_detect::detect_filter_lines (EDLines, one octave) on an 8-bit gray image: (n, 4) float32 x1 y1 x2 y2."""
    g = np.ascontiguousarray(gray, np.uint8)
    out = np.zeros((cap, 4), np.float32)
    n = lib().oracle_edlines_detect(g.ctypes.data_as(C.POINTER(C.c_ubyte)), g.shape[1], g.shape[0], C.c_float(length_thres), out.ctypes.data_as(C.POINTER(C.c_float)), cap)
    if n < 0:
        raise RuntimeError("edlines oracle: more than %d segments" % cap)
    return out[:n].copy()
