"""ctypes binding of oracle/edge_oracle.cpp (test infrastructure only)."""
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
        so = os.path.join(HERE, "liboracle_edge.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "edge_oracle.cpp")):
            subprocess.check_call(["make", "-s", "-C", HERE, "liboracle_edge.so"])
        _lib = C.CDLL(so)
    return _lib


def _u8(a):
    return np.ascontiguousarray(a, np.uint8)


def bgr_to_gray(bgr):
    bgr = _u8(bgr)
    out = np.zeros(bgr.shape[:2], np.uint8)
    lib().oracle_bgr_to_gray(bgr.ctypes.data_as(C.POINTER(C.c_ubyte)), out.size, out.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return out


def canny_roi(gray, roi, low=80, high=200):
    gray = _u8(gray)
    l, t, w, h = [int(v) for v in roi]
    out = np.zeros((h, w), np.uint8)
    lib().oracle_canny_roi(gray.ctypes.data_as(C.POINTER(C.c_ubyte)), gray.shape[1], gray.shape[0], l, t, w, h, int(low), int(high), out.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return out


def dist_l2_3x3(edges255):
    e = _u8(edges255)
    out = np.zeros(e.shape, np.float32)
    lib().oracle_dist_l2_3x3(e.ctypes.data_as(C.POINTER(C.c_ubyte)), e.shape[1], e.shape[0], out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def edge_distance_map(gray, roi, low=80, high=200):
    gray = _u8(gray)
    l, t, w, h = [int(v) for v in roi]
    out = np.zeros((h, w), np.float32)
    lib().oracle_edge_distance_map(gray.ctypes.data_as(C.POINTER(C.c_ubyte)), gray.shape[1], gray.shape[0], l, t, w, h, int(low), int(high), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out
