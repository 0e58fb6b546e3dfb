"""cs_atan2 (cube_slam_wu_amd/csrc/cs_atan2.h): correctly rounded, IEEE special cases, agreement with glibc."""
import ctypes as C
import math

import mpmath
import numpy as np

from oracle import oracle_py


def _both(y, x):
    L = oracle_py.lib()
    L.oracle_set_atan2_mode(1)
    a = L.oracle_atan2(C.c_double(y), C.c_double(x))
    L.oracle_set_atan2_mode(0)
    b = L.oracle_atan2(C.c_double(y), C.c_double(x))
    L.oracle_set_atan2_mode(1)
    return a, b


def test_correctly_rounded_against_mpmath():
    mpmath.mp.prec = 300
    rng = np.random.default_rng(0)
    ys = np.concatenate([rng.uniform(-1500, 1500, 4000), np.ldexp(rng.uniform(-1, 1, 1000), rng.integers(-60, 60, 1000))])
    xs = np.concatenate([rng.uniform(-1500, 1500, 4000), np.ldexp(rng.uniform(-1, 1, 1000), rng.integers(-60, 60, 1000))])
    bad = 0
    for y, x in zip(ys, xs):
        got, _ = _both(float(y), float(x))
        exact = mpmath.atan2(mpmath.mpf(float(y)), mpmath.mpf(float(x)))
        if got != float(exact):  # float(mpf) rounds to nearest
            bad += 1
    assert bad == 0


def test_agreement_with_glibc_is_one_ulp_and_rare():
    rng = np.random.default_rng(1)
    n, mism = 200000, 0
    for y, x in zip(rng.uniform(-1500, 1500, n), rng.uniform(-1500, 1500, n)):
        a, b = _both(float(y), float(x))
        if a != b:
            mism += 1
            assert abs(a - b) <= math.ulp(b) * 1.0000001
    # glibc 2.35 dropped its correctly-rounded slow path: ~9e-4 of calls are off by one ulp (measured)
    assert mism / n < 5e-3


def test_special_values_follow_ieee():
    vals = [0.0, -0.0, 1.0, -1.0, math.inf, -math.inf, math.nan, 1e-310, -1e-310, 1e308, -1e308, 5e-324, 3.0, 1e-200, 1e200]
    for y in vals:
        for x in vals:
            a, b = _both(y, x)
            if math.isnan(b):
                assert math.isnan(a)
            else:
                assert a == b and math.copysign(1, a) == math.copysign(1, b), (y, x, a, b)


def _check_lib():
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle_atan2.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(path)])
    L = C.CDLL(path)
    L.atan2_check_compare.restype = C.c_longlong
    L.atan2_check_compare.argtypes = [C.c_longlong, C.c_ulonglong, C.c_int, C.POINTER(C.c_longlong)]
    L.atan2_check_fast_value.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return L


def test_fast_path_returns_the_double_double_result_bit_for_bit():
    """cs_atan2 = light evaluation + rounding test, double-double evaluation when the test fails.  Built with and without
    the light path (oracle/atan2_check*.cpp), the two must agree on every input: uniform scene coordinates, integer and
    half-pixel differences (what the sweep feeds it), wide exponent spreads, quotients next to the table points."""
    L = _check_lib()
    for kind in range(5):
        fb = C.c_longlong(0)
        n = 4_000_000
        assert L.atan2_check_compare(n, 99 + kind, kind, C.byref(fb)) == 0
        assert fb.value / n < 5e-4          # the rounding test rejects ~4e-5 of the calls (bound 2^-68, rounding at 2^-53)


def test_fast_path_error_stays_four_times_below_its_bound():
    """The rounding test is only sound if the light evaluation's error is below the bound it assumes (2^-68 relative)."""
    L = _check_lib()
    mpmath.mp.prec = 300
    rng = np.random.default_rng(5)
    worst = mpmath.mpf(0)
    cases = []
    for _ in range(3000):
        big = 1 + rng.random()
        cases.append((big * rng.random(), big))
        i = rng.integers(0, 257)
        cases.append((min(max(big * ((i + (rng.random() - 0.5) * 1.02) / 256), 1e-9), big), big))   # edges of the table cells
        big = 2 - rng.random() * 1e-6
        cases.append((big * rng.random(), big))
        big = 1 + rng.random()
        cases.append((big * 2.0 ** (-rng.random() * 60), big))
    for small, big in cases:
        hi, lo = C.c_double(), C.c_double()
        L.atan2_check_fast_value(float(small), float(big), C.byref(hi), C.byref(lo))
        exact = mpmath.atan(mpmath.mpf(float(small)) / mpmath.mpf(float(big)))
        worst = max(worst, abs(mpmath.mpf(hi.value) + mpmath.mpf(lo.value) - exact) / exact)
    assert worst < mpmath.mpf(2) ** -70
