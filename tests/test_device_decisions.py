"""The device kernels take some of the reference's comparisons in a cheaper form and fall back to the exact evaluation inside a
margin (cube_slam_wu_amd/csrc/detect_kernels.hip: atan2_float, vp_step, vp_support_multi pass 1, ls_pair_pass; cs_geom.h:
sqrt_lt_bound / sqrt_le_bound).  The GPU parity tests show the results are unchanged on their inputs; these tests pin the error
bounds the margins are built on, on the CPU, from the constants in the source."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cube_slam_wu_amd", "csrc", "detect_kernels.hip")


def _atan2_float_coefficients():
    """The polynomial of atan2_float, read from the source (highest power first)."""
    text = open(SRC).read()
    body = text[text.index("__device__ __forceinline__ float atan2_float("):]
    body = body[: body.index("return at;")]
    first = re.search(r"float at = (-?[0-9.]+)f;", body).group(1)
    rest = re.findall(r"at = __builtin_fmaf\(at, q2, (-?[0-9.]+)f\);", body)
    assert len(rest) == 5
    return [np.float32(first)] + [np.float32(c) for c in rest]


def _fma32(a, b, c):
    # float32 operands: the product is exact in float64, one rounding of the sum, then to float32
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def atan2_float_model(fy, fx, rcp_ulp=0):
    """atan2_float in numpy float32; rcp_ulp moves the reciprocal by that many ulps (the hardware reciprocal is within 1 ulp)."""
    co = _atan2_float_coefficients()
    PI_F, HPI_F = np.float32(3.14159274), np.float32(1.57079637)
    ay, ax = np.abs(fy), np.abs(fx)
    hi, lo = np.maximum(ax, ay), np.minimum(ax, ay)
    rcp = (np.float32(1.0) / hi).astype(np.float32)
    if rcp_ulp:
        rcp = np.nextafter(rcp, np.float32(np.inf if rcp_ulp > 0 else -np.inf)).astype(np.float32)
    q = (lo * rcp).astype(np.float32)
    q2 = (q * q).astype(np.float32)
    at = np.full_like(q, co[0])
    for c in co[1:]:
        at = _fma32(at, q2, np.full_like(q, c))
    at = (at * q).astype(np.float32)
    at = np.where(ay > ax, HPI_F - at, at).astype(np.float32)
    at = np.where(fx < 0, PI_F - at, at).astype(np.float32)
    at = np.where(fy < 0, -at, at).astype(np.float32)
    return at


@pytest.mark.parametrize("rcp_ulp", [-1, 0, 1])
def test_float_atan2_error_bound(rcp_ulp):
    """|atan2_float - atan2| < 2.5e-6 rad in every octant, at every magnitude a pixel difference can have."""
    rng = np.random.default_rng(7)
    ang = np.concatenate([np.linspace(-np.pi, np.pi, 400001), rng.uniform(-np.pi, np.pi, 400000)])
    # around the octant boundaries, where the branches of the range reduction meet
    for k in range(-4, 5):
        ang = np.concatenate([ang, k * np.pi / 4 + np.linspace(-1e-4, 1e-4, 20001)])
    worst = 0.0
    for mag in (1e-3, 0.7, 33.0, 1.0e3, 4.0e5, 2.0e9):
        r = mag * rng.uniform(0.5, 2.0, ang.size)
        y64, x64 = r * np.sin(ang), r * np.cos(ang)
        fy, fx = y64.astype(np.float32), x64.astype(np.float32)
        got = atan2_float_model(fy, fx, rcp_ulp).astype(np.float64)
        ref = np.arctan2(fy.astype(np.float64), fx.astype(np.float64))      # of the float arguments: the conversion error is budgeted apart
        d = np.abs(got - ref)
        d = np.minimum(d, 2 * np.pi - d)                                      # -pi and +pi are the same direction
        worst = max(worst, float(d.max()))
    assert worst < 2.5e-6, worst


def test_cross_dot_pass_never_drops_an_inlier():
    """Pass 1 of vp_support_multi: |u x d| > tan(thre + 1e-4) |u . d| + 2e-6 (|dx| + |dy|) must not hold for any segment whose
    circular angle distance to the direction d is below thre (15 and 10 degrees in the reference), near or far vanishing points."""
    rng = np.random.default_rng(11)
    n = 2_000_000
    for thre_deg in (15.0, 10.0):
        thre = thre_deg / 180.0 * np.pi
        tan_f = np.float32(np.tan(np.float32(np.float32(thre) + np.float32(1e-4))))
        la = rng.uniform(-np.pi / 2, np.pi / 2, n)                                  # segment angle (double on the device, float for cos / sin)
        # directions concentrated around the inlier boundary, both sides
        kind = rng.integers(0, 3, n)
        off = np.where(kind == 0, rng.uniform(-1.2, 1.2, n) * thre,
                       np.where(kind == 1, (1 + rng.uniform(-1e-4, 1e-4, n)) * thre * rng.choice([-1, 1], n), rng.uniform(-np.pi / 2, np.pi / 2, n)))
        phi = la + off + rng.choice([0.0, np.pi], n)                                  # d may point either way along the line
        dist = 10.0 ** rng.uniform(-1, 9, n)
        dx, dy = dist * np.cos(phi), dist * np.sin(phi)
        raw = np.arctan2(dy, dx)
        nrm = np.where(raw > np.pi / 2, raw - np.pi, np.where(raw < -np.pi / 2, raw + np.pi, raw))
        df = np.abs(la - nrm)
        df = np.minimum(df, np.pi - df)
        inlier = df < thre
        laf = la.astype(np.float32)
        ux, uy = np.cos(laf).astype(np.float32), np.sin(laf).astype(np.float32)
        for eps in (0.0, 2.5e-7, -2.5e-7):                                             # cosf / sinf: 2 ulp
            uxe, uye = (ux + np.float32(eps)).astype(np.float32), (uy - np.float32(eps)).astype(np.float32)
            fx, fy = dx.astype(np.float32), dy.astype(np.float32)
            cr = np.abs(_fma32(fx, uye, -(fy * uxe).astype(np.float32)))
            dt = np.abs(_fma32(fx, uxe, (fy * uye).astype(np.float32)))
            slack = (np.float32(2.0e-6) * (np.abs(fx) + np.abs(fy)).astype(np.float32)).astype(np.float32)
            out_for_sure = cr > _fma32(np.full_like(dt, tan_f), dt, slack)
            assert not np.any(out_for_sure & inlier)
        # and it does its job: most clear outliers are dropped
        clear_out = df > thre + 0.05
        assert out_for_sure[clear_out].mean() > 0.99


SQRT_TEST = r"""
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <random>
#include "cs_geom.h"
int main() {
  std::mt19937_64 g(5);
  std::uniform_real_distribution<double> u(0.0, 1.0);
  long bad = 0, n = 0;
  auto check = [&](double t) {
    const double lt = cs::sqrt_lt_bound(t), le = cs::sqrt_le_bound(t);
    // walk the doubles around both bounds
    for (double c : {lt, le}) {
      double x = c;
      for (int k = 0; k < 6 && x > 0 && std::isfinite(x); k++) x = std::nextafter(x, 0.0);
      for (int k = 0; k < 13; k++) {
        if (!(x >= 0)) break;
        n++;
        if ((std::sqrt(x) < t) != (x < lt)) bad++;
        if ((std::sqrt(x) > t) != (x > le)) bad++;
        x = std::nextafter(x, std::numeric_limits<double>::infinity());
      }
    }
    for (double x : {0.0, 1e-300, 1.0, 399.99999999999994, 400.0, 400.00000000000006, 1e300, std::numeric_limits<double>::infinity(), std::nan("")}) {
      n++;
      if ((std::sqrt(x) < t) != (x < lt)) bad++;
      if ((std::sqrt(x) > t) != (x > le)) bad++;
    }
  };
  for (double t : {20.0, 30.0, 0.0, -1.0, 1.0, 2.0, 1e-160, 1e-170, 4.9e-324, 1e154, 1.3e154, 1.4e154, 1e200, std::numeric_limits<double>::infinity(), std::nan("")}) check(t);
  for (int i = 0; i < 200000; i++) check(std::pow(10.0, -3 + 9 * u(g)));
  for (int i = 0; i < 2000; i++) check(std::pow(10.0, -320 + 640 * u(g)));
  std::printf("%ld %ld\n", n, bad);
  return bad != 0;
}
"""


def test_squared_length_bounds_are_exact(tmp_path):
    """sqrt(x) < t <=> x < sqrt_lt_bound(t) and sqrt(x) > t <=> x > sqrt_le_bound(t), for the doubles on both sides of each bound."""
    src = tmp_path / "sqrt_bounds.cpp"
    src.write_text(SQRT_TEST)
    exe = tmp_path / "sqrt_bounds"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "cube_slam_wu_amd", "csrc"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    n, bad = (int(v) for v in out.stdout.split())
    assert out.returncode == 0 and bad == 0 and n > 1_000_000, out.stdout
