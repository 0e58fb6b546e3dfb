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


def _vp_support_exact(mx, my, la, vpx, vpy, thre):
    """VP_support_edge_infos as the reference evaluates it (object_3d_util.cpp:548-619), in float64: (inliers, arg-max, arg-min)."""
    inl, base, hi, lo, i_hi, i_lo = [], None, None, None, -1, -1
    for i in range(len(mx)):
        raw = np.arctan2(my[i] - vpy, mx[i] - vpx)
        nrm = raw - np.pi if raw > np.pi / 2 else (raw + np.pi if raw < -np.pi / 2 else raw)
        d = abs(la[i] - nrm)
        d = min(d, np.pi - d)
        if not d < thre:
            continue
        inl.append(i)
        if base is None:
            base = hi = lo = raw
            i_hi = i_lo = i
            continue
        sh = raw + 2 * np.pi if raw - base < -np.pi else (raw - 2 * np.pi if raw - base > np.pi else raw)
        if sh > hi:
            hi, i_hi = sh, i
        if sh < lo:
            lo, i_lo = sh, i
    return inl, i_hi, i_lo


def _vp_support_device_logic(mx, my, la, vpx, vpy, thre, stats):
    """The decision logic of vp_support_multi / vp_step (pass 1 in float, float angles with margins, cross-product order,
    float64 evaluation inside the margins), step by step as the kernel takes it."""
    f32 = np.float32
    PI_F, HPI_F, M_IN, M_ORD = f32(3.14159274), f32(1.57079637), f32(1.0e-5), f32(2.0e-5)
    thre_f = f32(thre)
    tan_f = f32(np.tan(f32(thre_f + f32(1e-4))))

    def exact_unwrapped(i, ib):
        raw = np.arctan2(my[i] - vpy, mx[i] - vpx)
        if i == ib:
            return raw
        base = np.arctan2(my[ib] - vpy, mx[ib] - vpx)
        return raw + 2 * np.pi if raw - base < -np.pi else (raw - 2 * np.pi if raw - base > np.pi else raw)

    have, ib, i_hi, i_lo = False, 0, -1, -1
    base_f = hi_f = lo_f = f32(0)
    hx = hy = lx = ly = 0.0
    inl_list = []
    for i in range(len(mx)):
        dxd, dyd = mx[i] - vpx, my[i] - vpy
        fx, fy = f32(dxd), f32(dyd)
        # pass 1
        laf = f32(la[i])
        ux, uy = f32(np.cos(laf)), f32(np.sin(laf))
        cr = abs(f32(np.float64(fx) * np.float64(uy) - np.float64(f32(fy * ux))))
        dt = abs(f32(np.float64(fx) * np.float64(ux) + np.float64(f32(fy * uy))))
        slack = f32(f32(2.0e-6) * f32(abs(fx) + abs(fy)))
        if cr > f32(np.float64(tan_f) * np.float64(dt) + np.float64(slack)):
            stats["dropped"] += 1
            continue
        # pass 2
        at = atan2_float_model(np.array([fy], f32), np.array([fx], f32))[0]
        hi_mag = max(abs(fx), abs(fy))
        usable = hi_mag > 0 and hi_mag < f32(3.0e38)
        nr = at - PI_F if at > HPI_F else (at + PI_F if at < -HPI_F else at)
        df = abs(f32(laf - nr))
        df = min(df, f32(PI_F - df))
        inl = bool(usable and df < f32(thre_f - M_IN))
        if not (inl or (usable and df > f32(thre_f + M_IN))):
            stats["exact_inlier"] += 1
            raw = np.arctan2(dyd, dxd)
            nrm = raw - np.pi if raw > np.pi / 2 else (raw + np.pi if raw < -np.pi / 2 else raw)
            d = abs(la[i] - nrm)
            inl = min(d, np.pi - d) < thre
            at = f32(raw)
        if not inl:
            continue
        inl_list.append(i)
        if not have:
            have, ib, i_hi, i_lo = True, i, i, i
            base_f = hi_f = lo_f = at
            hx = lx = dxd
            hy = ly = dyd
            continue
        d = f32(at - base_f)
        sh = f32(at + f32(2) * PI_F) if d < -PI_F else (f32(at - f32(2) * PI_F) if d > PI_F else at)
        if abs(f32(abs(d) - PI_F)) < M_ORD:
            stats["exact_unwrap"] += 1
            sh = f32(exact_unwrapped(i, ib))
        if sh > f32(hi_f + M_ORD):
            hi_f, i_hi, hx, hy = sh, i, dxd, dyd
        elif sh > f32(hi_f - M_ORD):
            c = hx * dyd - hy * dxd
            lim = 1.0e-12 * ((abs(dxd) + abs(dyd)) * (abs(hx) + abs(hy)))
            stats["cross_order"] += 1
            if c > lim:
                greater = True
            elif c < -lim:
                greater = False
            else:
                stats["exact_order"] += 1
                greater = exact_unwrapped(i, ib) > exact_unwrapped(i_hi, ib)
            if greater:
                hi_f, i_hi, hx, hy = sh, i, dxd, dyd
        if sh < f32(lo_f - M_ORD):
            lo_f, i_lo, lx, ly = sh, i, dxd, dyd
        elif sh < f32(lo_f + M_ORD):
            c = lx * dyd - ly * dxd
            lim = 1.0e-12 * ((abs(dxd) + abs(dyd)) * (abs(lx) + abs(ly)))
            stats["cross_order"] += 1
            if c < -lim:
                less = True
            elif c > lim:
                less = False
            else:
                stats["exact_order"] += 1
                less = exact_unwrapped(i, ib) < exact_unwrapped(i_lo, ib)
            if less:
                lo_f, i_lo, lx, ly = sh, i, dxd, dyd
    return inl_list, i_hi, i_lo


def test_vp_support_decision_logic_equals_the_exact_evaluation():
    """Inlier set, arg-max and arg-min of the device's decision procedure against the plain float64 evaluation of the reference's
    loop, for vanishing points inside the ROI, around the image and far away (10^4 .. 10^9 px, where all segments lie within a
    fraction of a milliradian and the float angles cannot order them), with duplicated and collinear segments mixed in."""
    rng = np.random.default_rng(23)
    stats = dict(dropped=0, exact_inlier=0, exact_unwrap=0, exact_order=0, cross_order=0)
    n_cases = 0
    for case in range(700):
        m = int(rng.integers(3, 70))
        mx, my = rng.uniform(200, 500, m), rng.uniform(100, 300, m)
        la = rng.uniform(-np.pi / 2, np.pi / 2, m)
        kind = case % 4
        if kind == 0:
            vpx, vpy = rng.uniform(150, 550), rng.uniform(50, 350)                    # inside the segment cloud: wrap-arounds
        elif kind == 1:
            vpx, vpy = rng.uniform(-2000, 3000), rng.uniform(-1500, 2000)
        else:
            r, a = 10.0 ** rng.uniform(4, 9), rng.uniform(0, 2 * np.pi)
            vpx, vpy = r * np.cos(a), r * np.sin(a)
        # many segments pointing at the vanishing point (what a real vanishing point has), some exactly duplicated
        for i in range(m):
            if rng.random() < 0.5:
                raw = np.arctan2(my[i] - vpy, mx[i] - vpx)
                nrm = raw - np.pi if raw > np.pi / 2 else (raw + np.pi if raw < -np.pi / 2 else raw)
                la[i] = nrm + rng.uniform(-0.3, 0.3)
        if m > 6:
            mx[5], my[5], la[5] = mx[2], my[2], la[2]
        if m > 12:   # a segment whose angle sits on the inlier threshold to within 1e-7 rad, and (vanishing point inside the cloud) a
            # segment diametrically opposite the first one, so that the unwrap decision sits on +-pi
            raw = np.arctan2(my[9] - vpy, mx[9] - vpx)
            nrm = raw - np.pi if raw > np.pi / 2 else (raw + np.pi if raw < -np.pi / 2 else raw)
            la[9] = nrm + (15.0 / 180.0 * np.pi) * (1 + rng.uniform(-1e-7, 1e-7))
            la[9] = la[9] - np.pi if la[9] > np.pi / 2 else la[9]
            if kind == 0:
                mx[11], my[11] = 2 * vpx - mx[0] + rng.uniform(-1e-5, 1e-5), 2 * vpy - my[0]
                raw0 = np.arctan2(my[0] - vpy, mx[0] - vpx)
                nrm0 = raw0 - np.pi if raw0 > np.pi / 2 else (raw0 + np.pi if raw0 < -np.pi / 2 else raw0)
                la[0] = nrm0
                la[11] = nrm0
        for thre_deg in (15.0, 10.0):
            thre = thre_deg / 180.0 * np.pi
            ref = _vp_support_exact(mx, my, la, vpx, vpy, thre)
            got = _vp_support_device_logic(mx, my, la, vpx, vpy, thre, stats)
            assert got == ref, (case, thre_deg)
            n_cases += 1
    # every branch of the procedure was taken, and the exact evaluations stay the exception
    assert stats["dropped"] > 10000 and stats["cross_order"] > 500
    assert stats["exact_inlier"] > 50 and stats["exact_unwrap"] > 20 and stats["exact_order"] > 50
    assert stats["exact_inlier"] + stats["exact_unwrap"] + stats["exact_order"] < 0.05 * n_cases * 35
    assert n_cases == 1400


def test_line_pair_test_decisions_equal_the_exact_evaluation():
    """ls_t12 / ls_t3 (the three tests of merge_break_lines, object_3d_util.cpp:464-497) as line_setup_kernel takes them since round 5:
    every row carries its angle as a FLOAT (atan2_float of its end points); the angle-difference test and the test of the would-be merged
    segment's angle are taken on floats, with the exact evaluation (float64 atan2 of the rows' end points) inside a margin of 1.5e-5 rad;
    end-point gaps are compared squared against the squared bound.  The same decision as sqrt / atan2 in float64 for every pair,
    including pairs built to sit on the thresholds.  The margin is read from the source."""
    rng = np.random.default_rng(31)
    f32 = np.float32
    text = open(SRC).read()
    margin = f32(float(re.search(r"#define LS_MARGIN ([0-9.e+-]+)f", text).group(1)))
    assert margin == f32(1.5e-5)
    PI_F = f32(3.14159274)
    dist_thre, ang_thre = 20.0, 5.0 / 180.0 * np.pi
    # sqrt(x) < 20 <=> x < bound, bound = the smallest double whose rounded root reaches 20 (cs_geom.h sqrt_lt_bound; the C++
    # function itself is checked in test_squared_length_bounds_are_exact): one ulp below 400, whose root already rounds to 20
    dist_sq_bound = 400.0
    while np.sqrt(np.nextafter(dist_sq_bound, 0.0)) >= dist_thre:
        dist_sq_bound = np.nextafter(dist_sq_bound, 0.0)
    assert dist_sq_bound == 399.99999999999994
    n_exact1 = n_exact3 = n_pass = 0
    worst_af = 0.0
    for trial in range(60000):
        x1, y1 = rng.uniform(0, 600), rng.uniform(0, 300)
        a = rng.uniform(-np.pi / 2, np.pi / 2)
        la_ = rng.uniform(15, 120)
        A = (x1, y1, x1 + la_ * np.cos(a), y1 + la_ * np.sin(a))
        if A[2] < A[0]:
            A = (A[2], A[3], A[0], A[1])
        # B continues A after a gap, with a small angle change; every 4th pair sits on a threshold
        gap = rng.uniform(0, 30) if trial % 4 else dist_thre * (1 + rng.uniform(-1e-9, 1e-9))
        b = a + (rng.uniform(-0.15, 0.15) if trial % 4 != 1 else ang_thre * rng.choice([-1, 1]) * (1 + rng.uniform(-1e-7, 1e-7)))
        bx, by = A[2] + gap * np.cos(a), A[3] + gap * np.sin(a)
        lb = rng.uniform(15, 120)
        B = (bx, by, bx + lb * np.cos(b), by + lb * np.sin(b))
        if B[2] < B[0]:
            B = (B[2], B[3], B[0], B[1])
        angA, angB = np.arctan2(A[3] - A[1], A[2] - A[0]), np.arctan2(B[3] - B[1], B[2] - B[0])
        # the rows' float angles, as the kernel stores them
        afA = atan2_float_model(np.array([A[3] - A[1]], f32), np.array([A[2] - A[0]], f32))[0]
        afB = atan2_float_model(np.array([B[3] - B[1]], f32), np.array([B[2] - B[0]], f32))[0]
        worst_af = max(worst_af, abs(float(afA) - angA), abs(float(afB) - angB))

        def exact():
            diff = abs(angA - angB)
            if min(diff, np.pi - diff) >= ang_thre:
                return False
            d_ab = np.sqrt((A[2] - B[0]) ** 2 + (A[3] - B[1]) ** 2)
            d_ba = np.sqrt((B[2] - A[0]) ** 2 + (B[3] - A[1]) ** 2)
            if not (d_ab < dist_thre or d_ba < dist_thre):
                return False
            sx, sy = (A[0], A[1]) if A[0] < B[0] else (B[0], B[1])
            ex, ey = (A[2], A[3]) if A[2] > B[2] else (B[2], B[3])
            t = abs(angA - np.arctan2(ey - sy, ex - sx))
            return min(t, np.pi - t) < ang_thre

        def device():
            nonlocal n_exact1, n_exact3
            th = f32(ang_thre)
            df = abs(f32(afA - afB))
            df = min(df, f32(PI_F - df))
            if df < f32(th - margin):
                p1 = True
            elif df > f32(th + margin):
                p1 = False
            else:
                n_exact1 += 1
                diff = abs(angA - angB)
                p1 = min(diff, np.pi - diff) < ang_thre
            if not p1:
                return False
            d_ab = (A[2] - B[0]) ** 2 + (A[3] - B[1]) ** 2
            d_ba = (B[2] - A[0]) ** 2 + (B[3] - A[1]) ** 2
            if not (d_ab < dist_sq_bound or d_ba < dist_sq_bound):
                return False
            sx, sy = (A[0], A[1]) if A[0] < B[0] else (B[0], B[1])
            ex, ey = (A[2], A[3]) if A[2] > B[2] else (B[2], B[3])
            dy, dx = ey - sy, ex - sx
            at = atan2_float_model(np.array([dy], f32), np.array([dx], f32))[0]
            tf = abs(f32(afA - at))
            mf = min(tf, f32(PI_F - tf))
            if mf < f32(th - margin):
                return True
            if mf > f32(th + margin):
                return False
            n_exact3 += 1
            t = abs(angA - np.arctan2(dy, dx))
            return min(t, np.pi - t) < ang_thre

        e = exact()
        assert device() == e, trial
        n_pass += e
    # a float difference of two row angles is within 2 x 2.5e-6 (+ rounding) of the exact difference: a third of the margin
    assert worst_af < 2.5e-6 and 2 * worst_af + 1e-6 < float(margin)
    # (a quarter of the pairs is built ON the angle threshold: 15 000 exact evaluations of the first test by construction)
    assert n_pass > 5000 and 15000 <= n_exact1 < 15200 and 0 < n_exact3 < 3000
