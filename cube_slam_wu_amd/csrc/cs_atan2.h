// cs_atan2.h -- one atan2() for host and device.
//
// Why this exists.  The reference scores every cuboid proposal with angles from libm atan2()
// (detect_3d_cuboid/src/object_3d_util.cpp:573 VP_support_edge_infos, :702
// box_edge_alignment_angle_error, box_proposal_detail.cpp:313) and then ranks proposals on those
// doubles (object_3d_util.cpp:748-786 partial_sort).  The ranking contract is integer-exact, so the
// HIP kernels and any CPU checker must agree on atan2 to the last bit.  Device libm (ocml) and glibc
// do not.  cs_atan2() is evaluated in double-double arithmetic (error ~2^-83 relative, i.e.
// correctly rounded except when the true value lies within ~2^-30 ulp of a rounding boundary),
// uses only +,-,*,/ and fma -- all IEEE-exact on x86-64 and gfx950 -- and is therefore
// bit-reproducible across the two.  tests/test_atan2.py measures its agreement with glibc.
//
// Two evaluations, one result.  The double-double evaluation costs ~230 FP64 instructions; a lighter one (finer table,
// one division, the cubic series in plain doubles: ~90 instructions) is accurate to ~2^-70 and comes with its error
// bound: when both ends of [value - bound, value + bound] round to the same double, that double IS the correctly
// rounded result and is returned; otherwise (about 2^-15 of the calls) the double-double evaluation decides.  Both
// paths consist of IEEE-exact operations only, so host and device still take the same path and return the same bits.
//
// Build rule for every TU that includes this header: -ffp-contract=off (explicit fma() calls below
// are intended; implicit contraction is not).
#pragma once
#include "cs_atan2_tab.h"

#if defined(__HIPCC__)
#define CS_HD __host__ __device__ __forceinline__
#define CS_HD_COLD inline __host__ __device__ __attribute__((noinline))   /* rarely taken: out of line, so that it does not set the caller's register budget */
#else
#define CS_HD static inline
#define CS_HD_COLD static
#endif

namespace cs {

struct dd_t {
  double hi, lo;
};

CS_HD dd_t dd_fast_two_sum(double a, double b) {  // requires |a| >= |b| (or a == 0)
  double s = a + b;
  double e = b - (s - a);
  return dd_t{s, e};
}

CS_HD dd_t dd_two_sum(double a, double b) {
  double s = a + b;
  double bb = s - a;
  double e = (a - (s - bb)) + (b - bb);
  return dd_t{s, e};
}

CS_HD dd_t dd_two_prod(double a, double b) {
  double p = a * b;
  double e = __builtin_fma(a, b, -p);
  return dd_t{p, e};
}

CS_HD dd_t dd_add(dd_t a, dd_t b) {
  dd_t s = dd_two_sum(a.hi, b.hi);
  dd_t t = dd_two_sum(a.lo, b.lo);
  s.lo = s.lo + t.hi;
  s = dd_fast_two_sum(s.hi, s.lo);
  s.lo = s.lo + t.lo;
  return dd_fast_two_sum(s.hi, s.lo);
}

CS_HD dd_t dd_add_d(dd_t a, double b) {
  dd_t s = dd_two_sum(a.hi, b);
  s.lo = s.lo + a.lo;
  return dd_fast_two_sum(s.hi, s.lo);
}

CS_HD dd_t dd_neg(dd_t a) { return dd_t{-a.hi, -a.lo}; }

CS_HD dd_t dd_mul(dd_t a, dd_t b) {
  dd_t p = dd_two_prod(a.hi, b.hi);
  p.lo = p.lo + (a.hi * b.lo + a.lo * b.hi);
  return dd_fast_two_sum(p.hi, p.lo);
}

CS_HD dd_t dd_mul_d(dd_t a, double b) {
  dd_t p = dd_two_prod(a.hi, b);
  p.lo = p.lo + a.lo * b;
  return dd_fast_two_sum(p.hi, p.lo);
}

CS_HD dd_t dd_div(dd_t n, dd_t d) {
  double q1 = n.hi / d.hi;
  // r = n - q1*d, exactly enough
  dd_t p = dd_two_prod(q1, d.hi);
  double r = ((n.hi - p.hi) - p.lo) + n.lo - q1 * d.lo;
  double q2 = r / d.hi;
  return dd_fast_two_sum(q1, q2);
}

CS_HD long long cs_bits(double x) {
  long long b;
  __builtin_memcpy(&b, &x, sizeof(b));
  return b;
}

CS_HD double cs_from_bits(long long b) {
  double x;
  __builtin_memcpy(&x, &b, sizeof(x));
  return x;
}

// atan of a double-double in [0, 1], returned as double-double.
CS_HD dd_t dd_atan_unit(dd_t num, dd_t den, double q_approx) {
  static const double tab[CS_ATAN_TAB_N][2] = CS_ATAN_TAB_INIT;
  int i = (int)(q_approx * 64.0 + 0.5);
  i = i < 0 ? 0 : (i > 64 ? 64 : i);
  double c = (double)i * (1.0 / 64.0);
  // t = (num - c*den) / (den + c*num);  atan(num/den) = atan(c) + atan(t), |t| <= ~1/128
  dd_t cn = dd_mul_d(num, c);
  dd_t cd = dd_mul_d(den, c);
  dd_t N = dd_add(num, dd_neg(cd));
  dd_t D = dd_add(den, cn);
  dd_t t = dd_div(N, D);
  dd_t t2 = dd_mul(t, t);
  double u = t2.hi;
  // tail of the odd series beyond -1/3: u/5 - u^2/7 + u^3/9 - u^4/11 + u^5/13 - u^6/15
  double s = u * (1.0 / 5.0 + u * (-1.0 / 7.0 + u * (1.0 / 9.0 + u * (-1.0 / 11.0 + u * (1.0 / 13.0 + u * (-1.0 / 15.0))))));
  dd_t A = dd_add_d(dd_t{CS_DD_M1_3_HI, CS_DD_M1_3_LO}, s);
  dd_t B = dd_mul(t2, A);
  dd_t C = dd_mul(t, B);
  dd_t at = dd_add(t, C);
  return dd_add(dd_t{tab[i][0], tab[i][1]}, at);
}

// Fast path.  0 < small <= big, big in [1, 2).  atan(small / big) = atan(c) + atan(t), c = i / 256 next to the quotient,
// t = (small - c big) / (big + c small), |t| <= 2^-9 (+ the slack of the single-precision quotient that picks i: any
// neighbouring i is as good).  Numerator and denominator are kept as unevaluated sums of two doubles, t = th + tl comes
// from one division and its remainder, atan(t) = t - t^3/3 + t^5/5 - t^7/7 with the corrections in plain doubles
// (|t^3| <= 2^-18 |t|: their rounding errors stay below 2^-70 |t|; the series ends below 2^-75 |t|).  Returns the value as
// a double-double; relative error < 2^-69.
CS_HD dd_t dd_atan_fast(double small, double big) {
  static const double ftab[CS_ATAN_FTAB_N][2] = CS_ATAN_FTAB_INIT;
  const float qf = (float)small / (float)big;
  int i = (int)(qf * 256.0f + 0.5f);
  i = i < 0 ? 0 : (i > 256 ? 256 : i);
  const double c = (double)i * (1.0 / 256.0);
  const dd_t pb = dd_two_prod(c, big);
  const dd_t ns = dd_two_sum(small, -pb.hi);
  const double Nl = ns.lo - pb.lo;
  const dd_t ps = dd_two_prod(c, small);
  const dd_t ds = dd_two_sum(big, ps.hi);
  const double Dl = ds.lo + ps.lo;
  const double inv = 1.0 / ds.hi;
  const double th = ns.hi * inv;
  const double r = __builtin_fma(-th, ds.hi, ns.hi) + (Nl - th * Dl);
  const double tl = r * inv;
  // the error bound below assumes |t| <= 2^-9 (+ slack); should the table index ever be off, decline (NaN fails the caller's
  // rounding test, which sends the call to the double-double evaluation)
  if (!(th <= 0x1.1p-9 && th >= -0x1.1p-9)) return dd_t{__builtin_nan(""), 0.0};
  const double u = th * th;
  const double corr = th * (u * (-1.0 / 3.0 + u * (1.0 / 5.0 + u * (-1.0 / 7.0))));
  const dd_t s = dd_two_sum(ftab[i][0], th);
  const double lo = s.lo + (ftab[i][1] + (tl + corr));
  return dd_fast_two_sum(s.hi, lo);
}

// The double-double evaluation of the first-quadrant angle, reflected into the octant (small / big scaled, lo / hi the
// unscaled magnitudes in the same order).
CS_HD_COLD double cs_atan2_dd(double small, double big, double lo_mag, double hi_mag, bool swap, bool xneg) {
  double q = small / big;
  dd_t a;
  if (q < 0x1p-900) {
    // atan(q) == q to far beyond double precision; keep the plain quotient of the originals.
    a = dd_t{lo_mag / hi_mag, 0.0};
  } else {
    a = dd_atan_unit(dd_t{small, 0.0}, dd_t{big, 0.0}, q);
  }
  if (swap) a = dd_add(dd_t{CS_DD_PI_2_HI, CS_DD_PI_2_LO}, dd_neg(a));
  if (xneg) a = dd_add(dd_t{CS_DD_PI_HI, CS_DD_PI_LO}, dd_neg(a));
  return a.hi;
}

// IEEE-754 atan2 semantics (C11 F.10.1.4) for zeros, infinities and NaN.
CS_HD double cs_atan2(double y, double x) {
  if (x != x || y != y) return x + y;
  const long long SIGN = (long long)0x8000000000000000ULL;
  long long by = cs_bits(y), bx = cs_bits(x);
  bool yneg = (by & SIGN) != 0, xneg = (bx & SIGN) != 0;
  double ay = cs_from_bits(by & ~SIGN), ax = cs_from_bits(bx & ~SIGN);
  const double INF = __builtin_huge_val();
  double r;
  if (ay == 0.0) {
    r = xneg ? CS_DD_PI_HI : 0.0;
  } else if (ax == 0.0) {
    r = CS_DD_PI_2_HI;
  } else if (ax == INF) {
    if (ay == INF) r = xneg ? CS_DD_3PI_4_HI : CS_DD_PI_4_HI;
    else r = xneg ? CS_DD_PI_HI : 0.0;
  } else if (ay == INF) {
    r = CS_DD_PI_2_HI;
  } else {
    // finite, non-zero.  Scale the larger magnitude into [1,2) (exact power-of-two scaling).
    bool swap = ay > ax;
    double big = swap ? ay : ax, small = swap ? ax : ay;
    int eb = (int)((cs_bits(big) >> 52) & 0x7ff);
    if (eb == 0) {  // subnormal big: pre-scale both by 2^600
      big *= 0x1p600; small *= 0x1p600;
      eb = (int)((cs_bits(big) >> 52) & 0x7ff);
    }
    // multiply by 2^(1023-eb) in two exact steps to stay inside the exponent range
    int sh = 1023 - eb;
    int sh1 = sh / 2, sh2 = sh - sh1;
    double f1 = cs_from_bits((long long)(1023 + sh1) << 52), f2 = cs_from_bits((long long)(1023 + sh2) << 52);
    big = big * f1 * f2;
    small = small * f1 * f2;  // may underflow when the ratio is below ~2^-1000: atan ~ ratio
    bool done = false;
#ifndef CS_ATAN2_NO_FAST   /* (the checker builds one copy without the fast path to compare against) */
    if (small >= 0x1p-200) {
      // fast path + rounding test.  The octant reflections subtract from constants known to 2^-106, and the result is never
      // smaller than the value it was reflected from, so one relative bound on the final value covers everything.
      dd_t f = dd_atan_fast(small, big);
      if (swap) { dd_t v = dd_two_sum(CS_DD_PI_2_HI, -f.hi); v.lo = v.lo + (CS_DD_PI_2_LO - f.lo); f = dd_fast_two_sum(v.hi, v.lo); }
      if (xneg) { dd_t v = dd_two_sum(CS_DD_PI_HI, -f.hi); v.lo = v.lo + (CS_DD_PI_LO - f.lo); f = dd_fast_two_sum(v.hi, v.lo); }
      const double bound = f.hi * 0x1p-68;
      const double up = f.hi + (f.lo + bound), dn = f.hi + (f.lo - bound);
      if (up == dn) { r = up; done = true; }
    }
#endif
    if (!done) {
#ifdef CS_ATAN2_ON_FALLBACK   /* checker hook: counts how often the rounding test defers to the double-double evaluation */
      CS_ATAN2_ON_FALLBACK;
#endif
      r = cs_atan2_dd(small, big, swap ? ax : ay, swap ? ay : ax, swap, xneg);
    }
  }
  return yneg ? -r : r;
}

}  // namespace cs
