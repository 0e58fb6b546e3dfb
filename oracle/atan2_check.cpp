// atan2_check.cpp -- TEST INFRASTRUCTURE (only tests/ load it): cs_atan2() built twice, with (here) and without
// (atan2_check_dd.cpp) its fast path, to check that the rounding test makes the two return the same bits; the fast
// evaluation exposed as a double-double so that tests/test_atan2.py can measure its error against mpmath.
#include <cstdint>
static long long g_fallbacks = 0;
#define CS_ATAN2_ON_FALLBACK (g_fallbacks++)
#include "../cube_slam_wu_amd/csrc/cs_atan2.h"

extern "C" {
double atan2_check_dd(double y, double x);
double atan2_check_fast(double y, double x) { return cs::cs_atan2(y, x); }
void atan2_check_fast_value(double small, double big, double* hi, double* lo) {
  cs::dd_t v = cs::dd_atan_fast(small, big);
  *hi = v.hi; *lo = v.lo;
}

// n pseudo-random argument pairs of the given kind; returns the number of pairs on which the two builds differ and, in
// *fallbacks, how many the rounding test sent to the double-double evaluation.
//   kind 0: uniform in [-1500, 1500]^2   1: differences of integer pixel coordinates   2: half-pixel grid (segment mid points)
//   kind 3: exponents spread over 2^-60 .. 2^60   4: ratios next to the table points i / 256
long long atan2_check_compare(long long n, unsigned long long seed, int kind, long long* fallbacks) {
  unsigned long long s = seed * 6364136223846793005ULL + 1442695040888963407ULL;
  auto rnd = [&]() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(s >> 11) * (1.0 / 9007199254740992.0); };
  long long bad = 0;
  g_fallbacks = 0;
  for (long long k = 0; k < n; k++) {
    double y, x;
    if (kind == 0) { y = (rnd() * 2 - 1) * 1500; x = (rnd() * 2 - 1) * 1500; }
    else if (kind == 1) { y = (double)((long long)(rnd() * 2483) - 1241); x = (double)((long long)(rnd() * 2483) - 1241); }
    else if (kind == 2) { y = ((long long)(rnd() * 4966) - 2483) * 0.5; x = ((long long)(rnd() * 4966) - 2483) * 0.5 + (rnd() - 0.5) * 1e-9 * (double)(k & 1); }
    else if (kind == 3) {
      union { double d; uint64_t u; } a, b;
      a.d = rnd() + 1.0; b.d = rnd() + 1.0;
      a.u += (uint64_t)((long long)(rnd() * 120) - 60) << 52; b.u += (uint64_t)((long long)(rnd() * 120) - 60) << 52;
      y = (k & 1) ? -a.d : a.d; x = (k & 2) ? -b.d : b.d;
    } else {
      x = (rnd() + 0.5) * 1000; y = x * ((double)(long long)(rnd() * 257) / 256.0) * (1.0 + (rnd() - 0.5) * 1e-12);
      if (k & 1) { double t = x; x = y; y = t; }
      if (k & 2) x = -x;
    }
    const double a = cs::cs_atan2(y, x), b = atan2_check_dd(y, x);
    uint64_t ua, ub;
    __builtin_memcpy(&ua, &a, 8); __builtin_memcpy(&ub, &b, 8);
    bad += (ua != ub);
  }
  if (fallbacks) *fallbacks = g_fallbacks;
  return bad;
}
}  // extern "C"
