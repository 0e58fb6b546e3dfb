// cs_nfa.h -- -log10 of the number of false alarms of a binomial tail, the a-contrario test both line detectors of the reference use:
//   EDLines validation   line_lbd/include/line_lbd/descriptor.hpp:650-848 (nfa / log_gamma_*)
//   LSD rectangles       line_lbd/libs/lsd.cpp:1098-1148 (LineSegmentDetectorImpl::nfa) -- the same routine except that its first
//                        term is (n + 1) where the other has log_gamma(n + 1); kept, `gamma_first_term` = false selects it.
// Host only (libm's log / exp / pow / sinh, as the reference).
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>

namespace cs {
inline double lgamma_lanczos(double x) {
  static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
  double a = (x + 0.5) * std::log(x + 5.5) - (x + 5.5), b = 0.0;
  for (int n = 0; n < 7; n++) { a -= std::log(x + (double)n); b += q[n] * std::pow(x, (double)n); }
  return a + std::log(b);
}
inline double lgamma_windschitl(double x) { return 0.918938533204673 + (x - 0.5) * std::log(x) - x + 0.5 * x * std::log(x * std::sinh(1 / x) + 1 / (810.0 * std::pow(x, 6.0))); }
inline double lgamma_pick(double x) { return x > 15.0 ? lgamma_windschitl(x) : lgamma_lanczos(x); }
inline bool nearly_equal(double a, double b) {
  if (a == b) return true;
  double m = std::max(std::fabs(a), std::fabs(b));
  if (m < DBL_MIN) m = DBL_MIN;
  return std::fabs(a - b) / m <= 100.0 * DBL_EPSILON;
}
inline double minus_log10_nfa(int n, int k, double p, double logNT, bool gamma_first_term) {
  if (n == 0 || k == 0) return -logNT;
  if (n == k) return -logNT - (double)n * std::log10(p);
  const double ratio = p / (1.0 - p);
  const double first = gamma_first_term ? lgamma_pick((double)n + 1.0) : ((double)n + 1.0);
  const double log_first = first - lgamma_pick((double)k + 1.0) - lgamma_pick((double)(n - k) + 1.0) + (double)k * std::log(p) + (double)(n - k) * std::log(1.0 - p);
  double term = std::exp(log_first);
  if (nearly_equal(term, 0.0)) return ((double)k > (double)n * p) ? -log_first / 2.30258509299404568402 - logNT : -logNT;
  double tail = term;
  for (int i = k + 1; i <= n; i++) {
    const double bin = (double)(n - i + 1) / (double)i, mult = bin * ratio;
    term *= mult;
    tail += term;
    if (bin < 1.0) {
      const double err = term * ((1.0 - std::pow(mult, (double)(n - i + 1))) / (1.0 - mult) - 1.0);
      if (err < 0.1 * std::fabs(-std::log10(tail) - logNT) * tail) break;
    }
  }
  return -std::log10(tail) - logNT;
}
}  // namespace cs
