// cs_fast_atan.h -- OpenCV's fastAtan2 (degrees; third party: restated from its published polynomial), the host copy.  The device copy
// is lsd_kernels.hip's.  Field::grow (lsd_host.cpp) relies on how far it can be from the true angle: tools/microbench/lsd_atan_bound.cpp
// walks every quotient through every branch (0.00956 degrees at worst).
#pragma once
#include <cfloat>
#include <cmath>

namespace cs {
inline float fast_atan2_deg(float y, float x) {
  constexpr double kPi = 3.1415926535897932384626433832795;
  static const float p1 = 0.9997878412794807f * (float)(180 / kPi), p3 = -0.3258083974640975f * (float)(180 / kPi), p5 = 0.1555786518463281f * (float)(180 / kPi),
                     p7 = -0.04432655554792128f * (float)(180 / kPi);
  const float ax = std::fabs(x), ay = std::fabs(y);
  const bool steep = ay > ax;
  const float c = steep ? ax / (ay + (float)DBL_EPSILON) : ay / (ax + (float)DBL_EPSILON);
  const float c2 = c * c;
  float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  if (steep) a = 90.f - a;
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}
}  // namespace cs
