// How far OpenCV's fastAtan2 polynomial (csrc/cs_fast_atan.h, the function the LSD host stage calls) can be from the true angle:
// every float quotient c in [0, 1] (all 2^30 of them with stride 1), through both branches (|y| <= |x| and the steep one) and all four
// quadrants.  Field::grow (csrc/lsd_host.cpp) decides a neighbour's alignment from the cos / sin sums directly when the angle between
// the sums and the neighbour is further than kGrowMargin = 0.2 degrees from the tolerance; that is sound as long as this bound (plus
// 3e-5 degrees for the float cos / sin of the neighbour) stays far below the margin.
//   lsd_atan_bound [stride]      stride 1: exhaustive (45 s on 8 threads); the CPU test suite runs stride 61
// Exit code 1 if the bound exceeds 0.0105 degrees.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../cube_slam_wu_amd/csrc/cs_fast_atan.h"

int main(int argc, char** argv) {
  const unsigned stride = argc > 1 ? (unsigned)std::max(1, atoi(argv[1])) : 1u;
  const double kPi = 3.1415926535897932384626433832795;
  const int NT = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<double> worst(NT, 0.0);
  std::vector<std::thread> th;
  for (int t = 0; t < NT; t++)
    th.emplace_back([&, t] {
      double w = 0;
      auto check = [&](float y, float x) {
        double truth = std::atan2((double)y, (double)x) * 180 / kPi;
        if (truth < 0) truth += 360;
        double e = std::fabs((double)cs::fast_atan2_deg(y, x) - truth);
        if (e > 180) e = 360 - e;
        if (e > w) w = e;
      };
      for (unsigned long long b = (unsigned long long)t * stride; b <= 0x3f800000ull; b += (unsigned long long)NT * stride) {
        const unsigned bits = (unsigned)b;
        float c;
        std::memcpy(&c, &bits, 4);
        for (float y : {c, -c})
          for (float x : {1.f, -1.f}) { check(y, x); check(x, y); }      // (second call: the steep branch, roles swapped)
      }
      worst[t] = w;
    });
  for (auto& x : th) x.join();
  double w = 0;
  for (double x : worst) w = std::max(w, x);
  printf("fastAtan2: at most %.6f degrees from the true angle (quotients with stride %u, 8 sign / branch cases each)\n", w, stride);
  return w < 0.0105 ? 0 : 1;
}
