// TEST INFRASTRUCTURE (tests/test_lsd_oracle.py): the two planes the device hands to the LSD host stage -- modulus (double) and
// level-line angle (the float fastAtan2 returns, degrees; -1024 where undefined) -- computed by the CPU restatement, so that the
// product's sequential half (csrc/lsd_host.cpp, linked from lsd_host_check.cpp) can be held to the restatement WITHOUT a GPU.
// A translation unit of its own: the restatement and the product both keep their constants in anonymous namespaces.
#include <cstdio>
#include <cstdlib>

#include "../../oracle/lsd_oracle.cpp"

extern "C" void oracle_planes_f(const unsigned char* gray, int w, int h, int* Ws, int* Hs, float** deg, double** mod) {
  static Lsd L;
  static std::vector<float> D;
  const double prec = kPi * 22.5 / 180, rho = 2.0 / std::sin(prec);
  L.scale_image(gray, w, h);
  L.ll_angle(rho);
  const int W = L.W, H = L.H;
  D.assign((size_t)W * H, -1024.f);
  for (int y = 0; y < H - 1; y++)
    for (int x = 0; x < W - 1; x++) {
      const size_t addr = (size_t)y * W + x;
      if (L.angles[addr] == NOTDEF) continue;
      const double DA = L.img[addr + W + 1] - L.img[addr], BC = L.img[addr + 1] - L.img[addr + W];
      const double gx = DA + BC, gy = DA - BC;
      D[addr] = fast_atan2(float(gx), float(-gy));
      if ((double)D[addr] * DEG_TO_RADS != L.angles[addr]) { fprintf(stderr, "float angle x pi / 180 differs from the restatement's double\n"); abort(); }
    }
  *Ws = W; *Hs = H; *deg = D.data(); *mod = L.modgrad.data();
}
