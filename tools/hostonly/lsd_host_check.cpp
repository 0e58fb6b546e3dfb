// Host-only check of the LSD producer's sequential half (csrc/lsd_host.cpp: region growing without the fastAtan2 chain, cached cos /
// sin, state-byte scans, interval form of rect_nfa's test) against the CPU restatement, image by image, bit for bit -- and its clock.
// No GPU: the planes come from the restatement (lsd_planes_from_oracle.cpp), the device entry points of lsd_host.cpp are stubs that
// are never called.  tests/test_lsd_oracle.py builds and runs it on the reference's 58 frames.
//   lsd_host_check <raw gray file> <width> <height> <images> <repeats>        (images back to back, 8 bit)
#include "../../cube_slam_wu_amd/csrc/lsd_host.cpp"

#include <cstdio>

void cs_set_error_ba(const std::string& s) { fprintf(stderr, "%s\n", s.c_str()); }
extern "C" void* cs_internal_detector_stream(cs_detector*) { return nullptr; }
extern "C" int cs_internal_detector_device(cs_detector*) { return 0; }
extern "C" void** cs_internal_detector_lsd_slot(cs_detector*, void (*)(void*)) { return nullptr; }
extern "C" void* cs_internal_detector_lines_mutex(cs_detector*) { return nullptr; }
extern "C" void cs_internal_detector_parallel(cs_detector*, int, void (*)(int, void*), void*) {}
extern "C" void cs_internal_detector_parallel_long(cs_detector*, int, void (*)(int, void*), void*) {}
namespace cs { void launch_lsd_maps(const unsigned char*, int, int, int, int, const LsdGauss&, const LsdScaleTab&, double, double*, char*, size_t, hipStream_t, int) {} }
extern "C" void oracle_planes_f(const unsigned char* gray, int w, int h, int* Ws, int* Hs, float** deg, double** mod);
extern "C" int lsd_oracle_detect(const unsigned char* gray, int img_w, int img_h, double length_thres, float* out4, int cap);

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const int w = atoi(argv[2]), h = atoi(argv[3]), ni = atoi(argv[4]), reps = std::max(1, atoi(argv[5]));
  std::vector<unsigned char> g((size_t)w * h * ni);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(g.data(), 1, g.size(), f) != g.size()) return 2;
  fclose(f);
  std::vector<float> out(4 * 20000), ref(4 * 20000);
  double t_all = 0;
  int bad = 0;
  long nseg = 0;
  for (int i = 0; i < ni; i++) {
    const unsigned char* gi = g.data() + (size_t)w * h * i;
    int Ws, Hs, n = 0;
    float* deg;
    double* mod;
    oracle_planes_f(gi, w, h, &Ws, &Hs, &deg, &mod);
    double best = 1e30;
    for (int r = 0; r < reps; r++) { const double t0 = lsd_now_ms(); lsd_host_stage(w, h, Ws, Hs, deg, mod, 15.0, out.data(), 20000, &n); best = std::min(best, lsd_now_ms() - t0); }
    t_all += best;
    const int nr = lsd_oracle_detect(gi, w, h, 15.0, ref.data(), 20000);
    if (n != nr || memcmp(out.data(), ref.data(), 16 * (size_t)n)) { bad++; printf("image %d differs (%d vs %d segments)\n", i, n, nr); }
    nseg += n;
  }
  printf("%d images, %ld segments, %d differ from the restatement, host stage %.3f ms per image\n", ni, nseg, bad, t_all / ni);
  return bad != 0;
}
