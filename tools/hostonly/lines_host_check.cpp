// Host-only check of the EDLines producer's sequential half (csrc/lines_host.cpp: routing, fitting, validation on the packed map, the
// gradient and direction maps evaluated from dx / dy where they are read) against the CPU restatement, image by image, bit for bit.
// No GPU: the packed map comes from the restatement (lines_map_from_oracle.cpp; repacked to the device's three bytes per pixel); the device entry points are stubs that are never called.
//   lines_host_check <raw gray file> <width> <height> <images>        (images back to back, 8 bit)
#include "../../cube_slam_wu_amd/csrc/lines_host.cpp"

#include <cstdio>

void cs_set_error_ba(const std::string& s) { fprintf(stderr, "%s\n", s.c_str()); }
extern "C" void* cs_internal_detector_stream(cs_detector*) { return nullptr; }
extern "C" int cs_internal_detector_device(cs_detector*) { return 0; }
extern "C" void** cs_internal_detector_lines_slot(cs_detector*, void (*)(void*)) { return nullptr; }
extern "C" void* cs_internal_detector_lines_mutex(cs_detector*) { return nullptr; }
extern "C" void cs_internal_detector_parallel(cs_detector*, int, void (*)(int, void*), void*) {}
extern "C" void cs_internal_detector_parallel_long(cs_detector*, int, void (*)(int, void*), void*) {}
namespace cs { void launch_lines_maps(const unsigned char*, int, int, const LineMaps&, const int[3], int, int, int, hipStream_t, int) {} }
extern "C" void oracle_lines_packed_map(const uint8_t* gray, int w, int h, int* out);
extern "C" int oracle_edlines_detect(const uint8_t* gray, int w, int h, float length_thres, float* out4, int cap);

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const int w = atoi(argv[2]), h = atoi(argv[3]), ni = atoi(argv[4]);
  std::vector<unsigned char> g((size_t)w * h * ni);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(g.data(), 1, g.size(), f) != g.size()) return 2;
  fclose(f);
  std::vector<float> out(4 * 20000), ref(4 * 20000);
  std::vector<int> pk((size_t)w * h);
  int bad = 0;
  long nseg = 0;
  const EdParams P;
  for (int i = 0; i < ni; i++) {
    const unsigned char* gi = g.data() + (size_t)w * h * i;
    oracle_lines_packed_map(gi, w, h, pk.data());
    std::vector<unsigned char> p3;
    maps_pack3(pk.data(), pk.size(), p3);
    Maps M;
    M.W = w; M.H = h; M.grad_thr = P.grad_thr; M.p3 = p3.data();
    int n = 0;
    if (lines_host_stage(M, P, 15.0, out.data(), 20000, &n) != CS_OK) { bad++; continue; }
    const int nr = oracle_edlines_detect(gi, w, h, 15.0f, ref.data(), 20000);
    if (n != nr || memcmp(out.data(), ref.data(), 16 * (size_t)n)) { bad++; printf("image %d differs (%d vs %d segments)\n", i, n, nr); }
    nseg += n;
  }
  printf("%d images, %ld segments, %d differ from the restatement\n", ni, nseg, bad);
  return bad != 0;
}
