// bcr_factor_probe -- where the time of bcr_factor_kernel goes: the kernel built with -DBCR_PROBE stamps the 100 MHz clock at every barrier
// (wave w of workgroup 0); this harness solves one banded system by block cyclic reduction and prints, per wave, the intervals between the
// stamps of the LAST factor launch (one block: nothing else on the device).  Build: tools/microbench/build.sh.  Development tool, not shipped.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#define BCR_PROBE 1
#include "../../cube_slam_wu_amd/csrc/bcr_kernels.hip"

#include <cmath>
#include <vector>

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 128 * 3, LD = argc > 2 ? atoi(argv[2]) : 120, reps = argc > 3 ? atoi(argv[3]) : 3;
  const int bw = LD - 1;
  std::vector<double> A((size_t)n * LD, 0.0), b(n), diag(n, 1.0);
  unsigned long long s = 12345;
  auto rnd = [&]() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return ((s >> 11) * (1.0 / 9007199254740992.0)) * 2 - 1; };
  for (int c = 0; c < n; c++)
    for (int d = 1; d <= bw && c + d < n; d++) { double v = rnd(); A[(size_t)c * LD + d] = v; diag[c] += std::fabs(v); diag[c + d] += std::fabs(v); }
  for (int c = 0; c < n; c++) { A[(size_t)c * LD] = diag[c]; b[c] = rnd(); }
  double *dS, *dL, *dr; int* dinfo; long long* dprobe;
  hipMalloc(&dS, A.size() * 8); hipMalloc(&dL, cs::ba_bcr_workspace_doubles(n, 128) * 8); hipMalloc(&dr, n * 8); hipMalloc(&dinfo, 96); hipMalloc(&dprobe, 8 * 64 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(cs::g_bcr_probe), &dprobe, sizeof(dprobe));
  if (!cs::ba_bcr_ok(n, LD)) { printf("block cyclic reduction refused this shape\n"); return 1; }
  hipStream_t st; hipStreamCreate(&st);
  for (int r = 0; r < reps; r++) {
    hipMemcpyAsync(dS, A.data(), A.size() * 8, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(dr, b.data(), n * 8, hipMemcpyHostToDevice, st);
    hipMemsetAsync(dinfo, 0, 96, st);
    hipMemsetAsync(dprobe, 0, 8 * 64 * 8, st);
    cs::ba_launch_bcr(dS, dL, n, LD, 128, dr, dinfo, st);
    hipStreamSynchronize(st);
  }
  std::vector<long long> p(8 * 64);
  hipMemcpy(p.data(), dprobe, p.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> x(n);
  hipMemcpy(x.data(), dr, n * 8, hipMemcpyDeviceToHost);
  double rmax = 0, bmax = 0;
  for (int i = 0; i < n; i++) {
    double acc = 0;
    for (int j = std::max(0, i - bw); j <= std::min(n - 1, i + bw); j++) acc += (j <= i ? A[(size_t)j * LD + (i - j)] : A[(size_t)i * LD + (j - i)]) * x[j];
    rmax = std::max(rmax, std::fabs(acc - b[i])); bmax = std::max(bmax, std::fabs(b[i]));
  }
  printf("residual %.3e\n", rmax / bmax);
  // (the buffer holds the stamps of the last factor launch that had a workgroup 0 -- every launch overwrites them)
  for (int w = 0; w < 8; w++) {
    int k = 0; while (k < 64 && p[w * 64 + k]) k++;
    printf("wave %d: %d stamps, total %.2f us; intervals (us):", w, k, k > 1 ? (p[w * 64 + k - 1] - p[w * 64]) * 0.01 : 0.0);
    for (int q = 1; q < k; q++) printf(" %.2f", (p[w * 64 + q] - p[w * 64 + q - 1]) * 0.01);
    printf("\n");
  }
  return 0;
}
