// potf2_bench.cpp -- times the banded solver's 32 x 32 diagonal-block routine (Cholesky factor + inverse of the factor) alone: one workgroup,
// `reps` calls on the same block from LDS, device clock around them; checks L L^T = A and X L = I.  Build: tools/microbench/build.sh.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../cube_slam_wu_amd/csrc/ba_kernels.hip"

// latency probes: a chain of dependent f64 multiply-adds, of reciprocal estimates, of rsq estimates, of LDS write -> barrier -> read trips
__global__ __launch_bounds__(256) void latency_kernel(double* out, long long* ticks, double seed) {
  __shared__ double buf[512];
  const int tid = threadIdx.x;
  double a = seed + tid * 1e-9, b = 1.0000001;
  long long t0 = wall_clock64();
  for (int i = 0; i < 1024; i++) a = fma(a, b, 1e-12);
  long long t1 = wall_clock64();
  double r = a;
  for (int i = 0; i < 256; i++) r = __builtin_amdgcn_rcp(r) + 1.5;
  long long t2 = wall_clock64();
  double q = r;
  for (int i = 0; i < 256; i++) q = __builtin_amdgcn_rsq(q) + 1.5;
  long long t3 = wall_clock64();
  double z = q;
  for (int i = 0; i < 256; i++) { buf[(i & 1) * 256 + tid] = z; __syncthreads(); z = buf[(i & 1) * 256 + ((tid + 64) & 255)] + 1.0; }
  long long t4 = wall_clock64();
  double y = z;
  for (int i = 0; i < 256; i++) { buf[(i & 1) * 256 + tid] = y; __builtin_amdgcn_wave_barrier(); y = buf[(i & 1) * 256 + (tid ^ 1)] + 1.0; }
  long long t5 = wall_clock64();
  out[tid] = y;
  if (tid == 0) { ticks[0] = t1 - t0; ticks[1] = t2 - t1; ticks[2] = t3 - t2; ticks[3] = t4 - t3; ticks[4] = t5 - t4; }
}
__global__ __launch_bounds__(256) void potf2_bench_kernel(const double* A, double* Lout, double* Xout, long long* ticks, int reps, int nb) {
  __shared__ double U[cs::BS][cs::BS + 1];
  __shared__ double Dl[cs::BS][cs::BS + 1];
  __shared__ double X[cs::BS][cs::BS + 1];
  __shared__ double colbuf[2 * 256];
  const int tid = threadIdx.x;
  for (int e = tid; e < cs::BS * cs::BS; e += 256) U[e >> 5][e & 31] = A[e];
  __syncthreads();
  long long esc; cs::band_clock_init(&esc);
  const long long t0 = wall_clock64();
  bool bad = false;
  for (int r = 0; r < reps; r++) {
    bad |= cs::band_potf2_inv_k0(U, nb, Dl, X, colbuf);
    __syncthreads();
  }
  const long long t1 = wall_clock64();
  if (tid == 0) { ticks[0] = t1 - t0; ticks[1] = bad; }
  for (int e = tid; e < cs::BS * cs::BS; e += 256) { Lout[e] = Dl[e >> 5][e & 31]; Xout[e] = X[e >> 5][e & 31]; }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200, nb = argc > 2 ? atoi(argv[2]) : 32;
  const int n = 32;
  std::vector<double> A(n * n), M(n * n);
  srand(7);
  for (auto& v : M) v = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = (i == j) ? 4.0 : 0.0; for (int k = 0; k < n; k++) s += M[i * n + k] * M[j * n + k]; A[i * n + j] = s; }
  double *dA, *dL, *dX; long long* dt;
  hipMalloc(&dA, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&dX, n * n * 8); hipMalloc(&dt, 16);
  hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
  {
    double* dout; long long* dtk; hipMalloc(&dout, 256 * 8); hipMalloc(&dtk, 64);
    for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(latency_kernel, dim3(1), dim3(256), 0, 0, dout, dtk, 1.0); hipDeviceSynchronize(); }
    long long t[5]; hipMemcpy(t, dtk, 40, hipMemcpyDeviceToHost);
    printf("latency (ns): dependent f64 fma %.1f  rcp+add %.1f  rsq+add %.1f  LDS write/s_barrier/read (4 waves) %.1f  LDS write/read (wave-local) %.1f\n", t[0] * 10.0 / 1024, t[1] * 10.0 / 256, t[2] * 10.0 / 256, t[3] * 10.0 / 256, t[4] * 10.0 / 256);
  }
  for (int it = 0; it < 3; it++) {
    hipLaunchKernelGGL(potf2_bench_kernel, dim3(1), dim3(256), 0, 0, dA, dL, dX, dt, reps, nb);
    hipDeviceSynchronize();
    long long t[2]; hipMemcpy(t, dt, 16, hipMemcpyDeviceToHost);
    std::vector<double> L(n * n), X(n * n);
    hipMemcpy(L.data(), dL, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(X.data(), dX, n * n * 8, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < nb; i++) for (int j = 0; j <= i; j++) { double s = 0; for (int k = 0; k <= j; k++) s += L[i * n + k] * L[j * n + k]; e1 = fmax(e1, fabs(s - A[i * n + j])); }
    for (int i = 0; i < nb; i++) for (int j = 0; j < nb; j++) { double s = 0; for (int k = 0; k < nb; k++) s += X[i * n + k] * L[k * n + j]; e2 = fmax(e2, fabs(s - (i == j))); }
    printf("potf2 + inverse: %.3f us per call (%d calls)  bad %lld  |L L^T - A| %.2e  |X L - I| %.2e\n", t[0] * 0.01 / reps, reps, t[1], e1, e2);
  }
  return 0;
}
