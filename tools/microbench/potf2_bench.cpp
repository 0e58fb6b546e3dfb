// potf2_bench.cpp -- times the banded solver's 32 x 32 diagonal-block routine (Cholesky factor + inverse of the factor) alone: one workgroup,
// `reps` calls on the same block from LDS, device clock around them; checks L L^T = A and X L = I.  Build: tools/microbench/build.sh.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../cube_slam_wu_amd/csrc/ba_kernels.hip"


// the routine's rounds under a stopwatch (a copy of band_potf2_inv4b_impl with clock reads; thread 0's accumulated phase times)
namespace cs {
__device__ __attribute__((noinline)) bool potf2_probe(const double (*U)[BS + 1], int nb, double (*Dl)[BS + 1], double (*X)[BS + 1], double* colbuf, long long* ph) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = lane & 31;
  const bool lower = lane < BS;
  double v[BS / 4];
#pragma unroll
  for (int i = 0; i < BS / 4; i++) { const int q = 4 * i + w; const double u = U[row][q]; v[i] = (lower && row < nb && q <= row) ? u : ((q == row) ? 1.0 : 0.0); }
  int bad = 0;
  long long tp = wall_clock64();
#define PH(k) do { long long tn = wall_clock64(); ph[k] += tn - tp; tp = tn; } while (0)
#pragma unroll
  for (int i0 = 0; i0 < BS / 4; i0++) {
    const int c0 = 4 * i0;
    double* buf = colbuf + (i0 & 1) * 256;
    buf[w * 64 + lane] = v[i0];
    __syncthreads();
    PH(0);
    double P[4][4], m[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) P[i][j] = buf[j * 64 + c0 + i];
#pragma unroll
    for (int j = 0; j < 4; j++) m[j] = buf[j * 64 + lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PH(1);
    double d[4], inv[4], g[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      d[j] = P[j][j];
      bad |= (c0 + j < nb) & !(d[j] > 0.0);
      inv[j] = band_rcp(d[j]);
#pragma unroll
      for (int i = j + 1; i < 4; i++) g[i][j] = P[i][j] * inv[j];
#pragma unroll
      for (int i = j + 1; i < 4; i++)
#pragma unroll
        for (int jj = j + 1; jj <= i; jj++) P[i][jj] = fma(-P[i][j], g[jj][j], P[i][jj]);
    }
    asm volatile("" :: "v"(inv[3]), "v"(g[3][2]));
    PH(2);
    double x[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { x[j] = m[j];
#pragma unroll
      for (int t = 0; t < j; t++) x[j] = fma(-x[t], g[j][t], x[j]); }
    const double dw = w == 0 ? d[0] : (w == 1 ? d[1] : (w == 2 ? d[2] : d[3]));
    const double xw = w == 0 ? x[0] : (w == 1 ? x[1] : (w == 2 ? x[2] : x[3]));
    v[i0] = xw * band_rsqrt(dw);
    double z[4];
#pragma unroll
    for (int t = 3; t >= 0; t--) { z[t] = x[t] * inv[t];
#pragma unroll
      for (int j = t + 1; j < 4; j++) z[t] = fma(-g[j][t], z[j], z[t]); }
    asm volatile("" :: "v"(z[0]), "v"(v[i0]));
    PH(3);
#pragma unroll
    for (int i = i0 + 1; i < BS / 4; i++) {
      const int q = 4 * i + w;
      double acc = v[i];
#pragma unroll
      for (int t = 0; t < 4; t++) acc = fma(-z[t], buf[t * 64 + q], acc);
      v[i] = acc;
    }
    if (i0 + 1 < BS / 4) asm volatile("" :: "v"(v[i0 + 1]));
    PH(4);
  }
#undef PH
#pragma unroll
  for (int i = 0; i < BS / 4; i++) { const int q = 4 * i + w; if (lower) Dl[row][q] = (row - q >= 0) ? v[i] : 0.0; else X[q][row] = (q - row >= 0) ? v[i] : 0.0; }
  return bad != 0;
}
}
__global__ __launch_bounds__(256) void potf2_probe_kernel(const double* A, long long* ph_out, int reps) {
  __shared__ double U[cs::BS][cs::BS + 1];
  __shared__ double Dl[cs::BS][cs::BS + 1];
  __shared__ double X[cs::BS][cs::BS + 1];
  __shared__ double colbuf[2 * 256];
  const int tid = threadIdx.x;
  for (int e = tid; e < cs::BS * cs::BS; e += 256) U[e >> 5][e & 31] = A[e];
  __syncthreads();
  long long ph[5] = {0, 0, 0, 0, 0};
  for (int r = 0; r < reps; r++) { cs::potf2_probe(U, 32, Dl, X, colbuf, ph); __syncthreads(); }
  if (tid == 0) for (int k = 0; k < 5; k++) ph_out[k] = ph[k];
}
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
  {
    long long* dph; hipMalloc(&dph, 40);
    hipLaunchKernelGGL(potf2_probe_kernel, dim3(1), dim3(256), 0, 0, dA, dph, reps); hipDeviceSynchronize();
    long long ph[5]; hipMemcpy(ph, dph, 40, hipMemcpyDeviceToHost);
    printf("per round (ns, clock reads included): write + barrier %.0f  broadcast reads %.0f  pivot block L D L^T %.0f  own entries + weights + rsqrt %.0f  trailing update %.0f\n",
           ph[0] * 10.0 / (8 * reps), ph[1] * 10.0 / (8 * reps), ph[2] * 10.0 / (8 * reps), ph[3] * 10.0 / (8 * reps), ph[4] * 10.0 / (8 * reps));
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
