// potf4_bench -- the diagonal sweep of bcr_factor_kernel (potf4_*: four columns per round, pivot block and trailing operands by v_readlane) alone:
// four waves, `reps` sweeps of one 32 x 32 block out of LDS, device clock around them; checks X L = I against a host Cholesky.
// Development tool, not shipped.  Build: tools/microbench/build.sh.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include "../../cube_slam_wu_amd/csrc/bcr_kernels.hip"

#include <cmath>
#include <vector>

__global__ __launch_bounds__(512) void potf4_kernel(const double* A, double* Xout, long long* ticks, int reps, int busy_m) {
  __shared__ double U[cs::BS][cs::BS + 1];
  __shared__ double X[cs::BS][cs::BS + 1];
  __shared__ double colbuf[2 * 256];
  __shared__ double scratch[8][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ws = __builtin_amdgcn_readfirstlane(w & 3);
  const bool isP = w < 4;
  for (int e = tid; e < cs::BS * cs::BS; e += 512) U[e >> 5][e & 31] = A[e];
  __syncthreads();
  const long long t0 = wall_clock64();
  cs::Potf4 S;
  double junk = lane;
  for (int r = 0; r < reps; r++) {
    if (isP) { cs::potf4_init(S, U, lane, ws); cs::potf4_pre(S, 0, colbuf, lane, ws); }
    __syncthreads();
#pragma unroll
    for (int i0 = 0; i0 < 7; i0++) {
      if (isP) { cs::potf4_post(S, i0, colbuf, lane, ws); cs::potf4_pre(S, i0 + 1, colbuf, lane, ws); }
      else if (busy_m) {   // the other set's kind of work: LDS operand reads + matrix-core products
        cs::bcr_v4d c = cs::bcr_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 8; s++) { const double a = U[(lane & 15) + 16 * (ws & 1)][4 * s + (lane >> 4)]; c = BCR_MFMA(a, a + junk, c); }
        junk += c[0];
      }
      __syncthreads();
    }
    if (isP) { cs::potf4_post(S, 7, colbuf, lane, ws); cs::potf4_store(S, reinterpret_cast<cs::bcr_blk>(&X[0][0]), lane, ws); }
    __syncthreads();
  }
  const long long t1 = wall_clock64();
  scratch[w][lane] = junk;
  if (tid == 0) ticks[0] = t1 - t0;
  for (int e = tid; e < cs::BS * cs::BS; e += 512) Xout[e] = X[e >> 5][e & 31];
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  const int n = 32;
  std::vector<double> A(n * n), M(n * n), L(n * n, 0.0);
  srand(7);
  for (auto& v : M) v = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = (i == j) ? 4.0 : 0.0; for (int k = 0; k < n; k++) s += M[i * n + k] * M[j * n + k]; A[i * n + j] = s; }
  for (int j = 0; j < n; j++) {
    double s = A[j * n + j]; for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
    L[j * n + j] = std::sqrt(s);
    for (int i = j + 1; i < n; i++) { double t = A[i * n + j]; for (int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k]; L[i * n + j] = t / L[j * n + j]; }
  }
  double *dA, *dX; long long* dt;
  hipMalloc(&dA, n * n * 8); hipMalloc(&dX, n * n * 8); hipMalloc(&dt, 16);
  hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
  for (int busy = 0; busy < 2; busy++)
    for (int it = 0; it < 2; it++) {
      hipLaunchKernelGGL(potf4_kernel, dim3(1), dim3(512), 0, 0, dA, dX, dt, reps, busy);
      hipDeviceSynchronize();
      long long t; hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
      std::vector<double> X(n * n);
      hipMemcpy(X.data(), dX, n * n * 8, hipMemcpyDeviceToHost);
      double e2 = 0;
      for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = 0; for (int k = 0; k < n; k++) s += X[i * n + k] * L[k * n + j]; e2 = fmax(e2, fabs(s - (i == j))); }
      printf("potf4 sweep (other wave set %s): %.3f us per 32 x 32 block (%d sweeps)  |X L - I| %.2e\n", busy ? "busy on the matrix cores" : "idle", t * 0.01 / reps, reps, e2);
    }
  return 0;
}
