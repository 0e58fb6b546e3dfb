// band_potf2.h -- the diagonal-block routine the reduced-system solvers share (ba_kernels.hip: the persistent banded Cholesky;
// bcr_kernels.hip: block cyclic reduction): Cholesky factor of a 32 x 32 block and the inverse of that factor in one sweep, by one
// workgroup of 256 threads out of LDS.  Held to a tolerance, not to bit parity (the reference's reduced solve is Eigen's LDLT,
// object_slam/Thirdparty/g2o/g2o/solvers/linear_solver_dense.h:65-113).
#pragma once
#include <hip/hip_runtime.h>

namespace cs {

enum { BS = 32 };

__device__ __forceinline__ double band_rdlane(double v, int l) {   // l uniform
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// 1 / sqrt(x): hardware estimate + two Goldschmidt steps (this solver is held to a tolerance, not to bit parity)
__device__ __forceinline__ double band_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  h = fma(h, r, h);
  return h + h;
}

// 1 / x: hardware estimate + two Newton steps
__device__ __forceinline__ double band_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(e, y, y);
  e = fma(-x, y, 1.0);
  return fma(e, y, y);
}
// The whole workgroup: Cholesky of the 32 x 32 block in U and the inverse of its factor, in one sweep over the columns, four columns
// per LDS round trip.  Lanes 0..31 of every wave own row r of L, lanes 32..63 column r of L^-1 (the identity rides along as 32 more
// rows, so the trailing update of L and the forward substitution of L^-1 are the same instructions); wave w keeps the columns
// q = 4 i + w of every lane's row, i.e. one column of every round.  A round eliminates the columns c0 .. c0 + 3 together: the four
// owners write them to LDS, one workgroup barrier, then every lane reads the 4 x 4 pivot block and its own row's four entries,
// factorises the pivot block as L D L^T in its registers (a chain of four reciprocals, no square root on it), carries its own
// entries through the same elimination (x_j = m_j - sum_t x_t g_jt) and updates the columns q it owns from lane q's four raw
// entries (weights G^-T (x / d)); the owner's final values are x_w / sqrt(d_w).  colbuf: 2 x 256 doubles.
// History (tools/microbench/potf2_bench.cpp, one workgroup alone): one wave sweeping column by column with one LDS broadcast line per
// column 6.5 us per block; the same spread over four waves 7.4 us (a workgroup barrier per column costs what the shorter update
// saves); this one 4.8 us -- in the factorisation of C4's reduced system (53 dependent steps) 1.07 -> 0.99 ms.
__device__ __forceinline__ bool band_potf2_inv4b_impl(const double (*U)[BS + 1], int nb, double (*Dl)[BS + 1], double (*X)[BS + 1], double* colbuf) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = lane & 31;
  const bool lower = lane < BS;
  double v[BS / 4];
#pragma unroll
  for (int i = 0; i < BS / 4; i++) {
    const int q = 4 * i + w;
    const double u = U[row][q];
    v[i] = (lower && row < nb && q <= row) ? u : ((q == row) ? 1.0 : 0.0);
  }
  int bad = 0;
#pragma unroll
  for (int i0 = 0; i0 < BS / 4; i0++) {
    const int c0 = 4 * i0;
    double* buf = colbuf + (i0 & 1) * 256;      // [column of the round][lane]
    buf[w * 64 + lane] = v[i0];
    __syncthreads();
    double P[4][4], m[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) P[i][j] = buf[j * 64 + c0 + i];     // lane c0 + i's entry of column c0 + j (uniform address)
#pragma unroll
    for (int j = 0; j < 4; j++) m[j] = buf[j * 64 + lane];
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
    double x[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      x[j] = m[j];
#pragma unroll
      for (int t = 0; t < j; t++) x[j] = fma(-x[t], g[j][t], x[j]);
    }
    // this wave's column of the round is final: x_w / sqrt(d_w)
    const double dw = w == 0 ? d[0] : (w == 1 ? d[1] : (w == 2 ? d[2] : d[3]));
    const double xw = w == 0 ? x[0] : (w == 1 ? x[1] : (w == 2 ? x[2] : x[3]));
    v[i0] = xw * band_rsqrt(dw);
    // trailing update of the owned later columns: v_q -= sum_t (x_t / d_t) x_t(lane q), and x(lane q) = G^-1 m(lane q) with the unit lower
    // G = (g_jt) -- so the weights are carried through G^-T once (z = G^-T (x / d)) and lane q's RAW entries are used as they are
    double z[4];
#pragma unroll
    for (int t = 3; t >= 0; t--) {
      z[t] = x[t] * inv[t];
#pragma unroll
      for (int j = t + 1; j < 4; j++) z[t] = fma(-g[j][t], z[j], z[t]);
    }
#pragma unroll
    for (int i = i0 + 1; i < BS / 4; i++) {
      const int q = 4 * i + w;
      double acc = v[i];
#pragma unroll
      for (int t = 0; t < 4; t++) acc = fma(-z[t], buf[t * 64 + q], acc);       // lane q's raw entries of the round's columns (uniform address)
      v[i] = acc;
    }
  }
#pragma unroll
  for (int i = 0; i < BS / 4; i++) {
    const int q = 4 * i + w;
    if (lower) Dl[row][q] = (row - q >= 0) ? v[i] : 0.0;
    else X[q][row] = (q - row >= 0) ? v[i] : 0.0;
  }
  return bad != 0;
}
}  // namespace cs
