// bcr_kernels.hip -- the reduced (pose) system solved by BLOCK CYCLIC REDUCTION, for gfx950 (MI355X).
//
// What it replaces: the reduced solve of g2o's BlockSolver (block_solver.hpp:441-453, `_linearSolver->solve(*_Hschur, ...)`; the reference
// constructs LinearSolverDense, solvers/linear_solver_dense.h:65-113).  The banded Cholesky of ba_kernels.hip eliminates 32 columns per
// dependent step -- 53 steps of ~13 us at C4 (n = 5 994, bandwidth 119), four fronts at once.  The arithmetic (0.2 Gflop) is nothing;
// the NUMBER of dependent steps is the cost.  Here the band is read as a block-tridiagonal matrix with blocks of Bv >= bandwidth
// unknowns (padded to B = 128) and eliminated in odd-even (nested dissection) order:
//
//   level l, blocks 0 .. N-1:  every EVEN block e is eliminated at once --
//     factor   L_e L_e^T = D_e, Linv_e = L_e^-1, y_e = Linv_e b_e                          (one workgroup per block, out of LDS)
//     panel    W_L = Linv_e A(e, e-1),  W_R = Linv_e A(e, e+1)                              (matrix cores, 16 columns per workgroup)
//     update   D'_j = D_r - W_R(r-1)^T W_R(r-1) - W_L(r+1)^T W_L(r+1),  r = 2 j + 1         (matrix cores, one 16 x 16 tile per workgroup)
//              A'(j+1, j) = - W_R(r+1)^T W_L(r+1),   b'_j = b_r - W_R(r-1)^T y - W_L(r+1)^T y
//   -- which leaves the N / 2 odd blocks as the next level's block-tridiagonal system; 47 -> 23 -> 11 -> 5 -> 2 -> 1 at C4: SIX levels
//   of one 128-column factorisation each instead of 53 steps, every level's work spread over the whole device.  Substitution runs back
//   up the levels:  x_e = Linv_e^T (y_e - W_L x_(e-1) - W_R x_(e+1)).
//
// It is the same Cholesky factorisation in another elimination order (a permutation P A P^T = L L^T), so it is as stable as the banded
// one; a non-positive pivot raises info[0] like there.  Kernels of one level are separate launches (stream order is the dependency;
// nothing here can spin or hang).
//
// Operand layouts.  v_mfma_f64_16x16x4_f64 takes A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15] and returns
// D[i = 4 g + (lane >> 4)][j = lane & 15] in component g -- so component g of a result tile IS the B (or transposed A) operand of
// depth step 4 m + g.  Two tiled layouts follow, both read and written as whole 512-byte lines:
//   TA(X)[m][s][lane] = X[16 m + (lane & 15)][4 s + (lane >> 4)]      X as the left factor       (Linv)
//   TB(X)[s][t][lane] = X[4 s + (lane >> 4)][16 t + (lane & 15)]      X as the right factor, and X^T as the left one (W_L, W_R)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "band_potf2.h"
#include "cs_hip_util.h"

namespace cs {

typedef double bcr_v4d __attribute__((ext_vector_type(4)));
enum { BCR_B = 128, BCR_NB = BCR_B / BS, BCR_NT = BCR_B / 16, BCR_NS = BCR_B / 4, BCR_BB = BCR_B * BCR_B, BCR_SLOTS = BCR_NB * (BCR_NB + 1) / 2,
       BCR_MAXLEV = 16, BCR_UPD_SLOTS = BCR_NT * (BCR_NT + 1) / 2 + BCR_NT * BCR_NT + 1 };
enum { BCR_LDS_DOUBLES = BCR_SLOTS * BS * (BS + 1) + 2 * BS * (BS + 1) + 512 + 2 * BCR_B };
typedef double (*bcr_blk)[BS + 1];

struct BcrLevel {
  int N, lvl;                 // blocks of this level; level index
  int packed;                 // level 0 only: D / YL / YR / b were filled by the caller (block form), there is no band
  // level 0 reads the band: S(r, c), r >= c, at Sb[c * LD + (r - c)]; block k holds the unknowns [k Bv, k Bv + Bv) (the last one fewer)
  const double* Sb; int LD, n, Bv;
  double* rhs;                // level-0 right-hand side in, solution out
  // this level's system (levels >= 1; level 0: D and YL / YR are filled by the pack workgroups of the factor launch)
  double* D;                  // N x B x B row-major, lower triangle (and whole diagonal tiles) valid
  double* YL; double* YR;     // per eliminated block: A(e, e-1), A(e, e+1) as B x B row-major [row of e][column of the neighbour]
  double* b;                  // N x B (levels >= 1)
  // what the elimination of this level produces
  double* Linv;               // per eliminated block, TA layout
  double* WL; double* WR;     // per eliminated block, TB layout
  double* y;                  // per eliminated block
  // next level's system
  double* Dn; double* YLn; double* YRn; double* bn;
  int* info;
};

__device__ __forceinline__ int bcr_valid_rows(const BcrLevel& P, int k) {    // level 0: unknowns of block k
  const int r = P.n - k * P.Bv;
  return r < P.Bv ? (r > 0 ? r : 0) : P.Bv;
}
// level-0 entry S(block k row R, block k column C), R >= C, padded with the identity
__device__ __forceinline__ double bcr_band_diag(const BcrLevel& P, int k, int vr, int R, int C) {
  if (R >= vr) return R == C ? 1.0 : 0.0;
  const int off = R - C;
  return off < P.LD ? P.Sb[(size_t)(k * P.Bv + C) * P.LD + off] : 0.0;
}
__device__ __forceinline__ double bcr_b_at(const BcrLevel& P, int k, int r) {
  if (P.lvl > 0 || P.packed) return P.b[(size_t)k * BCR_B + r];
  return r < bcr_valid_rows(P, k) ? P.rhs[k * P.Bv + r] : 0.0;
}

__device__ __forceinline__ double bcr_rowop(bcr_blk M, int t, int s, int lane) { return M[16 * t + (lane & 15)][4 * s + (lane >> 4)]; }
__device__ __forceinline__ double bcr_colop(bcr_blk M, int s, int t, int lane) { return M[4 * s + (lane >> 4)][16 * t + (lane & 15)]; }
__device__ __forceinline__ bcr_v4d bcr_ctile_load(bcr_blk M, int ti, int tj, int lane) {
  bcr_v4d c;
#pragma unroll
  for (int g = 0; g < 4; g++) c[g] = M[16 * ti + 4 * g + (lane >> 4)][16 * tj + (lane & 15)];
  return c;
}
__device__ __forceinline__ void bcr_ctile_store(bcr_blk M, int ti, int tj, int lane, bcr_v4d c) {
#pragma unroll
  for (int g = 0; g < 4; g++) M[16 * ti + 4 * g + (lane >> 4)][16 * tj + (lane & 15)] = c[g];
}
#define BCR_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__device__ __attribute__((noinline)) bool bcr_potf2_inv(const double (*U)[BS + 1], int nb, double (*Dl)[BS + 1], double (*X)[BS + 1], double* colbuf) { return band_potf2_inv4b_impl(U, nb, Dl, X, colbuf); }

// ---------------------------------------------------------------------------------------------------------------- level 0: pack --
// The couplings of the eliminated blocks and the diagonal blocks of the remaining ones, out of the band into plain row-major blocks
// (the panel / update kernels then read one layout at every level).  Runs beside the factor workgroups of level 0, off the chain.
__device__ void bcr_pack(const BcrLevel& P, int p) {
  const int k = p >> 3, which = (p >> 2) & 1, q = p & 3, tid = threadIdx.x;
  if (k >= P.N) return;
  const int Bv = P.Bv, LD = P.LD;
  const double* __restrict__ Sb = P.Sb;
  double v[16];
  if ((k & 1) == 0) {
    const int je = k >> 1;
    if (which == 0) {
      if (k == 0) return;
      double* __restrict__ Y = P.YL + (size_t)je * BCR_BB;
      const int vr = bcr_valid_rows(P, k);
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int idx = tid + 256 * u, c = idx & (BCR_B - 1), kr = 32 * q + (idx >> 7);
        const int off = Bv + kr - c;
        v[u] = (kr < vr && c < Bv && off < LD) ? Sb[(size_t)((k - 1) * Bv + c) * LD + off] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 16; u++) { const int idx = tid + 256 * u; Y[(32 * q + (idx >> 7)) * BCR_B + (idx & (BCR_B - 1))] = v[u]; }
    } else {
      if (k + 1 >= P.N) return;
      double* __restrict__ Y = P.YR + (size_t)je * BCR_BB;
      const int vrn = bcr_valid_rows(P, k + 1);
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int idx = tid + 256 * u, c = idx & (BCR_B - 1), kr = 32 * q + (idx >> 7);
        const int off = Bv + c - kr;
        v[u] = (c < vrn && kr < Bv && off < LD) ? Sb[(size_t)(k * Bv + kr) * LD + off] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 16; u++) { const int idx = tid + 256 * u; Y[(32 * q + (idx >> 7)) * BCR_B + (idx & (BCR_B - 1))] = v[u]; }
    }
  } else if (which == 0) {
    double* __restrict__ D = P.D + (size_t)k * BCR_BB;
    const int vr = bcr_valid_rows(P, k);
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int idx = tid + 256 * u, C = idx & (BCR_B - 1), R = 32 * q + (idx >> 7);
      v[u] = R >= C ? bcr_band_diag(P, k, vr, R, C) : bcr_band_diag(P, k, vr, C, R);
    }
#pragma unroll
    for (int u = 0; u < 16; u++) { const int idx = tid + 256 * u; D[(32 * q + (idx >> 7)) * BCR_B + (idx & (BCR_B - 1))] = v[u]; }
  }
}

// -------------------------------------------------------------------------------------------------------------------- factor --
// One workgroup per eliminated block: D = L L^T and Linv = L^-1, blocked right-looking over 32-column steps, all of it in LDS.  The
// lower block triangle lives in ten 32 x 32 slots; slot (i, j) holds A_ij -> L_ij (after the panel of step j) -> T_ij = sum_m L_im Linv_mj
// (accumulated by the steps m = j .. i-1) -> Linv_ij = -Linv_ii T_ij (step i).  Step k:
//   a  POTF2 + inverse of slot (k, k) (band_potf2.h)                                       -> X = Linv_kk
//   b  Linv_kj = -X T_kj (j < k);   d  L_ik = A_ik X^T (i > k);   slot (k, k) = X            (in place: results held back over a barrier)
//   e  A_ij -= L_ik L_jk^T (i >= j > k);   f' T_ij += L_ik Linv_kj (i > k > j)
//   f  T_ik = L_ik X (i > k)                                                                (in place)
// Wave w owns the 16 x 16 tile (w >> 1, w & 1) of every 32 x 32 block it is handed; operands come out of LDS eight depth steps at a time.
#define BCR_SLOT(i, j) (reinterpret_cast<bcr_blk>(slots + ((i) * ((i) + 1) / 2 + (j)) * (BS * (BS + 1))))
__device__ __forceinline__ void bcr_row8(bcr_blk M, int t, int lane, double (&a)[8]) {
#pragma unroll
  for (int s = 0; s < 8; s++) a[s] = M[16 * t + (lane & 15)][4 * s + (lane >> 4)];
}
__device__ __forceinline__ void bcr_col8(bcr_blk M, int t, int lane, double (&b)[8]) {
#pragma unroll
  for (int s = 0; s < 8; s++) b[s] = M[4 * s + (lane >> 4)][16 * t + (lane & 15)];
}
template <int K>
__device__ __forceinline__ void bcr_factor_step(double* slots, bcr_blk X, int tid, int lane, int ti, int tj) {
  // ---- b + d: three blocks (K of kind b, 3 - K of kind d), results kept in registers over the barrier
  {
    bcr_v4d acc[BCR_NB - 1];
    double xa[8], xb[8];
    bcr_row8(X, ti, lane, xa);
    bcr_row8(X, tj, lane, xb);
#pragma unroll
    for (int j = 0; j < K; j++) {
      double tc[8];
      bcr_col8(BCR_SLOT(K, j), tj, lane, tc);
      acc[j] = bcr_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 8; s++)
        if (s < 4 * (ti + 1)) acc[j] = BCR_MFMA(-xa[s], tc[s], acc[j]);
    }
#pragma unroll
    for (int i = K + 1; i < BCR_NB; i++) {
      double ar[8];
      bcr_row8(BCR_SLOT(i, K), ti, lane, ar);
      acc[i - 1] = bcr_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 8; s++)
        if (s < 4 * (tj + 1)) acc[i - 1] = BCR_MFMA(ar[s], xb[s], acc[i - 1]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < K; j++) bcr_ctile_store(BCR_SLOT(K, j), ti, tj, lane, acc[j]);
#pragma unroll
    for (int i = K + 1; i < BCR_NB; i++) bcr_ctile_store(BCR_SLOT(i, K), ti, tj, lane, acc[i - 1]);
    bcr_blk Skk = BCR_SLOT(K, K);
#pragma unroll
    for (int q = 0; q < 4; q++) { const int idx = tid + 256 * q; Skk[idx >> 5][idx & 31] = X[idx >> 5][idx & 31]; }
    __syncthreads();
  }
  if (K + 1 < BCR_NB) {
    // ---- e + f' (no slot both read and written) and f (in place: held back over the barrier)
    bcr_v4d facc[BCR_NB - 1];
    double xc[8];
    bcr_col8(X, tj, lane, xc);
#pragma unroll
    for (int i = K + 1; i < BCR_NB; i++) {
      double ar[8];
      bcr_row8(BCR_SLOT(i, K), ti, lane, ar);
#pragma unroll
      for (int j = K + 1; j <= i; j++) {
        double br[8];
        bcr_row8(BCR_SLOT(j, K), tj, lane, br);
        bcr_v4d c = bcr_ctile_load(BCR_SLOT(i, j), ti, tj, lane);
#pragma unroll
        for (int s = 0; s < 8; s++) c = BCR_MFMA(-ar[s], br[s], c);
        bcr_ctile_store(BCR_SLOT(i, j), ti, tj, lane, c);
      }
#pragma unroll
      for (int j = 0; j < K; j++) {
        double bc[8];
        bcr_col8(BCR_SLOT(K, j), tj, lane, bc);
        bcr_v4d c = bcr_ctile_load(BCR_SLOT(i, j), ti, tj, lane);
#pragma unroll
        for (int s = 0; s < 8; s++) c = BCR_MFMA(ar[s], bc[s], c);
        bcr_ctile_store(BCR_SLOT(i, j), ti, tj, lane, c);
      }
      facc[i - 1] = bcr_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 8; s++)
        if (s >= 4 * tj) facc[i - 1] = BCR_MFMA(ar[s], xc[s], facc[i - 1]);
    }
    __syncthreads();
#pragma unroll
    for (int i = K + 1; i < BCR_NB; i++) bcr_ctile_store(BCR_SLOT(i, K), ti, tj, lane, facc[i - 1]);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void bcr_factor_kernel(BcrLevel P, int n_elim) {
  if ((int)blockIdx.x >= n_elim) { bcr_pack(P, blockIdx.x - n_elim); return; }
  extern __shared__ double bcr_lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int je = blockIdx.x, e = 2 * je;
  double* slots = bcr_lds;
  bcr_blk Dl = reinterpret_cast<bcr_blk>(bcr_lds + BCR_SLOTS * BS * (BS + 1));
  bcr_blk X = reinterpret_cast<bcr_blk>(bcr_lds + (BCR_SLOTS + 1) * BS * (BS + 1));
  double* colbuf = bcr_lds + (BCR_SLOTS + 2) * BS * (BS + 1);
  double* bv = colbuf + 512;
  double* part = bv + BCR_B;

  // ---- load D_e (the ten lower blocks: 40 entries per thread, all in flight) and b_e
  {
    double v[BCR_SLOTS][4];
    if (P.lvl == 0 && !P.packed) {
      const int vr = bcr_valid_rows(P, e);
#pragma unroll
      for (int i = 0; i < BCR_NB; i++)
#pragma unroll
        for (int j = 0; j <= i; j++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int idx = tid + 256 * q, R = 32 * i + (idx & 31), C = 32 * j + (idx >> 5);
            v[i * (i + 1) / 2 + j][q] = R >= C ? bcr_band_diag(P, e, vr, R, C) : 0.0;
          }
#pragma unroll
      for (int i = 0; i < BCR_NB; i++)
#pragma unroll
        for (int j = 0; j <= i; j++)
#pragma unroll
          for (int q = 0; q < 4; q++) { const int idx = tid + 256 * q; BCR_SLOT(i, j)[idx & 31][idx >> 5] = v[i * (i + 1) / 2 + j][q]; }
    } else {
      const double* __restrict__ D = P.D + (size_t)e * BCR_BB;
#pragma unroll
      for (int i = 0; i < BCR_NB; i++)
#pragma unroll
        for (int j = 0; j <= i; j++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int idx = tid + 256 * q;
            v[i * (i + 1) / 2 + j][q] = D[(32 * i + (idx >> 5)) * BCR_B + 32 * j + (idx & 31)];
          }
#pragma unroll
      for (int i = 0; i < BCR_NB; i++)
#pragma unroll
        for (int j = 0; j <= i; j++)
#pragma unroll
          for (int q = 0; q < 4; q++) { const int idx = tid + 256 * q; BCR_SLOT(i, j)[idx >> 5][idx & 31] = v[i * (i + 1) / 2 + j][q]; }
    }
  }
  if (tid < BCR_B) bv[tid] = bcr_b_at(P, e, tid);
  __syncthreads();

  const int ti = (w >> 1) & 1, tj = w & 1;
  bool bad = false;
  bad |= bcr_potf2_inv(BCR_SLOT(0, 0), BS, Dl, X, colbuf);
  __syncthreads();
  bcr_factor_step<0>(slots, X, tid, lane, ti, tj);
  bad |= bcr_potf2_inv(BCR_SLOT(1, 1), BS, Dl, X, colbuf);
  __syncthreads();
  bcr_factor_step<1>(slots, X, tid, lane, ti, tj);
  bad |= bcr_potf2_inv(BCR_SLOT(2, 2), BS, Dl, X, colbuf);
  __syncthreads();
  bcr_factor_step<2>(slots, X, tid, lane, ti, tj);
  bad |= bcr_potf2_inv(BCR_SLOT(3, 3), BS, Dl, X, colbuf);
  __syncthreads();
  bcr_factor_step<3>(slots, X, tid, lane, ti, tj);
  static_assert(BCR_NB == 4, "four 32-column steps");
  if (bad && tid == 0) atomicMax(P.info, e + 1);

  // ---- Linv in the TA layout (row tile m: depth steps s < 4 (m + 1)), y = Linv b
  double* __restrict__ TA = P.Linv + (size_t)je * BCR_BB;
#pragma unroll
  for (int m = 0; m < BCR_NT; m++)
#pragma unroll
    for (int u = 0; u < m + 1; u++) {
      const int s = 4 * u + w;
      const int R = 16 * m + (lane & 15), C = 4 * s + (lane >> 4);
      TA[(size_t)(m * BCR_NS + s) * 64 + lane] = (&BCR_SLOT(m >> 1, 0)[R & 31][0])[(C >> 5) * (BS * (BS + 1)) + (C & 31)];   // (the slots of a block row are contiguous)
    }
  {
    const int R = tid & (BCR_B - 1), h = tid >> 7;
    const double* rowp = &BCR_SLOT(R >> 5, 0)[R & 31][0];
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
    for (int C = 64 * h; C < 64 * h + 64; C += 2) {
      const double l0 = C <= R ? rowp[(C >> 5) * (BS * (BS + 1)) + (C & 31)] : 0.0;
      const double l1 = C + 1 <= R ? rowp[((C + 1) >> 5) * (BS * (BS + 1)) + ((C + 1) & 31)] : 0.0;
      s0 = fma(l0, bv[C], s0); s1 = fma(l1, bv[C + 1], s1);
    }
    if (h) part[R] = s0 + s1;
    __syncthreads();
    if (!h) P.y[(size_t)je * BCR_B + R] = (s0 + s1) + part[R];
  }
}

// --------------------------------------------------------------------------------------------------------------------- panel --
// W = Linv Y for one side of one eliminated block, 16 columns per workgroup: wave w takes the row tiles w and 7 - w (36 depth steps
// together, Linv being lower triangular).  Every operand is requested before the first product.  Result straight into the TB layout.
__global__ __launch_bounds__(256) void bcr_panel_kernel(BcrLevel P) {
  const int t = blockIdx.x & 7, side = (blockIdx.x >> 3) & 1, je = blockIdx.x >> 4, e = 2 * je;
  if (side == 0 ? e == 0 : e + 1 >= P.N) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const double* __restrict__ Y = (side ? P.YR : P.YL) + (size_t)je * BCR_BB;
  const double* __restrict__ TA = P.Linv + (size_t)je * BCR_BB;
  double* __restrict__ W = (side ? P.WR : P.WL) + (size_t)je * BCR_BB;
  const int m0 = w, m1 = BCR_NT - 1 - w;
  double bop[BCR_NS], a0[16], a1[BCR_NS];
  // (row tile m0 = w needs s < 4 (w + 1) <= 16; row tile m1 = 7 - w needs s < 4 (8 - w))
#pragma unroll
  for (int s = 0; s < BCR_NS; s++) bop[s] = Y[(4 * s + (lane >> 4)) * BCR_B + 16 * t + (lane & 15)];
#pragma unroll
  for (int s = 0; s < 16; s++) a0[s] = TA[(size_t)(m0 * BCR_NS + (s < 4 * (m0 + 1) ? s : 0)) * 64 + lane];
#pragma unroll
  for (int s = 0; s < BCR_NS; s++) a1[s] = TA[(size_t)(m1 * BCR_NS + (s < 4 * (m1 + 1) ? s : 0)) * 64 + lane];
  bcr_v4d c0 = bcr_v4d{0.0, 0.0, 0.0, 0.0}, c1 = bcr_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < BCR_NS; s++) {
    if (s < 16 && s < 4 * (m0 + 1)) c0 = BCR_MFMA(a0[s & 15], bop[s], c0);
    if (s < 4 * (m1 + 1)) c1 = BCR_MFMA(a1[s], bop[s], c1);
  }
#pragma unroll
  for (int g = 0; g < 4; g++) {
    W[((size_t)(4 * m0 + g) * BCR_NT + t) * 64 + lane] = c0[g];
    W[((size_t)(4 * m1 + g) * BCR_NT + t) * 64 + lane] = c1[g];
  }
}

// -------------------------------------------------------------------------------------------------------------------- update --
// Next level's system.  Per remaining block r = 2 j + 1 (next-level block j): 36 tiles of D'_j, 64 tiles of the coupling to the next
// remaining block, one task for b'_j -- one workgroup each, the depth of 128 split over its four waves and summed through LDS in a
// fixed order.
__global__ __launch_bounds__(256) void bcr_update_kernel(BcrLevel P) {
  __shared__ double red[4][4][64];
  const int j = blockIdx.x / BCR_UPD_SLOTS, sl = blockIdx.x % BCR_UPD_SLOTS;
  const int r = 2 * j + 1;
  const bool has_r = r + 1 < P.N;              // the eliminated block behind r
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ND = BCR_NT * (BCR_NT + 1) / 2;
  const double* __restrict__ WRl = P.WR + (size_t)j * BCR_BB;          // W_R of e = 2 j
  const double* __restrict__ WLr = P.WL + (size_t)(j + 1) * BCR_BB;    // W_L of e = 2 j + 2
  const double* __restrict__ WRr = P.WR + (size_t)(j + 1) * BCR_BB;    // W_R of e = 2 j + 2
  if (sl < ND) {
    int a = 0, c = sl;
    while (c > a) { c -= a + 1; a++; }
    const int b = c;                           // tile (a, b), a >= b
    const double* __restrict__ Dr = P.D + (size_t)r * BCR_BB;
    double* __restrict__ Dn = P.Dn + (size_t)j * BCR_BB;
    double d[4], pa[8], pb[8], qa[8], qb[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { const int s = 8 * w + q; pa[q] = WRl[((size_t)s * BCR_NT + a) * 64 + lane]; pb[q] = WRl[((size_t)s * BCR_NT + b) * 64 + lane]; }
#pragma unroll
    for (int q = 0; q < 8; q++) { const int s = 8 * w + q; qa[q] = has_r ? WLr[((size_t)s * BCR_NT + a) * 64 + lane] : 0.0; qb[q] = has_r ? WLr[((size_t)s * BCR_NT + b) * 64 + lane] : 0.0; }
    if (w == 0)
#pragma unroll
      for (int g = 0; g < 4; g++) d[g] = Dr[(16 * a + 4 * g + (lane >> 4)) * BCR_B + 16 * b + (lane & 15)];
    bcr_v4d acc = bcr_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 8; q++) acc = BCR_MFMA(pa[q], pb[q], acc);
    if (has_r)
#pragma unroll
      for (int q = 0; q < 8; q++) acc = BCR_MFMA(qa[q], qb[q], acc);
#pragma unroll
    for (int g = 0; g < 4; g++) red[w][g][lane] = acc[g];
    __syncthreads();
    if (w == 0)
#pragma unroll
      for (int g = 0; g < 4; g++)
        Dn[(16 * a + 4 * g + (lane >> 4)) * BCR_B + 16 * b + (lane & 15)] = d[g] - ((red[0][g][lane] + red[1][g][lane]) + (red[2][g][lane] + red[3][g][lane]));
  } else if (sl < ND + BCR_NT * BCR_NT) {
    if (!has_r || r + 2 >= P.N) return;        // no next remaining block
    const int a = (sl - ND) >> 3, b = (sl - ND) & 7;
    // next-level blocks j and j + 1: the even one is eliminated there and owns the coupling
    const bool left_of_next = ((j + 1) & 1) == 0;          // YL of next-level block j + 1: [row of j + 1][column of j] = - W_R^T W_L
    const double* __restrict__ Aop = left_of_next ? WRr : WLr;
    const double* __restrict__ Bop = left_of_next ? WLr : WRr;
    double pa[8], pb[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { const int s = 8 * w + q; pa[q] = Aop[((size_t)s * BCR_NT + a) * 64 + lane]; pb[q] = Bop[((size_t)s * BCR_NT + b) * 64 + lane]; }
    bcr_v4d acc = bcr_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 8; q++) acc = BCR_MFMA(pa[q], pb[q], acc);
#pragma unroll
    for (int g = 0; g < 4; g++) red[w][g][lane] = acc[g];
    __syncthreads();
    if (w == 0) {
      double* __restrict__ Yn = left_of_next ? P.YLn + (size_t)((j + 1) >> 1) * BCR_BB : P.YRn + (size_t)(j >> 1) * BCR_BB;
#pragma unroll
      for (int g = 0; g < 4; g++)
        Yn[(16 * a + 4 * g + (lane >> 4)) * BCR_B + 16 * b + (lane & 15)] = -((red[0][g][lane] + red[1][g][lane]) + (red[2][g][lane] + red[3][g][lane]));
    }
  } else {
    // b'_j = b_r - W_R(2j)^T y(2j) - W_L(2j+2)^T y(2j+2): thread (c, h, g) sums the rows q = 4 g .. of side h for column c
    double* sh = &red[0][0][0];
    const int c = tid & (BCR_B - 1), h = tid >> 7;
    double sum = 0.0;
    if (h == 0 || has_r) {
      const double* __restrict__ W = h ? WLr : WRl;
      const double* __restrict__ y = P.y + (size_t)(j + h) * BCR_B;
      double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
      for (int q = 0; q < BCR_B; q += 4)
#pragma unroll
        for (int u = 0; u < 4; u++) s4[u] = fma(W[((size_t)(q >> 2) * BCR_NT + (c >> 4)) * 64 + u * 16 + (c & 15)], y[q + u], s4[u]);
      sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    if (h) sh[c] = sum;
    __syncthreads();
    if (!h) P.bn[(size_t)j * BCR_B + c] = bcr_b_at(P, r, c) - (sum + sh[c]);
  }
}

// -------------------------------------------------------------------------------------------------------------- substitution --
// x_e = Linv_e^T (y_e - W_L x_(e-1) - W_R x_(e+1)) for the eliminated blocks of one level, the neighbours' x already in rhs.
// Level-l block k is level-0 block (k + 1) 2^l - 1.  512 threads; every operand of both products is requested up front.
__global__ __launch_bounds__(512) void bcr_back_kernel(BcrLevel P) {
  __shared__ double xn[2][BCR_B], tv[BCR_B];
  const int je = blockIdx.x, e = 2 * je;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;      // 8 waves: wave w owns the depth steps / columns s = 4 w .. 4 w + 3
  const bool hasL = e > 0, hasR = e + 1 < P.N;
  const double* __restrict__ WL = P.WL + (size_t)je * BCR_BB;
  const double* __restrict__ WR = P.WR + (size_t)je * BCR_BB;
  const double* __restrict__ TA = P.Linv + (size_t)je * BCR_BB;
  const double* __restrict__ y = P.y + (size_t)je * BCR_B;
  if (tid < 2 * BCR_B) {
    const int side = tid >> 7, c = tid & (BCR_B - 1);
    const bool has = side ? hasR : hasL;
    double v = 0.0;
    if (has) {
      const int g = ((e + (side ? 1 : -1) + 1) << P.lvl) - 1;
      if (c < P.Bv && g * P.Bv + c < P.n) v = P.rhs[g * P.Bv + c];
    }
    xn[side][c] = v;
  }
  double wl[4][BCR_NT], wr[4][BCR_NT], la[4][BCR_NT], yv[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int s = 4 * w + q;
#pragma unroll
    for (int t = 0; t < BCR_NT; t++) {
      wl[q][t] = hasL ? WL[((size_t)s * BCR_NT + t) * 64 + lane] : 0.0;
      wr[q][t] = hasR ? WR[((size_t)s * BCR_NT + t) * 64 + lane] : 0.0;
      la[q][t] = t >= (s >> 2) ? TA[((size_t)t * BCR_NS + s) * 64 + lane] : 0.0;     // row tile m = t of column step s
    }
    yv[q] = y[4 * s + (lane >> 4)];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int s = 4 * w + q;
    double p = 0.0;
#pragma unroll
    for (int t = 0; t < BCR_NT; t++) p = fma(wl[q][t], xn[0][16 * t + (lane & 15)], p);
#pragma unroll
    for (int t = 0; t < BCR_NT; t++) p = fma(wr[q][t], xn[1][16 * t + (lane & 15)], p);
    p += __shfl_xor(p, 1); p += __shfl_xor(p, 2); p += __shfl_xor(p, 4); p += __shfl_xor(p, 8);
    if ((lane & 15) == 0) tv[4 * s + (lane >> 4)] = yv[q] - p;
  }
  __syncthreads();
  const int g0 = ((e + 1) << P.lvl) - 1;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int s = 4 * w + q;                    // columns 4 s .. 4 s + 3
    double p = 0.0;
#pragma unroll
    for (int m = 0; m < BCR_NT; m++) p = fma(la[q][m], tv[16 * m + (lane & 15)], p);
    p += __shfl_xor(p, 1); p += __shfl_xor(p, 2); p += __shfl_xor(p, 4); p += __shfl_xor(p, 8);
    const int c = 4 * s + (lane >> 4);
    if ((lane & 15) == 0 && c < P.Bv && g0 * P.Bv + c < P.n) P.rhs[g0 * P.Bv + c] = p;
  }
}

// ---------------------------------------------------------------------------------------------------------------------- host --
// Block size for a band: any Bv with bandwidth <= Bv <= 128 makes the band block tridiagonal; 128 gives the fewest blocks.
// (measured on MI355X, tools/bcr_try.sh: 640 unknowns at bandwidth 119 0.17 ms against the banded kernels' 0.20, 1 194: 0.23 / 0.35, 5 994: 0.36 / 0.87,
// 10 494: 0.57 / 1.46; narrow bands of a few hundred unknowns stay with the banded kernels -- ba_bcr_estimate_ms is what ba_host.cpp compares)
// The factor kernel's LDS (107 520 bytes) is above the 64 KB a kernel gets by default: raised once per DEVICE -- the attribute belongs to the
// current device's instance of the kernel (a handle on a second GPU, sharded ranks as threads on several devices).  Both ..._ok() below
// are asked by the structure phase on the handle's device; a refusal there keeps the handle on the persistent banded kernels.
static DynLdsOnce g_bcr_factor_lds;
static bool bcr_device_ready() { return g_bcr_factor_lds.set(reinterpret_cast<const void*>(bcr_factor_kernel), BCR_LDS_DOUBLES * (int)sizeof(double)); }
bool ba_bcr_ok(int n, int LD) {
  static const int on = getenv("CS_BAND_BCR") ? atoi(getenv("CS_BAND_BCR")) : 1;
  // (every level halves the block count: BCR_MAXLEV levels reach one block from at most 2^BCR_MAXLEV - 1 -- a larger system would stop short of it)
  const long long blocks = ((long long)n + BCR_B - 1) / BCR_B;
  return on != 0 && LD - 1 >= 1 && LD - 1 <= BCR_B && n > BCR_B && blocks < (1ll << BCR_MAXLEV) && bcr_device_ready();
}
static int bcr_levels(int n, int Bv, int* Ns) {
  int N = (n + Bv - 1) / Bv, L = 0;
  while (N >= 1 && L < BCR_MAXLEV) { Ns[L++] = N; N = N / 2; }
  return L;
}
double ba_bcr_estimate_ms(int n) {
  int Ns[BCR_MAXLEV];
  const int L = bcr_levels(n, BCR_B, Ns);
  return 0.048 * L + 0.045 + 2e-6 * n;
}
size_t ba_bcr_workspace_doubles(int n, int Bv) {
  int Ns[BCR_MAXLEV];
  const int L = bcr_levels(n, Bv, Ns);
  size_t tot = 0;
  for (int l = 0; l < L; l++) {
    const size_t N = Ns[l], ne = (N + 1) / 2;
    tot += N * BCR_BB + 2 * ne * BCR_BB + N * BCR_B + ne * BCR_BB + 2 * ne * BCR_BB + ne * BCR_B;
  }
  return tot + 64;
}
// Factorise and solve: Sb = the lower band (LD doubles per column), block size Bv (bandwidth <= Bv <= 128, or a matrix that is block
// tridiagonal in blocks of Bv whatever its band storage), rhs in / solution out, info[0] != 0: a non-positive pivot.  The band itself
// is left untouched.
static void bcr_run(const double* Sb, double* work, int n, int LD, int Bv, double* rhs, int* info, hipStream_t st, bool packed, void (*fill)(const BcrLevel&, void*, hipStream_t), void* fill_arg) {
  (void)bcr_device_ready();     // (a table look-up after the structure phase's ba_bcr_ok / ba_bcr_sep_ok; covers callers that come here on another device)
  int Ns[BCR_MAXLEV];
  const int L = bcr_levels(n, Bv, Ns);
  BcrLevel lev[BCR_MAXLEV];
  double* wp = work;
  for (int l = 0; l < L; l++) {
    BcrLevel& P = lev[l];
    const size_t N = Ns[l], ne = (N + 1) / 2;
    P.N = (int)N; P.lvl = l; P.packed = (l == 0 && packed) ? 1 : 0; P.Sb = Sb; P.LD = LD; P.n = n; P.Bv = Bv; P.rhs = rhs; P.info = info;
    P.D = wp; wp += N * BCR_BB;
    P.YL = wp; wp += ne * BCR_BB;
    P.YR = wp; wp += ne * BCR_BB;
    P.b = wp; wp += N * BCR_B;
    P.Linv = wp; wp += ne * BCR_BB;
    P.WL = wp; wp += ne * BCR_BB;
    P.WR = wp; wp += ne * BCR_BB;
    P.y = wp; wp += ne * BCR_B;
    P.Dn = P.YLn = P.YRn = P.bn = nullptr;
  }
  for (int l = 0; l + 1 < L; l++) { lev[l].Dn = lev[l + 1].D; lev[l].YLn = lev[l + 1].YL; lev[l].YRn = lev[l + 1].YR; lev[l].bn = lev[l + 1].b; }
  if (packed && L > 0) fill(lev[0], fill_arg, st);
  for (int l = 0; l < L; l++) {
    const BcrLevel& P = lev[l];
    const int ne = (P.N + 1) / 2, nr = P.N / 2;
    hipLaunchKernelGGL(bcr_factor_kernel, dim3(ne + ((l == 0 && !packed) ? 8 * P.N : 0)), dim3(256), BCR_LDS_DOUBLES * sizeof(double), st, P, ne);
    if (nr > 0) {
      hipLaunchKernelGGL(bcr_panel_kernel, dim3(16 * ne), dim3(256), 0, st, P);
      hipLaunchKernelGGL(bcr_update_kernel, dim3(nr * BCR_UPD_SLOTS), dim3(256), 0, st, P);
    }
  }
  for (int l = L - 1; l >= 0; l--) {
    const BcrLevel& P = lev[l];
    hipLaunchKernelGGL(bcr_back_kernel, dim3((P.N + 1) / 2), dim3(512), 0, st, P);
  }
}
void ba_launch_bcr(const double* Sb, double* work, int n, int LD, int Bv, double* rhs, int* info, hipStream_t st) {
  bcr_run(Sb, work, n, LD, Bv, rhs, info, st, false, nullptr, nullptr);
}

// ------------------------------------------------------------------------- the sharded solve's separator system, in block form --
// Separator mode of the sharded BA (ba_kernels.hip, "sharded reduced solve"): after the all-gather every rank holds the R messages
// [LL | RL | RR | tL | tR] (three wm x wm blocks, two wm-vectors) and solves the block-tridiagonal system of the R - 1 separators:
//   D_k = LL(rank k + 1) + RR(rank k),   A(k + 1, k) = RL(rank k + 1),   b_k = tL(rank k + 1) + tR(rank k)        (block k = separator Z_(k+1))
// The blocks (<= 128 wide, padded with the identity) go straight into the level-0 arrays; x comes back as N x 128 padded blocks.
struct BcrSepSrc { const double* msgs; size_t msg_doubles; int wm, R, ns; const int* sep_off; };
__global__ __launch_bounds__(256) void bcr_pack_sep_kernel(BcrLevel P, BcrSepSrc S) {
  const int k = blockIdx.x >> 4, which = (blockIdx.x >> 2) & 3, q = blockIdx.x & 3, tid = threadIdx.x;
  if (k >= P.N) return;
  auto width = [&](int z) { return (z + 1 < S.R ? S.sep_off[z + 1] : S.ns) - S.sep_off[z]; };     // of separator Z_z, z = 1 .. R - 1
  const int wm = S.wm;
  const size_t ww = (size_t)wm * wm;
  const int z = k + 1, w = width(z);
  const double* mz = S.msgs + (size_t)z * S.msg_doubles;          // message of rank z: Z_z is its left separator
  const double* mp = S.msgs + (size_t)(z - 1) * S.msg_doubles;    // message of rank z - 1: Z_z is its right separator
  if (which == 0) {
    double* D = P.D + (size_t)k * BCR_BB;
    for (int u = 0; u < 16; u++) {
      const int idx = tid + 256 * u, c = idx & (BCR_B - 1), r = 32 * q + (idx >> 7);
      const int lr = r >= c ? r : c, lc = r >= c ? c : r;
      D[r * BCR_B + c] = (lr < w) ? mz[(size_t)lr * wm + lc] + mp[2 * ww + (size_t)lr * wm + lc] : (r == c ? 1.0 : 0.0);
    }
  } else if (which == 1) {
    if ((k & 1) || k == 0) return;                                 // YL of an eliminated block: A(k, k - 1), rows Z_z, columns Z_(z-1) = RL of rank z - 1
    double* Y = P.YL + (size_t)(k >> 1) * BCR_BB;
    const int wc = width(z - 1);
    for (int u = 0; u < 16; u++) {
      const int idx = tid + 256 * u, c = idx & (BCR_B - 1), r = 32 * q + (idx >> 7);
      Y[r * BCR_B + c] = (r < w && c < wc) ? mp[ww + (size_t)r * wm + c] : 0.0;
    }
  } else if (which == 2) {
    if ((k & 1) || k + 1 >= P.N) return;                           // YR: A(k, k + 1) = A(k + 1, k)^T, A(k + 1, k) = RL of rank z (rows Z_(z+1), columns Z_z)
    double* Y = P.YR + (size_t)(k >> 1) * BCR_BB;
    const int wn = width(z + 1);
    for (int u = 0; u < 16; u++) {
      const int idx = tid + 256 * u, c = idx & (BCR_B - 1), r = 32 * q + (idx >> 7);
      Y[r * BCR_B + c] = (r < w && c < wn) ? mz[ww + (size_t)c * wm + r] : 0.0;
    }
  } else if (q == 0 && tid < BCR_B) {
    P.b[(size_t)k * BCR_B + tid] = tid < w ? mz[3 * ww + tid] + mp[3 * ww + wm + tid] : 0.0;
  }
}
static void bcr_fill_sep(const BcrLevel& P, void* arg, hipStream_t st) {
  hipLaunchKernelGGL(bcr_pack_sep_kernel, dim3(16 * P.N), dim3(256), 0, st, P, *static_cast<BcrSepSrc*>(arg));
}
// x of separator Z_z to its place in the solution vector (sep_col[z]: its first column), from the padded blocks
__global__ __launch_bounds__(128) void bcr_sep_scatter_kernel(const double* xpad, int R, int ns, const int* sep_off, const int* sep_col, double* x) {
  const int z = blockIdx.x + 1, l = threadIdx.x;
  const int w = (z + 1 < R ? sep_off[z + 1] : ns) - sep_off[z];
  if (l < w) x[sep_col[z] + l] = xpad[(size_t)(z - 1) * BCR_B + l];
}
bool ba_bcr_sep_ok(int wm, int R) {
  static const int on = getenv("CS_BAND_BCR") ? atoi(getenv("CS_BAND_BCR")) : 1;
  return on != 0 && wm >= 1 && wm <= BCR_B && R >= 2 && R - 1 < (1 << BCR_MAXLEV) && bcr_device_ready();
}
size_t ba_bcr_sep_workspace_doubles(int R) { return ba_bcr_workspace_doubles((R - 1) * BCR_B, BCR_B) + (size_t)(R - 1) * BCR_B; }
// msgs: the R gathered messages; x: the solution vector the separators' unknowns are scattered into; info[0] != 0: a non-positive pivot
void ba_launch_bcr_sep(const double* msgs, size_t msg_doubles, int wm, int R, const int* sep_off, const int* sep_col, int ns, double* work, double* x, int* info, hipStream_t st) {
  const int N = R - 1;
  if (N < 1) return;
  double* xpad = work + ba_bcr_workspace_doubles(N * BCR_B, BCR_B);
  BcrSepSrc src{msgs, msg_doubles, wm, R, ns, sep_off};
  bcr_run(nullptr, work, N * BCR_B, 0, BCR_B, xpad, info, st, true, bcr_fill_sep, &src);
  hipLaunchKernelGGL(bcr_sep_scatter_kernel, dim3(N), dim3(128), 0, st, xpad, R, ns, sep_off, sep_col, x);
}

}  // namespace cs
