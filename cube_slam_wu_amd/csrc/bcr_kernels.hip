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

// ---------------------------------------------------------------------------------------------------------------- level 0: pack --
// The couplings of the eliminated blocks and the diagonal blocks of the remaining ones, out of the band into plain row-major blocks
// (the panel / update kernels then read one layout at every level).  Runs beside the factor workgroups of level 0, off the chain.
template <int NT>
__device__ void bcr_pack(const BcrLevel& P, int p) {
  const int k = p >> 3, which = (p >> 2) & 1, q = p & 3, tid = threadIdx.x;
  if (k >= P.N) return;
  const int Bv = P.Bv, LD = P.LD;
  const double* __restrict__ Sb = P.Sb;
  constexpr int NU = 4096 / NT;       // entries of a 32-row quarter per thread
  double v[NU];
  if ((k & 1) == 0) {
    const int je = k >> 1;
    if (which == 0) {
      if (k == 0) return;
      double* __restrict__ Y = P.YL + (size_t)je * BCR_BB;
      const int vr = bcr_valid_rows(P, k);
#pragma unroll
      for (int u = 0; u < NU; u++) {
        const int idx = tid + NT * u, c = idx & (BCR_B - 1), kr = 32 * q + (idx >> 7);
        const int off = Bv + kr - c;
        v[u] = (kr < vr && c < Bv && off < LD) ? Sb[(size_t)((k - 1) * Bv + c) * LD + off] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < NU; u++) { const int idx = tid + NT * u; Y[(32 * q + (idx >> 7)) * BCR_B + (idx & (BCR_B - 1))] = v[u]; }
    } else {
      if (k + 1 >= P.N) return;
      double* __restrict__ Y = P.YR + (size_t)je * BCR_BB;
      const int vrn = bcr_valid_rows(P, k + 1);
#pragma unroll
      for (int u = 0; u < NU; u++) {
        const int idx = tid + NT * u, c = idx & (BCR_B - 1), kr = 32 * q + (idx >> 7);
        const int off = Bv + c - kr;
        v[u] = (c < vrn && kr < Bv && off < LD) ? Sb[(size_t)(k * Bv + kr) * LD + off] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < NU; u++) { const int idx = tid + NT * u; Y[(32 * q + (idx >> 7)) * BCR_B + (idx & (BCR_B - 1))] = v[u]; }
    }
  } else if (which == 0) {
    double* __restrict__ D = P.D + (size_t)k * BCR_BB;
    const int vr = bcr_valid_rows(P, k);
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int idx = tid + NT * u, C = idx & (BCR_B - 1), R = 32 * q + (idx >> 7);
      v[u] = R >= C ? bcr_band_diag(P, k, vr, R, C) : bcr_band_diag(P, k, vr, C, R);
    }
#pragma unroll
    for (int u = 0; u < NU; u++) { const int idx = tid + NT * u; D[(32 * q + (idx >> 7)) * BCR_B + (idx & (BCR_B - 1))] = v[u]; }
  }
}

// -------------------------------------------------------------------------------------------------------------------- factor --
// One workgroup of EIGHT waves per eliminated block: D = L L^T and Linv = L^-1 over four 32-column steps, all of it in LDS.  Round 6: the
// chain of the kernel is the four 32 x 32 diagonal sweeps (a dependent chain of 128 pivots) -- everything else is matrix-core work that
// does not have to wait for them.  So the waves are two sets:
//   P (waves 0-3)  the diagonal sweep of step K (potf4: Cholesky factor + inverse, four columns per LDS round trip), then only the two block
//                  products the next sweep waits for:  L(K+1,K) = A(K+1,K) X_K^T  and  A(K+1,K+1) -= L(K+1,K) L(K+1,K)^T
//   M (waves 4-7)  the rest of step K, beside sweep K + 1:  the other panels L(i,K), the trailing updates A(i,j) -= L(i,K) L(j,K)^T, and the
//                  running products of the inverse  T(i,j) = sum_m L(i,m) Linv(m,j)  ->  Linv(i,j) = -X_i T(i,j)
// Every hand-over between the sets is one of the workgroup barriers both execute in lockstep (the M set's work is dealt into the intervals
// between the sweep's barriers); a SIMD holds one wave of each set, so the sweep's FP64 vector chain and the other set's matrix-core
// instructions issue side by side.  Storage: the ten lower blocks A(i,j) (A(i,j), i > j, is reused for T(i,j) once L(i,j) has been formed from
// it; A(K,K) receives X_K), six blocks L(i,j) (L(i,j), later Linv(i,j)).  No block is read and written by different waves inside one interval.
enum { F8_SLOT = BS * (BS + 1), F8_NA = BCR_NB * (BCR_NB + 1) / 2, F8_NL = BCR_NB * (BCR_NB - 1) / 2, F8_COLBUF = 2 * 4 * 64,
       BCR_LDS_DOUBLES = (F8_NA + F8_NL) * F8_SLOT + F8_COLBUF + 4 * BCR_B, BCR_FACTOR_THREADS = 512 };
#define F8_A(i, j) (reinterpret_cast<bcr_blk>(lds + ((i) * ((i) + 1) / 2 + (j)) * F8_SLOT))
#define F8_L(i, j) (reinterpret_cast<bcr_blk>(lds + (F8_NA + (i) * ((i) - 1) / 2 + (j)) * F8_SLOT))
#define F8_ZERO (bcr_v4d{0.0, 0.0, 0.0, 0.0})

__device__ __forceinline__ void bcr_row8(bcr_blk M, int t, int lane, double (&a)[8]) {
#pragma unroll
  for (int s = 0; s < 8; s++) a[s] = M[16 * t + (lane & 15)][4 * s + (lane >> 4)];
}
__device__ __forceinline__ void bcr_col8(bcr_blk M, int t, int lane, double (&b)[8]) {
#pragma unroll
  for (int s = 0; s < 8; s++) b[s] = M[4 * s + (lane >> 4)][16 * t + (lane & 15)];
}
// c (+/-)= rows ti of MA times  BCOL ? columns tj of MB : rows tj of MB transposed,  depth steps [s0, s1) of the 32
template <bool NEG, bool BCOL>
__device__ __forceinline__ bcr_v4d f8_mma(bcr_v4d c, bcr_blk MA, int ti, bcr_blk MB, int tj, int lane, int s0, int s1) {
  double a[8], b[8];
  bcr_row8(MA, ti, lane, a);
  if (BCOL) bcr_col8(MB, tj, lane, b); else bcr_row8(MB, tj, lane, b);
#pragma unroll
  for (int s = 0; s < 8; s++)
    if (s >= s0 && s < s1) c = BCR_MFMA(NEG ? -a[s] : a[s], b[s], c);
  return c;
}
// the block operations of a step K (a wave computes the 16 x 16 tile (ti, tj) of the block):
//   D  L(i,K) = A(i,K) X_K^T                    E  A(i,j) -= L(i,K) L(j,K)^T              F  T(i,K) = L(i,K) X_K          (into A(i,K))
//   G  T(i,j) += L(i,K) Linv(K,j)  (j < K < i)   B  Linv(K,j) = -X_K T(K,j)               (into L(K,j))
__device__ __forceinline__ void f8_D(double* lds, int i, int K, int ti, int tj, int lane) {
  bcr_ctile_store(F8_L(i, K), ti, tj, lane, f8_mma<false, false>(F8_ZERO, F8_A(i, K), ti, F8_A(K, K), tj, lane, 0, 4 * (tj + 1)));
}
__device__ __forceinline__ void f8_E(double* lds, int i, int j, int K, int ti, int tj, int lane) {
  bcr_ctile_store(F8_A(i, j), ti, tj, lane, f8_mma<true, false>(bcr_ctile_load(F8_A(i, j), ti, tj, lane), F8_L(i, K), ti, F8_L(j, K), tj, lane, 0, 8));
}
__device__ __forceinline__ void f8_F(double* lds, int i, int K, int ti, int tj, int lane) {
  bcr_ctile_store(F8_A(i, K), ti, tj, lane, f8_mma<false, true>(F8_ZERO, F8_L(i, K), ti, F8_A(K, K), tj, lane, 4 * tj, 8));
}
__device__ __forceinline__ void f8_G(double* lds, int i, int j, int K, int ti, int tj, int lane) {
  bcr_ctile_store(F8_A(i, j), ti, tj, lane, f8_mma<false, true>(bcr_ctile_load(F8_A(i, j), ti, tj, lane), F8_L(i, K), ti, F8_L(K, j), tj, lane, 0, 8));
}
__device__ __forceinline__ void f8_B(double* lds, int K, int j, int ti, int tj, int lane) {
  bcr_ctile_store(F8_L(K, j), ti, tj, lane, f8_mma<true, true>(F8_ZERO, F8_A(K, K), ti, F8_A(K, j), tj, lane, 0, 4 * (ti + 1)));
}

// The M set's products beside a sweep are ISSUED in one interval and STORED at the head of the next: the interval then holds the operand loads
// and the issue of the eight matrix-core instructions only -- shorter than a round of the sweep, which the set would otherwise hold up at
// the barrier (measured: 0.7 us per whole product against 0.5 us per round) -- and nothing reads the tile before the barrier after its store.
struct F8Pend { bcr_v4d c; int off; };      // off: the destination slot's offset in doubles, -1 = nothing pending
#define F8_OFF_A(i, j) (((i) * ((i) + 1) / 2 + (j)) * F8_SLOT)
#define F8_OFF_L(i, j) ((F8_NA + (i) * ((i) - 1) / 2 + (j)) * F8_SLOT)
__device__ __forceinline__ void f8_flush(double* lds, F8Pend& q, int ti, int tj, int lane) {
  if (q.off >= 0) bcr_ctile_store(reinterpret_cast<bcr_blk>(lds + q.off), ti, tj, lane, q.c);
  q.off = -1;
}
__device__ __forceinline__ F8Pend f8_E_issue(double* lds, int i, int j, int K, int ti, int tj, int lane) {
  return F8Pend{f8_mma<true, false>(bcr_ctile_load(F8_A(i, j), ti, tj, lane), F8_L(i, K), ti, F8_L(j, K), tj, lane, 0, 8), F8_OFF_A(i, j)};
}
__device__ __forceinline__ F8Pend f8_F_issue(double* lds, int i, int K, int ti, int tj, int lane) {
  return F8Pend{f8_mma<false, true>(F8_ZERO, F8_L(i, K), ti, F8_A(K, K), tj, lane, 4 * tj, 8), F8_OFF_A(i, K)};
}
__device__ __forceinline__ F8Pend f8_G_issue(double* lds, int i, int j, int K, int ti, int tj, int lane) {
  return F8Pend{f8_mma<false, true>(bcr_ctile_load(F8_A(i, j), ti, tj, lane), F8_L(i, K), ti, F8_L(K, j), tj, lane, 0, 8), F8_OFF_A(i, j)};
}

// ---- the diagonal sweep (four waves), four columns per round as band_potf2.h's routine: lanes 0..31 of every wave own row r of the block,
// lanes 32..63 row r of an identity that rides along (what the elimination turns it into is column r of L^-1); wave ws keeps the column
// q = 4 i + ws of every lane's row.  Round i0 eliminates the columns 4 i0 .. 4 i0 + 3 together: their owners write them to LDS (potf4_pre),
// one workgroup barrier, then (potf4_post) every lane reads its own row's four entries -- four ordinary LDS reads -- and takes everything that
// is the same for all lanes OUT OF ITS WAVE'S REGISTERS with v_readlane: the 4 x 4 pivot block (lane c0 + i's entries) and, for the trailing
// update of a later column q, lane q's raw entries.  (Measured, tools/microbench/bcr_factor_probe.cpp: the uniform-address LDS reads of the
// first form -- 14 per round here, 68 in an eight-column round -- cost ~35 cycles each with four waves issuing the same reads, and WERE the
// round: 0.7 us per four-column round, 1.6 us per eight-column round.)  Only the inverse is kept.
struct Potf4 { double v[8]; int bad; };
__device__ __forceinline__ void potf4_init(Potf4& S, bcr_blk U, int lane, int ws) {
  const int row = lane & 31;
  const bool lower = lane < BS;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int q = 4 * i + ws;
    const double x = U[row][q];
    S.v[i] = (lower && q <= row) ? x : (q == row ? 1.0 : 0.0);
  }
  S.bad = 0;
}
__device__ __forceinline__ void potf4_pre(const Potf4& S, int i0, double* colbuf, int lane, int ws) { colbuf[(i0 & 1) * 256 + ws * 64 + lane] = S.v[i0]; }
__device__ __forceinline__ void potf4_post(Potf4& S, int i0, const double* colbuf, int lane, int ws) {
  const double* buf = colbuf + (i0 & 1) * 256;
  const int c0 = 4 * i0;
  double m[4], P[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++) m[j] = buf[j * 64 + lane];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) P[i][j] = band_rdlane(m[j], c0 + i);       // lane c0 + i's entry of column c0 + j
  double d[4], inv[4], g[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    d[j] = P[j][j];
    S.bad |= !(d[j] > 0.0);
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
  const double dw = ws == 0 ? d[0] : (ws == 1 ? d[1] : (ws == 2 ? d[2] : d[3]));
  const double xw = ws == 0 ? x[0] : (ws == 1 ? x[1] : (ws == 2 ? x[2] : x[3]));
  S.v[i0] = xw * band_rsqrt(dw);
  double z[4];
#pragma unroll
  for (int t = 3; t >= 0; t--) {
    z[t] = x[t] * inv[t];
#pragma unroll
    for (int j = t + 1; j < 4; j++) z[t] = fma(-g[j][t], z[j], z[t]);
  }
#pragma unroll
  for (int i = i0 + 1; i < 8; i++) {
    const int q = 4 * i + ws;            // (wave-uniform: ws comes from a readfirstlane)
    double acc = S.v[i];
#pragma unroll
    for (int t = 0; t < 4; t++) acc = fma(-z[t], band_rdlane(m[t], q), acc);      // lane q's raw entries of the round's columns
    S.v[i] = acc;
  }
}
// the inverse of the factor, X[q][row] (lower triangular, the rest zero), into the diagonal block's own slot
__device__ __forceinline__ void potf4_store(const Potf4& S, bcr_blk X, int lane, int ws) {
  if (lane < BS) return;
  const int row = lane & 31;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int q = 4 * i + ws;
    X[q][row] = (q >= row) ? S.v[i] : 0.0;
  }
}
// -DBCR_PROBE (tools/microbench/bcr_factor_probe.cpp only): wave w of workgroup 0 stamps the constant 100 MHz clock at every barrier
#ifdef BCR_PROBE
__device__ long long* g_bcr_probe = nullptr;
#define BCR_STAMP() do { if (g_bcr_probe && blockIdx.x == 0 && lane == 0 && probe_k < 64) g_bcr_probe[w * 64 + probe_k++] = wall_clock64(); } while (0)
#else
#define BCR_STAMP() do { } while (0)
#endif
__global__ __launch_bounds__(BCR_FACTOR_THREADS) void bcr_factor_kernel(BcrLevel P, int n_elim) {
  if ((int)blockIdx.x >= n_elim) { bcr_pack<BCR_FACTOR_THREADS>(P, blockIdx.x - n_elim); return; }
  extern __shared__ double bcr_lds[];
  double* lds = bcr_lds;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#ifdef BCR_PROBE
  int probe_k = 0;
#endif
  BCR_STAMP();
  const bool isP = w < 4;
  const int ws = __builtin_amdgcn_readfirstlane(w & 3), ti = ws >> 1, tj = ws & 1;
  const int je = blockIdx.x, e = 2 * je;
  double* colbuf = lds + (F8_NA + F8_NL) * F8_SLOT;
  double* bv = colbuf + F8_COLBUF;
  double* part = bv + BCR_B;          // 3 x 128

  // ---- load D_e (the ten lower blocks: 20 entries per thread, all in flight) and b_e
  {
    double v[F8_NA][2];
    if (P.lvl == 0 && !P.packed) {
      const int vr = bcr_valid_rows(P, e);
#pragma unroll
      for (int i = 0; i < BCR_NB; i++)
#pragma unroll
        for (int j = 0; j <= i; j++)
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const int idx = tid + BCR_FACTOR_THREADS * q, R = 32 * i + (idx & 31), C = 32 * j + (idx >> 5);
            v[i * (i + 1) / 2 + j][q] = R >= C ? bcr_band_diag(P, e, vr, R, C) : 0.0;
          }
#pragma unroll
      for (int i = 0; i < BCR_NB; i++)
#pragma unroll
        for (int j = 0; j <= i; j++)
#pragma unroll
          for (int q = 0; q < 2; q++) { const int idx = tid + BCR_FACTOR_THREADS * q; F8_A(i, j)[idx & 31][idx >> 5] = v[i * (i + 1) / 2 + j][q]; }
    } else {
      const double* __restrict__ D = P.D + (size_t)e * BCR_BB;
#pragma unroll
      for (int i = 0; i < BCR_NB; i++)
#pragma unroll
        for (int j = 0; j <= i; j++)
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const int idx = tid + BCR_FACTOR_THREADS * q;
            v[i * (i + 1) / 2 + j][q] = D[(32 * i + (idx >> 5)) * BCR_B + 32 * j + (idx & 31)];
          }
#pragma unroll
      for (int i = 0; i < BCR_NB; i++)
#pragma unroll
        for (int j = 0; j <= i; j++)
#pragma unroll
          for (int q = 0; q < 2; q++) { const int idx = tid + BCR_FACTOR_THREADS * q; F8_A(i, j)[idx >> 5][idx & 31] = v[i * (i + 1) / 2 + j][q]; }
    }
  }
  if (tid < BCR_B) bv[tid] = bcr_b_at(P, e, tid);
  BCR_STAMP(); __syncthreads();

  Potf4 S;
  int bad = 0;
  // One sweep = init pre(0) | B | post(0) pre(1) | B | ... | post(6) pre(7) | B | post(7) store | B  (P set); the M set runs the statements
  // M0 .. M7 in the eight intervals behind the sweep's first eight barriers.
#ifdef BCR_EXPERIMENT_NO_M      /* timing experiment of tools/microbench only: the sweeps without the other set's work (results wrong) */
#define F8_MWORK(x) (void)0
#else
#define F8_MWORK(x) x
#endif
#define F8_ROUND(r, MS)                                                                                                     \
  if (isP) { potf4_post(S, r, colbuf, lane, ws); potf4_pre(S, (r) + 1, colbuf, lane, ws); } else { f8_flush(lds, pend, ti, tj, lane); F8_MWORK(MS); } \
  BCR_STAMP(); __syncthreads();
#define F8_SWEEP(K, M0, M1, M2, M3, M4, M5, M6, M7)                                                                        \
  if (isP) { potf4_init(S, F8_A(K, K), lane, ws); potf4_pre(S, 0, colbuf, lane, ws); }                                     \
  BCR_STAMP(); __syncthreads();                                                                                            \
  F8_ROUND(0, M0) F8_ROUND(1, M1) F8_ROUND(2, M2) F8_ROUND(3, M3) F8_ROUND(4, M4) F8_ROUND(5, M5) F8_ROUND(6, M6)            \
  if (isP) { potf4_post(S, 7, colbuf, lane, ws); potf4_store(S, F8_A(K, K), lane, ws); bad |= S.bad; } else { f8_flush(lds, pend, ti, tj, lane); F8_MWORK(M7); } \
  BCR_STAMP(); __syncthreads();
#define F8_NONE (void)0
  F8Pend pend{F8_ZERO, -1};

  // ---- step 0
  F8_SWEEP(0, F8_NONE, F8_NONE, F8_NONE, F8_NONE, F8_NONE, F8_NONE, F8_NONE, F8_NONE)
  if (isP) f8_D(lds, 1, 0, ti, tj, lane); else f8_D(lds, 2, 0, ti, tj, lane);
  BCR_STAMP(); __syncthreads();
  if (isP) f8_E(lds, 1, 1, 0, ti, tj, lane); else f8_D(lds, 3, 0, ti, tj, lane);
  BCR_STAMP(); __syncthreads();
  // ---- step 1 (beside its sweep: the rest of step 0)
  F8_SWEEP(1, pend = f8_E_issue(lds, 2, 1, 0, ti, tj, lane), pend = f8_E_issue(lds, 2, 2, 0, ti, tj, lane), pend = f8_E_issue(lds, 3, 1, 0, ti, tj, lane),
           pend = f8_E_issue(lds, 3, 2, 0, ti, tj, lane), pend = f8_E_issue(lds, 3, 3, 0, ti, tj, lane), pend = f8_F_issue(lds, 1, 0, ti, tj, lane),
           pend = f8_F_issue(lds, 2, 0, ti, tj, lane), pend = f8_F_issue(lds, 3, 0, ti, tj, lane))
  if (isP) f8_D(lds, 2, 1, ti, tj, lane); else { f8_flush(lds, pend, ti, tj, lane); f8_D(lds, 3, 1, ti, tj, lane); }
  BCR_STAMP(); __syncthreads();
  if (isP) f8_E(lds, 2, 2, 1, ti, tj, lane); else f8_B(lds, 1, 0, ti, tj, lane);
  BCR_STAMP(); __syncthreads();
  // ---- step 2 (beside its sweep: the rest of step 1)
  F8_SWEEP(2, pend = f8_E_issue(lds, 3, 2, 1, ti, tj, lane), pend = f8_G_issue(lds, 2, 0, 1, ti, tj, lane), pend = f8_E_issue(lds, 3, 3, 1, ti, tj, lane),
           pend = f8_F_issue(lds, 2, 1, ti, tj, lane), pend = f8_F_issue(lds, 3, 1, ti, tj, lane), pend = f8_G_issue(lds, 3, 0, 1, ti, tj, lane), F8_NONE, F8_NONE)
  if (isP) f8_D(lds, 3, 2, ti, tj, lane); else f8_B(lds, 2, 0, ti, tj, lane);
  BCR_STAMP(); __syncthreads();
  if (isP) f8_E(lds, 3, 3, 2, ti, tj, lane); else f8_B(lds, 2, 1, ti, tj, lane);
  BCR_STAMP(); __syncthreads();
  // ---- step 3 (beside its sweep: the rest of step 2), then the last row of the inverse on all eight waves
  F8_SWEEP(3, pend = f8_G_issue(lds, 3, 0, 2, ti, tj, lane), pend = f8_G_issue(lds, 3, 1, 2, ti, tj, lane), pend = f8_F_issue(lds, 3, 2, ti, tj, lane),
           F8_NONE, F8_NONE, F8_NONE, F8_NONE, F8_NONE)
  static_assert(BCR_NB == 4, "four 32-column steps");
  if (isP) f8_B(lds, 3, 0, ti, tj, lane); else f8_B(lds, 3, 1, ti, tj, lane);
  if (isP ? ws < 2 : ws >= 2) f8_B(lds, 3, 2, ti, tj, lane);
  if (bad && tid == 0) atomicMax(P.info, e + 1);
  BCR_STAMP(); __syncthreads();
#undef F8_SWEEP
#undef F8_ROUND
#undef F8_MWORK
#undef F8_NONE

  // ---- Linv in the TA layout (row tile m: depth steps s < 4 (m + 1); s = 8 u + w lies in block column u), y = Linv b
  double* __restrict__ TA = P.Linv + (size_t)je * BCR_BB;
#pragma unroll
  for (int m = 0; m < BCR_NT; m++)
#pragma unroll
    for (int u = 0; u < (m + 2) / 2; u++) {
      const int s = 8 * u + w;
      bcr_blk Sl = (m >> 1) == u ? F8_A(m >> 1, m >> 1) : F8_L(m >> 1, u < (m >> 1) ? u : 0);
      if (s < 4 * (m + 1)) TA[(size_t)(m * BCR_NS + s) * 64 + lane] = Sl[16 * (m & 1) + (lane & 15)][4 * w + (lane >> 4)];
    }
  {
    const int R = tid & (BCR_B - 1), h = tid >> 7, bi = R >> 5;          // h: the block column (32 columns) this thread sums
    double s0 = 0.0, s1 = 0.0;
    if (h <= bi) {
      const double* rowp = lds + (h == bi ? bi * (bi + 1) / 2 + bi : F8_NA + bi * (bi - 1) / 2 + h) * F8_SLOT + (R & 31) * (BS + 1);
      const double* bh = bv + 32 * h;
#pragma unroll 8
      for (int C = 0; C < 32; C += 2) { s0 = fma(rowp[C], bh[C], s0); s1 = fma(rowp[C + 1], bh[C + 1], s1); }      // (the diagonal slot's upper triangle holds zeros)
    }
    if (h) part[(h - 1) * BCR_B + R] = s0 + s1;
    BCR_STAMP(); __syncthreads();
    const double yR = ((s0 + s1) + part[R]) + (part[BCR_B + R] + part[2 * BCR_B + R]);
    if (!h) P.y[(size_t)je * BCR_B + R] = yR;
    // The LAST level (one block, no neighbour): its substitution x = Linv^T y right here, out of the slots -- one launch less on the chain
    // (bcr_run leaves this level's bcr_back_kernel out).  Thread (column C, row block h) sums its 32 rows; the four parts in a fixed order.
    if (P.N == 1) {
      __syncthreads();                       // (every thread has read its parts)
      if (!h) bv[R] = yR;
      __syncthreads();
      const int C = R, bj = C >> 5;
      double t0 = 0.0, t1 = 0.0;
      if (h >= bj) {
        const double* colp = lds + (h == bj ? h * (h + 1) / 2 + h : F8_NA + h * (h - 1) / 2 + bj) * F8_SLOT + (C & 31);
        const double* yh = bv + 32 * h;
#pragma unroll 8
        for (int r = 0; r < 32; r += 2) { t0 = fma(colp[r * (BS + 1)], yh[r], t0); t1 = fma(colp[(r + 1) * (BS + 1)], yh[r + 1], t1); }      // (the diagonal slot's upper triangle holds zeros)
      }
      if (h) part[(h - 1) * BCR_B + C] = t0 + t1;
      __syncthreads();
      if (!h) {
        const double x = ((t0 + t1) + part[C]) + (part[BCR_B + C] + part[2 * BCR_B + C]);
        const int g0 = (1 << P.lvl) - 1;       // level-l block 0 is level-0 block 2^l - 1
        if (C < P.Bv && g0 * P.Bv + C < P.n) P.rhs[g0 * P.Bv + C] = x;
      }
    }
  }
  BCR_STAMP();
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
    hipLaunchKernelGGL(bcr_factor_kernel, dim3(ne + ((l == 0 && !packed) ? 8 * P.N : 0)), dim3(BCR_FACTOR_THREADS), BCR_LDS_DOUBLES * sizeof(double), st, P, ne);
    if (nr > 0) {
      hipLaunchKernelGGL(bcr_panel_kernel, dim3(16 * ne), dim3(256), 0, st, P);
      hipLaunchKernelGGL(bcr_update_kernel, dim3(nr * BCR_UPD_SLOTS), dim3(256), 0, st, P);
    }
  }
  for (int l = L - 1; l >= 0; l--) {
    const BcrLevel& P = lev[l];
    if (P.N == 1) continue;        // (the one block of the last level: substituted by its own factor workgroup)
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
