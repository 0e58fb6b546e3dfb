// sparse_kernels.hip -- numeric phase of the general sparse Cholesky of the reduced pose system (plan: ba_sparse.h).
//
//   sparse_chol_kernel     persistent; workgroup w takes the columns order[w], order[w + G], ... (level order: everything a column waits
//                          for sits earlier in the order, so a grid that is resident as a whole cannot deadlock).  Column j: the panel
//                          (rows = diagonal block, blocks below, right-hand side row) is gathered from the dense S into LDS, every column
//                          k with L(j, k) != 0 subtracts L(rows >= j of k, k) L(j, k)^T (rows found through a vertex -> panel-row map in
//                          LDS), the diagonal block is factorised, the rows below are scaled by L_jj^-T -- the right-hand side's row
//                          becomes y_j -- and the panel is written through to memory before the column's flag is raised.
//   sparse_back_kernel     L^T x = y the same way in the reverse order, a wavefront per column.
// Hand-offs between workgroups follow the banded solver's protocol (ba_kernels.hip): write-through stores, drained, then a flag; the
// consumer polls the flag, takes one agent-scope acquire and reads with plain loads.  Waits are bounded (~1 s): a starved grid fails.
#include <hip/hip_runtime.h>

#include "ba_sparse.h"

namespace cs {

namespace {
enum { SP_T = 256, SP_PF = 4, SB_PF = 3, SP_SPIN_LIMIT = 1 << 21, SP_TIMEOUT = 0x7fffffff };

__device__ __forceinline__ void sp_gstore(double* p, double v) {
  __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool sp_wait(const unsigned* flag, int* info) {   // false: aborted
  unsigned spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023u) == 0) {
      if (__hip_atomic_load(info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
      if (spins >= (unsigned)SP_SPIN_LIMIT) {
        __hip_atomic_store(info + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(info, (int)SP_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
  return true;
}
__device__ __forceinline__ void sp_publish(unsigned* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(SP_T) void sparse_chol_kernel(SparseView V, int panel_cap) {
  extern __shared__ double sp_lds[];
  double* P = sp_lds;                                   // the panel, row-major, dj doubles per row
  double* Ljk = P + panel_cap;                          // 2 x 81: L(j, k), two buffers
  double* Dj = Ljk + 162;                               // 81: L_jj (lower, row-major)
  int* map = reinterpret_cast<int*>(Dj + 81);           // vertex position -> first row of its block in the panel (N + 1 entries)
  __shared__ int s_ok;
  const int tid = threadIdx.x, N = V.N, n = V.n;
  for (int idx = blockIdx.x; idx < N; idx += gridDim.x) {
    const int j = V.order[idx], dj = V.ndim[j], cj = V.ncol[j], e0 = V.sptr[j], nent = V.sptr[j + 1] - e0, rows = V.prow[j];
    const int* rentj = V.rent + V.rbase[j];
    __syncthreads();                                    // (the previous column's LDS is free)
    if (tid == 0) map[j] = 0;
    for (int t = tid; t < nent; t += SP_T) map[V.srow[e0 + t]] = V.sroff[e0 + t];
    // ---- the panel's original entries
    for (int e = tid; e < rows * dj; e += SP_T) {
      const int rho = e / dj, b = e - rho * dj, t = rentj[rho];
      double v;
      if (t < 0) {
        v = rho >= b ? V.S[(size_t)(cj + rho) * n + cj + b] : 0.0;
      } else {
        const int i = V.srow[e0 + t];
        if (i == N) v = V.rhs[cj + b];
        else {
          const int rs = V.ncol[i] + (rho - V.sroff[e0 + t]), cs_ = cj + b;
          v = rs >= cs_ ? V.S[(size_t)rs * n + cs_] : V.S[(size_t)cs_ * n + rs];
        }
      }
      P[e] = v;
    }
    __syncthreads();
    // ---- the columns that update this one, in ascending order (a fixed order: the sums are reproducible), each waited for when it is
    // needed.  (Tried, both slower: L(j, k) of the next column loaded and the flag after it polled while the current one is applied, one
    // barrier per column instead of three -- 62 -> 73 ms on a 1 919-camera mesh; and the flags of all but the last three columns taken
    // together up front -- 16 -> 43 ms: the columns of the near-clique region are factorised side by side on different workgroups, each
    // applying the others' columns as they appear, and waiting AHEAD of need serialises them.)
    const bool tail = j >= V.tail_start;
    // (where an updating column's rows land in this panel is plan data, independent of its flag: the destinations of a thread's first
    // SP_PF elements of the NEXT update are looked up -- three dependent loads: row -> entry -> vertex -> panel row -- while the barrier
    // and the wait for that column's flag pass)
    int pf_dst[SP_PF], pf_src[SP_PF];
    auto prep = [&](int u) {
      const bool none = u >= V.rptr[j + 1] || V.rcol[u] >= V.tail_start;
      const int k = none ? 0 : V.rcol[u], ek = V.sptr[k], row0 = none ? 0 : V.sroff[ek + V.rpos[u]], R = none ? 0 : V.prow[k] - row0;
      const int* rentk = V.rent + V.rbase[k] + row0;
#pragma unroll
      for (int q = 0; q < SP_PF; q++) {
        const int e = tid + q * SP_T;
        pf_dst[q] = -1; pf_src[q] = 0;
        if (e < R * dj) {
          const int rho = e / dj, b = e - rho * dj, t = rentk[rho];
          const int i = V.srow[ek + t], a = row0 + rho - V.sroff[ek + t];
          pf_dst[q] = (map[i] + a) * dj + b; pf_src[q] = row0 + rho;
        }
      }
    };
    prep(V.rptr[j]);
    for (int u = V.rptr[j]; u < V.rptr[j + 1]; u++) {
      const int k = V.rcol[u], t0 = V.rpos[u], dk = V.ndim[k], ek = V.sptr[k];
      if (k >= V.tail_start) break;                     // (ascending: the rest are tail columns, whose part the dense factorisation does)
      if (tid == 0) {
        s_ok = sp_wait(V.done + k, V.info) ? 1 : 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      if (!s_ok) return;
      const double* Lk = V.L + V.poff[k];
      const int row0 = V.sroff[ek + t0], R = V.prow[k] - row0;
      if (tid < dj * dk) Ljk[tid] = Lk[(size_t)row0 * dk + tid];
      __syncthreads();
#pragma unroll
      for (int q = 0; q < SP_PF; q++) {
        if (pf_dst[q] < 0) continue;
        const int b = (tid + q * SP_T) % dj;
        const double* src = Lk + (size_t)pf_src[q] * dk;
        double s = 0.0;
        for (int c = 0; c < dk; c++) s = fma(src[c], Ljk[b * dk + c], s);
        P[pf_dst[q]] -= s;
      }
      const int* rentk = V.rent + V.rbase[k] + row0;
      for (int e = tid + SP_PF * SP_T; e < R * dj; e += SP_T) {
        const int rho = e / dj, b = e - rho * dj, t = rentk[rho];
        const int i = V.srow[ek + t], a = row0 + rho - V.sroff[ek + t];
        const double* src = Lk + (size_t)(row0 + rho) * dk;
        double s = 0.0;
        for (int c = 0; c < dk; c++) s = fma(src[c], Ljk[b * dk + c], s);
        P[(map[i] + a) * dj + b] -= s;
      }
      prep(u + 1);
      __syncthreads();
    }
    if (tail) {
      // ---- a column of the dense tail: its updated panel goes into the dense block (row-major, lower triangle), unfactorised
      for (int e = tid; e < rows * dj; e += SP_T) {
        const int rho = e / dj, b = e - rho * dj, t = rentj[rho];
        const int tc = V.tcol[j] + b;
        if (t < 0) { if (rho >= b) V.T[(size_t)(V.tcol[j] + rho) * V.n_tail + tc] = P[e]; }
        else {
          const int i = V.srow[e0 + t];
          if (i == N) V.rhs_t[tc] = P[e];
          else V.T[(size_t)(V.tcol[i] + rho - V.sroff[e0 + t]) * V.n_tail + tc] = P[e];
        }
      }
      continue;
    }
    // ---- the diagonal block (dj <= 9: one thread), then every row below times L_jj^-T
    if (tid == 0) {
      int bad = 0;
      for (int c = 0; c < dj; c++) {
        double d = P[c * dj + c];
        for (int q = 0; q < c; q++) d = fma(-Dj[c * 9 + q], Dj[c * 9 + q], d);
        if (!(d > 0.0)) { bad = 1; d = 1.0; }
        const double l = sqrt(d), inv = 1.0 / l;
        Dj[c * 9 + c] = l;
        for (int r = c + 1; r < dj; r++) {
          double v = P[r * dj + c];
          for (int q = 0; q < c; q++) v = fma(-Dj[r * 9 + q], Dj[c * 9 + q], v);
          Dj[r * 9 + c] = v * inv;
        }
      }
      if (bad) atomicCAS(V.info, 0, cj + 1);
    }
    __syncthreads();
    double* Lj = V.L + V.poff[j];
    for (int rho = tid; rho < rows; rho += SP_T) {
      double x[9];
      if (rho < dj) {
        for (int c = 0; c < dj; c++) x[c] = c <= rho ? Dj[rho * 9 + c] : 0.0;
      } else {
        for (int c = 0; c < dj; c++) {
          double v = P[rho * dj + c];
          for (int q = 0; q < c; q++) v = fma(-x[q], Dj[c * 9 + q], v);
          x[c] = v / Dj[c * 9 + c];
        }
      }
      for (int c = 0; c < dj; c++) sp_gstore(Lj + (size_t)rho * dj + c, x[c]);
    }
    sp_publish(V.done + j);
  }
}

// x_j = L_jj^-T (y_j - sum over the blocks below of L(i, j)^T x_i), columns in the reverse of the factorisation's order, a wavefront each.
// The lanes first wait for the flags of ALL the vertices below, 64 at a time (the last one to arrive is what the column waits for
// anyway), then share the panel's rows: lane l takes rows l, l + 64, ...; partial sums meet in a fixed shuffle tree (reproducible).
// (First form: lane 0 waited for the entries one after the other and the rows of an entry went to di <= 9 lanes -- 8.6 ms of a 17 ms
// solve on the 1 920-camera mesh, four times the factorisation.)
__global__ __launch_bounds__(64) void sparse_back_kernel(SparseView V) {
  const int lane = threadIdx.x, N = V.N;
  for (int idx = N - 1 - (int)blockIdx.x; idx >= 0; idx -= (int)gridDim.x) {
    const int j = V.order[idx], dj = V.ndim[j], cj = V.ncol[j], e0 = V.sptr[j], nent = V.sptr[j + 1] - e0 - 1;   // (without the right-hand side's entry)
    if (j >= V.tail_start) continue;                    // (solved by the dense factorisation; sparse_tail_scatter_kernel published it)
    const double* Lj = V.L + V.poff[j];
    // the factorisation is complete (kernel boundary).  What does not depend on the vertices below -- a lane's first SB_PF rows of the
    // panel and where their x sits -- is requested BEFORE the wait for them
    const int* rentj = V.rent + V.rbase[j];
    const int nrows = V.prow[j] - dj - 1;               // rows below the diagonal block, without the right-hand side's
    double lpre[SB_PF][9];
    int xat[SB_PF];
#pragma unroll
    for (int q = 0; q < SB_PF; q++) {
      const int r = lane + 64 * q;
      xat[q] = -1;
      if (r < nrows) {
        const int rho = dj + r, t = rentj[rho];
        xat[q] = V.srow[e0 + t] * 9 + (rho - V.sroff[e0 + t]);
        const double* lrow = Lj + (size_t)rho * dj;
#pragma unroll
        for (int c = 0; c < 9; c++) lpre[q][c] = c < dj ? lrow[c] : 0.0;
      }
    }
    int ok = 1;
    for (int t = lane; t < nent; t += 64) ok &= sp_wait(V.xdone + V.srow[e0 + t], V.info) ? 1 : 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (__ballot(!ok)) return;
    double acc[9];
    for (int c = 0; c < 9; c++) acc[c] = 0.0;
#pragma unroll
    for (int q = 0; q < SB_PF; q++) {
      if (xat[q] < 0) continue;
      const double xi = V.xs[xat[q]];
#pragma unroll
      for (int c = 0; c < 9; c++) acc[c] = fma(lpre[q][c], xi, acc[c]);
    }
    for (int r = lane + 64 * SB_PF; r < nrows; r += 64) {
      const int rho = dj + r, t = rentj[rho], i = V.srow[e0 + t];
      const double xi = V.xs[(size_t)i * 9 + (rho - V.sroff[e0 + t])];
      const double* lrow = Lj + (size_t)rho * dj;
      for (int c = 0; c < dj; c++) acc[c] = fma(lrow[c], xi, acc[c]);
    }
    for (int c = 0; c < dj; c++) {
      double v = acc[c];
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
      acc[c] = v;
    }
    if (lane == 0) {
      const double* yrow = Lj + (size_t)(V.prow[j] - 1) * dj;   // the right-hand side's row: y_j
      double x[9];
      for (int c = dj - 1; c >= 0; c--) {
        double v = yrow[c] - acc[c];
        for (int q = c + 1; q < dj; q++) v = fma(-Lj[(size_t)q * dj + c], x[q], v);
        x[c] = v / Lj[(size_t)c * dj + c];
      }
      for (int c = 0; c < dj; c++) { sp_gstore(V.xs + (size_t)j * 9 + c, x[c]); V.rhs[cj + c] = x[c]; }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(V.xdone + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// The blocks of the plan's pattern in the dense S, zeroed before a trial's assembly (the Schur kernels accumulate into the blocks the
// graph has, a subset of the pattern; nothing else of S is ever written, so the n x n array -- 1 GB at 11 514 unknowns, 2.7 GB at 18 426 --
// is cleared once per structure phase and only the pattern per trial): one workgroup per column.
__global__ __launch_bounds__(256) void sparse_zero_pattern_kernel(SparseView V, double* S) {
  const int j = blockIdx.x, dj = V.ndim[j], cj = V.ncol[j], e0 = V.sptr[j], rows = V.prow[j] - 1, n = V.n;   // (without the right-hand side's row)
  const int* rentj = V.rent + V.rbase[j];
  for (int e = threadIdx.x; e < rows * dj; e += 256) {
    const int rho = e / dj, b = e - rho * dj, t = rentj[rho];
    if (t < 0) { if (rho >= b) S[(size_t)(cj + rho) * n + cj + b] = 0.0; }
    else {
      const int rs = V.ncol[V.srow[e0 + t]] + (rho - V.sroff[e0 + t]), cs_ = cj + b;
      if (rs >= cs_) S[(size_t)rs * n + cs_] = 0.0; else S[(size_t)cs_ * n + rs] = 0.0;
    }
  }
}

// the dense tail's solution to where the substitution reads it (by position) and to the solution vector; its flags are raised
__global__ __launch_bounds__(64) void sparse_tail_scatter_kernel(SparseView V) {
  const int j = V.tail_start + blockIdx.x, c = threadIdx.x;
  if (j >= V.N) return;
  if (c < V.ndim[j]) { const double x = V.rhs_t[V.tcol[j] + c]; V.xs[(size_t)j * 9 + c] = x; V.rhs[V.ncol[j] + c] = x; }
  if (c == 0) V.xdone[j] = 1u;
}
}  // namespace

int sparse_max_panel_doubles() { return 16384; }   // 128 KB of the 160 KB LDS: panels of up to ~1 800 rows x 9
static size_t sparse_lds_bytes(int panel_cap, int N) { return (size_t)(panel_cap + 243) * sizeof(double) + (size_t)(N + 2) * sizeof(int); }

// the factorisation's workgroups wait for each other: the grid must be resident as a whole.  Decided ONCE, in the structure phase (device
// properties, the kernel's LDS attribute and the occupancy query are not per-trial work), and kept with the handle.
bool sparse_grids(int panel_cap, int N, SparseGrids* g) {
  int dev = 0, occ = 0, occ_b = 0;
  hipDeviceProp_t prop;
  g->chol = g->back = 0;
  if (N <= 0 || hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
  const size_t lds = sparse_lds_bytes(panel_cap, N);
  if (lds > 156 * 1024) return false;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(sparse_chol_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, sparse_chol_kernel, SP_T, lds) != hipSuccess || occ < 1) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_b, sparse_back_kernel, 64, 0) != hipSuccess || occ_b < 1) occ_b = 1;
  g->chol = std::min(N, occ * prop.multiProcessorCount);
  g->back = std::min(N, occ_b * prop.multiProcessorCount);
  return g->chol > 0;
}

// (false: no launch -- grids that sparse_grids() did not produce, or a failed clear / launch; the caller reports it instead of waiting
// on flags nobody raises)
bool launch_sparse_cholesky(const SparseView& V, int max_panel_doubles, const SparseGrids& g, hipStream_t st) {
  if (g.chol <= 0 || g.chol > V.N) return false;
  if (hipMemsetAsync(V.done, 0, sizeof(unsigned) * (size_t)V.N, st) != hipSuccess) return false;
  if (hipMemsetAsync(V.xdone, 0, sizeof(unsigned) * (size_t)(V.N + 1), st) != hipSuccess) return false;
  if (V.n_tail > 0 && hipMemsetAsync(V.T, 0, sizeof(double) * ((size_t)V.n_tail * V.n_tail + V.n_tail), st) != hipSuccess) return false;
  hipLaunchKernelGGL(sparse_chol_kernel, dim3(g.chol), dim3(SP_T), sparse_lds_bytes(max_panel_doubles, V.N), st, V, max_panel_doubles);
  return hipGetLastError() == hipSuccess;
}
void launch_sparse_zero_pattern(const SparseView& V, double* S, hipStream_t st) {
  if (V.N > 0) hipLaunchKernelGGL(sparse_zero_pattern_kernel, dim3(V.N), dim3(256), 0, st, V, S);
}
bool launch_sparse_backsolve(const SparseView& V, const SparseGrids& g, hipStream_t st) {
  if (g.back <= 0 || g.back > V.N) return false;
  if (V.n_tail > 0) hipLaunchKernelGGL(sparse_tail_scatter_kernel, dim3(V.N - V.tail_start), dim3(64), 0, st, V);
  hipLaunchKernelGGL(sparse_back_kernel, dim3(g.back), dim3(64), 0, st, V);
  return hipGetLastError() == hipSuccess;
}

}  // namespace cs
