// ba_sparse.h -- general sparse Cholesky of the reduced pose system, for graphs reverse Cuthill-McKee cannot band (2-D covisibility
// meshes, many loop closures): what the reference leaves to Eigen::SimplicialLDLT behind a fill-reducing block ordering
// (g2o/solvers/linear_solver_eigen.h:94-232, blockOrdering + AMD).  Host side: the SYMBOLIC phase, once per structure --
//   * minimum-degree elimination of the block graph (vertices = free cameras / cuboids, 6 or 9 unknowns each); eliminating a vertex joins
//     its remaining neighbours, which at that moment are exactly the rows of its column of L;
//   * per column j its row structure, per row i the columns k < i that update it (left-looking), a processing order by elimination-tree level;
//   * storage: column j of L is a dense PANEL, rows = [the diagonal block | the blocks below it | one row for the right-hand side], dj wide.
// Device side (sparse_kernels.hip): one persistent kernel factorises (a workgroup per column, columns taken in level order, a column waits
// for the flags of the columns that update it), the right-hand side riding along as a row of every panel; a second one substitutes backwards.
// The values come from the densely assembled S (ba_types.h: S[r n + c], lower triangle): the sparse path replaces rocSOLVER's potrf / potrs only.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <iterator>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace cs {

struct SparsePlan {
  int N = 0;                          // vertices (elimination positions 0 .. N - 1; position N is the right-hand side's pseudo-vertex, 1 wide)
  std::vector<int> ndim, ncol;        // per position: unknowns, first column in S (and in the right-hand side)
  std::vector<int> sptr, srow, sroff; // per position: entries below the diagonal (positions ascending, the pseudo-vertex N last) and their first row in the panel
  std::vector<int> prow, rbase, rent; // rows of the panel; first index of the panel's rows in rent; per row: the entry it belongs to (-1: diagonal block)
  std::vector<long long> poff;        // first value of the panel in L
  std::vector<int> rptr, rcol, rpos;  // per position j: the columns k < j with L(j, k) != 0 (ascending) and j's entry index in column k
  std::vector<int> order;             // processing order: by level (longest chain of updating columns), then by position
  long long nvals = 0; int max_panel = 0, levels = 0;
  // the dense tail: positions >= tail_start (the top of the elimination tree, where the columns have become a near-clique) are not
  // factorised column by column -- their panels, updated by the columns before them, are written into a dense n_tail x n_tail block that
  // a dense Cholesky takes; tcol: first row / column of a tail position in that block
  int tail_start = 0, n_tail = 0;
  std::vector<int> tcol;
  double flops = 0;                   // multiply-adds of the numeric factorisation
};

// adj: symmetric adjacency of the block graph over ids 0 .. adj.size() - 1; `verts`: the ids that take part (free vertices), dim / col per id.
// Returns false when the plan would not pay (fill above max_fill of the dense triangle) or a panel would not fit the kernel's LDS.
inline bool sparse_plan_build(const std::vector<std::vector<int>>& adj, const std::vector<int>& verts, const std::vector<int>& dim, const std::vector<int>& col,
                              int max_panel_doubles, double max_fill, SparsePlan& P, int max_tail_unknowns = 0) {
  const int N = (int)verts.size();
  P = SparsePlan();
  P.N = N;
  if (N == 0) return false;
  std::vector<int> local(adj.size(), -1);
  for (int a = 0; a < N; a++) local[verts[a]] = a;
  // ---- minimum degree (weighted by unknowns) on the plain elimination graph, the vertices' neighbourhoods as BIT ROWS: eliminating a
  // vertex ORs its row into its neighbours' (N / 64 words each) -- with sorted lists and set_union the same sweep took 230 ms at 3 072
  // vertices (a 2-D mesh's fill), a tenth of that this way.  N is a few thousand at most (the kernel's LDS map bounds it).
  const int Wd = (N + 63) / 64;
  std::vector<unsigned long long> bits((size_t)N * Wd, 0ull), mask9(Wd, 0ull), S(Wd);
  for (int a = 0; a < N; a++) {
    if (dim[verts[a]] == 9) mask9[a >> 6] |= 1ull << (a & 63);
    for (int w : adj[verts[a]]) { const int b2 = local[w]; if (b2 >= 0 && b2 != a) bits[(size_t)a * Wd + (b2 >> 6)] |= 1ull << (b2 & 63); }
  }
  std::vector<int> pos(N, -1), at(N);
  std::vector<char> gone(N, 0);
  std::vector<std::vector<int>> colstruct(N);      // by elimination position: local ids of the rows below
  std::vector<long long> deg(N);
  auto weight = [&](int a) {                        // unknowns of the neighbours: 6 per vertex, 9 for the cuboids
    const unsigned long long* r = bits.data() + (size_t)a * Wd;
    long long c6 = 0, c9 = 0;
    for (int w = 0; w < Wd; w++) { c6 += __builtin_popcountll(r[w]); c9 += __builtin_popcountll(r[w] & mask9[w]); }
    return 6 * c6 + 3 * c9;
  };
  for (int a = 0; a < N; a++) deg[a] = weight(a);
  for (int step = 0; step < N; step++) {
    int best = -1;
    for (int a = 0; a < N; a++) if (!gone[a] && (best < 0 || deg[a] < deg[best])) best = a;
    pos[best] = step; at[step] = best; gone[best] = 1;
    const unsigned long long* rb = bits.data() + (size_t)best * Wd;
    std::vector<int>& cs_ = colstruct[step];
    for (int w = 0; w < Wd; w++) {
      S[w] = rb[w];
      for (unsigned long long m = rb[w]; m; m &= m - 1) cs_.push_back(64 * w + __builtin_ctzll(m));
    }
    const unsigned long long bbit = 1ull << (best & 63);
    for (int u : cs_) {
      unsigned long long* ru = bits.data() + (size_t)u * Wd;
      for (int w = 0; w < Wd; w++) ru[w] |= S[w];
      ru[u >> 6] &= ~(1ull << (u & 63));
      ru[best >> 6] &= ~bbit;
      deg[u] = weight(u);
    }
  }
  // ---- by position
  P.ndim.resize(N + 1); P.ncol.resize(N + 1);
  for (int j = 0; j < N; j++) { P.ndim[j] = dim[verts[at[j]]]; P.ncol[j] = col[verts[at[j]]]; }
  P.ndim[N] = 1; P.ncol[N] = 0;
  P.sptr.assign(N + 1, 0); P.prow.resize(N); P.poff.resize(N); P.rbase.resize(N);
  long long nvals = 0, nrows = 0, n_unk = 0;
  double flops = 0, dense_tri = 0;
  for (int j = 0; j < N; j++) n_unk += P.ndim[j];
  dense_tri = 0.5 * (double)n_unk * (double)n_unk;
  for (int j = 0; j < N; j++) {
    std::vector<int> rows;
    for (int u : colstruct[j]) rows.push_back(pos[u]);
    std::sort(rows.begin(), rows.end());
    rows.push_back(N);
    P.sptr[j + 1] = P.sptr[j] + (int)rows.size();
    int r = P.ndim[j];
    for (int i : rows) { P.srow.push_back(i); P.sroff.push_back(r); r += P.ndim[i]; }
    P.prow[j] = r; P.poff[j] = nvals; P.rbase[j] = (int)nrows;
    nvals += (long long)r * P.ndim[j]; nrows += r;
    P.max_panel = std::max(P.max_panel, r * P.ndim[j]);
    flops += 0.5 * (double)r * (double)r * P.ndim[j];
  }
  P.nvals = nvals; P.flops = flops;
  if (P.max_panel > max_panel_doubles) return false;
  if ((double)nvals > max_fill * dense_tri) return false;
  P.rent.resize(nrows);
  for (int j = 0; j < N; j++) {
    int* re = P.rent.data() + P.rbase[j];
    for (int a = 0; a < P.ndim[j]; a++) re[a] = -1;
    for (int t = P.sptr[j]; t < P.sptr[j + 1]; t++)
      for (int a = 0; a < P.ndim[P.srow[t]]; a++) re[P.sroff[t] + a] = t - P.sptr[j];
  }
  // ---- who updates whom (left-looking), levels, order
  std::vector<int> cnt(N + 1, 0);
  for (int k = 0; k < N; k++) for (int t = P.sptr[k]; t < P.sptr[k + 1]; t++) if (P.srow[t] < N) cnt[P.srow[t] + 1]++;
  P.rptr.assign(N + 1, 0);
  for (int j = 0; j < N; j++) P.rptr[j + 1] = P.rptr[j] + cnt[j + 1];
  P.rcol.resize(P.rptr[N]); P.rpos.resize(P.rptr[N]);
  std::vector<int> fill(P.rptr.begin(), P.rptr.end() - 1);
  for (int k = 0; k < N; k++)
    for (int t = P.sptr[k]; t < P.sptr[k + 1]; t++) { const int i = P.srow[t]; if (i < N) { P.rcol[fill[i]] = k; P.rpos[fill[i]] = t - P.sptr[k]; fill[i]++; } }
  // ---- the dense tail: the longest suffix of positions whose columns hold at least tail_density of the positions after them
  static const double tail_density = [] { const char* e = getenv("CS_BA_SPARSE_TAIL_DENSITY"); return e ? atof(e) : 0.45; }();   // (swept on the survey-flight meshes: flat between 0.3 and 0.8)
  P.tail_start = N; P.n_tail = 0; P.tcol.assign(N + 1, 0);
  if (max_tail_unknowns > 0) {
    int c = N, unk = 0;
    while (c > 0) {
      const int j = c - 1, below = P.sptr[j + 1] - P.sptr[j] - 1;
      if ((double)below < tail_density * (double)(N - 1 - j) || unk + P.ndim[j] > max_tail_unknowns) break;
      unk += P.ndim[j]; c--;
    }
    if (N - c >= 24) {
      P.tail_start = c; P.n_tail = unk;
      int t = 0;
      for (int j = c; j < N; j++) { P.tcol[j] = t; t += P.ndim[j]; }
    }
  }
  std::vector<int> level(N, 0);
  for (int j = 0; j < N; j++)
    for (int u = P.rptr[j]; u < P.rptr[j + 1]; u++) if (P.rcol[u] < P.tail_start) level[j] = std::max(level[j], level[P.rcol[u]] + 1);
  P.order.resize(N);
  for (int j = 0; j < N; j++) P.order[j] = j;
  std::stable_sort(P.order.begin(), P.order.end(), [&](int a, int b) { return level[a] < level[b]; });
  for (int j = 0; j < N; j++) P.levels = std::max(P.levels, level[j] + 1);
  if (P.n_tail > 0) {   // (the tail's own factorisation is the dense block's: what stays here is the columns before it)
    double f = 0;
    for (int j = 0; j < P.tail_start; j++) f += 0.5 * (double)P.prow[j] * (double)P.prow[j] * P.ndim[j];
    P.flops = f;
  }
  return true;
}

// device side (sparse_kernels.hip)
struct SparseView {
  int N, n;                           // vertices; unknowns (S is n x n)
  const int *ndim, *ncol, *sptr, *srow, *sroff, *prow, *rbase, *rent, *rptr, *rcol, *rpos, *order;
  const long long* poff;
  const int* tcol; int tail_start, n_tail;   // the dense tail (SparsePlan)
  double* T; double* rhs_t;           // its n_tail x n_tail block (row-major, lower; zeroed by the launcher) and right-hand side
  const double* S; double* rhs;       // assembled system; right-hand side in, solution out
  double* L; double* xs;              // panels; the solution by elimination position (N x 9)
  unsigned* done; unsigned* xdone;    // per position: factorised / solved (zeroed by the launcher)
  int* info;                          // [0]: first non-positive pivot + 1 (0x7fffffff: a wait timed out), [1]: abort word
};
struct SparseGrids { int chol, back; };   // workgroups of the factorisation (co-resident) and of the substitution
bool sparse_grids(int max_panel_doubles, int N, SparseGrids* g);   // structure phase, once per plan; false: the factorisation does not fit the device
int sparse_max_panel_doubles();
bool launch_sparse_cholesky(const SparseView& V, int max_panel_doubles, const SparseGrids& g, hipStream_t st);     // factorisation (+ the tail's block assembled); false: not launched
bool launch_sparse_backsolve(const SparseView& V, const SparseGrids& g, hipStream_t st);
void launch_sparse_zero_pattern(const SparseView& V, double* S, hipStream_t st);                // S's pattern blocks <- 0 (before a trial's assembly)                           // after the tail's x sits in rhs_t: L^T x = y

}  // namespace cs
