// sparse_plan_check -- host-only check of the symbolic phase of the general sparse reduced solve (csrc/ba_sparse.h) on random block graphs:
// the plan's pattern must be exactly the fill of eliminating the vertices in the plan's order (dense boolean elimination), the panels' row
// offsets consistent, every update list complete, and the processing order a topological order of the update dependencies.
//   hipcc -O2 -std=c++17 tools/microbench/sparse_plan_check.cpp -o build_tmp/sparse_plan_check && build_tmp/sparse_plan_check
#include "../../cube_slam_wu_amd/csrc/ba_sparse.h"

#include <cstdio>
#include <cstdlib>
#include <set>

int main() {
  unsigned long long s = 99;
  auto rnd = [&](int m) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (int)((s >> 33) % (unsigned)m); };
  int checked = 0;
  for (int trial = 0; trial < 40; trial++) {
    const int NV = 20 + rnd(120), nfree = NV - rnd(5);
    std::vector<std::vector<int>> adj(NV);
    std::vector<int> dim(NV), col(NV), verts;
    int c = 0;
    for (int v = 0; v < NV; v++) { dim[v] = rnd(3) ? 6 : 9; col[v] = c; c += dim[v]; if (v < nfree) verts.push_back(v); }
    auto link = [&](int a, int b) { if (a != b && a < nfree && b < nfree) { adj[a].push_back(b); adj[b].push_back(a); } };
    if (trial % 2 == 0) {   // a grid: a mesh
      const int w = 5 + rnd(6);
      for (int v = 0; v < nfree; v++) { if ((v + 1) % w) link(v, v + 1 < nfree ? v + 1 : v); if (v + w < nfree) link(v, v + w); if (v + w + 1 < nfree && (v + 1) % w) link(v, v + w + 1); }
    } else {                // a chain with random long links
      for (int v = 0; v + 1 < nfree; v++) { link(v, v + 1); if (v + 2 < nfree) link(v, v + 2); }
      for (int q = 0; q < nfree / 4; q++) link(rnd(nfree), rnd(nfree));
    }
    for (auto& a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
    cs::SparsePlan P;
    if (!cs::sparse_plan_build(adj, verts, dim, col, 1 << 30, 2.0, P, trial % 3 ? 400 : 0)) { printf("trial %d: no plan\n", trial); return 1; }
    const int N = P.N;
    if (N != nfree) { printf("trial %d: N\n", trial); return 1; }
    // which vertex sits at which position: by its column
    std::vector<int> vert_at(N, -1);
    for (int j = 0; j < N; j++) for (int v : verts) if (col[v] == P.ncol[j] && dim[v] == P.ndim[j]) vert_at[j] = v;
    std::vector<int> pos(NV, -1);
    for (int j = 0; j < N; j++) { if (vert_at[j] < 0 || pos[vert_at[j]] >= 0) { printf("trial %d: not a permutation\n", trial); return 1; } pos[vert_at[j]] = j; }
    // dense boolean elimination in the plan's order
    std::vector<std::set<int>> g(N);
    for (int v : verts) for (int w : adj[v]) if (pos[w] >= 0) g[pos[v]].insert(pos[w]);
    for (int j = 0; j < N; j++) {
      std::set<int> below;
      for (int w : g[j]) if (w > j) below.insert(w);
      std::vector<int> plan_rows(P.srow.begin() + P.sptr[j], P.srow.begin() + P.sptr[j + 1]);
      if (plan_rows.empty() || plan_rows.back() != N) { printf("trial %d: column %d lacks the right-hand side's entry\n", trial, j); return 1; }
      plan_rows.pop_back();
      if (plan_rows != std::vector<int>(below.begin(), below.end())) { printf("trial %d: column %d: pattern differs from the fill\n", trial, j); return 1; }
      for (int a : below) for (int b : below) if (a != b) g[a].insert(b);
      int r = P.ndim[j];
      for (int t = P.sptr[j]; t < P.sptr[j + 1]; t++) { if (P.sroff[t] != r) { printf("trial %d: row offsets\n", trial); return 1; } r += P.ndim[P.srow[t]]; }
      if (r != P.prow[j]) { printf("trial %d: panel rows\n", trial); return 1; }
      for (int q = 0; q < P.prow[j]; q++) {
        const int t = P.rent[P.rbase[j] + q];
        if (q < P.ndim[j] ? t != -1 : (t < 0 || q < P.sroff[P.sptr[j] + t] || q >= P.sroff[P.sptr[j] + t] + P.ndim[P.srow[P.sptr[j] + t]])) { printf("trial %d: row -> entry map\n", trial); return 1; }
      }
    }
    // update lists and order
    std::vector<int> where(N);
    for (int q = 0; q < N; q++) where[P.order[q]] = q;
    for (int j = 0; j < N; j++) {
      std::set<int> want;
      for (int k = 0; k < j; k++) for (int t = P.sptr[k]; t < P.sptr[k + 1]; t++) if (P.srow[t] == j) want.insert(k);
      std::vector<int> have(P.rcol.begin() + P.rptr[j], P.rcol.begin() + P.rptr[j + 1]);
      if (have != std::vector<int>(want.begin(), want.end())) { printf("trial %d: update list of %d\n", trial, j); return 1; }
      for (int u = P.rptr[j]; u < P.rptr[j + 1]; u++) {
        if (P.srow[P.sptr[P.rcol[u]] + P.rpos[u]] != j) { printf("trial %d: entry index of %d in %d\n", trial, j, P.rcol[u]); return 1; }
        if (P.rcol[u] < P.tail_start && where[P.rcol[u]] >= where[j]) { printf("trial %d: order is not topological\n", trial); return 1; }   // (tail columns do not wait for each other)
      }
    }
    // the dense tail: a suffix of positions, its columns laid end to end
    if (P.tail_start < 0 || P.tail_start > N || (P.n_tail == 0) != (P.tail_start == N)) { printf("trial %d: tail bounds\n", trial); return 1; }
    { int t = 0; for (int j = P.tail_start; j < N; j++) { if (P.tcol[j] != t) { printf("trial %d: tail columns\n", trial); return 1; } t += P.ndim[j]; } if (t != P.n_tail || t > 400) { printf("trial %d: tail size\n", trial); return 1; } }
    checked++;
  }
  printf("sparse plan: %d random block graphs, pattern = fill, offsets / update lists / order consistent\n", checked);
  return 0;
}
