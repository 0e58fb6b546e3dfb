// band_win.h -- the banded Cholesky of the reduced pose system with a WINDOW-RESIDENT front (included by ba_kernels.hip, after the
// helpers of the cooperative kernels: BS, BandView, band_gload / band_gstore, band_wait_ge, band_acquire, band_rcp, band_rsqrt, ba_v4d).
//
// The cooperative kernels (band_chol_coop_kernel, band_chol_nested_kernel) spread one 32-column elimination step over ceil(bw / 16)
// workgroups: of a 13 us step, 9 are hand-offs between compute units in different XCDs (gathering rows others wrote, the team's barrier,
// publishing the diagonal block).  For bandwidths up to 128 the whole ACTIVE WINDOW of a front -- the (D + 1) x (D + 1) lower triangle
// of 32 x 32 blocks a right-looking elimination still has to update, D = floor((bw + 31) / 32) <= 4 -- fits the registers of ONE
// workgroup as matrix-core accumulators (15 blocks x 8 KB), so a front needs no other workgroup at all:
//
//   one workgroup = 12 wavefronts: waves 0..3 the diagonal team, waves 4..11 own the window's blocks, two each (slot (d, m) holds block
//   (J + d, J), J = m mod (D + 1 - d); when a block has served as a panel block its slot takes the block entering the window).
//   (12 waves = 3 per SIMD = 170 registers each: with 8 waves and four blocks each the compiler spilled accumulator tiles.)
//   Step k:   panel      L(k + d, k)^T = L_kk^-1 A(k + d, k)^T, 16 tiles of 16 x 16 over waves 0..7          (operands staged in LDS)
//             look-ahead U = A(k + 1, k + 1) - L(k + 1, k) L(k + 1, k)^T by three waves of the team, then the team factorises it
//                        (POTF2 + inverse, four columns per LDS round trip) WHILE
//             trailing   the block waves update their blocks, block(I, J) -= L(I, k) L(J, k)^T, one depth-4 slice per round
//                        of the factorisation (the workgroup barriers of the factorisation's rounds are the only synchronisation)
//             post       blocks of column k + 1 go to the staging area, their slots load the entering blocks, L_kk / L_kk^-1 / y_k
//                        and the panel go to memory.
//   The chain of a step is panel + look-ahead + POTF2; everything else runs beside it.  The right-hand side rides along in LDS.
//
// Orders: one front (the sharded solve's interiors), two fronts (A forward from the top, B in the mirrored index space from the
// bottom; B then eliminates the middle -- both dump what they have accumulated on the middle as (value - original), B reloads the sum),
// and the nested order (a separator block C in the middle, two fronts per half; C's rows ride along with the front next to C in
// separator-row workgroups (win_sep_rows) that READ the front's published panel blocks, one step behind it and off its chain; C's
// Schur complement is accumulated behind those by win_schur_accum, then C is assembled and factorised by the same front routine).

enum { WIN_T = 768, WIN_DMAX = 4, WIN_NS = 2, WIN_W = BS * (WIN_DMAX + 1), WIN_POOL = 14304 + BS * BS };
#define WIN_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

struct WinBlk { ba_v4d t[2][2]; };      // a 32 x 32 block as 2 x 2 accumulator tiles: t[ti][tj][g] = entry (16 ti + (lane >> 4) + 4 g, 16 tj + (lane & 15))

struct WinLds {
  double (*U)[BS + 1];        // the next diagonal block
  double (*Dl)[BS + 1];       // [2][32]: L_kk, by the step's parity
  double (*X)[BS + 1];        // [2][32]: L_kk^-1
  double (*St)[BS + 1];       // [WIN_DMAX][32]: the panel's operands A(k + d, k), row-major
  double* Pn;                 // [WIN_DMAX][1024]: the panel L(k + d, k), depth-major and swizzled (win_sw)
  double* colbuf;             // 512
  double* bwin;               // (WIN_DMAX + 1) * 32: the right-hand side of the window's rows
  double* yk;                 // 32
  double* zblk;               // 1024 zeros: the operand of a slot that takes no update in a step
  double* pool;
};
__device__ __forceinline__ WinLds win_carve(double* pool) {
  WinLds M;
  double* p = pool;
  M.pool = pool;
  M.U = reinterpret_cast<double (*)[BS + 1]>(p); p += BS * (BS + 1);
  M.Dl = reinterpret_cast<double (*)[BS + 1]>(p); p += 2 * BS * (BS + 1);
  M.X = reinterpret_cast<double (*)[BS + 1]>(p); p += 2 * BS * (BS + 1);
  M.St = reinterpret_cast<double (*)[BS + 1]>(p); p += WIN_DMAX * BS * (BS + 1);
  M.Pn = p; p += WIN_DMAX * BS * BS;
  M.colbuf = p; p += 512;
  M.bwin = p; p += (WIN_DMAX + 1) * BS;
  M.yk = p; p += BS;
  M.zblk = p; p += BS * BS;
  return M;
}
// entry (row r, depth t) of a depth-major 32 x 32 operand block: the four depth indices a matrix-core issue reads land in both halves
// of the banks (two-way = the minimum for 64 lanes x 8 bytes)
__device__ __forceinline__ int win_sw(int t, int r) { return t * BS + (r ^ ((t & 1) << 4)); }

// where the entries of a run come from: the band through a view
struct WinSrc {
  BandView v; int nv, bw;                  // rows / columns at or beyond nv do not exist
  const double* zero;
};
__device__ __forceinline__ double win_entry(const WinSrc& S, int i, int j) {
  const bool ok = j <= i && i - j <= S.bw && i < S.nv;
  return band_gload(ok ? S.v.base + (long long)i * S.v.si + (long long)j * S.v.sj : S.zero);
}
__device__ __forceinline__ double win_rhs(const WinSrc& S, int i) {
  return band_gload(i < S.nv ? S.v.rb + (long long)i * S.v.sr : S.zero);
}
__device__ __forceinline__ void win_load_block(const WinSrc& S, int I, int J, WinBlk& B, int li, int lk) {
  // one pointer per lane, constant strides from there
  const int i0 = BS * I + lk, j0 = BS * J + li, dij = i0 - j0;
  const double* p0 = S.v.base + (long long)i0 * S.v.si + (long long)j0 * S.v.sj;
  const long long si4 = 4 * S.v.si, sj16 = 16 * S.v.sj;
#pragma unroll
  for (int ti = 0; ti < 2; ti++)
#pragma unroll
    for (int tj = 0; tj < 2; tj++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int dd = dij + 16 * ti + 4 * g - 16 * tj;          // i - j
        const bool ok = dd >= 0 && dd <= S.bw && i0 + 16 * ti + 4 * g < S.nv;
        B.t[ti][tj][g] = band_gload(ok ? p0 + (4 * ti + g) * si4 + tj * sj16 : S.zero);
      }
}
__device__ __forceinline__ void win_zero_block(WinBlk& B) {
#pragma unroll
  for (int ti = 0; ti < 2; ti++)
#pragma unroll
    for (int tj = 0; tj < 2; tj++) B.t[ti][tj] = ba_v4d{0.0, 0.0, 0.0, 0.0};
}
__device__ __forceinline__ void win_stage(const WinBlk& B, double (*dst)[BS + 1], int li, int lk) {
#pragma unroll
  for (int ti = 0; ti < 2; ti++)
#pragma unroll
    for (int tj = 0; tj < 2; tj++)
#pragma unroll
      for (int g = 0; g < 4; g++) dst[16 * ti + lk + 4 * g][16 * tj + li] = B.t[ti][tj][g];
}
// one depth-4 slice (ks) of block -= L(I, k) L(J, k)^T; diag: the block lies on the diagonal (I == J), its upper tile is never read
__device__ __forceinline__ void win_chunk(WinBlk& B, const double* PI, const double* PJ, int ks, bool diag, int li, int lk) {
  const int t = 4 * ks + lk, sx = (t & 1) << 4, o = t * BS;
  const double a0 = -PI[o + (li ^ sx)], a1 = -PI[o + ((16 + li) ^ sx)];
  const double b0 = PJ[o + (li ^ sx)], b1 = PJ[o + ((16 + li) ^ sx)];
  B.t[0][0] = WIN_MFMA(a0, b0, B.t[0][0]);
  B.t[1][0] = WIN_MFMA(a1, b0, B.t[1][0]);
  B.t[1][1] = WIN_MFMA(a1, b1, B.t[1][1]);
  if (!diag) B.t[0][1] = WIN_MFMA(a0, b1, B.t[0][1]);
}
// tile (ti, tj) of the TRANSPOSED panel block: out[g] = L(I, k)(row 16 tj + li, column 16 ti + lk + 4 g) = sum_t L_kk^-1(column, t) A(row, t)
__device__ __forceinline__ ba_v4d win_panel_tile(const double (*X)[BS + 1], const double (*A)[BS + 1], int ti, int tj, int li, int lk) {
  ba_v4d acc = ba_v4d{0.0, 0.0, 0.0, 0.0};
  const int ksn = ti ? 8 : 4;                     // L_kk^-1(c, t) = 0 for t > c
  for (int ks = 0; ks < ksn; ks++) acc = WIN_MFMA(X[16 * ti + li][4 * ks + lk], A[16 * tj + li][4 * ks + lk], acc);
  return acc;
}

// ---- the diagonal block: band_potf2_inv4b_impl cut into its rounds, so that the other waves can work between the barriers
struct WinPotf { double v[BS / 4]; int bad; };
__device__ __forceinline__ void win_potf_init(WinPotf& P, const double (*U)[BS + 1], int nb, int lane, int w) {
  const int row = lane & 31;
  const bool lower = lane < BS;
#pragma unroll
  for (int i = 0; i < BS / 4; i++) {
    const int q = 4 * i + w;
    const double u = U[row][q];
    P.v[i] = (lower && row < nb && q <= row) ? u : ((q == row) ? 1.0 : 0.0);
  }
  P.bad = 0;
}
__device__ __forceinline__ void win_potf_put(const WinPotf& P, int i0, double* colbuf, int lane, int w) { colbuf[(i0 & 1) * 256 + w * 64 + lane] = P.v[i0]; }
__device__ __forceinline__ void win_potf_round(WinPotf& Q, int i0, const double* colbuf, int nb, int lane, int w) {
  const int c0 = 4 * i0;
  const double* buf = colbuf + (i0 & 1) * 256;
  double P[4][4], m[4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) P[i][j] = buf[j * 64 + c0 + i];
#pragma unroll
  for (int j = 0; j < 4; j++) m[j] = buf[j * 64 + lane];
  double d[4], inv[4], g[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    d[j] = P[j][j];
    Q.bad |= (c0 + j < nb) & !(d[j] > 0.0);
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
  const double dw = w == 0 ? d[0] : (w == 1 ? d[1] : (w == 2 ? d[2] : d[3]));
  const double xw = w == 0 ? x[0] : (w == 1 ? x[1] : (w == 2 ? x[2] : x[3]));
  Q.v[i0] = xw * band_rsqrt(dw);
  double z[4];
#pragma unroll
  for (int t = 3; t >= 0; t--) {
    z[t] = x[t] * inv[t];
#pragma unroll
    for (int j = t + 1; j < 4; j++) z[t] = fma(-g[j][t], z[j], z[t]);
  }
#pragma unroll
  for (int i = 0; i < BS / 4; i++) {
    if (i <= i0) continue;
    const int q = 4 * i + w;
    double acc = Q.v[i];
#pragma unroll
    for (int t = 0; t < 4; t++) acc = fma(-z[t], buf[t * 64 + q], acc);
    Q.v[i] = acc;
  }
}
__device__ __forceinline__ void win_potf_fin(const WinPotf& P, double (*Dl)[BS + 1], double (*X)[BS + 1], int lane, int w) {
  const int row = lane & 31;
  const bool lower = lane < BS;
#pragma unroll
  for (int i = 0; i < BS / 4; i++) {
    const int q = 4 * i + w;
    if (lower) Dl[row][q] = (row - q >= 0) ? P.v[i] : 0.0;
    else X[q][row] = (q - row >= 0) ? P.v[i] : 0.0;
  }
}

// Workgroup barrier of the fronts' loops: LDS traffic only.  (__syncthreads() is a workgroup-scope fence + barrier: it waits for the
// wave's global loads and stores as well -- the entering blocks' loads and the factor's stores are meant to stay in flight across barriers;
// where memory has to be drained -- before a step is published -- the code says so with s_waitcnt vmcnt(0).)
__device__ __forceinline__ void win_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- one run of a front: column blocks [kb0, kb1) of a view
struct WinRun {
  WinSrc src;
  int D;                          // block reach of the band: floor((bw + 31) / 32), 1 .. WIN_DMAX
  int kb0, kb1;
  const double* Linv;             // inverted diagonal blocks, block kb at + kb * 1024 (row-major)
  unsigned* pub;                  // published step count (block kb done -> kb + 1); null: nobody follows this front
  // dump of what the run has accumulated on the window behind its last column (value - original), for the run that continues there:
  // entry (i, j) of this view -> own coordinates [(j - x) W + (i - x)], x = 32 kb1, or (dump_mirror > 0) the coordinates of the
  // mirrored view, (i', j') = (mirror - 1 - j, mirror - 1 - i), relative to dump_xb
  const double* dump; const double* dumpr; int dump_mirror, dump_xb;
  int* info;
  long long* prof;                // optional phase clock of the team's first thread: panel, look-ahead, rounds, post (wall clock ticks), steps
};

__device__ __forceinline__ void win_dump_delta(const WinRun& R, int i, int j, double delta) {
  const WinSrc& S = R.src;
  if (i >= S.nv || j > i) return;
  int ri, rj;
  if (R.dump_mirror > 0) { ri = R.dump_mirror - 1 - j - R.dump_xb; rj = R.dump_mirror - 1 - i - R.dump_xb; }
  else { ri = i - BS * R.kb1; rj = j - BS * R.kb1; }
  if (rj < 0 || ri < 0 || ri >= WIN_W || rj >= WIN_W) return;
  band_gstore(R.dump + (long long)rj * WIN_W + ri, delta);
}
__device__ __forceinline__ void win_dump_entry(const WinRun& R, int i, int j, double val) {
  const WinSrc& S = R.src;
  if (i >= S.nv || j > i) return;
  const bool inb = i - j <= S.bw;
  const double orig = inb ? band_gload(S.v.base + (long long)i * S.v.si + (long long)j * S.v.sj) : 0.0;
  int ri, rj;
  if (R.dump_mirror > 0) { ri = R.dump_mirror - 1 - j - R.dump_xb; rj = R.dump_mirror - 1 - i - R.dump_xb; }
  else { ri = i - BS * R.kb1; rj = j - BS * R.kb1; }
  if (rj < 0 || ri < 0 || ri >= WIN_W || rj >= WIN_W) return;
  band_gstore(R.dump + (long long)rj * WIN_W + ri, val - orig);
}

// The join of two fronts: what both have accumulated on the middle (their dumps, value - original, in the continuing front's own
// coordinates relative to x) goes INTO the band and the right-hand side -- the middle's entries are still the originals --, in a fixed
// order (original + A's + B's), so that the run that continues there is an ordinary one.
__device__ __forceinline__ void win_merge_dumps(const WinSrc& S, int x, const double* dA, const double* dB, const double* drA, const double* drB) {
  const int m = S.nv - x;                   // the middle: rows / columns x .. nv - 1 (m <= WIN_W)
  for (int e = threadIdx.x; e < m * (S.bw + 1); e += WIN_T) {
    const int rj = e / (S.bw + 1), dd = e % (S.bw + 1), ri = rj + dd;
    if (ri >= m) continue;
    const double* p = S.v.base + (long long)(x + ri) * S.v.si + (long long)(x + rj) * S.v.sj;
    const long long o = (long long)rj * WIN_W + ri;
    band_gstore(p, (band_gload(p) + band_gload(dA + o)) + band_gload(dB + o));
  }
  for (int e = threadIdx.x; e < m; e += WIN_T) {
    const double* p = S.v.rb + (long long)(x + e) * S.v.sr;
    band_gstore(p, (band_gload(p) + band_gload(drA + e)) + band_gload(drB + e));
  }
}

// The two kinds of waves run DIFFERENT loops with the same sequence of workgroup barriers (per step: after the panel, after the
// look-ahead, one per round of the factorisation -- or one, in a run's last step --, after the post phase), so that the register
// allocation of one does not carry the other's state: the block waves hold 4 x 32 accumulator registers across the loop, the team the
// factorisation's rounds.
// (WIN_OPAQUE at the top of a loop body: the lane's indices are redefined there as far as the compiler can tell, so the dozens of
// per-lane LDS / global addresses a step uses are recomputed in the step -- a few integer operations -- instead of being hoisted out
// of the loop and spilled to scratch, which is what the register allocator did with them.)
#define WIN_OPAQUE() asm volatile("" : "+v"(tid), "+v"(lane), "+v"(li), "+v"(lk))
__device__ __forceinline__ void win_front_run(const WinLds& M, const WinRun& R) {
  int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool team = wv < 4;
  const int D = R.D, NB1 = D + 1, nv = R.src.nv, bw = R.src.bw;
  const int nslot = (D * (D + 3)) / 2;             // sum over d < D of (D + 1 - d): blocks at distance D never rest in registers
  if (R.kb0 >= R.kb1) return;
  const bool pub = R.pub != nullptr;
  for (int e = tid; e < NB1 * BS; e += WIN_T) {
    const int I = R.kb0 + (e >> 5);
    M.bwin[(I % NB1) * BS + (e & 31)] = win_rhs(R.src, BS * I + (e & 31));
  }
  for (int e = tid; e < BS * BS; e += WIN_T) M.zblk[e] = 0.0;
  // the panel of step k: 4 D tiles of 16 x 16 over the 8 waves.  Wave w: tile (ti, tj) = (w >> 2, w & 1) of the blocks at distance
  // dA = ((w >> 1) & 1) + 1 and dA + 2 -- the two share the L_kk^-1 operand; a SIMD's two waves (w, w + 4) together issue 8 + 16 of
  // the 64-cycle matrix-core instructions (ti = 0 needs only the first 16 depth indices: L_kk^-1 is lower triangular)
  auto panel = [&](int k) {
    const int k0 = BS * k, nb = min(BS, nv - k0);
    const double (*Xk)[BS + 1] = M.X + (k & 1) * BS;
    const int ti = wv >> 2, tj = wv & 1, dA = ((wv >> 1) & 1) + 1, dB = dA + 2;
    if (wv >= 8 || dA > D) return;
    const bool two = dB <= D;
    const double (*SA)[BS + 1] = M.St + (dA - 1) * BS;
    const double (*SB)[BS + 1] = M.St + ((two ? dB : dA) - 1) * BS;
    ba_v4d accA = ba_v4d{0.0, 0.0, 0.0, 0.0}, accB = accA;
    double xa[8], sa[8], sb[8];
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      xa[ks] = Xk[16 * ti + li][4 * ks + lk]; sa[ks] = SA[16 * tj + li][4 * ks + lk]; sb[ks] = SB[16 * tj + li][4 * ks + lk];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ks++) { accA = WIN_MFMA(xa[ks], sa[ks], accA); accB = WIN_MFMA(xa[ks], sb[ks], accB); }
    if (ti) {
#pragma unroll
      for (int ks = 4; ks < 8; ks++) { accA = WIN_MFMA(xa[ks], sa[ks], accA); accB = WIN_MFMA(xa[ks], sb[ks], accB); }
    }
    const int r = 16 * tj + li, c0 = 16 * ti + lk;
    auto put = [&](int d, const ba_v4d& acc) {
      const int i = k0 + BS * d + r;
      const double* q0 = R.src.v.base + (long long)i * R.src.v.si + (long long)(k0 + c0) * R.src.v.sj;
      const long long sj4 = 4 * R.src.v.sj;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int c = c0 + 4 * g;
        M.Pn[(d - 1) * BS * BS + win_sw(c, r)] = acc[g];
        if (c < nb && i < nv && i - (k0 + c) <= bw) band_gstore(q0 + g * sj4, acc[g]);
      }
    };
    put(dA, accA);
    if (two) put(dB, accB);
  };
  auto publish = [&](int k) { if (tid == 0) __hip_atomic_store(R.pub, (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

  if (team) {
    // =============================================================== the diagonal team (waves 0..3)
#ifndef WIN_X_NOPRIO
    __builtin_amdgcn_s_setprio(3);                   // the chain of a step runs here: ahead of the block waves on the same SIMDs
#endif
    win_bar();                                 // (P1) the first diagonal block is in U
    {
      WinPotf pf;
      const int nb = min(BS, nv - BS * R.kb0);
      win_potf_init(pf, M.U, nb, lane, wv);
#pragma unroll
      for (int r = 0; r < BS / 4; r++) {
        win_potf_put(pf, r, M.colbuf, lane, wv);
        win_bar();
        win_potf_round(pf, r, M.colbuf, nb, lane, wv);
      }
      win_potf_fin(pf, M.Dl + (R.kb0 & 1) * BS, M.X + (R.kb0 & 1) * BS, lane, wv);
      if (pf.bad && tid == 0) atomicCAS(R.info, 0, BS * R.kb0 + 1);
    }
    win_bar();                                 // (P2)
    win_bar();                                 // (P3)
    long long tp[4] = {0, 0, 0, 0}, t_prev = R.prof ? wall_clock64() : 0;
#define WIN_TICK(q) do { if (R.prof) { const long long t_now = wall_clock64(); tp[q] += t_now - t_prev; t_prev = t_now; } } while (0)
    for (int k = R.kb0; k < R.kb1; k++) {
      WIN_OPAQUE();
      const bool last = k + 1 >= R.kb1;
      panel(k);
      win_bar();                               // (B1)
      WIN_TICK(0);
      if ((!last || R.dump) && wv < 3) {
        // look-ahead: the next diagonal block takes this column's update in LDS (three tiles; the upper one is never read)
        const int ti = wv > 0, tj = wv > 1;
        ba_v4d acc = ba_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
          const int t = 4 * ks + lk;
          acc = WIN_MFMA(M.Pn[win_sw(t, 16 * ti + li)], M.Pn[win_sw(t, 16 * tj + li)], acc);
        }
#pragma unroll
        for (int g = 0; g < 4; g++) M.U[16 * ti + lk + 4 * g][16 * tj + li] -= acc[g];
      }
      win_bar();                               // (B2)
      WIN_TICK(1);
      double nr = 0.0;                                // the right-hand side of the entering rows: in flight across the window
      if (tid < BS) nr = win_rhs(R.src, BS * (k + NB1) + tid);
      if (!last) {
        WinPotf pf;
        const int nb1 = min(BS, nv - BS * (k + 1));
        win_potf_init(pf, M.U, nb1, lane, wv);
#pragma unroll
        for (int r = 0; r < BS / 4; r++) {
          win_potf_put(pf, r, M.colbuf, lane, wv);
          if (r == BS / 4 - 1 && pub) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          win_bar();
          if (r == BS / 4 - 1 && pub) publish(k);
          win_potf_round(pf, r, M.colbuf, nb1, lane, wv);
        }
        win_potf_fin(pf, M.Dl + ((k + 1) & 1) * BS, M.X + ((k + 1) & 1) * BS, lane, wv);
        if (pf.bad && tid == 0) atomicCAS(R.info, 0, BS * (k + 1) + 1);
        WIN_TICK(2);
      } else {
        win_bar();                                     // (the block waves' y_k)
        if (pub) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        win_bar();
        if (pub) publish(k);
      }
      if (tid < BS) M.bwin[(k % NB1) * BS + tid] = nr;
      win_bar();                               // (Bend)
      WIN_TICK(3);
    }
    if (R.prof && tid == 0) { for (int q = 0; q < 4; q++) R.prof[q] += tp[q]; R.prof[4] += R.kb1 - R.kb0; }
#undef WIN_TICK
  } else {
    // =============================================================== the block waves (waves 4..11)
    // A slot holds only what the eliminated columns have SUBTRACTED from its block (it starts at zero: re-using a slot moves no
    // data); the blocks' original entries travel from memory straight to the staging areas -- column k + 1's D panel blocks to St,
    // the diagonal block (k + 2, k + 2) to U, requested at the start of step k's window by all block waves together and written
    // before its last round -- and a block that leaves its slot is ADDED to what lies there.  (Loading the entering blocks into
    // the slots costs a wait per step whatever is tried: the compiler merges "kept" and "reloaded" with register copies, and a copy
    // needs the data.)
    WinBlk blk[WIN_NS];
    int sd[WIN_NS], sp[WIN_NS], sJ[WIN_NS];
    bool sok[WIN_NS];
#pragma unroll
    for (int q = 0; q < WIN_NS; q++) {
      int s = (wv - 4) + 8 * q;
      sok[q] = s < nslot;
      int d = 0;
      if (sok[q]) { while (s >= NB1 - d) { s -= NB1 - d; d++; } } else s = 0;
      sd[q] = d; sp[q] = NB1 - d;
      sJ[q] = R.kb0 + (((s - R.kb0) % sp[q]) + sp[q]) % sp[q];
      // (the first column's blocks and the first two diagonal blocks take nothing from a slot: their slots start one period on)
      if (sJ[q] == R.kb0 || (d == 0 && sJ[q] == R.kb0 + 1)) sJ[q] += sp[q];
      win_zero_block(blk[q]);
    }
    // original entries of column kc's panel blocks (b = 0 .. D - 1: distance b + 1) and of the diagonal block kd (b = D): word u of
    // this thread is entry e = lt + 512 u of the D + 1 blocks laid end to end
    const int lt = tid - 256;
    double lv[2 * (WIN_DMAX + 1)];
    auto orig_fetch = [&](int kc, int kd, bool with_panel, bool with_diag) {
#pragma unroll
      for (int u = 0; u < 2 * (WIN_DMAX + 1); u++) {
        const int e = (tid - 256) + 512 * u, b = e >> 10, w = e & 1023, r = w >> 5, c = w & 31;
        const bool isd = b == D;
        const int i = isd ? BS * kd + r : BS * (kc + b + 1) + r, j = isd ? BS * kd + c : BS * kc + c;
        const bool want = b <= D && (isd ? with_diag : with_panel);
        lv[u] = band_gload(want && j <= i && i - j <= bw && i < nv ? R.src.v.base + (long long)i * R.src.v.si + (long long)j * R.src.v.sj : R.src.zero);
      }
    };
    auto orig_put = [&](bool with_panel, bool with_diag) {
#pragma unroll
      for (int u = 0; u < 2 * (WIN_DMAX + 1); u++) {
        const int e = (tid - 256) + 512 * u, b = e >> 10, w = e & 1023, r = w >> 5, c = w & 31;
        if (b < D && with_panel) M.St[b * BS + r][c] = lv[u];
        if (b == D && with_diag) M.U[r][c] = lv[u];
      }
    };
    (void)lt;
    orig_fetch(R.kb0, R.kb0, true, true);
    orig_put(true, true);
    win_bar();                                 // (P1)
    orig_fetch(R.kb0, R.kb0 + 1, false, true);
#pragma unroll
    for (int r = 0; r < BS / 4; r++) win_bar();
    win_bar();                                 // (P2)
    orig_put(false, true);
    win_bar();                                 // (P3)
    for (int k = R.kb0; k < R.kb1; k++) {
      WIN_OPAQUE();
      const int par = k & 1, k0 = BS * k, nb = min(BS, nv - k0);
      const bool last = k + 1 >= R.kb1;
      const double (*Xk)[BS + 1] = M.X + par * BS;
      const double (*Dk)[BS + 1] = M.Dl + par * BS;
      panel(k);
      win_bar();                               // (B1)
      win_bar();                               // (B2)
      if (!last) orig_fetch(k + 1, k + 2, true, true);
#pragma unroll 1
      for (int r = 0; r < BS / 4; r++) {
        WIN_OPAQUE();
        if (!last) {
          if (r == BS / 4 - 1 && pub) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          win_bar();
        }
#ifndef WIN_X_NOEXTRA
        if (r == 0 && wv < 8) {
          // L_kk and its inverse to memory (4 + 4 words per thread); y_k = L_kk^-1 b_k (wave 4: lane = row + 32 x half of the terms)
          const int t2 = tid - 256, rr0 = t2 >> 5, cc = t2 & 31;
          double xv[4], dv[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { xv[u] = Xk[rr0 + 8 * u][cc]; dv[u] = Dk[rr0 + 8 * u][cc]; }
          const double* lp = R.Linv + (size_t)k * BS * BS + t2;
          const double* dp = R.src.v.base + (long long)(k0 + rr0) * R.src.v.si + (long long)(k0 + cc) * R.src.v.sj;
          const long long si8 = 8 * R.src.v.si;
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int rr = rr0 + 8 * u;
            band_gstore(lp + 256 * u, xv[u]);
            if (rr < nb && cc <= rr && rr - cc <= bw) band_gstore(dp + u * si8, dv[u]);
          }
          if (wv == 4) {
            const int row = lane & 31, h = lane >> 5;
            const double* bk = M.bwin + (k % NB1) * BS + 16 * h;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int c8 = 0; c8 < 16; c8 += 8) {
              double xr[8], bv[8];
#pragma unroll
              for (int c = 0; c < 8; c++) { xr[c] = Xk[row][16 * h + c8 + c]; bv[c] = bk[c8 + c]; }
#pragma unroll
              for (int c = 0; c < 8; c += 4) { s0 = fma(xr[c], bv[c], s0); s1 = fma(xr[c + 1], bv[c + 1], s1); s2 = fma(xr[c + 2], bv[c + 2], s2); s3 = fma(xr[c + 3], bv[c + 3], s3); }
            }
            double y = (s0 + s1) + (s2 + s3);
            y += __shfl_xor(y, 32);
            if (h == 0) {
              M.yk[row] = y;
              if (row < nb) band_gstore(&R.src.v.rb[(long long)(k0 + row) * R.src.v.sr], y);
            }
          }
        }
#endif
        if (r == 0 && last) win_bar();    // (no rounds' barriers in a run's last step: y_k for the other waves)
#ifndef WIN_X_NOEXTRA
        if (r == 1 && wv - 3 <= D) {
          // b(k + d) -= L(k + d, k) y_k: wave 4 + (d - 1), lane = row + 32 x half of the terms
          const int d = wv - 3, row = lane & 31, h = lane >> 5;
          const double* P = M.Pn + (d - 1) * BS * BS;
          double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
          for (int c8 = 0; c8 < 16; c8 += 8) {
            double pv[8], yv[8];
#pragma unroll
            for (int c = 0; c < 8; c++) { pv[c] = P[win_sw(16 * h + c8 + c, row)]; yv[c] = M.yk[16 * h + c8 + c]; }
#pragma unroll
            for (int c = 0; c < 8; c += 4) { s0 = fma(pv[c], yv[c], s0); s1 = fma(pv[c + 1], yv[c + 1], s1); s2 = fma(pv[c + 2], yv[c + 2], s2); s3 = fma(pv[c + 3], yv[c + 3], s3); }
          }
          double a = (s0 + s1) + (s2 + s3);
          a += __shfl_xor(a, 32);
          if (h == 0) M.bwin[((k + d) % NB1) * BS + row] -= a;
        }
#endif
#pragma unroll
        for (int q = 0; q < WIN_NS; q++) {
          const int dI = sJ[q] + sd[q] - k, dJ = sok[q] ? sJ[q] - k : 0;
#ifndef WIN_X_NOCHUNK
          // (tried unconditional, an idle slot multiplying a block of zeros, which rids the loop of the register copies the compiler
          // makes around the conditional: slower, 9.2 -> 10.4 us of rounds per step -- the FP64 matrix-core instructions of the
          // block waves and the team's FP64 vector instructions do not overlap on a SIMD, every issue counts)
          if (dJ >= 1 && dI <= D && BS * sJ[q] < nv) win_chunk(blk[q], M.Pn + (dI - 1) * BS * BS, M.Pn + (dJ - 1) * BS * BS, r, sd[q] == 0, li, lk);
#endif
        }
        if (r == BS / 4 - 2 && !last) orig_put(true, true);      // (St is free since the panel, U since the factorisation began)
      }
      if (!last) {
        // blocks of column k + 1 (and the diagonal block after it) leave their slots: added to the originals in St / U
#pragma unroll
        for (int q = 0; q < WIN_NS; q++) {
          if (!sok[q]) continue;
          const bool pan = sd[q] > 0 && sJ[q] == k + 1, dg = sd[q] == 0 && sJ[q] == k + 2;
          if (pan || dg) {
            double (*dst)[BS + 1] = dg ? M.U : M.St + (sd[q] - 1) * BS;
#pragma unroll
            for (int ti = 0; ti < 2; ti++)
#pragma unroll
              for (int tj = 0; tj < 2; tj++)
#pragma unroll
                for (int g = 0; g < 4; g++) dst[16 * ti + lk + 4 * g][16 * tj + li] += blk[q].t[ti][tj][g];
            sJ[q] += sp[q];
          }
        }
#pragma unroll
        for (int q = 0; q < WIN_NS; q++) {
          // (a slot that was just emptied starts again from zero: a multiplication instead of `if (...) blk = 0`, same reason as above)
          const bool gone = sok[q] && ((sd[q] > 0 && sJ[q] == k + 1 + sp[q]) || (sd[q] == 0 && sJ[q] == k + 2 + sp[q]));
          const double keep = gone ? 0.0 : 1.0;
#pragma unroll
          for (int ti = 0; ti < 2; ti++)
#pragma unroll
            for (int tj = 0; tj < 2; tj++) blk[q].t[ti][tj] *= keep;
        }
      } else {
        if (pub) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        win_bar();
      }
      win_bar();                               // (Bend)
    }
    // ---- the window behind the last column, for the run that continues there: this wave's blocks ARE (value - original)
    if (R.dump) {
#pragma unroll
      for (int q = 0; q < WIN_NS; q++) {
        if (!sok[q] || sJ[q] < R.kb1 || sJ[q] + sd[q] > R.kb1 + D) continue;
#pragma unroll
        for (int ti = 0; ti < 2; ti++)
#pragma unroll
          for (int tj = 0; tj < 2; tj++)
#pragma unroll
            for (int g = 0; g < 4; g++) win_dump_delta(R, BS * (sJ[q] + sd[q]) + 16 * ti + lk + 4 * g, BS * sJ[q] + 16 * tj + li, blk[q].t[ti][tj][g]);
      }
    }
  }
  // ---- ... the diagonal block in U, the untouched block at distance D, the right-hand side
  if (R.dump) {
    const int kb1 = R.kb1;
    for (int e = tid; e < BS * BS; e += WIN_T) win_dump_entry(R, BS * kb1 + (e >> 5), BS * kb1 + (e & 31), M.U[e >> 5][e & 31]);
    for (int e = tid; e < BS * BS; e += WIN_T) win_dump_entry(R, BS * (kb1 + D) + (e >> 5), BS * kb1 + (e & 31), win_entry(R.src, BS * (kb1 + D) + (e >> 5), BS * kb1 + (e & 31)));
    for (int e = tid; e < NB1 * BS; e += WIN_T) {
      const int I = kb1 + (e >> 5), i = BS * I + (e & 31);
      if (i < nv) {
        const double orig = band_gload(R.src.v.rb + (long long)i * R.src.v.sr);
        const int ri = R.dump_mirror > 0 ? R.dump_mirror - 1 - i - R.dump_xb : i - BS * kb1;
        if (ri >= 0 && ri < WIN_W) band_gstore(R.dumpr + ri, M.bwin[(I % NB1) * BS + (e & 31)] - orig);
      }
    }
  }
}

// ---- rows of the separator block C below a front (nested order): this workgroup owns ncb blocks of 32 rows of C and keeps
// (C rows) x (the front's window columns k .. k + D) in registers.  Per step of the front it READS what the front has published --
// L_kk^-1 and the panel blocks L(k + d, k) from the band --, forms L(C, k) = A'(C, k) L_kk^-T (written to lc, where the Schur
// accumulators and the substitution read it) and applies (C, J) -= L(C, k) L(J, k)^T.  Nothing flows back to the front.
struct WinSep {
  BandView v; int nv, bw, D, KB;          // the front's view; it publishes KB steps
  const double* Linv; unsigned* pub;
  int cb0, ncb, wc, qflip;                // rows 32 cb0 .. of C (in the half's index space)
  const double* a_base; long long a_sq, a_sj; int m0, ms;   // A(q, j) = a_base + q a_sq + j a_sj, inside the band iff m0 + q + ms j <= bw
  const double* lc; unsigned* done; const double* zero;
};
__device__ __forceinline__ void win_sep_rows(const WinLds& M, const WinSep& R) {
  int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = R.D, NB1 = D + 1, nv = R.nv, bw = R.bw;
  double (*X)[BS + 1] = M.X;
  double (*St)[BS + 1] = M.St;             // [2][32]: the blocks (C, k) as operands
  double* Pn = M.Pn;                       // the front's panel
  double* PnC = reinterpret_cast<double*>(M.Dl);   // [2][1024]: L(C, k), depth-major swizzled
  WinBlk blk[1]; int scb[1], sJ[1]; bool sok[1];      // 2 (D + 1) <= 10 slots over the 12 waves; a slot holds what has been subtracted from its block
#pragma unroll
  for (int q = 0; q < 1; q++) {
    const int s = wv;
    sok[q] = s < R.ncb * NB1;
    scb[q] = sok[q] ? s / NB1 : 0; sJ[q] = sok[q] ? s % NB1 : 0;
    win_zero_block(blk[q]);
  }
  // what the front published for step k: L_kk^-1 (2 words per thread) and D panel blocks (2 D words); and the original entries of
  // the blocks (C rows, column block k) (3 words: zero once the columns have left the band)
  double fx[2], fp[2 * WIN_DMAX], fc[3];
  auto fetch = [&](int k) {
#pragma unroll
    for (int u = 0; u < 2; u++) fx[u] = band_gload(tid + WIN_T * u < BS * BS ? R.Linv + (size_t)k * BS * BS + tid + WIN_T * u : R.zero);
#pragma unroll
    for (int d = 1; d <= WIN_DMAX; d++)
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int e = tid + WIN_T * u, r = e & 31, t = e >> 5, i = BS * (k + d) + r, j = BS * k + t;
        const bool ok = e < BS * BS && d <= D && i < nv && j < nv && i - j <= bw;
        fp[2 * (d - 1) + u] = band_gload(ok ? R.v.base + (long long)i * R.v.si + (long long)j * R.v.sj : R.zero);
      }
#pragma unroll
    for (int u = 0; u < 3; u++) {
      const int e = tid + WIN_T * u, cbi = e >> 10, r = (e >> 5) & 31, c = e & 31, q = BS * (R.cb0 + cbi) + r, j = BS * k + c;
      const bool ok = cbi < R.ncb && j < nv && R.m0 + q + R.ms * j <= bw;
      fc[u] = band_gload(ok ? R.a_base + (long long)q * R.a_sq + (long long)j * R.a_sj : R.zero);
    }
  };
  bool have = false;
  for (int k = 0; k < R.KB; k++) {
    WIN_OPAQUE();
    if (!have) {
      if (tid == 0) { band_wait_ge(R.pub, (unsigned)(k + 1)); band_acquire(); }
      win_bar();
      fetch(k);
    }
    win_bar();             // (the previous step's readers of X / Pn / St / PnC are through)
#pragma unroll
    for (int u = 0; u < 2; u++) { const int e = tid + WIN_T * u; if (e < BS * BS) X[e >> 5][e & 31] = fx[u]; }
#pragma unroll
    for (int d = 1; d <= WIN_DMAX; d++)
#pragma unroll
      for (int u = 0; u < 2; u++) { const int e = tid + WIN_T * u; if (d <= D && e < BS * BS) Pn[(d - 1) * BS * BS + win_sw(e >> 5, e & 31)] = fp[2 * (d - 1) + u]; }
#pragma unroll
    for (int u = 0; u < 3; u++) { const int e = tid + WIN_T * u; if ((e >> 10) < R.ncb) St[(e >> 10) * BS + ((e >> 5) & 31)][e & 31] = fc[u]; }
    win_bar();
#pragma unroll
    for (int q = 0; q < 1; q++)
      if (sok[q] && sJ[q] == k) {
        double (*dst)[BS + 1] = St + scb[q] * BS;
#pragma unroll
        for (int ti = 0; ti < 2; ti++)
#pragma unroll
          for (int tj = 0; tj < 2; tj++)
#pragma unroll
            for (int g = 0; g < 4; g++) dst[16 * ti + lk + 4 * g][16 * tj + li] += blk[q].t[ti][tj][g];
      }
    win_bar();
    const int nb = min(BS, nv - BS * k);
    if (wv < 4 * R.ncb) {
      const int cbi = wv >> 2, ti = (wv >> 1) & 1, tj = wv & 1;
      const ba_v4d acc = win_panel_tile(X, St + cbi * BS, ti, tj, li, lk);
      const int r = 16 * tj + li, qq = BS * (R.cb0 + cbi) + r;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int c = 16 * ti + lk + 4 * g;
        PnC[cbi * BS * BS + win_sw(c, r)] = acc[g];
        if (c < nb) band_gstore(&R.lc[(long long)(BS * k + c) * R.wc + (R.qflip ? R.wc - 1 - qq : qq)], acc[g]);
      }
    }
    win_bar();
    // the next step's operands, if the front is already there
    {
      int* seen = reinterpret_cast<int*>(M.yk);      // one answer for the workgroup
      if (tid == 0) *seen = (k + 1 < R.KB) && __hip_atomic_load(R.pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(k + 2);
      win_bar();
      have = *seen != 0;
      if (have) { band_acquire(); fetch(k + 1); }
    }
#pragma unroll
    for (int q = 0; q < 1; q++) {
      if (!sok[q]) continue;
      const int dJ = sJ[q] - k;
      if (dJ >= 1 && dJ <= D && BS * sJ[q] < nv) {
#pragma unroll
        for (int ks = 0; ks < 8; ks++) win_chunk(blk[q], PnC + scb[q] * BS * BS, Pn + (dJ - 1) * BS * BS, ks, false, li, lk);
      }
      if (sJ[q] == k) sJ[q] += NB1;
      if (dJ == 0) win_zero_block(blk[q]);
    }
    // lc of this step is complete
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    win_bar();
    if (tid == 0) __hip_atomic_fetch_add(R.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- the kernel
struct WinHalf {
  BandView fv, rv;          // the half's index space [0, nh) (C below it) and the mirrored one (u = nh - 1 - v)
  int nh, K1, K2, qflip;    // front A: blocks [0, K1) of fv; front B: blocks [0, K2) of rv, then -- joined with A's dump -- the rest of rv's first nh - 32 K1 columns
  const double* Linv_f; const double* Linv_r; const double* lc;
  const double* dumpA; const double* dumpB; const double* dumprA; const double* dumprB;
};
struct WinNested {
  WinHalf h[2];
  int n, bw, wc, c0, LD, D, NCR, GS, nested;     // nested: C = [c0, c0 + wc); NCR separator-row workgroups and GS Schur accumulators per half
  const double* Sb; const double* rhs; const double* SC; const double* rhsC; const double* LinvC; const double* part;
  long long* prof;
  const double* zero; int* info; unsigned* bars;  // bars: [half: A done, separator-row steps, B's published steps] x 2, [6] accumulators done, [8] C assembled
};

// Schur accumulator of C (band_chol_nested_kernel's team 2, 256 of the workgroup's threads): 16 rows of sum_t L(i, t) L(j, t) and
// sum_t L(i, t) y(t) over everything front B eliminates, behind the separator-row workgroups
__device__ __forceinline__ void win_schur_accum(double* buf, const WinNested& P, const WinHalf& H, int hid, int wi) {
  const int tid = threadIdx.x, wc = P.wc, ldb = wc + 1, Th = H.nh - BS * H.K1, nstep = (Th + BS - 1) / BS;
  const bool act = tid < 256;
  const int tt8 = (tid & 255) >> 3, l8 = tid & 7, lane = tid & 63, wv4 = (tid >> 6) & 3, li = lane & 15, lk = lane >> 4;
  unsigned* crc = P.bars + 3 * hid + 1;
  ba_v4d acc[2];
  acc[0] = ba_v4d{0.0, 0.0, 0.0, 0.0}; acc[1] = acc[0];
  double accy = 0.0;
  for (int sidx = 0; sidx < nstep; sidx++) {
    if (tid == 0) { band_wait_ge(crc, (unsigned)(sidx + 1) * (unsigned)P.NCR); band_acquire(); }
    __syncthreads();
    if (act) {
      const int t = sidx * BS + tt8;
      const double* row = H.lc + (long long)t * wc + l8;
      double vals[16];
#pragma unroll
      for (int u = 0; u < 16; u++) vals[u] = band_gload((t < Th && 8 * u + l8 < wc) ? row + 8 * u : P.zero);
      const double yv = band_gload((t < Th && l8 == 0) ? H.rv.rb + (long long)t * H.rv.sr : P.zero);
#pragma unroll
      for (int u = 0; u < 16; u++) if (8 * u < wc) buf[tt8 * ldb + l8 + 8 * u] = vals[u];
      if (l8 == 0) buf[tt8 * ldb + wc] = yv;
    }
    __syncthreads();
    if (act) {
#pragma unroll
      for (int t0 = 0; t0 < BS; t0 += 4) {
        const double* bt = buf + (t0 + lk) * ldb;
        const double a = bt[wi * 16 + li];
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int ct = wv4 + 4 * q;
          if (16 * ct < wc) acc[q] = WIN_MFMA(a, bt[16 * ct + li], acc[q]);
        }
      }
      if (tid < 16) {
#pragma unroll 8
        for (int tt = 0; tt < BS; tt++) accy = fma(buf[tt * ldb + wi * 16 + tid], buf[tt * ldb + wc], accy);
      }
    }
  }
  if (act) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int ct = wv4 + 4 * q;
      if (16 * ct >= wc) continue;
#pragma unroll
      for (int g = 0; g < 4; g++) band_gstore(&P.part[((size_t)hid * wc + wi * 16 + lk + 4 * g) * ldb + 16 * ct + li], acc[q][g]);
    }
    if (tid < 16) band_gstore(&P.part[((size_t)hid * wc + wi * 16 + tid) * ldb + wc], accy);
  }
}

__global__ __launch_bounds__(WIN_T) void band_chol_win_kernel(WinNested P) {
  __shared__ double pool[WIN_POOL];
  const WinLds M = win_carve(pool);
  const int tid = threadIdx.x;
  const int per = P.nested ? 2 + P.NCR + P.GS : 2;
  const int hid = blockIdx.x / per, r = blockIdx.x % per;
  const WinHalf& H = P.h[hid];
  unsigned* bars = P.bars + 3 * hid;
  const int nh = H.nh, bw = P.bw;
  const int nvA = H.K2 > 0 ? nh - BS * H.K2 : nh, nvB = nh - BS * H.K1, KB = (nvB + BS - 1) / BS;
  // roles: 0 = front A, 1 = front B (two runs), 2 .. = separator rows, then the Schur accumulators (the first of half 0 factorises C)
  int nruns = 0;
  if (r == 0) nruns = 1;
  else if (r == 1) nruns = H.K2 > 0 ? 2 : 0;
  else if (r < 2 + P.NCR) {
    WinSep S;
    const int cq = r - 2, wcb = P.wc / BS;
    S.v = H.rv; S.nv = nvB; S.bw = bw; S.D = P.D; S.KB = KB; S.Linv = H.Linv_r; S.pub = bars + 2;
    S.cb0 = 2 * cq; S.ncb = min(2, wcb - 2 * cq); S.wc = P.wc; S.qflip = H.qflip;
    // A(q, u): entry (nh + q, nh - 1 - u) of the half's space, inside the band iff q + 1 + u <= bw
    S.a_base = H.fv.base + (long long)nh * H.fv.si + (long long)(nh - 1) * H.fv.sj; S.a_sq = H.fv.si; S.a_sj = -H.fv.sj; S.m0 = 1; S.ms = 1;
    S.lc = H.lc; S.done = bars + 1; S.zero = P.zero;
    if (S.ncb > 0) win_sep_rows(M, S);
    return;
  } else {
    const int wi = r - 2 - P.NCR, wc = P.wc;
    win_schur_accum(pool, P, H, hid, wi);
    band_grid_sync(P.bars + 6, 2u * (unsigned)P.GS);
    if (hid != 0) return;
    if (tid < 256) {
      const int ldb = wc + 1, rr = tid >> 4, cg = tid & 15, i = wi * 16 + rr;
      for (int j = cg; j <= wc; j += 16) {
        const double s = band_gload(&P.part[(size_t)i * ldb + j]) + band_gload(&P.part[((size_t)wc + i) * ldb + j]);
        if (j < wc) {
          if (j <= i) band_gstore(&P.SC[(size_t)j * wc + i], ((i - j <= bw) ? band_gload(&P.Sb[(size_t)(P.c0 + j) * P.LD + (i - j)]) : 0.0) - s);
        } else {
          band_gstore(&P.rhsC[i], band_gload(&P.rhs[P.c0 + i]) - s);
        }
      }
    }
    band_grid_sync(P.bars + 8, (unsigned)P.GS);
    if (wi != 0) return;
    nruns = 1;
  }
  for (int ri = 0; ri < nruns; ri++) {
    WinRun R;
    R.D = P.D; R.info = P.info; R.prof = (P.prof && r == 1 && hid == 0) || (P.prof && !P.nested && r == 0 && H.K2 == 0) ? P.prof : nullptr; R.pub = nullptr; R.dump = nullptr; R.dumpr = nullptr; R.dump_mirror = 0; R.dump_xb = 0;
    R.src.bw = bw; R.src.zero = P.zero;
    if (r == 0) {
      // front A: the half's own index space from its far end
      R.src.v = H.fv; R.src.nv = nvA; R.kb0 = 0; R.kb1 = H.K1; R.Linv = H.Linv_f;
      if (H.K2 > 0) { R.dump = H.dumpA; R.dumpr = H.dumprA; R.dump_mirror = nh; R.dump_xb = BS * H.K2; }
    } else if (r == 1) {
      // front B: the mirrored space, next to C (nested order); then the middle, joined with what A left there
      R.src.v = H.rv; R.src.nv = nvB; R.Linv = H.Linv_r;
      R.pub = P.nested ? bars + 2 : nullptr;
      if (ri == 0) { R.kb0 = 0; R.kb1 = H.K2; R.dump = H.dumpB; R.dumpr = H.dumprB; }
      else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) { band_wait_ge(bars + 0, 1u); band_acquire(); }
        __syncthreads();
        win_merge_dumps(R.src, BS * H.K2, H.dumpA, H.dumpB, H.dumprA, H.dumprB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        band_acquire();
        R.kb0 = H.K2; R.kb1 = KB;
      }
    } else {
      const int wc = P.wc;
      R.src.v = BandView{P.SC, 1, (long long)wc, P.rhsC, 1}; R.src.nv = wc; R.src.bw = wc - 1; R.D = wc / BS;
      R.kb0 = 0; R.kb1 = wc / BS; R.Linv = P.LinvC;
    }
    win_front_run(M, R);
    if (r == 0 && H.K2 > 0) band_grid_arrive(bars + 0);
  }
}
