// ba_kernels.hip -- HIP kernels of the g2o bundle-adjustment iteration for gfx950 (MI355X).
//
// One LM linearisation = computeActiveErrors + buildSystem + Schur complement + back-substitution
// (object_slam/Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:61-114, block_solver.hpp:353-560):
//   ba_chi2_*            residuals + (robust) chi2, fixed-shape block reduction (deterministic);
//   ba_lin_cam_kernel    one workgroup per camera: residual, analytic 2x6 Jacobian
//                        (types_six_dof_expmap.cpp:148-184), rho'-weighted J^T Omega J and -J^T Omega e
//                        accumulated per lane and reduced with wavefront shuffles -> A_ii (6x6), b_i;
//   ba_lin_pt_kernel     one lane per landmark over its (contiguous) edges: A_jj (3x3), b_j in registers, the
//                        6x3 H_pl block of every edge streamed out once;
//   ba_cub_edge_kernel / ba_odom_edge_kernel   numeric central-difference Jacobians, delta = 1e-9
//                        (base_binary_edge.hpp:130-205) through oplus + computeError, then the quadratic form
//                        (base_binary_edge.hpp:54-120) into per-edge blocks; ba_accum_pose_kernel gathers them;
//   ba_prep_kernel       D_j^-1 = (H_ll,j + lambda I)^-1, D^-1 b_l, W D^-1 per edge (block_solver.hpp:385-407);
//   ba_cam_rhs_kernel / ba_cub_scatter_kernel / ba_offdiag_kernel   reduced system S = H_pp (+lambda) and
//                        b_schur = b_p - sum W D^-1 b_l (:373-439);
//   ba_schur_kernel      one wavefront per covisible camera pair (i1 <= i2): S_{i1 i2} -= sum_j (W D^-1)_{i1 j} W_{i2 j}^T
//                        over the landmarks both cameras see (:409-431);
//   ba_backsub_kernel    x_l = D^-1 (b_l - W^T x_p) (:457-482);
//   ba_update_*          oplus on every vertex (sparse_optimizer.cpp:422-435).
// All FP64.  No atomics on the data path except the (unique-pair) off-diagonal pose blocks.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <utility>

#include "ba_types.h"
#include "band_potf2.h"

namespace cs {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;
}

// rho, rho' of a projection edge's squared error (cs_robust.h): Huber-or-none from the delta alone, any kernel through the kind array
__device__ __forceinline__ void proj_rho(const int* rk, int k, double delta, double e, double& rho0, double& rho1) {
  if (rk) robust_rho(rk[k], delta, e, rho0, rho1); else huber_rho(e, delta, rho0, rho1);
}

struct ProjLin {
  double e[2];     // error
  double Jp[6];    // 2x3  d e / d point
  double Jc[12];   // 2x6  d e / d camera (rot 0..2, trans 3..5)
  double Wm[4];    // rho' * Omega
  double r[2];     // -rho' * Omega e
  double chi;      // rho0
};

__device__ __forceinline__ void proj_linearize(const Pose& T, const double* R, const double* X, const double* uv, const double* info, const double* intr, double huber, const int* rk, int k, ProjLin& L) {
  double pc[3];
  proj_error(T, X, uv, intr, L.e, pc);
  double x = pc[0], y = pc[1], z = pc[2], z_2 = z * z, fx = intr[0], fy = intr[1];
  double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double s = 0;
#pragma unroll
      for (int q = 0; q < 3; q++) s += (-1. / z * tmp[3 * i + q]) * R[3 * q + j];
      L.Jp[3 * i + j] = s;
    }
  L.Jc[0] = x * y / z_2 * fx; L.Jc[1] = -(1 + (x * x / z_2)) * fx; L.Jc[2] = y / z * fx; L.Jc[3] = -1. / z * fx; L.Jc[4] = 0; L.Jc[5] = x / z_2 * fx;
  L.Jc[6] = (1 + y * y / z_2) * fy; L.Jc[7] = -x * y / z_2 * fy; L.Jc[8] = -x / z * fy; L.Jc[9] = 0; L.Jc[10] = -1. / z * fy; L.Jc[11] = y / z_2 * fy;
  double c = L.e[0] * (info[0] * L.e[0] + info[1] * L.e[1]) + L.e[1] * (info[2] * L.e[0] + info[3] * L.e[1]);
  double rho1;
  proj_rho(rk, k, huber, c, L.chi, rho1);
#pragma unroll
  for (int i = 0; i < 4; i++) L.Wm[i] = rho1 * info[i];
  L.r[0] = -(info[0] * L.e[0] + info[1] * L.e[1]) * rho1;
  L.r[1] = -(info[2] * L.e[0] + info[3] * L.e[1]) * rho1;
}

// ---------------------------------------------------------------------------------------------------
// (device bodies with the block index / block count as arguments: ba_chi2_kernel below runs both kinds of block in ONE launch)
__device__ __forceinline__ void chi2_proj_block(const BaView& v, int bid, int nblocks, double* ws) {
  double acc = 0;
  for (int k = bid * 256 + threadIdx.x; k < v.n_proj; k += nblocks * 256) {
    Pose T = pose_load(v.cams + 7 * v.pm_cam[k]);
    double e[2], pc[3];
    proj_error(T, v.points + 3 * v.pm_pt[k], v.pm_uv + 2 * k, v.intr_u ? v.intr_u : v.pm_intr + 4 * k, e, pc);
    const double* info = v.info_u ? v.info_u : v.pm_info + 4 * k;
    double c = e[0] * (info[0] * e[0] + info[1] * e[1]) + e[1] * (info[2] * e[0] + info[3] * e[1]);
    double rho0, rho1;
    proj_rho(v.pm_rk, k, v.pm_huber[k], c, rho0, rho1);
    acc += rho0;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) v.chi_partial[bid] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

__device__ __forceinline__ double quad_form(const double* e, const double* info, int n) {
  double s = 0;
  for (int i = 0; i < n; i++) {
    double t = 0;
    for (int j = 0; j < n; j++) t += info[n * i + j] * e[j];
    s += e[i] * t;
  }
  return s;
}

// one lane per cuboid / odometry edge; partial sums appended after the projection partials
__device__ __forceinline__ void chi2_pose_edges_wave(const BaView& v, int partial_off, int bid, int lane) {
  int k = bid * 64 + lane;
  double c = 0;
  if (k < v.n_cub3) {
    double e[9];
    if (v.ce_active[k]) cuboid_edge_error(pose_load(v.cams + 7 * v.ce_cam[k]), cube_load(v.cubes + 10 * v.ce_cub[k]), cube_load(v.ce_meas + 10 * k), e);
    if (v.ce_active[k]) c = quad_form(e, v.ce_info + 81 * k, 9);
    if (v.ce_active[k] && v.ce_rk && v.ce_rk[k]) { double r1; robust_rho(v.ce_rk[k], v.ce_rdelta[k], c, c, r1); }
  } else if (k < v.n_cub) {
    const int q = k - v.n_cub3;
    double e[4];
    if (v.ce_active[k]) {
      cuboid_proj_error(pose_load(v.cams + 7 * v.ce_cam[k]), cube_load(v.cubes + 10 * v.ce_cub[k]), v.pe_K + 9 * (size_t)q, v.pe_meas + 4 * (size_t)q, e);
      c = quad_form(e, v.pe_info + 16 * (size_t)q, 4);
      if (v.ce_rk && v.ce_rk[k]) { double r1; robust_rho(v.ce_rk[k], v.ce_rdelta[k], c, c, r1); }
    }
  } else if (k < v.n_cub + v.n_odom) {
    int q = k - v.n_cub;
    double e[6];
    if (v.oe_active[q]) {
      odom_edge_error(pose_load(v.cams + 7 * v.oe_i[q]), pose_load(v.cams + 7 * v.oe_j[q]), pose_load(v.oe_meas + 7 * q), e);
      c = quad_form(e, v.oe_info + 36 * q, 6);
      if (v.oe_rk && v.oe_rk[q]) { double r1; robust_rho(v.oe_rk[q], v.oe_rdelta[q], c, c, r1); }
    }
  }
  c = wave_sum(c);
  if (lane == 0) v.chi_partial[partial_off + bid] = c;
}
// chi2 of every active edge in one launch: the blocks of the cuboid / odometry edges FIRST (a wave per 64 edges: few, long -- four SE3 logs
// per cuboid edge), the projection edges' blocks behind them.  The partial sums keep their places: [0, nb_proj) projection blocks, then one
// per 64 pose edges -- the same values in the same slots as the two launches this replaces (round 6: one dispatch less on every trial's chain).
__global__ __launch_bounds__(256) void ba_chi2_kernel(BaView v, int nb_proj, int nb_pose) {
  __shared__ double ws[4];
  const int nb4 = (nb_pose + 3) / 4;
  if ((int)blockIdx.x < nb4) {
    const int pb = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pb < nb_pose) chi2_pose_edges_wave(v, nb_proj, pb, threadIdx.x & 63);
    return;
  }
  chi2_proj_block(v, blockIdx.x - nb4, nb_proj, ws);
}

// x^T (lambda x + b) of the LM gain ratio (optimization_algorithm_levenberg.cpp:117-126, computeScale :182-189) over
// the pose increments (x in v.rhs, b in bcam / bcub) and the landmark increments (xl, bl); fixed-shape reduction:
// SCALE_BLOCKS partial sums, added up by the host in order.
enum { SCALE_BLOCKS = 256 };
__device__ __forceinline__ void scale_block(const BaView& v, const double* __restrict__ lamp, double* partial, int bid, double* ws) {
  const double lambda_lm = lamp[0], lambda_pose = lamp[1];     // (lambda of the trial, read from device memory: the launch sequence of a trial is a fixed graph)
  double acc = 0;
  const int n = v.np + v.nc + v.no;
  for (int i = bid * 256 + threadIdx.x; i < n; i += SCALE_BLOCKS * 256) {
    if (i < v.np) {
      if (v.pt_free[i])
#pragma unroll
        for (int d = 0; d < 3; d++) { const double x = v.xl[3 * (size_t)i + d]; acc += x * (lambda_lm * x + v.bl[3 * (size_t)i + d]); }
    } else if (i < v.np + v.nc) {
      const int c = i - v.np, col = v.cam_col[c];
      if (col >= 0)
#pragma unroll
        for (int d = 0; d < 6; d++) { const double x = v.rhs[col + d]; acc += x * (lambda_pose * x + v.bcam[6 * c + d]); }
    } else {
      const int o = i - v.np - v.nc, col = v.cub_col[o];
      if (col >= 0)
#pragma unroll
        for (int d = 0; d < 9; d++) { const double x = v.rhs[col + d]; acc += x * (lambda_pose * x + v.bcub[9 * o + d]); }
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[bid] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
__global__ __launch_bounds__(256) void ba_scale_kernel(BaView v, const double* __restrict__ lamp, double* partial) {
  __shared__ double ws[4];
  scale_block(v, lamp, partial, blockIdx.x, ws);
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_lin_cam_kernel(BaView v) {
  int c = blockIdx.x;
  if (v.cam_col[c] < 0) return;
  __shared__ double red[4][27];
  Pose T = pose_load(v.cams + 7 * c);
  double R[9];
  pose_rotmat(T, R);
  double A[21], b[6];
#pragma unroll
  for (int i = 0; i < 21; i++) A[i] = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) b[i] = 0;
  int e0 = v.cam_ptr[c], e1 = v.cam_ptr[c + 1];
  for (int k = e0 + threadIdx.x; k < e1; k += 256) {
    ProjLin L;
    proj_linearize(T, R, v.points + 3 * v.cm_pt[k], v.cm_uv + 2 * k, v.info_u ? v.info_u : v.cm_info + 4 * k, v.intr_u ? v.intr_u : v.cm_intr + 4 * k, v.cm_huber[k], v.cm_rk, k, L);
    // JW = Jc^T W (6x2)
    int q = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      double jw0 = L.Jc[i] * L.Wm[0] + L.Jc[6 + i] * L.Wm[2];
      double jw1 = L.Jc[i] * L.Wm[1] + L.Jc[6 + i] * L.Wm[3];
#pragma unroll
      for (int j = i; j < 6; j++) A[q++] += jw0 * L.Jc[j] + jw1 * L.Jc[6 + j];
      b[i] += L.Jc[i] * L.r[0] + L.Jc[6 + i] * L.r[1];
    }
  }
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 21; i++) { double s = wave_sum(A[i]); if (lane == 0) red[w][i] = s; }
#pragma unroll
  for (int i = 0; i < 6; i++) { double s = wave_sum(b[i]); if (lane == 0) red[w][21 + i] = s; }
  __syncthreads();
  if (threadIdx.x < 27) {
    double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (threadIdx.x < 21) {
      // upper-triangular index -> (i, j)
      int t = threadIdx.x, i = 0;
      while (t >= 6 - i) { t -= 6 - i; i++; }
      int j = i + t;
      v.Hcam[36 * c + 6 * i + j] = s;
      v.Hcam[36 * c + 6 * j + i] = s;
    } else {
      v.bcam[6 * c + (threadIdx.x - 21)] = s;
    }
  }
}

// Landmark side of the projection edges: A_jj (3x3), b_j and the 6x3 H_pl block of every edge.  One LANE PER EDGE (edges are stored
// landmark-major, so a wavefront reads its 64 edge records -- uv, information, intrinsics, kernel width, camera id -- as coalesced
// streams; a lane per landmark read them 80-byte-strided and fetched every line ~3 times: 415 MB for 150 MB of records by PMC).
// A workgroup owns LIN_PT_GROUP consecutive landmarks = one contiguous run of edges; every edge's J_p^T W J_p / J_p^T r terms go to
// LDS and the landmark's first lane adds them up in edge order -- the order, and therefore every bit, of the sequential loop this
// replaces.  Groups with more than LIN_PT_CAP edges (landmarks with very long tracks) walk their edges per landmark instead.
enum { LIN_PT_GROUP = 48, LIN_PT_CAP = 512 };

__device__ __forceinline__ void lin_pt_edge(const BaView& v, int k, bool free_pt, double* H6, double* b3) {
  const int c = v.pm_cam[k];
  Pose T = pose_load(v.cams + 7 * c);
  double R[9];
  pose_rotmat(T, R);
  ProjLin L;
  proj_linearize(T, R, v.points + 3 * v.pm_pt[k], v.pm_uv + 2 * k, v.info_u ? v.info_u : v.pm_info + 4 * k, v.intr_u ? v.intr_u : v.pm_intr + 4 * k, v.pm_huber[k], v.pm_rk, k, L);
  double pw[6];  // Jp^T W (3x2)
#pragma unroll
  for (int i = 0; i < 3; i++) {
    pw[2 * i] = L.Jp[i] * L.Wm[0] + L.Jp[3 + i] * L.Wm[2];
    pw[2 * i + 1] = L.Jp[i] * L.Wm[1] + L.Jp[3 + i] * L.Wm[3];
  }
  int q = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = i; j < 3; j++) H6[q++] = pw[2 * i] * L.Jp[j] + pw[2 * i + 1] * L.Jp[3 + j];
    b3[i] = L.Jp[i] * L.r[0] + L.Jp[3 + i] * L.r[1];
  }
  double* Wk = v.W + 18 * (size_t)k;
  const bool both = free_pt && v.cam_col[c] >= 0;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const double jw0 = L.Jc[i] * L.Wm[0] + L.Jc[6 + i] * L.Wm[2];
    const double jw1 = L.Jc[i] * L.Wm[1] + L.Jc[6 + i] * L.Wm[3];
#pragma unroll
    for (int j = 0; j < 3; j++) Wk[3 * i + j] = both ? (jw0 * L.Jp[j] + jw1 * L.Jp[3 + j]) : 0.0;
  }
}

__global__ __launch_bounds__(256) void ba_lin_pt_kernel(BaView v) {
  __shared__ double part[9][LIN_PT_CAP];     // [term][edge of the group]: consecutive lanes, consecutive words
  const int p0 = blockIdx.x * LIN_PT_GROUP, p1 = min(v.np, p0 + LIN_PT_GROUP);
  if (p0 >= v.np) return;
  const int e0 = v.pt_ptr[p0], e1 = v.pt_ptr[p1];
  const bool staged = e1 - e0 <= LIN_PT_CAP;
  if (staged) {
    for (int k = e0 + threadIdx.x; k < e1; k += 256) {
      double H6[6], b3[3];
      lin_pt_edge(v, k, v.pt_free[v.pm_pt[k]] != 0, H6, b3);
#pragma unroll
      for (int i = 0; i < 6; i++) part[i][k - e0] = H6[i];
#pragma unroll
      for (int i = 0; i < 3; i++) part[6 + i][k - e0] = b3[i];
    }
    __syncthreads();
  }
  const int p = p0 + threadIdx.x;
  if (threadIdx.x >= LIN_PT_GROUP || p >= p1) return;
  double H[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
  for (int k = v.pt_ptr[p]; k < v.pt_ptr[p + 1]; k++) {
    double H6[6], b3[3];
    if (staged) {
#pragma unroll
      for (int i = 0; i < 6; i++) H6[i] = part[i][k - e0];
#pragma unroll
      for (int i = 0; i < 3; i++) b3[i] = part[6 + i][k - e0];
    } else {
      lin_pt_edge(v, k, v.pt_free[p] != 0, H6, b3);
    }
#pragma unroll
    for (int i = 0; i < 6; i++) H[i] += H6[i];
#pragma unroll
    for (int i = 0; i < 3; i++) b[i] += b3[i];
  }
  double* Hp = v.Hll + 9 * (size_t)p;
  Hp[0] = H[0]; Hp[1] = H[1]; Hp[2] = H[2]; Hp[3] = H[1]; Hp[4] = H[3]; Hp[5] = H[4]; Hp[6] = H[2]; Hp[7] = H[4]; Hp[8] = H[5];
  v.bl[3 * p] = b[0]; v.bl[3 * p + 1] = b[1]; v.bl[3 * p + 2] = b[2];
}

// ---- numeric-Jacobian edges -------------------------------------------------------------------------
// One edge per 16 lanes: lane d evaluates the two perturbed errors of Jacobian column d (lane 15 the unperturbed
// error), the columns meet in LDS, and lane i then forms row i of J^T Omega J and of -J^T Omega e.
// NA / NB = tangent dimensions of the two vertices, D = error dimension; J is stored J[d][r] (column d, row r).
template <int D, int NA, int NB>
// w = rho' of the edge's robust kernel (1 without one: the products below are then exactly the unweighted ones): Omega is weighted
// entry by entry like robustInformation() (base_edge.h:96-102), which also weights b = -J^T (rho' Omega) e (base_binary_edge.hpp:99).
__device__ __forceinline__ void edge_rows_from_columns(const double (*J)[D], const double* e0, const double* __restrict__ info, int i,
                                                       double* Haa, double* Hbb, double* Hab, double* ba, double* bb, double w) {
  constexpr int N = NA + NB;
  if (i >= N) return;
  double t[D];   // t = J(:, i)^T (w Omega)
#pragma unroll
  for (int j = 0; j < D; j++) { double s = 0; for (int k = 0; k < D; k++) s = fma(J[i][k], w * info[D * k + j], s); t[j] = s; }
  double bi = 0;
#pragma unroll
  for (int k = 0; k < D; k++) bi = fma(t[k], e0[k], bi);
  if (i < NA) ba[i] = -bi; else bb[i - NA] = -bi;
  for (int j = 0; j < N; j++) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < D; k++) s = fma(t[k], J[j][k], s);
    if (i < NA) { if (j < NA) Haa[i * NA + j] = s; else Hab[i * NB + (j - NA)] = s; }
    else if (j >= NA) Hbb[(i - NA) * NB + (j - NA)] = s;
  }
}

// 32 lanes per edge: lane (d, s), d = column 0..14 (camera tangent 0-5, cuboid tangent 6-14), s = 0 / 1 evaluates the error at
// +delta / -delta of that column (lane (15, 0): the unperturbed error), so the two oplus + computeError chains of a column run side
// by side instead of one after the other; the pair meets through a lane exchange.
__device__ __forceinline__ double lane_xor16(double x) {
  const int lo = __shfl_xor(__double2loint(x), 16), hi = __shfl_xor(__double2hiint(x), 16);
  return __hiloint2double(hi, lo);
}

// (one instance per edge class -- IS3D: EdgeSE3Cuboid, the edges [0, n_cub3) of the combined list; else EdgeSE3CuboidProj, [n_cub3, n_cub) --
// so that the bounding-box path's eight unrolled corners do not set the register budget of the 9-dim path: 196 -> 168 VGPRs, three
// wavefronts per SIMD instead of two)
__device__ __forceinline__ void odom_edge_block(const BaView& v, int blk, int tid, double (*J)[16][6]);
// odom_blocks (IS3D instance only): that many workgroups IN FRONT of the cuboid edges' do the odometry edges instead, eight per workgroup (round 6,
// last: their ~250 short wavefronts used to be a launch of their own behind this one on the side stream -- 20-38 us that ba_accum_pose_kernel waited
// for; on a stream of their own they queue for wave slots behind this kernel's and the chi2 kernel's workgroups; as this launch's FIRST workgroups
// they are dispatched first and are done long before the cuboid edges)
template <bool IS3D>
__global__ __launch_bounds__(128, 3) void ba_cub_edge_kernel(BaView v, int odom_blocks) {
  __shared__ double Jbuf[2 * 4 * 16 * 6];      // the cuboid edges' [4][16][9] column store, or two odometry column stores of [4][16][6]
  static_assert(2 * 4 * 16 * 6 >= 4 * 16 * 9, "one buffer for both kinds of workgroup");
  if (IS3D && (int)blockIdx.x < odom_blocks) {
    odom_edge_block(v, 2 * blockIdx.x + (threadIdx.x >> 6), threadIdx.x & 63, reinterpret_cast<double (*)[16][6]>(Jbuf + 4 * 16 * 6 * (threadIdx.x >> 6)));
    return;
  }
  double (*J)[16][9] = reinterpret_cast<double (*)[16][9]>(Jbuf);    // [edge in block][column d (15 = e0)][row]
  const int sub = threadIdx.x >> 5, h = threadIdx.x & 31, d = h & 15, sgn = h >> 4;
  const int k = (IS3D ? 0 : v.n_cub3) + ((int)blockIdx.x - (IS3D ? odom_blocks : 0)) * 4 + sub;
  const bool live = k < (IS3D ? v.n_cub3 : v.n_cub);
  const double delta = 1e-9, scalar = 1.0 / (2 * delta);
  const double step = sgn ? -delta : delta;
  constexpr bool is3d = IS3D;
  double (*Jq)[4] = reinterpret_cast<double (*)[4]>(&J[sub][0][0]);   // the 4-row view of this edge's column store
  if (!IS3D && live) {
    const int q = k - v.n_cub3;
    Pose T = pose_load(v.cams + 7 * v.ce_cam[k]);
    Cube cube = cube_load(v.cubes + 10 * v.ce_cub[k]);
    const double* K = v.pe_K + 9 * (size_t)q;
    const double* meas = v.pe_meas + 4 * (size_t)q;
    const bool act = v.ce_active[k] != 0;
    const bool fa = act && v.cam_col[v.ce_cam[k]] >= 0, fb = act && v.cub_col[v.ce_cub[k]] >= 0;
    double e1[4] = {0, 0, 0, 0};
    if (d == 15) {
      if (sgn == 0) cuboid_proj_error(T, cube, K, meas, e1);
    } else if (d < 6) {
      double add[6] = {0, 0, 0, 0, 0, 0};
      add[d] = step;
      if (fa) cuboid_proj_error(cam_oplus(T, add), cube, K, meas, e1);
    } else {
      double add[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      add[d - 6] = step;
      if (fb) cuboid_proj_error(T, cube_oplus(cube, add), K, meas, e1);
    }
    const bool on = d < 6 ? fa : fb;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const double e2 = lane_xor16(e1[r]);      // the partner's error: -delta for the s = 0 lanes
      if (sgn == 0) Jq[d][r] = d == 15 ? (act ? e1[r] : 0.0) : (on ? scalar * (e1[r] - e2) : 0.0);
    }
  }
  if (IS3D && live) {
    Pose T = pose_load(v.cams + 7 * v.ce_cam[k]);
    Cube cube = cube_load(v.cubes + 10 * v.ce_cub[k]);
    Cube meas = cube_load(v.ce_meas + 10 * k);
    const bool act = v.ce_active[k] != 0;  // sharded BA: the edge belongs to another rank -> zero blocks
    const bool fa = act && v.cam_col[v.ce_cam[k]] >= 0, fb = act && v.cub_col[v.ce_cub[k]] >= 0;
    double e1[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // Two phases.  (1) The unperturbed error: min_log_error's four yaw candidates on four lanes (h = 0 .. 3) instead of a loop on one;
    // the norms meet through lane exchanges, the winner by the reference's rule (first minimum, strict <; g2o_Object.h:76-114).
    // (2) The 30 perturbed evaluations: a 1e-9 step moves a candidate's norm by ~1e-7 at most, so where the winner leads the runner-up
    // by more than 1e-3 (1 + norm) every perturbed evaluation has the same winner and computes THAT candidate only -- the same
    // instructions on the same operands as the loop would run for it, hence the same bits; otherwise (a near tie: yaw near 45 degrees
    // off with a square footprint) the full loop.  Two candidate evaluations per wavefront instead of four: 135 -> ~80 us at C4.
    Cube esti0;
    esti0.pose = pose_mul(pose_inv(T), meas.pose);
    esti0.scale[0] = meas.scale[0]; esti0.scale[1] = meas.scale[1]; esti0.scale[2] = meas.scale[2];
    double ec[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double nc = 0.0;
    if (h < 4) nc = cube_log_error_candidate(cube, esti0, h, ec);
    const int gbase = threadIdx.x & 32;          // first lane of this edge's 32 within the wavefront
    double n4[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { const int lo = __shfl(__double2loint(nc), gbase + i), hi = __shfl(__double2hiint(nc), gbase + i); n4[i] = __hiloint2double(hi, lo); }
    int win = 0;
    double best_n = n4[0];
#pragma unroll
    for (int i = 1; i < 4; i++) if (n4[i] < best_n) { best_n = n4[i]; win = i; }
    bool clear = best_n == best_n;               // (a NaN leader: the loop decides)
#pragma unroll
    for (int i = 0; i < 4; i++) if (i != win && !(n4[i] - best_n > 1e-3 * (1.0 + best_n))) clear = false;
    // the unperturbed error: the winner's, fetched from the lane that holds it (every lane takes part in the exchange: an inactive
    // source lane would read as zero)
#pragma unroll
    for (int r = 0; r < 9; r++) {
      const int lo = __shfl(__double2loint(ec[r]), gbase + win), hi = __shfl(__double2hiint(ec[r]), gbase + win);
      if (d == 15) e1[r] = __hiloint2double(hi, lo);
    }
    if (d < 6) {
      double add[6] = {0, 0, 0, 0, 0, 0};
      add[d] = step;
      if (fa) {
        const Pose Tp = cam_oplus(T, add);
        if (clear) { Cube es; es.pose = pose_mul(pose_inv(Tp), meas.pose); es.scale[0] = meas.scale[0]; es.scale[1] = meas.scale[1]; es.scale[2] = meas.scale[2]; cube_log_error_candidate(cube, es, win, e1); }
        else cuboid_edge_error(Tp, cube, meas, e1);
      }
    } else if (d < 15) {
      double add[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      add[d - 6] = step;
      if (fb) {
        if (clear) cube_log_error_candidate(cube_oplus(cube, add), esti0, win, e1);
        else cuboid_edge_error(T, cube_oplus(cube, add), meas, e1);
      }
    }
    const bool on = d < 6 ? fa : fb;
#pragma unroll
    for (int r = 0; r < 9; r++) {
      const double e2 = lane_xor16(e1[r]);
      if (sgn == 0) J[sub][d][r] = d == 15 ? (act ? e1[r] : 0.0) : (on ? scalar * (e1[r] - e2) : 0.0);
    }
  }
  __syncthreads();
  double w = 1.0;      // rho' of the edge's kernel at its chi2 (every lane of the edge computes the same value)
  if (live && sgn == 0 && v.ce_rk && v.ce_rk[k]) {
    const double c = is3d ? quad_form(J[sub][15], v.ce_info + 81 * (size_t)k, 9) : quad_form(Jq[15], v.pe_info + 16 * (size_t)(k - v.n_cub3), 4);
    double r0;
    robust_rho(v.ce_rk[k], v.ce_rdelta[k], c, r0, w);
  }
  if (live && is3d && sgn == 0)
    edge_rows_from_columns<9, 6, 9>(J[sub], J[sub][15], v.ce_info + 81 * (size_t)k, d, v.ce_Hcc + 36 * (size_t)k, v.ce_Hoo + 81 * (size_t)k,
                                    v.ce_Hco + 54 * (size_t)k, v.ce_bc + 6 * (size_t)k, v.ce_bo + 9 * (size_t)k, w);
  if (live && !is3d && sgn == 0)
    edge_rows_from_columns<4, 6, 9>(Jq, Jq[15], v.pe_info + 16 * (size_t)(k - v.n_cub3), d, v.ce_Hcc + 36 * (size_t)k, v.ce_Hoo + 81 * (size_t)k,
                                    v.ce_Hco + 54 * (size_t)k, v.ce_bc + 6 * (size_t)k, v.ce_bo + 9 * (size_t)k, w);
}

// (four odometry edges per 64 threads; blk = which four, tid = 0 .. 63, J: the four edges' column store.  Every thread of the workgroup reaches the barrier.)
__device__ __forceinline__ void odom_edge_block(const BaView& v, int blk, int tid, double (*J)[16][6]) {
  const int sub = tid >> 4, d = tid & 15;
  const int k = blk * 4 + sub;
  const bool live = k < v.n_odom;
  const double delta = 1e-9, scalar = 1.0 / (2 * delta);
  if (live) {
    Pose T1 = pose_load(v.cams + 7 * v.oe_i[k]), T2 = pose_load(v.cams + 7 * v.oe_j[k]), M = pose_load(v.oe_meas + 7 * k);
    const bool act = v.oe_active[k] != 0;
    const bool fa = act && v.cam_col[v.oe_i[k]] >= 0, fb = act && v.cam_col[v.oe_j[k]] >= 0;
    double e1[6], e2[6], add[6] = {0, 0, 0, 0, 0, 0};
    if (d == 15) {
      odom_edge_error(T1, T2, M, e1);
      for (int r = 0; r < 6; r++) J[sub][15][r] = act ? e1[r] : 0.0;
    } else if (d < 6) {
      if (fa) {
        add[d] = delta; odom_edge_error(cam_oplus(T1, add), T2, M, e1);
        add[d] = -delta; odom_edge_error(cam_oplus(T1, add), T2, M, e2);
      }
      for (int r = 0; r < 6; r++) J[sub][d][r] = fa ? scalar * (e1[r] - e2[r]) : 0.0;
    } else if (d < 12) {
      if (fb) {
        add[d - 6] = delta; odom_edge_error(T1, cam_oplus(T2, add), M, e1);
        add[d - 6] = -delta; odom_edge_error(T1, cam_oplus(T2, add), M, e2);
      }
      for (int r = 0; r < 6; r++) J[sub][d][r] = fb ? scalar * (e1[r] - e2[r]) : 0.0;
    }
  }
  __syncthreads();
  if (live) {
    double w = 1.0;
    if (v.oe_rk && v.oe_rk[k]) { double r0; robust_rho(v.oe_rk[k], v.oe_rdelta[k], quad_form(J[sub][15], v.oe_info + 36 * (size_t)k, 6), r0, w); }
    edge_rows_from_columns<6, 6, 6>(J[sub], J[sub][15], v.oe_info + 36 * (size_t)k, d, v.oe_Hii + 36 * (size_t)k, v.oe_Hjj + 36 * (size_t)k,
                                    v.oe_Hij + 36 * (size_t)k, v.oe_bi + 6 * (size_t)k, v.oe_bj + 6 * (size_t)k, w);
  }
}

__global__ __launch_bounds__(64) void ba_odom_edge_kernel(BaView v) {
  __shared__ double J[4][16][6];
  odom_edge_block(v, blockIdx.x, threadIdx.x, J);
}

// gather the numeric-edge blocks into the pose vertices' A_ii / b_i (fixed order: deterministic)
__global__ __launch_bounds__(128) void ba_accum_pose_kernel(BaView v, int zero_cam_first) {
  int vid = blockIdx.x, t = threadIdx.x;
  if (vid < v.nc) {
    int c = vid;
    if (v.cam_col[c] < 0 || t >= 42) return;
    double s = zero_cam_first ? 0.0 : ((t < 36) ? v.Hcam[36 * c + t] : v.bcam[6 * c + t - 36]);
    for (int q = v.cam_ce_ptr[c]; q < v.cam_ce_ptr[c + 1]; q++) { int k = v.cam_ce_idx[q]; s += (t < 36) ? v.ce_Hcc[36 * (size_t)k + t] : v.ce_bc[6 * (size_t)k + t - 36]; }
    for (int q = v.cam_oei_ptr[c]; q < v.cam_oei_ptr[c + 1]; q++) { int k = v.cam_oei_idx[q]; s += (t < 36) ? v.oe_Hii[36 * (size_t)k + t] : v.oe_bi[6 * (size_t)k + t - 36]; }
    for (int q = v.cam_oej_ptr[c]; q < v.cam_oej_ptr[c + 1]; q++) { int k = v.cam_oej_idx[q]; s += (t < 36) ? v.oe_Hjj[36 * (size_t)k + t] : v.oe_bj[6 * (size_t)k + t - 36]; }
    if (t < 36) v.Hcam[36 * c + t] = s; else v.bcam[6 * c + t - 36] = s;
  } else {
    int o = vid - v.nc;
    if (o >= v.no || v.cub_col[o] < 0 || t >= 90) return;
    double s = 0;
    for (int q = v.cub_ce_ptr[o]; q < v.cub_ce_ptr[o + 1]; q++) { int k = v.cub_ce_idx[q]; s += (t < 81) ? v.ce_Hoo[81 * (size_t)k + t] : v.ce_bo[9 * (size_t)k + t - 81]; }
    if (t < 81) v.Hcub[81 * o + t] = s; else v.bcub[9 * o + t - 81] = s;
  }
}

// ---- Schur complement ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_prep_kernel(BaView v, const double* __restrict__ lamp) {
  const double lambda = lamp[0];
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= v.np) return;
  double D[9], Di[9];
  // a landmark without edges on this rank (sharded BA: it lives elsewhere) gets a zero increment
  bool free_pt = v.pt_free[p] != 0 && v.pt_ptr[p + 1] > v.pt_ptr[p];
  if (free_pt) {
#pragma unroll
    for (int i = 0; i < 9; i++) D[i] = v.Hll[9 * (size_t)p + i];
    D[0] += lambda; D[4] += lambda; D[8] += lambda;
    inv3x3(D, Di);
  } else {
#pragma unroll
    for (int i = 0; i < 9; i++) Di[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < 9; i++) v.Dinv[9 * (size_t)p + i] = Di[i];
  double b[3] = {v.bl[3 * p], v.bl[3 * p + 1], v.bl[3 * p + 2]}, db[3];
  mat3_vec(Di, b, db);
  v.dbl[3 * p] = db[0]; v.dbl[3 * p + 1] = db[1]; v.dbl[3 * p + 2] = db[2];
}

// WD = W D^-1, one thread per (projection edge, row of its 6 x 3 block): reads and writes are contiguous across the wave
__global__ __launch_bounds__(256) void ba_wd_kernel(BaView v) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= 6 * (size_t)v.n_proj) return;
  const int k = (int)(t / 6);
  const double* Di = v.Dinv + 9 * (size_t)v.pm_pt[k];
  const double w0 = v.W[3 * t], w1 = v.W[3 * t + 1], w2 = v.W[3 * t + 2];
#pragma unroll
  for (int c = 0; c < 3; c++) v.WD[3 * t + c] = w0 * Di[c] + w1 * Di[3 + c] + w2 * Di[6 + c];
}

// camera part of the reduced system: S_cc = A_cc + lambda I, b_schur,c = b_c - sum_e W_e (D^-1 b_l)
__global__ __launch_bounds__(256) void ba_cam_rhs_kernel(BaView v, const double* __restrict__ lamp) {
  const double lambda = lamp[0];
  int c = blockIdx.x;
  int col = v.cam_col[c];
  if (col < 0) return;
  __shared__ double red[4][6];
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int k = v.cam_ptr[c] + threadIdx.x; k < v.cam_ptr[c + 1]; k += 256) {
    int pm = v.cm_pm[k];
    const double* Wk = v.W + 18 * (size_t)pm;
    const double* db = v.dbl + 3 * v.cm_pt[k];
#pragma unroll
    for (int r = 0; r < 6; r++) acc[r] += Wk[3 * r] * db[0] + Wk[3 * r + 1] * db[1] + Wk[3 * r + 2] * db[2];
  }
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < 6; r++) { double s = wave_sum(acc[r]); if (lane == 0) red[w][r] = s; }
  __syncthreads();
  if (threadIdx.x < 6) {
    double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    v.rhs[col + threadIdx.x] = v.bcam[6 * c + threadIdx.x] - s;
  }
  if (threadIdx.x < 36) {
    int i = threadIdx.x / 6, j = threadIdx.x % 6;
    if (i >= j) *ba_S_at(v, col + i, col + j) = v.Hcam[36 * c + threadIdx.x] + ((i == j && col >= v.lam_lo && col < v.lam_hi) ? lambda : 0.0);
  }
}

__global__ __launch_bounds__(128) void ba_cub_scatter_kernel(BaView v, const double* __restrict__ lamp) {
  const double lambda = lamp[0];
  int o = blockIdx.x, t = threadIdx.x;
  int col = v.cub_col[o];
  if (col < 0) return;
  if (t < 81) {
    int i = t / 9, j = t % 9;
    if (i >= j) *ba_S_at(v, col + i, col + j) = v.Hcub[81 * o + t] + ((i == j && col >= v.lam_lo && col < v.lam_hi) ? lambda : 0.0);
  } else if (t < 90) {
    v.rhs[col + t - 81] = v.bcub[9 * o + t - 81];
  }
}

// off-diagonal pose blocks: camera-cuboid (6x9) per cuboid edge, camera-camera (6x6) per odometry edge
__device__ __forceinline__ void ba_offdiag_body(const BaView& v, int k, int t) {
  if (k < v.n_cub) {
    if (v.elim) return;       // the cuboids are not part of the reduced system
    int ca = v.cam_col[v.ce_cam[k]], cb = v.cub_col[v.ce_cub[k]];
    if (ca < 0 || cb < 0 || t >= 54) return;
    int i = t / 9, j = t % 9;
    double val = v.ce_Hco[54 * (size_t)k + t];
    // block (ca+i, cb+j): stored in the lower triangle, whichever vertex comes later in the ordering
    if (cb > ca) atomicAdd(ba_S_at(v, cb + j, ca + i), val); else atomicAdd(ba_S_at(v, ca + i, cb + j), val);
  } else {
    int q = k - v.n_cub;
    if (q >= v.n_odom) return;
    int ca = v.cam_col[v.oe_i[q]], cb = v.cam_col[v.oe_j[q]];
    if (ca < 0 || cb < 0 || t >= 36) return;
    int i = t / 6, j = t % 6;
    double val = v.oe_Hij[36 * (size_t)q + t];
    if (cb > ca) atomicAdd(ba_S_at(v, cb + j, ca + i), val); else atomicAdd(ba_S_at(v, ca + i, cb + j), val);
  }
}
__global__ __launch_bounds__(64) void ba_offdiag_kernel(BaView v) { ba_offdiag_body(v, blockIdx.x, threadIdx.x); }

// One wavefront per covisible camera pair.  Every lane walks its own landmarks of the pair (entries q0 + lane,
// + 64, ...), accumulating the whole 6 x 6 block W_a D^-1 W_b^T in registers (each lane streams two contiguous
// 144-byte records per entry); the 64 partial blocks are summed through an LDS tile in a fixed order.
__global__ __launch_bounds__(128) void ba_schur_kernel(BaView v) {
  __shared__ double red[2][64][37];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 2 + wv;
  if (pair < v.n_pairs) {
    const int q0 = v.pair_ptr[pair], q1 = v.pair_ptr[pair + 1];
    double acc[36];
#pragma unroll
    for (int o = 0; o < 36; o++) acc[o] = 0.0;
    for (int q = q0 + lane; q < q1; q += 64) {
      const double* x = v.WD + 18 * (size_t)v.ent_a[q];
      const double* y = v.W + 18 * (size_t)v.ent_b[q];
      double xv[18], yv[18];
#pragma unroll
      for (int i = 0; i < 18; i++) { xv[i] = x[i]; yv[i] = y[i]; }
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) acc[6 * r + c] = fma(xv[3 * r + 2], yv[3 * c + 2], fma(xv[3 * r + 1], yv[3 * c + 1], fma(xv[3 * r], yv[3 * c], acc[6 * r + c])));
    }
#pragma unroll
    for (int o = 0; o < 36; o++) red[wv][lane][o] = acc[o];
  }
  __syncthreads();
  if (pair < v.n_pairs && lane < 36) {
    double s = 0;
#pragma unroll 8
    for (int j = 0; j < 64; j++) s += red[wv][j][lane];
    const int r = lane / 6, c = lane % 6;
    const int i1 = v.pair_i1[pair], i2 = v.pair_i2[pair];
    // element (i1 + r, i2 + c) of the symmetric S, i1 <= i2: its lower-triangle home is (i2 + c, i1 + r)
    if (i1 != i2 || c >= r) *ba_S_at(v, i2 + c, i1 + r) -= s;
  }
}

// ---- fused Schur product on the matrix cores ----------------------------------------------------------------------------
// block_solver.hpp:385-431 for one segment of landmarks that are seen by the same k cameras.  Their Hpl blocks form, per
// landmark j, a (6k x 3) matrix W_j (rows: camera slot a, row r of its 6 x 3 block; contiguous in the point-major W array), and
//     sum_j W_j D_j^-1 W_j^T   (6k x 6k)       sum_j W_j D_j^-1 b_l,j   (6k)
// is ONE matrix product whose contraction index runs over (landmark, landmark coordinate): A = [W_j D_j^-1]_j (6k x 3m),
// B = [W_j | b_l,j]_j^T (3m x (6k + 1)).  v_mfma_f64_16x16x4_f64 takes the K = 3 coordinates of one landmark (padded to 4) per
// issue; a wavefront walks its segment, keeping the upper block triangle of the product in accumulator registers
// (MT x MT tiles of 16 x 16), i.e. the cross-landmark reduction happens inside the matrix core, not through LDS or atomics.
// Operand lane map of the f64 16x16x4 form: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// D[row = (lane >> 4) + 4 reg][col = lane & 15].
// Every W record is read from HBM exactly once (the pair-major kernel re-read it once per camera pair), D^-1 is written once for
// the back-substitution, and W D^-1 never exists in memory.
typedef double ba_v4d __attribute__((ext_vector_type(4)));

template <int MT>
struct BaSegOperands {
  double Dm[9];        // H_ll of the landmark (lambda not yet added)
  double w[MT][3];     // W row 16 t + i
  double bcol[MT];     // B operand of N-tile u
  int p;
};

// all loads unconditional (clamped addresses, values selected afterwards): no exec-masked branches, so the compiler can count
// the loads in flight and the prefetch of the next landmark really overlaps this one's products
template <int MT>
__device__ __forceinline__ void ba_seg_load(const BaView& v, int p, int e0, int rows, int i, int kk, BaSegOperands<MT>& o) {
  o.p = p;
  const double* H = v.Hll + 9 * (size_t)p;
#pragma unroll
  for (int q = 0; q < 9; q++) o.Dm[q] = H[q];
  const double* Wp = v.W + 18 * (size_t)e0;
  const int k3 = kk < 3 ? kk : 0;
  const double blv = v.bl[3 * (size_t)p + k3];
#pragma unroll
  for (int t = 0; t < MT; t++) {
    const int row = 16 * t + i;
    const bool in = row < rows;
    const double* wr = Wp + 3 * (in ? row : rows - 1);
    const double w0 = wr[0], w1 = wr[1], w2 = wr[2];
    o.w[t][0] = in ? w0 : 0.0; o.w[t][1] = in ? w1 : 0.0; o.w[t][2] = in ? w2 : 0.0;
    // B: column `row` of [W_j^T | b_l]: W(row, kk) below `rows`, the right-hand-side column at `rows`
    const double wk = k3 == 0 ? w0 : (k3 == 1 ? w1 : w2);
    o.bcol[t] = kk < 3 ? (in ? wk : (row == rows ? blv : 0.0)) : 0.0;
  }
}

template <int MT>
__device__ __forceinline__ void ba_schur_segment(const BaView& v, double lambda, int seg, int k) {
  const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4, rows = 6 * k;
  const int q0 = v.seg_ptr[seg], q1 = v.seg_ptr[seg + 1];
  // the segment's landmark ids and first-edge slots, one per lane, broadcast with readlane inside the loop (no dependent
  // scalar loads on the chain)
  int my_p = 0, my_e0 = 0;
  if (q0 + lane < q1) { my_p = v.run_lm[q0 + lane]; my_e0 = v.pt_ptr[my_p]; }
  ba_v4d acc[MT][MT];
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int u = 0; u < MT; u++) acc[t][u] = ba_v4d{0.0, 0.0, 0.0, 0.0};
  // ping-pong operand buffers: while landmark q is multiplied, the loads of landmark q + 1 are in flight
  BaSegOperands<MT> bufA, bufB;
  const int n = q1 - q0;
  auto load = [&](BaSegOperands<MT>& o, int q) {
    const int qq = q < n ? q : n - 1;            // past the end: re-load the last landmark (harmless, keeps the loop branch-free)
    ba_seg_load<MT>(v, __builtin_amdgcn_readlane(my_p, qq), __builtin_amdgcn_readlane(my_e0, qq), rows, i, kk, o);
  };
  auto product = [&](const BaSegOperands<MT>& o) {
    double D[9], Di[9];
#pragma unroll
    for (int e = 0; e < 9; e++) D[e] = o.Dm[e];
    D[0] += lambda; D[4] += lambda; D[8] += lambda;
    inv3x3(D, Di);
    if (lane < 9) v.Dinv[9 * (size_t)o.p + lane] = Di[lane];       // for the back-substitution (block_solver.hpp:457-482)
    const int k3 = kk < 3 ? kk : 0;
    double a[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) {
      const double wd = o.w[t][0] * Di[k3] + o.w[t][1] * Di[3 + k3] + o.w[t][2] * Di[6 + k3];   // (W D^-1)(row, kk), as ba_wd_kernel
      a[t] = kk < 3 ? wd : 0.0;
    }
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
      for (int u = t; u < MT; u++) acc[t][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], o.bcol[u], acc[t][u], 0, 0, 0);
  };
  load(bufA, 0);
  for (int q = 0; q < n; q += 2) {
    load(bufB, q + 1);
    product(bufA);
    if (q + 1 < n) {            // wave-uniform
      load(bufA, q + 2);
      product(bufB);
    }
  }
  // flush: the k (k + 1) / 2 blocks (a <= b; diagonal blocks: upper triangle c >= r, what the reduced system stores) and
  // the k coefficient vectors
  const int tile0 = v.seg_tile[seg], slot0 = v.seg_slot[seg];
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int u = t; u < MT; u++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int R = 16 * t + (lane >> 4) + 4 * g, Cc = 16 * u + (lane & 15);
        if (R >= rows) continue;
        const int sa = R / 6, r = R - 6 * sa;
        const double val = acc[t][u][g];
        if (Cc < rows) {
          const int sb = Cc / 6, c = Cc - 6 * sb;
          if (sa < sb || (sa == sb && c >= r)) v.part_tiles[36 * (size_t)(tile0 + sa * k - sa * (sa - 1) / 2 + (sb - sa)) + 6 * r + c] = val;
        } else if (Cc == rows) {
          v.part_coef[6 * (size_t)(slot0 + sa) + r] = val;
        }
      }
}

// one instantiation per tile count (the segments are sorted by k, so each launch covers a contiguous range): MT = 1 for k <= 2
// (12 rows + the right-hand-side column), 2 for k <= 5 (30 + 1), 3 for k <= 7 (42 + 1), 4 for k <= 10 (60 + 1), 5 for k <= BA_FUSED_KMAX = 13 (78 + 1)
template <int MT>
__global__ __launch_bounds__(256) void ba_schur_fused_kernel(BaView v, const double* __restrict__ lamp, int seg_begin, int seg_end) {
  const double lambda = lamp[0];
  const int seg = __builtin_amdgcn_readfirstlane(seg_begin + blockIdx.x * 4 + (threadIdx.x >> 6));   // wave-uniform: scalar loads below
  if (seg >= seg_end) return;
  ba_schur_segment<MT>(v, lambda, seg, v.seg_k[seg]);
}

// ---- linearisation + Schur product in one pass over the projection edges (round 5) ------------------------------------------------
// ba_lin_pt_kernel wrote every edge's H_pl block (163 MB at C4) for ba_schur_fused_kernel to read back two kernels later, and spent
// 70-140 us of the linearisation phase doing so.  Here the segment's wavefront linearises its own edges -- a lane per edge, a chunk of
// floor(64 / k) landmarks at a time (types_six_dof_expmap.cpp:148-184, base_binary_edge.hpp:54-120: the same proj_linearize and the
// same products as ba_lin_pt_kernel, hence the same bits) -- keeps the blocks in LDS as the matrix-core operands, sums H_ll / b_l per
// landmark in edge order, inverts the damped block once per landmark (not once per lane) and runs ba_schur_segment's products.  H_pl,
// H_ll, b_l and D^-1 are still written, once, for the back-substitution (block_solver.hpp:457-482); nothing is read back.  A lane's
// camera is its slot's -- the same for every landmark of the segment -- so its pose and rotation matrix are loaded and formed once.
// Used by cs_ba_optimize from the second iteration on (the first needs H_ll for lambda's initial value before any trial), on unsharded
// graphs without long tracks or host-evaluated edges; the classic pair of kernels serves everything else and the inspection entries.
enum { LS_PLANES = 18, LS_DCOLS = 16, LS_WAVE_DOUBLES = LS_PLANES * 64 + 12 * LS_DCOLS };   // 10.75 KB per wavefront: three workgroups per CU
template <int MT>
__device__ __forceinline__ void ba_lin_schur_segment(const BaView& v, double lambda, int seg, int k, double* lds) {
  const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4, rows = 6 * k;
  const int q0 = v.seg_ptr[seg], q1 = v.seg_ptr[seg + 1], n = q1 - q0;
  double (*Wl)[64] = reinterpret_cast<double (*)[64]>(lds);                  // [18][edge of the chunk]
  double (*Dl)[LS_DCOLS] = reinterpret_cast<double (*)[LS_DCOLS]>(lds + LS_PLANES * 64);   // [12][landmark of the chunk]: D^-1 (9), b_l (3)
  // (round 6: the run entry's first edge and the slot's camera come from tables of their own -- run_e0 = pt_ptr[run_lm], seg_cam = pm_cam of the
  // segment's first landmark -- so that the head of a segment is three dependent round trips (segment scalars -> these -> pose / records)
  // instead of five; with three waves per SIMD the kernel is those round trips)
  int my_p = 0, my_e0 = 0;
  if (q0 + lane < q1) { my_p = v.run_lm[q0 + lane]; my_e0 = v.run_e0[q0 + lane]; }
  ba_v4d acc[MT][MT];
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int u = 0; u < MT; u++) acc[t][u] = ba_v4d{0.0, 0.0, 0.0, 0.0};
  const int C = min((int)LS_DCOLS, 64 / k);             // landmarks per chunk
  const int lc = lane / k, sa_l = lane - lc * k;        // this lane's (landmark of the chunk, camera slot)
  // the slot's camera (the same for every landmark of the segment)
  const int cam = v.seg_cam[v.seg_slot[seg] + (lc < C ? sa_l : 0)];
  const Pose T = pose_load(v.cams + 7 * cam);
  double R[9];
  pose_rotmat(T, R);
  const bool cam_free = v.cam_col[cam] >= 0;
  // operand rows of this lane, per 16-row tile: row = 16 t + i = 6 a + r
  int w_off[MT]; bool w_in[MT];
#pragma unroll
  for (int t = 0; t < MT; t++) {
    const int row = 16 * t + i;
    w_in[t] = row < rows;
    const int rr = w_in[t] ? row : 0, a = rr / 6, r = rr - 6 * a;
    w_off[t] = (3 * r) * 64 + a;                        // plane 3 r (+ c), edge a (+ l k)
  }
  const int k3 = kk < 3 ? kk : 0;
  // an edge's record (point, measurement, information, intrinsics, kernel width): requested one chunk ahead, with clamped
  // indices -- every load unconditional, so that the requests of the next chunk are in flight under this chunk's products
  struct EdgeRec { double X[3], uv[2], info[4], intr[4], huber; int p, e; };
  auto fetch = [&](int c0) {
    EdgeRec r;
    const int ncl = min(C, n - c0);
    const int src = (c0 < n && lc < ncl) ? c0 + lc : 0;
    r.p = __shfl(my_p, src); r.e = __shfl(my_e0, src) + ((c0 < n && lc < ncl) ? sa_l : 0);
#pragma unroll
    for (int q = 0; q < 3; q++) r.X[q] = v.points[3 * (size_t)r.p + q];
#pragma unroll
    for (int q = 0; q < 2; q++) r.uv[q] = v.pm_uv[2 * (size_t)r.e + q];
#pragma unroll
    for (int q = 0; q < 4; q++) { r.info[q] = v.info_u ? v.info_u[q] : v.pm_info[4 * (size_t)r.e + q]; r.intr[q] = v.intr_u ? v.intr_u[q] : v.pm_intr[4 * (size_t)r.e + q]; }
    r.huber = v.pm_huber[r.e];
    return r;
  };
  EdgeRec rec = fetch(0);
  for (int c0 = 0; c0 < n; c0 += C) {
    const int ncl = min(C, n - c0);
    // ---- the chunk's edges, a lane each
    double h9[9];          // this edge's J_p^T W J_p (upper triangle, 6) and J_p^T r (3)
    const int p_edge = rec.p;
    {
      const int p = rec.p, e = rec.e;
#pragma unroll
      for (int q = 0; q < 9; q++) h9[q] = 0.0;
      if (lc < ncl) {
        ProjLin L;
        proj_linearize(T, R, rec.X, rec.uv, rec.info, rec.intr, rec.huber, v.pm_rk, e, L);
        double pw[6];  // Jp^T W (3x2)
#pragma unroll
        for (int q = 0; q < 3; q++) {
          pw[2 * q] = L.Jp[q] * L.Wm[0] + L.Jp[3 + q] * L.Wm[2];
          pw[2 * q + 1] = L.Jp[q] * L.Wm[1] + L.Jp[3 + q] * L.Wm[3];
        }
        int o = 0;
#pragma unroll
        for (int q = 0; q < 3; q++) {
#pragma unroll
          for (int j = q; j < 3; j++) h9[o++] = pw[2 * q] * L.Jp[j] + pw[2 * q + 1] * L.Jp[3 + j];
          h9[6 + q] = L.Jp[q] * L.r[0] + L.Jp[3 + q] * L.r[1];
        }
        // (H_pl goes out as 18 eight-byte stores per lane.  Round 5 measured the alternative -- the chunk's records through LDS as runs of 18 k
        // contiguous doubles, 64 consecutive doubles per store instruction --: the same HBM traffic by the counters (251 MB read, 227 MB written per
        // launch at C4: L2 merges the partial lines either way) and a kernel 7 % slower for the index arithmetic; dropped.  What the kernel
        // over-fetches is its INPUT: a landmark's k edge records are 80 / 160 bytes in a stream ordered by landmark id, and the landmarks of a
        // segment are not neighbours in it.)
        // Round 6: H_pl does not go to memory at all here (231 MB written per launch at C4, 259 MB read back by the back-substitution): the only
        // reader behind this kernel, x_l = D^-1 (b_l - H_pl^T x_p), forms the blocks again from the same state with the same arithmetic
        // (backsub_lin_point) -- 20 bytes of edge record instead of a 144-byte block per edge, the same bits.
        const bool both = cam_free && v.pt_free[p] != 0;
#pragma unroll
        for (int q = 0; q < 6; q++) {
          const double jw0 = L.Jc[q] * L.Wm[0] + L.Jc[6 + q] * L.Wm[2];
          const double jw1 = L.Jc[q] * L.Wm[1] + L.Jc[6 + q] * L.Wm[3];
#pragma unroll
          for (int j = 0; j < 3; j++) {
            const double w = both ? (jw0 * L.Jp[j] + jw1 * L.Jp[3 + j]) : 0.0;
            Wl[3 * q + j][lane] = w;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- per landmark: H_ll, b_l summed in edge order (the landmark's first lane collects its k edges' terms through lane exchanges:
    // 0 + h_0 + h_1 + ..., the order and the bits of ba_lin_pt_kernel's loop); the damped block's inverse
    {
      double H[6], b[3];
#pragma unroll
      for (int q = 0; q < 6; q++) H[q] = h9[q];
#pragma unroll
      for (int q = 0; q < 3; q++) b[q] = h9[6 + q];
      for (int a = 1; a < k; a++) {
        const int from = (lane + a) & 63;
#pragma unroll
        for (int q = 0; q < 9; q++) {
          const int lo = __shfl(__double2loint(h9[q]), from), hi = __shfl(__double2hiint(h9[q]), from);
          const double t = __hiloint2double(hi, lo);
          if (q < 6) H[q] += t; else b[q - 6] += t;
        }
      }
      if (sa_l == 0 && lc < ncl) {
        double* Hp = v.Hll + 9 * (size_t)p_edge;
        Hp[0] = H[0]; Hp[1] = H[1]; Hp[2] = H[2]; Hp[3] = H[1]; Hp[4] = H[3]; Hp[5] = H[4]; Hp[6] = H[2]; Hp[7] = H[4]; Hp[8] = H[5];
        v.bl[3 * (size_t)p_edge] = b[0]; v.bl[3 * (size_t)p_edge + 1] = b[1]; v.bl[3 * (size_t)p_edge + 2] = b[2];
        double D[9] = {H[0], H[1], H[2], H[1], H[3], H[4], H[2], H[4], H[5]}, Di[9];
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        inv3x3(D, Di);
#pragma unroll
        for (int q = 0; q < 9; q++) { v.Dinv[9 * (size_t)p_edge + q] = Di[q]; Dl[q][lc] = Di[q]; }
#pragma unroll
        for (int q = 0; q < 3; q++) Dl[9 + q][lc] = b[q];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    rec = fetch(c0 + C);
    // ---- the products, landmark by landmark (ba_schur_segment's, operands out of LDS)
    for (int l = 0; l < ncl; l++) {
      double Di[9];
#pragma unroll
      for (int q = 0; q < 9; q++) Di[q] = Dl[q][l];
      const double blv = Dl[9 + k3][l];
      double a[MT], bcol[MT];
#pragma unroll
      for (int t = 0; t < MT; t++) {
        const double* wp = lds + w_off[t] + l * k;
        const double x0 = wp[0], x1 = wp[64], x2 = wp[128];
        const double w0 = w_in[t] ? x0 : 0.0, w1 = w_in[t] ? x1 : 0.0, w2 = w_in[t] ? x2 : 0.0;
        const double wd = w0 * Di[k3] + w1 * Di[3 + k3] + w2 * Di[6 + k3];
        a[t] = kk < 3 ? wd : 0.0;
        const double wk = k3 == 0 ? w0 : (k3 == 1 ? w1 : w2);
        bcol[t] = kk < 3 ? (w_in[t] ? wk : (16 * t + i == rows ? blv : 0.0)) : 0.0;
      }
#pragma unroll
      for (int t = 0; t < MT; t++)
#pragma unroll
        for (int u = t; u < MT; u++) acc[t][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], bcol[u], acc[t][u], 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // flush: as ba_schur_segment
  const int tile0 = v.seg_tile[seg], slot0 = v.seg_slot[seg];
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int u = t; u < MT; u++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int Rr = 16 * t + (lane >> 4) + 4 * g, Cc = 16 * u + (lane & 15);
        if (Rr >= rows) continue;
        const int sa = Rr / 6, r = Rr - 6 * sa;
        const double val = acc[t][u][g];
        if (Cc < rows) {
          const int sb = Cc / 6, c = Cc - 6 * sb;
          if (sa < sb || (sa == sb && c >= r)) v.part_tiles[36 * (size_t)(tile0 + sa * k - sa * (sa - 1) / 2 + (sb - sa)) + 6 * r + c] = val;
        } else if (Cc == rows) {
          v.part_coef[6 * (size_t)(slot0 + sa) + r] = val;
        }
      }
}
template <int MT>
__global__ __launch_bounds__(256) void ba_lin_schur_kernel(BaView v, const double* __restrict__ lamp, int seg_begin, int seg_end, double lam_val) {
  __shared__ double lds[4][LS_WAVE_DOUBLES];
  const double lambda = lamp ? lamp[0] : lam_val;      // (by value: the trial's prologue runs beside this kernel, BaSidePrologue)
  const int seg = __builtin_amdgcn_readfirstlane(seg_begin + blockIdx.x * 4 + (threadIdx.x >> 6));
  if (seg >= seg_end) return;
  ba_lin_schur_segment<MT>(v, lambda, seg, v.seg_k[seg], lds[threadIdx.x >> 6]);
}

// Long tracks (BA_FUSED_KMAX < k <= BA_LONG_KMAX cameras: the tail of a real map, landmarks seen from dozens of key frames): the same
// segment, the same partial blocks and vectors for the destination schedule, formed with plain multiply-adds -- a wavefront per
// segment, its landmarks one after the other (such landmarks rarely share their camera set: usually one), W and W D^-1 of the landmark
// in LDS, the k (k + 1) / 2 blocks spread over the lanes.  Without it ONE such landmark sent the whole problem to the pair-major path.
__global__ __launch_bounds__(256) void ba_schur_long_kernel(BaView v, const double* __restrict__ lamp, int seg_begin, int seg_end) {
  __shared__ double Wl[4][6 * BA_LONG_KMAX * 3], WDl[4][6 * BA_LONG_KMAX * 3];
  const double lambda = lamp[0];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int seg = __builtin_amdgcn_readfirstlane(seg_begin + blockIdx.x * 4 + wv);
  if (seg >= seg_end) return;
  const int k = v.seg_k[seg], rows = 6 * k, q0 = v.seg_ptr[seg], q1 = v.seg_ptr[seg + 1], tile0 = v.seg_tile[seg], slot0 = v.seg_slot[seg];
  const int npair = k * (k + 1) / 2;
  double* W = Wl[wv];
  double* WD = WDl[wv];
  for (int q = q0; q < q1; q++) {
    const int p = v.run_lm[q], e0 = v.pt_ptr[p];
    double D[9], Di[9];
#pragma unroll
    for (int e = 0; e < 9; e++) D[e] = v.Hll[9 * (size_t)p + e];
    D[0] += lambda; D[4] += lambda; D[8] += lambda;
    inv3x3(D, Di);
    if (lane < 9) v.Dinv[9 * (size_t)p + lane] = Di[lane];
    const double bl0 = v.bl[3 * (size_t)p], bl1 = v.bl[3 * (size_t)p + 1], bl2 = v.bl[3 * (size_t)p + 2];
    const double* Wp = v.W + 18 * (size_t)e0;            // the landmark's k edges, camera slots ascending: row 6 a + r at Wp + 3 (6 a + r)
    __builtin_amdgcn_wave_barrier();
    for (int row = lane; row < rows; row += 64) {
      const double w0 = Wp[3 * row], w1 = Wp[3 * row + 1], w2 = Wp[3 * row + 2];
      W[3 * row] = w0; W[3 * row + 1] = w1; W[3 * row + 2] = w2;
#pragma unroll
      for (int m = 0; m < 3; m++) WD[3 * row + m] = w0 * Di[m] + w1 * Di[3 + m] + w2 * Di[6 + m];       // (W D^-1)(row, m), as ba_wd_kernel
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool first = q == q0;
    // an item = one row of one block: (pair of slots, r) -> the six entries (r, 0 .. 5), 48 contiguous bytes of the tile
    for (int item = lane; item < 6 * npair; item += 64) {
      const int pi = item / 6, r = item - 6 * pi;
      // pair index -> (sa, sb), sa <= sb: pi = sa k - sa (sa - 1) / 2 + (sb - sa)
      const float tk = (float)(2 * k + 1);
      int sa = (int)((tk - __builtin_sqrtf(tk * tk - 8.0f * (float)pi)) * 0.5f);
      sa = max(0, min(k - 1, sa));
      while (sa > 0 && sa * k - sa * (sa - 1) / 2 > pi) sa--;
      while ((sa + 1) * k - (sa + 1) * sa / 2 <= pi) sa++;
      const int sb = sa + (pi - (sa * k - sa * (sa - 1) / 2));
      const double* x = WD + 3 * (6 * sa + r);
      const double x0 = x[0], x1 = x[1], x2 = x[2];
      const double* y = W + 18 * sb;
      double* dst = v.part_tiles + 36 * (size_t)(tile0 + pi) + 6 * r;
      double val[6];
#pragma unroll
      for (int c = 0; c < 6; c++) val[c] = fma(x2, y[3 * c + 2], fma(x1, y[3 * c + 1], x0 * y[3 * c]));
      const int c_lo = sa == sb ? r : 0;                 // (diagonal blocks: the upper triangle is what the reduced system stores)
      if (first) {                                       // (wave-uniform: the segment's first landmark stores, the others add -- no load before a plain store)
#pragma unroll
        for (int c = 0; c < 6; c++) if (c >= c_lo) dst[c] = val[c];
      } else {
#pragma unroll
        for (int c = 0; c < 6; c++) if (c >= c_lo) dst[c] += val[c];
      }
    }
    for (int row = lane; row < rows; row += 64) {
      const double val = fma(WD[3 * row + 2], bl2, fma(WD[3 * row + 1], bl1, WD[3 * row] * bl0));
      double* dst = v.part_coef + 6 * (size_t)slot0 + row;
      if (first) *dst = val; else *dst += val;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

// destination schedule, blocks: one wavefront per block of the reduced system sums the partial blocks written for it, in the
// order of the segments (fixed: the result does not depend on scheduling), and subtracts the sum (block_solver.hpp:409-431)
__global__ __launch_bounds__(256) void ba_schur_gather_kernel(BaView v) {
  const int pair = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
  if (pair >= v.n_gpairs) return;
  // (round 5: the partial blocks' indices are fetched 64 at a time, one per lane, and the blocks eight at a time -- all requests of a
  // batch in flight together; the additions stay in the schedule's order.  One index load + one block load per term, each waiting for
  // the one before, made this kernel 55 us of pure latency at C4.)
  const int q0 = v.gpair_ptr[pair], nq = v.gpair_ptr[pair + 1] - q0;
  const int l36 = lane < 36 ? lane : 0;
  double s = 0;
  for (int base = 0; base < nq; base += 64) {
    const int cnt = min(64, nq - base);
    const int my_t = v.gtile[q0 + base + (lane < cnt ? lane : 0)];
    for (int j = 0; j < cnt; j += 8) {
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int tq = __builtin_amdgcn_readlane(my_t, min(j + u, cnt - 1));
        x[u] = v.part_tiles[36 * (size_t)tq + l36];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) if (j + u < cnt) s += x[u];
    }
  }
  if (lane >= 36) return;
  const int r = lane / 6, c = lane % 6;
  const int i1 = v.gpair_i1[pair], i2 = v.gpair_i2[pair];
  if (i1 != i2 || c >= r) *ba_S_at(v, i2 + c, i1 + r) -= s;
}

// destination schedule, right-hand side: S_cc = A_cc + lambda I and b_schur,c = b_c - sum of the camera's partial W D^-1 b_l
__device__ __forceinline__ void ba_cam_rhs_fused_body(const BaView& v, const double* __restrict__ lamp, int c, int t) {
  const double lambda = lamp[0];
  const int col = v.cam_col[c];
  if (col < 0) return;
  {
    const int q0 = v.gcam_ptr[c], nq = v.gcam_ptr[c + 1] - q0;
    const int t6 = t < 6 ? t : 0;
    double s = 0;
    for (int base = 0; base < nq; base += 64) {       // (as ba_schur_gather_kernel: indices 64 at a time, vectors eight at a time, sums in order)
      const int cnt = min(64, nq - base);
      const int my_s = v.gslot[q0 + base + (t < cnt ? t : 0)];
      for (int j = 0; j < cnt; j += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = v.part_coef[6 * (size_t)__builtin_amdgcn_readlane(my_s, min(j + u, cnt - 1)) + t6];
#pragma unroll
        for (int u = 0; u < 8; u++) if (j + u < cnt) s += x[u];
      }
    }
    if (t < 6) v.rhs[col + t] = v.bcam[6 * c + t] - s;
  }
  if (t < 36) {
    const int i = t / 6, j = t % 6;
    if (i >= j) *ba_S_at(v, col + i, col + j) = v.Hcam[36 * c + t] + ((i == j && col >= v.lam_lo && col < v.lam_hi) ? lambda : 0.0);
  }
}
__global__ __launch_bounds__(64) void ba_cam_rhs_fused_kernel(BaView v, const double* __restrict__ lamp) { ba_cam_rhs_fused_body(v, lamp, blockIdx.x, threadIdx.x); }
// ... and the off-diagonal pose blocks in the same launch (round 6: the cameras' diagonal blocks / right-hand sides and the pose edges' off-diagonal
// blocks touch disjoint entries of S; two 5-7 us launches in a row on every trial's chain were one launch boundary too many)
__global__ __launch_bounds__(64) void ba_cam_rhs_offdiag_kernel(BaView v, const double* __restrict__ lamp) {
  if ((int)blockIdx.x < v.nc) ba_cam_rhs_fused_body(v, lamp, blockIdx.x, threadIdx.x);
  else ba_offdiag_body(v, (int)blockIdx.x - v.nc, threadIdx.x);
}

// ---- elimination of the cuboids (BaView::elim) ------------------------------------------------------------------------------
// A free cuboid is coupled to the cameras that observe it and to nothing else, so it leaves the reduced system exactly like a landmark
// (block_solver.hpp:385-431 with a 9 x 9 block): D_oo^-1 = (H_oo + lambda I)^-1, S(c_a, c_b) -= M_a D_oo^-1 M_b^T for every pair of
// observing cameras, b_schur(c_a) -= M_a D_oo^-1 b_o, with M_a = the 6 x 9 camera-cuboid block summed over the edges of slot a (a
// camera may hold an EdgeSE3Cuboid and an EdgeSE3CuboidProj to the same cuboid).  One wavefront per cuboid; the blocks and vectors go
// to the same partial arrays as the landmark segments' and are summed by the same destination schedule.
__global__ __launch_bounds__(256) void ba_cub_elim_kernel(BaView v, const double* __restrict__ lamp) {
  const double lambda = lamp[0];
  // M, G: 54 doubles per slot of the graph's widest cuboid (dynamic: sized for BA_ELIM_MAX_SLOTS = 64 slots they took 55 KB of a CU's LDS from the
  // landmark segments' kernel that runs beside this one; a C4 cuboid has ~20 observing cameras)
  extern __shared__ double elim_lds[];
  double (*M)[54] = reinterpret_cast<double (*)[54]>(elim_lds);
  double (*G)[54] = reinterpret_cast<double (*)[54]>(elim_lds + 54 * (size_t)v.elim_max_slots);
  __shared__ double A[9][9], Di[9][9], bo[9];
  const int o = blockIdx.x, t = threadIdx.x;
  if (v.cub_col[o] < 0 || !v.cub_mine[o]) return;      // another rank's cuboid: its partial blocks, M and D^-1 stay zero here
  const int s0 = v.cubS_ptr[o], ns = v.cubS_ptr[o + 1] - s0;
  // M_s: one thread per (slot, element), the slot's edges in edge order
  for (int e = t; e < ns * 54; e += 256) {
    const int sl = e / 54, el = e - 54 * sl;
    double sv = 0;
    for (int q = v.slotE_ptr[s0 + sl]; q < v.slotE_ptr[s0 + sl + 1]; q++) sv += v.ce_Hco[54 * (size_t)v.slotE_idx[q] + el];
    M[sl][el] = sv;
    v.cub_M[54 * (size_t)(s0 + sl) + el] = sv;
  }
  for (int e = t; e < 81; e += 256) A[e / 9][e % 9] = v.Hcub[81 * (size_t)o + e] + ((e / 9 == e % 9) ? lambda : 0.0);
  __syncthreads();
  // Cholesky A = L L^T, then Di = L^-T L^-1 (9 x 9).  Round 6: on the lanes of the first wave with L and X in LDS -- as one lane with L and X
  // as local arrays (dynamic indices: 1.3 KB of scratch per lane) it was ~1 500 dependent scratch accesses, most of the kernel's 80-95 us.
  // Every entry is formed by the same operations in the same order as before (a column of L by the lanes of its rows, a column of L^-1 by
  // one lane, an entry of Di by one lane): the same bits.
  __shared__ double Lm[9][9], Xm[9][9];
  if (t < 64) {
    bool fail = false;
    for (int j = 0; j < 9; j++) {
      double d = A[j][j];
      for (int k = 0; k < j; k++) d -= Lm[j][k] * Lm[j][k];
      if (!(d > 0.0)) { fail = true; d = 1.0; }
      const double r = sqrt(d);
      if (t == j) Lm[j][j] = r;
      if (t > j && t < 9) {
        double sv = A[t][j];
        for (int k = 0; k < j; k++) sv -= Lm[t][k] * Lm[j][k];
        Lm[t][j] = sv / r;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (t < 9) {         // X = L^-1 (lower triangular): lane c forms column c
      const int c = t;
      for (int i = 0; i < 9; i++) {
        double sv = (i == c) ? 1.0 : 0.0;
        for (int k = c; k < i; k++) sv -= Lm[i][k] * Xm[k][c];
        Xm[i][c] = (i >= c) ? sv / Lm[i][i] : 0.0;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int e = t; e < 81; e += 64) {
      const int i = e / 9, j = e - 9 * i;
      double sv = 0;
      for (int k = (i > j ? i : j); k < 9; k++) sv += Xm[k][i] * Xm[k][j];
      Di[i][j] = sv;
    }
    if (fail && t == 0) atomicExch(v.elim_fail, 1);
  }
  __syncthreads();
  for (int e = t; e < 81; e += 256) v.cub_Dinv[81 * (size_t)o + e] = Di[e / 9][e % 9];
  if (t < 9) { double sv = 0; for (int k = 0; k < 9; k++) sv += Di[t][k] * v.bcub[9 * (size_t)o + k]; bo[t] = sv; }
  __syncthreads();
  // G_s = M_s D^-1, the slot's share of the right-hand side
  for (int e = t; e < ns * 54; e += 256) {
    const int sl = e / 54, r = (e % 54) / 9, c = e % 9;
    double sv = 0;
    for (int k = 0; k < 9; k++) sv += M[sl][9 * r + k] * Di[k][c];
    G[sl][9 * r + c] = sv;
  }
  for (int e = t; e < ns * 6; e += 256) {
    const int sl = e / 6, r = e % 6;
    double sv = 0;
    for (int k = 0; k < 9; k++) sv += M[sl][9 * r + k] * bo[k];
    v.part_coef[6 * (size_t)(v.cub_coef[o] + sl) + r] = sv;
  }
  __syncthreads();
  // the slot pairs a <= b (slots are sorted by column): block(r, c) = sum_m G_a(r, m) M_b(c, m)
  const int npair = ns * (ns + 1) / 2;
  for (int e = t; e < npair * 36; e += 256) {
    const int pr = e / 36, rc = e % 36, r = rc / 6, c = rc % 6;
    // pair index -> (a, b), a <= b: pr = a ns - a (a - 1) / 2 + (b - a)  (closed form + correction, as ba_schur_long_kernel; the walk over
    // the rows it replaces was up to ns steps for every one of the block's 36 entries)
    const float tk = (float)(2 * ns + 1);
    int a = (int)((tk - __builtin_sqrtf(tk * tk - 8.0f * (float)pr)) * 0.5f);
    a = max(0, min(ns - 1, a));
    while (a > 0 && a * ns - a * (a - 1) / 2 > pr) a--;
    while ((a + 1) * ns - (a + 1) * a / 2 <= pr) a++;
    const int b = a + (pr - (a * ns - a * (a - 1) / 2));
    double sv = 0;
    for (int m = 0; m < 9; m++) sv += G[a][9 * r + m] * M[b][9 * c + m];
    v.part_tiles[36 * (size_t)(v.cub_tile[o] + pr) + rc] = sv;
  }
}

// x_o = D_oo^-1 (b_o - sum_s M_s^T x_cam(s))   (block_solver.hpp:457-482 for the cuboid blocks)
__device__ __forceinline__ void cub_backsub_wave(const BaView& v, int o, int t, double* cl) {     // one wavefront; cl: 9 doubles of LDS of its own
  const int col = v.cub_col[o];
  if (col < 0) return;
  if (t < 9) {
    double acc = v.bcub[9 * (size_t)o + t];
    for (int sl = v.cubS_ptr[o]; sl < v.cubS_ptr[o + 1]; sl++) {
      const double* Ms = v.cub_M + 54 * (size_t)sl;
      const double* xp = v.rhs + v.cam_col[v.cubS_cam[sl]];
      double sv = 0;
      for (int r = 0; r < 6; r++) sv += Ms[9 * r + t] * (-xp[r]);
      acc += sv;
    }
    cl[t] = acc;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (t < 9) {
    double sv = 0;
    for (int k = 0; k < 9; k++) sv += v.cub_Dinv[81 * (size_t)o + 9 * t + k] * cl[k];
    v.rhs[col + t] = sv;
  }
}
__global__ __launch_bounds__(64) void ba_cub_backsub_kernel(BaView v) {
  __shared__ double cl[9];
  cub_backsub_wave(v, blockIdx.x, threadIdx.x, cl);
}

// (Round 5 tried the H_pl records as one coalesced stream through LDS, a lane per edge, the landmark's lane summing in edge order: 84 us
// against 80 at C4 -- the 163 MB read is not what this kernel waits for -- and was dropped.)
__device__ __forceinline__ void backsub_point(const BaView& v, int p) {
  if (p >= v.np) return;
  double cl[3] = {v.bl[3 * p], v.bl[3 * p + 1], v.bl[3 * p + 2]};
  for (int k = v.pt_ptr[p]; k < v.pt_ptr[p + 1]; k++) {
    int col = v.cam_col[v.pm_cam[k]];
    if (col < 0) continue;
    const double* Wk = v.W + 18 * (size_t)k;
    const double* xp = v.rhs + col;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double s = 0;
#pragma unroll
      for (int r = 0; r < 6; r++) s += Wk[3 * r + c] * (-xp[r]);
      cl[c] += s;
    }
  }
  double x[3];
  mat3_vec(v.Dinv + 9 * (size_t)p, cl, x);
  v.xl[3 * p] = x[0]; v.xl[3 * p + 1] = x[1]; v.xl[3 * p + 2] = x[2];
}

// The same with H_pl formed on the spot (the fused linearise-in-Schur trials, which never write it): per edge the camera, the point, the
// measurement -- proj_linearize and the J_c^T (rho' Omega) J_p product exactly as ba_lin_schur_segment / lin_pt_edge form them, then the same
// sums in the same order as the kernel above: the same x_l bit for bit (tests/test_ba_gpu.py holds the LM run to the classic pair).
__device__ __forceinline__ void backsub_lin_point(const BaView& v, int p) {
  if (p >= v.np) return;
  double cl[3] = {v.bl[3 * p], v.bl[3 * p + 1], v.bl[3 * p + 2]};
  const bool free_pt = v.pt_free[p] != 0;
  const double X[3] = {v.points[3 * (size_t)p], v.points[3 * (size_t)p + 1], v.points[3 * (size_t)p + 2]};
  for (int k = v.pt_ptr[p]; k < v.pt_ptr[p + 1]; k++) {
    const int c = v.pm_cam[k];
    int col = v.cam_col[c];
    if (col < 0) continue;
    Pose T = pose_load(v.cams + 7 * c);
    double R[9];
    pose_rotmat(T, R);
    ProjLin L;
    proj_linearize(T, R, X, v.pm_uv + 2 * (size_t)k, v.info_u ? v.info_u : v.pm_info + 4 * (size_t)k, v.intr_u ? v.intr_u : v.pm_intr + 4 * (size_t)k, v.pm_huber[k], v.pm_rk, k, L);
    double Wk[18];
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const double jw0 = L.Jc[q] * L.Wm[0] + L.Jc[6 + q] * L.Wm[2];
      const double jw1 = L.Jc[q] * L.Wm[1] + L.Jc[6 + q] * L.Wm[3];
#pragma unroll
      for (int j = 0; j < 3; j++) Wk[3 * q + j] = free_pt ? (jw0 * L.Jp[j] + jw1 * L.Jp[3 + j]) : 0.0;      // (the camera is free here)
    }
    const double* xp = v.rhs + col;
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
      double s = 0;
#pragma unroll
      for (int r = 0; r < 6; r++) s += Wk[3 * r + cc] * (-xp[r]);
      cl[cc] += s;
    }
  }
  double x[3];
  mat3_vec(v.Dinv + 9 * (size_t)p, cl, x);
  v.xl[3 * p] = x[0]; v.xl[3 * p + 1] = x[1]; v.xl[3 * p + 2] = x[2];
}

// Both back-substitutions of a trial in ONE launch: the eliminated cuboids' blocks first (a wave per cuboid: x_o from the cameras' x_p), the
// landmarks' behind them -- the two read the cameras' increments only and write disjoint outputs.  LIN: H_pl formed on the spot (above).
template <bool LIN>
__global__ __launch_bounds__(256) void ba_backsub_all_kernel(BaView v, int n_cub_blocks) {
  __shared__ double cl[4][9];
  if ((int)blockIdx.x < n_cub_blocks) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o < v.no) cub_backsub_wave(v, o, threadIdx.x & 63, cl[threadIdx.x >> 6]);
    return;
  }
  const int p = (blockIdx.x - n_cub_blocks) * 256 + threadIdx.x;
  if (LIN) backsub_lin_point(v, p); else backsub_point(v, p);
}

// bak_*: non-null = the push of the LM trial rides along (OptimizableGraph::push before the update, optimization_algorithm_levenberg.cpp:104-
// 113): every vertex's estimate goes to its backup slot before it is changed -- three device-to-device copies of the whole state per trial
// (~50 us at C4) become a few more stores of a kernel that reads the state anyway
__device__ __forceinline__ void update_vertex(const BaView& v, int i, double* bak_cams, double* bak_points, double* bak_cubes) {
  if (i < v.np) {
    if (bak_points) { bak_points[3 * i] = v.points[3 * i]; bak_points[3 * i + 1] = v.points[3 * i + 1]; bak_points[3 * i + 2] = v.points[3 * i + 2]; }
    if (v.pt_free[i]) { v.points[3 * i] += v.xl[3 * i]; v.points[3 * i + 1] += v.xl[3 * i + 1]; v.points[3 * i + 2] += v.xl[3 * i + 2]; }
    return;
  }
  i -= v.np;
  if (i < v.nc) {
    if (bak_cams) for (int q = 0; q < 7; q++) bak_cams[7 * i + q] = v.cams[7 * i + q];
    int col = v.cam_col[i];
    if (col >= 0) pose_store(cam_oplus(pose_load(v.cams + 7 * i), v.rhs + col), v.cams + 7 * i);
    return;
  }
  i -= v.nc;
  if (i < v.no) {
    if (bak_cubes) for (int q = 0; q < 10; q++) bak_cubes[10 * i + q] = v.cubes[10 * i + q];
    int col = v.cub_col[i];
    if (col >= 0) cube_store(cube_oplus(cube_load(v.cubes + 10 * i), v.rhs + col), v.cubes + 10 * i);
  }
}
__global__ __launch_bounds__(256) void ba_update_kernel(BaView v, double* bak_cams, double* bak_points, double* bak_cubes) {
  update_vertex(v, blockIdx.x * 256 + threadIdx.x, bak_cams, bak_points, bak_cubes);
}
// the LM scale term and the update of a trial in ONE launch: x^T (lambda x + b) reads the increments and the right-hand sides, the update reads
// the increments and writes the estimates -- independent; the SCALE_BLOCKS partial sums keep their places
__global__ __launch_bounds__(256) void ba_scale_update_kernel(BaView v, const double* __restrict__ lamp, double* partial, double* bak_cams, double* bak_points, double* bak_cubes) {
  __shared__ double ws[4];
  if (blockIdx.x < SCALE_BLOCKS) { scale_block(v, lamp, partial, blockIdx.x, ws); return; }
  update_vertex(v, (blockIdx.x - SCALE_BLOCKS) * 256 + threadIdx.x, bak_cams, bak_points, bak_cubes);
}


// ------------------------------------------------------------------ banded Cholesky of the reduced system --
// The pose graph of a trajectory is banded once its vertices are ordered by reverse Cuthill-McKee: cameras are
// coupled to the few neighbours they share landmarks with and to the cuboids they observe.  S is stored as a lower
// band (LD = bandwidth + 1 doubles per column): n * bw^2 flops instead of n^3 / 3 (0.35 Gflop instead of 385 Gflop at
// C4, n = 10494, bw = 182).  So little arithmetic that the factorisation is a chain of dependent 32-column steps and
// its cost is their latency; it therefore runs as ONE persistent (co-resident) kernel, left-looking over column blocks:
//   band_step               a worker workgroup's step: gather the block row's history (64-row x bw strip of L: the block's 32
//                           rows + its own 16 panel rows) through LDS, multiply its rows against the block rows, scale by
//                           the inverse of the diagonal block's factor, meet the team at a grid barrier.  The right-hand
//                           side rides along as one more row below the band, so L y = b costs no extra step.
//   band_diag_phase         one extra workgroup per front factorises the diagonal blocks (POTF2 + inverse in the registers
//                           of one wave) one step ahead of the workers and publishes the inverses.
//   band_chol_coop_kernel   two fronts (forward from the top, reverse from the bottom) + the middle block.
//   band_chol_nested_kernel four fronts: a separator block splits the band in two halves, each eliminated at both ends;
//                           the separator's rows ride along, its Schur complement is accumulated beside the fronts.
//   band_backsolve_kernel   L^T x = y per front with the inverted diagonal blocks (two small mat-vecs per step, every load
//                           that does not depend on x prefetched one step ahead); band_sep_* handle the separator.

// Hand-offs between workgroups (other CUs, mostly other XCDs -- whose L2s are not coherent with each other): every shared double is written
// with a write-through (sc1) store and read with an sc1 load that bypasses the reader's L1 -- 8-byte agent-scope relaxed atomics on both
// sides, the second of the valid forms of MI355X_MICROARCH.md -- so a producer only drains its stores (s_waitcnt vmcnt(0)) before it
// raises a counter or a flag, and a consumer only polls: no buffer_wbl2 (a write-back of the XCD's whole L2, >= 1.7 us) per release and
// no buffer_inv per acquire, which is what most of a step's chain used to consist of.  BAND_WT 0 restores plain accesses + agent fences.
#ifndef BAND_WT
#define BAND_WT 1
#endif
#ifndef BAND_WT_LOADS_
#define BAND_WT_LOADS_ 0     // 1: sc1 loads instead of acquire fence + plain loads (measured slower: the strip gather re-reads every entry many times and sc1 loads never hit)
#endif
__device__ __forceinline__ void band_release() {
#if !BAND_WT
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
}
__device__ __forceinline__ void band_acquire() {
#if !(BAND_WT && BAND_WT_LOADS_)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}


// Bounded wait on a monotone counter.  The persistent kernels below need their whole team resident; the host checks that
// against the occupancy query before choosing the banded path (ba_band_fits_device), but a GPU shared with another process's
// persistent kernel can still starve a team.  Such a launch must fail, not hang: after ~1 s of polling the waiter raises the
// abort word of the launch's info block (all barrier counters and flags live in that one 96-byte, 256-byte-aligned block:
// info[19]), marks the factorisation as failed (info[0] = INT_MAX) and every later wait of the launch returns at once.
enum { BAND_ABORT_SLOT = 19, BAND_SPIN_LIMIT = 1 << 21, BAND_TIMEOUT_INFO = 0x7fffffff };
__device__ __forceinline__ void band_wait_ge(unsigned* ctr, unsigned target) {
  unsigned spins = 0;
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023u) == 0) {
      unsigned* base = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned long long>(ctr) & ~127ull);
      if (__hip_atomic_load(base + BAND_ABORT_SLOT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      if (spins >= (unsigned)BAND_SPIN_LIMIT) {
        __hip_atomic_store(base + BAND_ABORT_SLOT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(base, (unsigned)BAND_TIMEOUT_INFO, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
}

// All workgroups of the (co-resident) grid arrive; thread 0 spins on the monotone counter.  Producer side: every
// wave drains its stores, lane 0 writes the XCD's L2 back (agent-scope release; the explicit wait restates the one the
// compiler may drop after buffer_wbl2) and arrives; consumer side: relaxed poll, one agent-scope acquire for the CU.
__device__ __forceinline__ void band_grid_sync(unsigned* bar, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    band_release();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    band_wait_ge(bar, target);
    band_acquire();
  }
  __syncthreads();
}

// Out-of-line, one instance per kernel: a device function with a single calling kernel keeps no callee-saved registers
// (the compiler specialises its convention); shared between two kernels it would save and restore ~45 of them per call.
__device__ __attribute__((noinline)) bool band_potf2_inv_k0(const double (*U)[BS + 1], int nb, double (*Dl)[BS + 1], double (*X)[BS + 1], double* colbuf) { return band_potf2_inv4b_impl(U, nb, Dl, X, colbuf); }
__device__ __attribute__((noinline)) bool band_potf2_inv_k1(const double (*U)[BS + 1], int nb, double (*Dl)[BS + 1], double (*X)[BS + 1], double* colbuf) { return band_potf2_inv4b_impl(U, nb, Dl, X, colbuf); }

typedef const __attribute__((address_space(1))) double* band_gptr;   // global address space: a noinline function would otherwise emit flat loads
__device__ __forceinline__ double band_gload(const double* p) {
#if BAND_WT && BAND_WT_LOADS_
  return __longlong_as_double((long long)__hip_atomic_load((const __attribute__((address_space(1))) unsigned long long*)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#else
  return *(band_gptr)(p);
#endif
}
__device__ __forceinline__ void band_gstore(const double* p, double v) {
#if BAND_WT
  __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)(const_cast<double*>(p)), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *const_cast<double*>(p) = v;
#endif
}
// plain store: ONLY in kernels whose workgroups hand nothing to each other within a launch (the substitution kernels: a front is
// walked by one workgroup, the kernel boundary publishes the result)
__device__ __forceinline__ void band_pstore(const double* p, double v) { *const_cast<double*>(p) = v; }

// panel rows per workgroup, per kernel.  16 in both: 8 rows in the nested kernel (twice the workgroups, half the product and
// panel work) was measured slower, 1.95 -> 2.15 ms at C4 -- what the halves save, the doubled teams lose in the barrier (2.5 ->
// 4.6 us), the gather (more workgroups behind the same L2) and the wait for the diagonal workgroup
enum { BAND_RW = 16, BAND_RW_NESTED = 16, BAND_NR = BS + BAND_RW + 8, BAND_NRP = 64, BAND_DC = 192, BAND_NV = BAND_DC * BAND_NRP / 256 };
__host__ __device__ constexpr int band_rw(int kid) { return kid ? BAND_RW_NESTED : BAND_RW; }

// An elimination front addresses the band through its own index space (v = 0 is where it starts): the forward front
// walks the matrix top-left to bottom-right (v = original index), the reverse front bottom-right to top-left
// (v = n - 1 - original index).  In both spaces the factor is lower triangular and banded; only the strides differ:
//   &L(i, j) = base + i * si + j * sj   (i >= j, i - j <= bw),      &rhs(v) = rb + v * sr
//   forward: base = Sb, si = 1, sj = bw              reverse: base = Sb + (n - 1) LD, si = -bw, sj = -1
// (Every global pointer of the persistent kernels is const-qualified: the ONLY ways to write through one are band_gstore -- the
// write-through store the hand-offs rely on -- and band_pstore, a plain store for kernels whose workgroups exchange nothing within a
// launch.  A plain `p[i] = v` in a step routine does not compile, so a future edit cannot slip a store past the hand-off protocol.)
struct BandView {
  const double* base; long long si, sj;
  const double* rb; long long sr;
};
struct BandSeg {            // one run of history columns [jlo, jhi) of a view; flip: the step's rows are re-indexed i -> n - 1 - i
  BandView v; int jlo, jhi, flip;
};
struct BandSegX : BandSeg { // ... of a front that carries separator rows (BandAug): their history of view column j >= lc_jmin
  int lc_t0, lc_jmin;       // is column lc_t0 + j of aug.lc
};
// Rows of a separator block that ride below a front (nested elimination, see band_chol_nested_kernel): row q of the
// separator against column j of the front's view.  A(q, j) is an entry of the band (zero outside it); the factor's
// entries fill in, so they live in their own array.  wc == 0: no such rows.
struct BandAug {
  const double* a_base; long long a_sq, a_sj;   // &A(q, j) = a_base + q a_sq + j a_sj, inside the band iff m0 + q + ms j <= bw
  int m0, ms;
  const double* lc; int wc, qflip, t0;          // &L(q, j) = lc + (t0 + j) wc + (qflip ? wc - 1 - q : q)   (t0: the view being eliminated)
};
struct BandLds { double* R; double (*U)[BS + 1]; double (*Dl)[BS + 1]; double (*X)[BS + 1]; double* colbuf; int* rowidx; };

// Strip gather + product of one step: U(rr, c) = A(rr, k0 + c) - sum_j L(rr, j) L(k0 + c, j) for the 32 block rows and
// this workgroup's rows, the history columns j coming from one or two segments (the second one only in the middle
// phase of the two-sided elimination: the other front's columns).  A thread gathers ONE row (rr = tid & 63) at every
// fourth column, so the addresses are a pointer walk, and all BAND_NV loads are unconditional (masked lanes read the
// zero word) and in flight together.
__device__ __forceinline__ int band_seg_jmin(const BandSeg&) { return 0; }
__device__ __forceinline__ int band_seg_jmin(const BandSegX& s) { return s.lc_jmin; }
__device__ __forceinline__ int band_seg_t0(const BandSeg&) { return 0; }
__device__ __forceinline__ int band_seg_t0(const BandSegX& s) { return s.lc_t0; }
template <bool AUG, class SEG, int RW>
__device__ __forceinline__ void band_gather_gemm_impl(const SEG& s0, const SEG& s1, int nseg, const BandAug& aug, const double* zero, int n, int bw, int k0, int nb, const int* rowidx,
                                                           double* R, double (*U)[BS + 1], long long* tp, long long* t_prev) {
  constexpr int NR = BS + RW + 8, NRP = BAND_NRP, NOWN = RW + 8, NRT = NOWN / 8, NV = BAND_NV, DC = BAND_DC;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c1 = tid & 31, rg1 = tid >> 5;      // one-column mapping of the finishing pass
#define BAND_TICK(k) do { if (tp) { long long t_now = wall_clock64(); tp[k] += t_now - *t_prev; *t_prev = t_now; } } while (0)
  // the block's own columns (A of U = A - sum), in the step's own view s0.v: loads in flight while the strip is gathered
  double aval[NRT];
#pragma unroll
  for (int m = 0; m < NRT; m++) {
    const int i = rowidx[BS + rg1 + 8 * m], dlt = i - (k0 + c1), qa = -16 - i;   // i <= -16: separator row qa
    const bool ok = c1 < nb && (i == -2 || (i >= 0 && dlt >= 0 && dlt <= bw) || (AUG && i <= -16 && aug.m0 + qa + aug.ms * (k0 + c1) <= bw));
    const double* ptr = (i == -2) ? s0.v.rb + (long long)(k0 + c1) * s0.v.sr
                      : (AUG && i <= -16) ? aug.a_base + (long long)qa * aug.a_sq + (long long)(k0 + c1) * aug.a_sj
                                          : s0.v.base + (long long)i * s0.v.si + (long long)(k0 + c1) * s0.v.sj;
    aval[m] = band_gload(ok ? ptr : zero);
  }
  static_assert(NOWN <= 32 && NRP >= BS + 32, "two 16-row MFMA tiles of own rows");
  ba_v4d macc[2][2];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int u = 0; u < 2; u++) macc[t][u] = ba_v4d{0.0, 0.0, 0.0, 0.0};
  const int row0 = lane < NR ? rowidx[lane] : -1;            // the row this thread gathers (in the step's view)
  for (int sg = 0; sg < nseg; sg++) {
    const SEG S = sg ? s1 : s0;
    const int my_i = (S.flip && row0 >= 0) ? n - 1 - row0 : row0;
    const bool sep = AUG && my_i <= -16;                           // a separator row: history in aug.lc
    const int qc = aug.qflip ? aug.wc - 1 - (-16 - my_i) : (-16 - my_i);
    const int jmin = my_i >= 0 ? max(S.jlo, my_i - bw) : (sep ? max(S.jlo, band_seg_jmin(S)) : S.jlo);   // in-band columns of that row: j >= i - bw
    double vals[NV];
    auto fetch = [&](int j0) {
      const int jend = (my_i == -1) ? 0 : min(S.jhi, j0 + DC);
      int j = j0 + wv;
      const double* p = (my_i == -2) ? S.v.rb + (long long)j * S.v.sr
                      : sep ? aug.lc + (long long)(band_seg_t0(S) + j) * aug.wc + qc
                            : S.v.base + (long long)(my_i >= 0 ? my_i : 0) * S.v.si + (long long)j * S.v.sj;
      const long long step = (my_i == -2) ? 4 * S.v.sr : (sep ? 4LL * aug.wc : 4 * S.v.sj);
#pragma unroll
      for (int u = 0; u < NV; u++) {
        const bool ok = j >= jmin && j < jend;
        vals[u] = band_gload(ok ? p : zero);
        p += step; j += 4;
      }
    };
    if (S.jlo < S.jhi) fetch(S.jlo);
    BAND_TICK(6);
    for (int j0 = S.jlo; j0 < S.jhi; j0 += DC) {
      const int jn = min(DC, S.jhi - j0);
#pragma unroll
      for (int u = 0; u < NV; u++) R[(wv + 4 * u) * NRP + lane] = vals[u];
      __syncthreads();
      BAND_TICK(7);
      if (j0 + DC < S.jhi) fetch(j0 + DC);
      BAND_TICK(6);
      // this wave's quarter of the depth on the matrix cores: acc(own rows, block columns) += sum_d R(d, rows) R(d, cols) as
      // v_mfma_f64_16x16x4_f64 tiles (own rows padded to 32: 2 x 2 tiles), four depth indices per issue; of every 16 depth indices
      // wave w takes 4 w .. 4 w + 3.  Rows of R past the chunk's jn were stored as zeros by the gather, so the last step may
      // overrun jn.  Operand lane map: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15].
      {
        const int li = lane & 15, lk = lane >> 4;
        for (int d0 = 4 * wv; d0 < jn; d0 += 16) {
          const double* rj = R + (d0 + lk) * NRP;
          const double a0 = rj[BS + li], a1 = rj[BS + 16 + li], b0 = rj[li], b1 = rj[16 + li];
          macc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, macc[0][0], 0, 0, 0);
          macc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, macc[0][1], 0, 0, 0);
          macc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, macc[1][0], 0, 0, 0);
          macc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, macc[1][1], 0, 0, 0);
        }
      }
      __syncthreads();
      BAND_TICK(8);
    }
  }
  // partial sums of the 4 waves -> LDS (over the strip buffer), then U = A - sum.  D[row = (lane >> 4) + 4 g][col = lane & 15].
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int g = 0; g < 4; g++) R[(wv * NRP + BS + 16 * t + (lane >> 4) + 4 * g) * BS + 16 * u + (lane & 15)] = macc[t][u][g];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < NRT; m++) {
    const int rr = BS + rg1 + 8 * m;
    const double sum = (R[(0 * NRP + rr) * BS + c1] + R[(1 * NRP + rr) * BS + c1]) + (R[(2 * NRP + rr) * BS + c1] + R[(3 * NRP + rr) * BS + c1]);
    U[rr][c1] = aval[m] - sum;
  }
  __syncthreads();
  BAND_TICK(1);
#undef BAND_TICK
}
// out-of-line instances: fronts without / with separator rows
__device__ __attribute__((noinline)) void band_gather_gemm_plain_k0(BandSeg s0, BandSeg s1, int nseg, const double* zero, int n, int bw, int k0, int nb, const int* rowidx,
                                                                    double* R, double (*U)[BS + 1], long long* tp, long long* t_prev) {
  const BandAug noaug{zero, 0, 0, 0, 0, nullptr, 0, 0, 0};
  band_gather_gemm_impl<false, BandSeg, band_rw(0)>(s0, s1, nseg, noaug, zero, n, bw, k0, nb, rowidx, R, U, tp, t_prev);
}
__device__ __attribute__((noinline)) void band_gather_gemm_plain_k1(BandSeg s0, BandSeg s1, int nseg, const double* zero, int n, int bw, int k0, int nb, const int* rowidx,
                                                                    double* R, double (*U)[BS + 1], long long* tp, long long* t_prev) {
  const BandAug noaug{zero, 0, 0, 0, 0, nullptr, 0, 0, 0};
  band_gather_gemm_impl<false, BandSeg, band_rw(1)>(s0, s1, nseg, noaug, zero, n, bw, k0, nb, rowidx, R, U, tp, t_prev);
}
__device__ __attribute__((noinline)) void band_gather_gemm_sep(BandSegX s0, BandSegX s1, int nseg, BandAug aug, const double* zero, int n, int bw, int k0, int nb, const int* rowidx,
                                                               double* R, double (*U)[BS + 1], long long* tp, long long* t_prev) {
  band_gather_gemm_impl<true, BandSegX, band_rw(1)>(s0, s1, nseg, aug, zero, n, bw, k0, nb, rowidx, R, U, tp, t_prev);
}

// arrive without waiting (a workgroup that leaves the kernel): the release half of band_grid_sync
__device__ __forceinline__ void band_grid_arrive(unsigned* bar) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    band_release();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// One elimination step of `view`: block column k0 (nb wide), rows up to i_end, history segments s0 (the view's own
// columns) and s1.  Workgroup roles: cw < 0: panel rows own0 = k0 + nb + RW w ... of the view (has_rhs: plus the
// right-hand side); cw >= 0: separator rows RW cw ... of aug.  Every workgroup factorises the diagonal block itself.
template <int KID>
__device__ __forceinline__ void band_gather_gemm(const BandSeg& s0, const BandSeg& s1, int nseg, const BandAug&, const double* zero, int n, int bw, int k0, int nb, const int* rowidx,
                                                 double* R, double (*U)[BS + 1], long long* tp, long long* t_prev) {
  if (KID == 0) band_gather_gemm_plain_k0(s0, s1, nseg, zero, n, bw, k0, nb, rowidx, R, U, tp, t_prev);
  else band_gather_gemm_plain_k1(s0, s1, nseg, zero, n, bw, k0, nb, rowidx, R, U, tp, t_prev);
}
template <int KID>
__device__ __forceinline__ void band_gather_gemm(const BandSegX& s0, const BandSegX& s1, int nseg, const BandAug& aug, const double* zero, int n, int bw, int k0, int nb, const int* rowidx,
                                                 double* R, double (*U)[BS + 1], long long* tp, long long* t_prev) {
  band_gather_gemm_sep(s0, s1, nseg, aug, zero, n, bw, k0, nb, rowidx, R, U, tp, t_prev);
}
template <bool AUG, int KID, class SEG>
__device__ __forceinline__ void band_step(const BandLds& M, const BandView& view, const double* Linv, int k0, int nb, int i_end, const SEG& s0, const SEG& s1, int nseg,
                                          const BandAug& aug, int w, int cw, bool has_rhs, int n, int bw, const double* zero, unsigned* dflag, unsigned dtarget, long long* tp, long long* t_prev) {
  constexpr int RW = band_rw(KID), NR = BS + RW + 8;
  const int tid = threadIdx.x;
  const int own0 = k0 + nb + w * RW;
  if (tid < NR) {                                   // row (in the view's index space) of gathered row rr; -1 = none, -2 = right-hand side, <= -16: separator row
    int i = -1;
    if (tid < BS) i = tid < nb ? k0 + tid : -1;
    else if (tid - BS < RW) {
      if (cw >= 0) { const int q = cw * RW + (tid - BS); i = q < aug.wc ? -16 - q : -1; }
      else { const int q = own0 + (tid - BS); i = q < i_end ? q : -1; }
    }
    else if (tid - BS == RW && has_rhs) i = -2;
    M.rowidx[tid] = i;
  }
  __syncthreads();
  band_gather_gemm<KID>(s0, s1, nseg, aug, zero, n, bw, k0, nb, M.rowidx, M.R, M.U, tp, t_prev);
  // the inverse of the block's factor comes from the front's diagonal workgroup (band_diag_phase), normally before it is asked for
  if (tid == 0) {
    band_wait_ge(dflag, dtarget);
    band_acquire();
  }
  __syncthreads();
  {
    const double* Li = Linv + (size_t)(k0 / BS) * BS * BS;
    double xv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) xv[u] = band_gload(Li + tid + 256 * u);
#pragma unroll
    for (int u = 0; u < 4; u++) { const int e = tid + 256 * u; M.X[e >> 5][e & 31] = xv[u]; }
  }
  __syncthreads();
  if (tp) { long long t_now = wall_clock64(); tp[2] += t_now - *t_prev; *t_prev = t_now; }
  // own rows (and the right-hand side): row <- row * L^-T
  for (int e = tid; e < (RW + 1) * BS; e += 256) {
    const int q = e >> 5, cc = e & 31, rr = BS + q, i = M.rowidx[rr];
    if (cc < nb && i != -1) {
      // X is stored with its upper triangle zeroed: the full-length dot product in four independent chains
      double sa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int t = 0; t < BS; t += 4)
#pragma unroll
        for (int u = 0; u < 4; u++) sa[u] = fma(M.U[rr][t + u], M.X[cc][t + u], sa[u]);
      const double sacc = (sa[0] + sa[1]) + (sa[2] + sa[3]);
      if (i >= 0) { if (i - (k0 + cc) <= bw) band_gstore(&view.base[(long long)i * view.si + (long long)(k0 + cc) * view.sj], sacc); }
      else if (i == -2) band_gstore(&view.rb[(long long)(k0 + cc) * view.sr], sacc);
      else if (AUG) { const int qa = -16 - i; band_gstore(&aug.lc[(long long)(aug.t0 + k0 + cc) * aug.wc + (aug.qflip ? aug.wc - 1 - qa : qa)], sacc); }
    }
  }
  if (tp) { long long t_now = wall_clock64(); tp[4] += t_now - *t_prev; *t_prev = t_now; }
}
// ---- the diagonal workgroup of a front.  One extra workgroup per front factorises the diagonal blocks, so the 32-step
// chain of POTF2 leaves the other workgroups' step: while they still gather and multiply their strips of step k, it
// already holds sum_j L(k rows, j) L(k rows, j)^T over all history but the newest 32 columns (accumulated during step
// k - 1), adds those after the team's barrier, factorises, and publishes L_kk^-1 (and L_kk) with a monotone flag.

// acc(4 rg + m, 4 cg + t) += this wave's quarter of sum over columns j in [jlo, jhi) of S.v of L(k0 + row, j) L(k0 + col, j);
// rg = lane >> 3, cg = lane & 7; the four waves' partial sums meet when U is formed
template <int DC, class SEG>
__device__ __forceinline__ void band_diag_accum(const SEG& S, int jlo, int jhi, int n, int bw, int k0, int nb, const double* zero, double* R, double (&acc)[4][4]) {
  constexpr int LDR = BS + 1;
  const int tid = threadIdx.x, gr = tid & 31, ph = tid >> 5, lane = tid & 63, wv = tid >> 6, rg = lane >> 3, cg = lane & 7;
  const int i = k0 + gr, my_i = S.flip ? n - 1 - i : i;
  const int jmin = max(jlo, my_i - bw);
  const double* rowp = S.v.base + (long long)my_i * S.v.si;
  for (int j0 = jlo; j0 < jhi; j0 += DC) {
    const int jend = gr < nb ? min(jhi, j0 + DC) : 0;
    double vals[DC / 8];
#pragma unroll
    for (int u = 0; u < DC / 8; u++) {
      const int j = j0 + ph + 8 * u;
      vals[u] = band_gload((j >= jmin && j < jend) ? rowp + (long long)j * S.v.sj : zero);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < DC / 8; u++) R[(ph + 8 * u) * LDR + gr] = vals[u];
    __syncthreads();
    const int jn = min(DC, jhi - j0);
#pragma unroll 4
    for (int jj = wv; jj < jn; jj += 4) {
      const double* rj = R + jj * LDR;
      double rowv[4], colv[4];
#pragma unroll
      for (int m = 0; m < 4; m++) { rowv[m] = rj[4 * rg + m]; colv[m] = rj[4 * cg + m]; }
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int t = 0; t < 4; t++) acc[m][t] = fma(rowv[m], colv[t], acc[m][t]);
    }
  }
}
// The diagonal blocks k_begin, k_begin + 32, ... < k_end of `view` (one phase of a front).  History of block k0: the
// view's own columns [max(0, k0 - bw), k0) and, with two_seg, the columns [max(0, (n - min(k_end, k0 + nb + bw)) - bw), jhi1)
// of the other front's view v1 (rows re-indexed i -> n - 1 - i).  start: counter/target that opens the phase (may be null);
// bar: the team's step barrier, which reads G ep0 when the phase starts; flag: published block count, f0 at the start.
template <int KID>
__device__ __forceinline__ void band_diag_phase(const BandLds& M, const BandView& view, const double* Linv, int k_begin, int k_end, bool two_seg, const BandView& v1, int jhi1,
                                                unsigned* start, unsigned start_target, unsigned* bar, unsigned G, unsigned ep0, unsigned* flag, unsigned f0,
                                                int n, int bw, const double* zero, int* info, long long* dprof = nullptr) {
  const int tid = threadIdx.x, r = tid >> 3, cq = tid & 7, lane = tid & 63, wv = tid >> 6, rg = lane >> 3, cg = lane & 7;
  long long dt_prev = dprof ? wall_clock64() : 0;      // optional phase clock of this workgroup (CS_BAND_PROF): wait, newest block + U, POTF2, publish, look-ahead
#define BAND_DTICK(k) do { if (dprof && tid == 0) { const long long t_now = wall_clock64(); dprof[k] += t_now - dt_prev; dt_prev = t_now; } } while (0)
  auto wait_for = [&](unsigned* ctr, unsigned target) {
    if (tid == 0) {
      band_wait_ge(ctr, target);
      band_acquire();
    }
    __syncthreads();
  };
  auto seg1 = [&](int k0, int nb) { return BandSeg{v1, max(0, (n - min(k_end, k0 + nb + bw)) - bw), jhi1, 1}; };
  double acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; m++)
#pragma unroll
    for (int t = 0; t < 4; t++) acc[m][t] = 0.0;
  unsigned s = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += BS, s++) {
    const int nb = min(BS, k_end - k0);
    const BandSeg s0{view, max(0, k0 - bw), k0, 0};
    double av[4];                                // the block's own entries (lower triangle)
    auto load_block = [&]() {
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int c = 4 * cq + t;
        const bool ok = r < nb && c <= r && r - c <= bw;
        av[t] = band_gload(ok ? view.base + (long long)(k0 + r) * view.si + (long long)(k0 + c) * view.sj : zero);
      }
    };
    if (s > 0) load_block();                     // in flight across the wait (nobody writes the block before this workgroup does)
    if (s == 0) {
      if (start) wait_for(start, start_target);  // (the gate may be what completes the matrix itself: the separator block)
      load_block();
      band_diag_accum<BAND_DC>(s0, s0.jlo, k0, n, bw, k0, nb, zero, M.R, acc);
      if (two_seg) { const BandSeg s1 = seg1(k0, nb); band_diag_accum<BAND_DC>(s1, s1.jlo, s1.jhi, n, bw, k0, nb, zero, M.R, acc); }
    } else {
      BAND_DTICK(4);
      wait_for(bar, (ep0 + s) * G);
      BAND_DTICK(0);
      band_diag_accum<BS>(s0, max(s0.jlo, k0 - BS), k0, n, bw, k0, nb, zero, M.R, acc);   // the newest block: 4 loads per thread
    }
    // U = A - sum of the four waves' partial sums (lower triangle)
    {
      __syncthreads();
      double* part = M.R;                       // [wave][32][33]
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int t = 0; t < 4; t++) part[(wv * BS + 4 * rg + m) * (BS + 1) + 4 * cg + t] = acc[m][t];
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int o = r * (BS + 1) + 4 * cq + t, ws = BS * (BS + 1);
        M.U[r][4 * cq + t] = av[t] - ((part[o] + part[ws + o]) + (part[2 * ws + o] + part[3 * ws + o]));
      }
    }
    __syncthreads();
    BAND_DTICK(1);
    {
      const bool bad = KID == 0 ? band_potf2_inv_k0(M.U, nb, M.Dl, M.X, M.colbuf) : band_potf2_inv_k1(M.U, nb, M.Dl, M.X, M.colbuf);
      if (bad && tid == 0) atomicCAS(info, 0, k0 + 1);
    }
    __syncthreads();
    BAND_DTICK(2);
    {
      const double* Li = Linv + (size_t)(k0 / BS) * BS * BS;
      for (int e = tid; e < BS * BS; e += 256) {
        const int rr = e >> 5, cc = e & 31;
        band_gstore(&Li[e], M.X[rr][cc]);
        if (rr < nb && cc <= rr && rr - cc <= bw) band_gstore(&view.base[(long long)(k0 + rr) * view.si + (long long)(k0 + cc) * view.sj], M.Dl[rr][cc]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      band_release();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(flag, f0 + s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    BAND_DTICK(3);
    // the next block's history except its newest 32 columns (those wait for the team's barrier)
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int t = 0; t < 4; t++) acc[m][t] = 0.0;
    const int k1 = k0 + BS;
    if (k1 < k_end) {
      const int nb1 = min(BS, k_end - k1);
      const BandSeg n0{view, max(0, k1 - bw), k1, 0};
      band_diag_accum<BAND_DC>(n0, n0.jlo, max(n0.jlo, k1 - BS), n, bw, k1, nb1, zero, M.R, acc);
      if (two_seg) { const BandSeg n1 = seg1(k1, nb1); band_diag_accum<BAND_DC>(n1, n1.jlo, n1.jhi, n, bw, k1, nb1, zero, M.R, acc); }
    }
  }
}

#undef BAND_DTICK
// (see band_potf2_inv_k0: a caller-side object whose address escapes before the first out-of-line call keeps those calls from
// being marked as tail-call candidates, which would switch the callees back to the save-everything convention)
__device__ __attribute__((noinline)) void band_clock_init(long long* t) { asm volatile("" : : "v"(t) : "memory"); *t = 0; }

// Two-sided ("burn at both ends") elimination.  Phase 1: team 0 eliminates the first K1 column blocks with the forward
// front while team 1 eliminates the last K2 column blocks with the reverse front -- the two regions are further apart
// than the bandwidth, so they never touch the same entries; each front also produces the rows of L that reach into the
// middle block M = [32 K1, n - 32 K2).  Phase 2: team 0 eliminates M, whose history now has two segments (the tails of
// both fronts).  The chain of dependent steps is halved.  K2 = 0 degenerates to the one-sided left-looking algorithm.
// bars: [team 0, team 1, both].
__global__ __launch_bounds__(256) void band_chol_coop_kernel(const double* Sb, const double* __restrict__ Linv_f, const double* __restrict__ Linv_r, const double* rhs, const double* zero, int n,
                                                             int LD, int K1, int K2, int* info, unsigned* bars, int G, long long* prof) {
  __shared__ double R[BAND_DC * BAND_NRP];   // strip chunk, transposed: R[jj * 64 + rr]; afterwards the 4 waves' partial sums
  __shared__ double U[BAND_NR][BS + 1];
  __shared__ double Dl[BS][BS + 1];
  __shared__ double X[BS][BS + 1];
  __shared__ double colbuf[2 * 256];
  __shared__ int rowidx[BAND_NR];
  const BandLds M{R, U, Dl, X, colbuf, rowidx};
  const int team = blockIdx.x / (G + 1), w = blockIdx.x % (G + 1);   // w == G: the front's diagonal workgroup
  const int tid = threadIdx.x;
  const int bw = LD - 1;
  const bool has_rhs = (w == G - 1);
  unsigned* dflag = bars + 14 + team;     // info[15 + team]
  long long t_escape; band_clock_init(&t_escape);
  const BandView fwd{Sb, 1, (long long)bw, rhs, 1};
  const BandView rev{Sb + (size_t)(n - 1) * LD, -(long long)bw, -1, rhs + (n - 1), -1};
  const int m_begin = BS * K1, m_end = n - BS * K2;      // the middle block (original indices)
  long long tp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = prof ? wall_clock64() : 0;   // optional phase clock (CS_BAND_PROF)
  const long long c_begin = prof ? clock64() : 0, w_begin = t_prev;
  const bool clocked = prof && team == 0;
  long long* tpp = clocked ? tp : nullptr;
#define BAND_TICK(k) do { if (clocked) { long long t_now = wall_clock64(); tp[k] += t_now - t_prev; t_prev = t_now; } } while (0)
  const BandSeg none{fwd, 0, 0, 0};
  const BandAug noaug{zero, 0, 0, 0, 0, nullptr, 0, 0, 0};
  // ---- phase 1: the two fronts, each with its own barrier (after every step, the last one included: the deferred
  // write of the diagonal block must not overtake a team-mate still reading A's block)
  unsigned ep = 0;   // barriers completed on this team's counter
  if (w == G) {
    const BandView view = team ? rev : fwd;
    band_diag_phase<0>(M, view, team ? Linv_r : Linv_f, 0, BS * (team ? K2 : K1), false, rev, 0, nullptr, 0, bars + team, (unsigned)G, 0, dflag, 0, n, bw, zero, info);
    // the middle block opens when both fronts have met (one-sided order: when the forward front's last barrier is through)
    if (team == 0) band_diag_phase<0>(M, fwd, Linv_f, m_begin, m_end, K2 > 0, rev, BS * K2, K2 > 0 ? bars + 2 : bars + 0, K2 > 0 ? 2u * (unsigned)G : (unsigned)K1 * (unsigned)G, bars + 0, (unsigned)G, (unsigned)K1, dflag, (unsigned)K1, n, bw, zero, info);
    return;
  }
  {
    const BandView view = team ? rev : fwd;
    const double* Linv = team ? Linv_r : Linv_f;
    const int Kt = team ? K2 : K1;
    for (int kb = 0; kb < Kt; kb++) {
      const int k0 = kb * BS;
      const BandSeg s0{view, max(0, k0 - bw), k0, 0};
      band_step<false, 0>(M, view, Linv, k0, BS, min(n, k0 + BS + bw), s0, none, 1, noaug, w, -1, has_rhs, n, bw, zero, dflag, ep + 1, tpp, &t_prev);
      ep++;
      band_grid_sync(bars + team, ep * (unsigned)G);
      BAND_TICK(5);
    }
    if (K2 > 0) {   // the fronts meet: the reverse team publishes and leaves, the forward team waits for it
      if (team) { band_grid_arrive(bars + 2); return; }
      band_grid_sync(bars + 2, 2u * (unsigned)G);
    }
  }
  // ---- phase 2: the middle block, forward front, history = tail of the forward front (+ M's own earlier columns) and
  // tail of the reverse front
  for (int k0 = m_begin; k0 < m_end; k0 += BS) {
    const int nb = min(BS, m_end - k0);
    const int i_end = min(m_end, k0 + nb + bw);
    const BandSeg s0{fwd, max(0, k0 - bw), k0, 0};
    const BandSeg s1{rev, max(0, (n - i_end) - bw), BS * K2, 1};     // rows i -> i' = n - 1 - i; columns of the reverse front within the band
    band_step<false, 0>(M, fwd, Linv_f, k0, nb, i_end, s0, s1, K2 > 0 ? 2 : 1, noaug, w, -1, has_rhs, n, bw, zero, dflag, ep + 1, tpp, &t_prev);
    ep++;
    band_grid_sync(bars + 0, ep * (unsigned)G);
    BAND_TICK(5);
  }
  if (prof && tid == 0 && (w == 0 || w == G - 1)) for (int k = 0; k < 9; k++) prof[(w == 0 ? 0 : 9) + k] = tp[k];
  if (prof && tid == 0 && w == 0) { prof[18] = clock64() - c_begin; prof[19] = wall_clock64() - w_begin; }
#undef BAND_TICK
}

// Nested elimination: four fronts.  A separator block C (wc >= bw columns in the middle of the matrix) splits the band
// into a left and a right half that touch only through C.  Each half is eliminated "at both ends" as above -- the left
// one in the matrix's own index space, the right one in the mirrored one (v = n - 1 - index), so that in both C lies
// below the half -- by two teams: team 0 the front that starts at the far end, team 1 the front that starts next to C
// and then the half's middle block.  The rows of C ride along with team 1 as extra panel rows (GC more workgroups): they
// fill in across everything team 1 eliminates, so their factor entries go to a dense array lc[t][q] (t: elimination
// column of the half, q: row of C).  A third team per half (GC workgroups, 16 rows of C each) accumulates C's Schur
// complement sum_t L(i, t) L(j, t) one step behind team 1, off the chain.  When both halves are done, C -- a dense
// wc x wc block in its own storage -- is factorised by the same step routine.  Chain of dependent steps: about
// n / (4 * 32) instead of n / (2 * 32).
struct BandHalf {
  BandView fv, rv;        // the half's index space [0, nh) (C at rows nh ...) and its reverse front's (u = nh - 1 - v)
  int nh, K1, K2, qflip;  // column blocks of the two fronts; qflip: C's rows appear in reverse order below this half
  const double* Linv_f; const double* Linv_r; const double* lc;
};
struct BandNested {
  BandHalf h[2];
  int n, bw, wc, c0, LD, G, GC, GS;     // C = [c0, c0 + wc); G workgroups per front, GC for C's rows (and C's own factorisation), GS Schur accumulators
  const double* Sb; const double* rhs; const double* SC; const double* rhsC; const double* LinvC; const double* part;
  const double* zero; int* info; unsigned* bars;   // bars: [half 0: team 0, team 1, join][half 1: ...][teams 1 + 2 of both halves][-][C team]
  long long* prof;                                 // optional phase stamps of half 0's team 1 (CS_BAND_PROF)
};
__device__ __forceinline__ const double* band_half_y(const BandHalf& H, int t) {   // right-hand side entry of elimination column t of team 1
  const int Trev = BS * H.K2;
  return t < Trev ? H.rv.rb + (long long)t * H.rv.sr : H.fv.rb + (long long)(BS * H.K1 + t - Trev) * H.fv.sr;
}
__global__ __launch_bounds__(256) void band_chol_nested_kernel(BandNested P) {
  __shared__ double R[BAND_DC * BAND_NRP];
  __shared__ double U[BAND_NR][BS + 1];
  __shared__ double Dl[BS][BS + 1];
  __shared__ double X[BS][BS + 1];
  __shared__ double colbuf[2 * 256];
  __shared__ int rowidx[BAND_NR];
  const BandLds M{R, U, Dl, X, colbuf, rowidx};
  const int tid = threadIdx.x;
  const int G = P.G, G1 = P.G + P.GC, per = (G + 1) + (G1 + 1) + P.GS;   // team 0 + its diagonal workgroup, team 1 + its, team 2
  const int hid = blockIdx.x / per, r = blockIdx.x % per;
  const int team = r < G + 1 ? 0 : (r < G + 1 + G1 + 1 ? 1 : 2), wi = team == 2 ? r - (G + 1) - (G1 + 1) : (team ? r - (G + 1) : r);
  const bool diag = (team == 0 && wi == G) || (team == 1 && wi == G1);   // the front's diagonal workgroup
  const int cw = (team == 1 && !diag && wi >= G) ? wi - G : -1;           // separator-row workgroup
  unsigned* dflag = P.bars + 9 + 2 * hid + (team == 1);                  // info[15 + 2 hid + team]
  long long t_escape; band_clock_init(&t_escape);
  const int w = cw < 0 ? wi : 0;
  const bool has_rhs = (cw < 0 && w == G - 1);
  const BandHalf H = P.h[hid];
  unsigned* bars = P.bars + 3 * hid;
  const int nh = H.nh, bw = P.bw, wc = P.wc;
  const int Trev = BS * H.K2, m_begin = BS * H.K1, m_end = nh - BS * H.K2;
  const double* zero = P.zero;
  const BandSeg none{H.fv, 0, 0, 0};
  const BandSegX nonex{{H.fv, 0, 0, 0}, 0, 0};
  const BandAug noaug{zero, 0, 0, 0, 0, nullptr, 0, 0, 0};
  // (the clock words are passed by address to the out-of-line step functions: with a caller-side object escaping, their calls
  // are not tail-call candidates and the functions keep the no-callee-saved-registers convention, see band_potf2_inv_k0)
  long long tp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
  long long* tpp = (P.prof && hid == 0 && team == 1 && (wi == 0 || wi == G1 - 1)) ? tp : nullptr;   // phase clock of a panel-row and a separator-row workgroup
  if (tpp) t_prev = wall_clock64();
  unsigned ep = 0;
  const bool stamp = P.prof && hid == 0 && team == 1 && wi == 0 && tid == 0;
#define BAND_STAMP(k) do { if (stamp) P.prof[k] = wall_clock64(); } while (0)
  BAND_STAMP(0);
  const int nslab = P.GS, GCC = P.GC;   // Schur slabs of 16 rows; workgroups of C's own factorisation
  if (diag) {
    if (team == 0) { band_diag_phase<1>(M, H.fv, H.Linv_f, 0, m_begin, false, H.rv, 0, nullptr, 0, bars + 0, (unsigned)G, 0, dflag, 0, nh, bw, zero, P.info); return; }
    band_diag_phase<1>(M, H.rv, H.Linv_r, 0, Trev, false, H.fv, 0, nullptr, 0, bars + 1, (unsigned)G1, 0, dflag, 0, nh, bw, zero, P.info, (P.prof && hid == 0) ? P.prof + 34 : nullptr);
    band_diag_phase<1>(M, H.fv, H.Linv_f, m_begin, m_end, true, H.rv, Trev, bars + 2, (unsigned)(G + G1), bars + 1, (unsigned)G1, (unsigned)H.K2, dflag, (unsigned)H.K2, nh, bw, zero, P.info);
    if (hid != 0) return;
    const BandView vcd{P.SC, 1, (long long)wc, P.rhsC, 1};
    const unsigned km = (unsigned)((m_end - m_begin + BS - 1) / BS);
    band_diag_phase<1>(M, vcd, P.LinvC, 0, wc, false, vcd, 0, P.bars + 8, (unsigned)GCC, P.bars + 8, (unsigned)GCC, 1, dflag, (unsigned)H.K2 + km, wc, wc - 1, zero, P.info);
    return;
  }
  if (team == 0) {
    for (int kb = 0; kb < H.K1; kb++) {
      const int k0 = kb * BS;
      const BandSeg s0{H.fv, max(0, k0 - bw), k0, 0};
      band_step<false, 1>(M, H.fv, H.Linv_f, k0, BS, min(nh, k0 + BS + bw), s0, none, 1, noaug, w, -1, has_rhs, nh, bw, zero, dflag, ep + 1, tpp, &t_prev);
      ep++;
      band_grid_sync(bars + 0, ep * (unsigned)G);
    }
    band_grid_arrive(bars + 2);
    return;
  }
  if (team == 2) {
    // Schur accumulator of C: 16 rows of sum_t L(i, t) L(j, t) and sum_t L(i, t) y(t) over everything team 1 eliminates,
    // one 32-column block behind it (waits on team 1's barrier counter without taking part in it)
    const int ldb = wc + 1, Th = nh - BS * H.K1, nstep = (Th + BS - 1) / BS;
    double* buf = R;                         // 32 x (wc + 1): rows of lc + the right-hand side entry
    const int tt8 = tid >> 3, l8 = tid & 7, lane = tid & 63, wv4 = tid >> 6, li = lane & 15, lk = lane >> 4;
    // the slab's 16 x wc block on the matrix cores: wave w takes the 16-column tiles w, w + 4, ... (wc <= 256: at most four), every
    // step is 8 x 16x16x4 issues per tile -- the multiply-add form read 17 LDS words per row of the step and per thread, and had
    // become the slowest team of the kernel once the fronts' hand-offs got cheaper
    ba_v4d acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = ba_v4d{0.0, 0.0, 0.0, 0.0};
    double accy = 0.0;                       // threads 0..15: the slab's share of sum_t L(i, t) y(t)
    for (int sidx = 0; sidx < nstep; sidx++) {
      if (tid == 0) {
        band_wait_ge(bars + 1, (unsigned)(sidx + 1) * (unsigned)G1);
        band_acquire();
      }
      __syncthreads();
      const int t = sidx * BS + tt8;
      const double* row = H.lc + (long long)t * wc + l8;
      double vals[32];
#pragma unroll
      for (int u = 0; u < 32; u++) vals[u] = band_gload((t < Th && 8 * u + l8 < wc) ? row + 8 * u : zero);
      const double yv = band_gload((t < Th && l8 == 0) ? band_half_y(H, t) : zero);
#pragma unroll
      for (int u = 0; u < 32; u++) if (8 * u < wc) buf[tt8 * ldb + l8 + 8 * u] = vals[u];
      if (l8 == 0) buf[tt8 * ldb + wc] = yv;
      __syncthreads();
#pragma unroll
      for (int t0 = 0; t0 < BS; t0 += 4) {
        const double* bt = buf + (t0 + lk) * ldb;
        const double a = bt[wi * 16 + li];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int ct = wv4 + 4 * q;
          if (16 * ct < wc) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bt[16 * ct + li], acc[q], 0, 0, 0);
        }
      }
      if (tid < 16) {
#pragma unroll 8
        for (int tt = 0; tt < BS; tt++) accy = fma(buf[tt * ldb + wi * 16 + tid], buf[tt * ldb + wc], accy);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int ct = wv4 + 4 * q;
      if (16 * ct >= wc) continue;
#pragma unroll
      for (int g = 0; g < 4; g++) band_gstore(&P.part[((size_t)hid * wc + wi * 16 + lk + 4 * g) * ldb + 16 * ct + li], acc[q][g]);
    }
    if (tid < 16) band_gstore(&P.part[((size_t)hid * wc + wi * 16 + tid) * ldb + wc], accy);
    band_grid_arrive(P.bars + 6);
    return;
  }
  // ---- team 1: the front next to C (reverse view of the half), C's rows below it
  {
    // A(q, u): entry (nh + q, nh - 1 - u) of the half's space, inside the band iff q + 1 + u <= bw
    const BandAug aug{H.fv.base + (long long)nh * H.fv.si + (long long)(nh - 1) * H.fv.sj, H.fv.si, -H.fv.sj, 1, 1, H.lc, wc, H.qflip, 0};
    for (int kb = 0; kb < H.K2; kb++) {
      const int k0 = kb * BS;
      const BandSegX s0{{H.rv, max(0, k0 - bw), k0, 0}, 0, 0};
      band_step<true, 1>(M, H.rv, H.Linv_r, k0, BS, min(nh, k0 + BS + bw), s0, nonex, 1, aug, w, cw, has_rhs, nh, bw, zero, dflag, ep + 1, tpp, &t_prev);
      ep++;
      band_grid_sync(bars + 1, ep * (unsigned)G1);
      if (tpp) { long long t_now = wall_clock64(); tp[5] += t_now - t_prev; t_prev = t_now; }
    }
  }
  if (tpp && tid == 0) for (int k = 0; k < 9; k++) P.prof[16 + (wi == 0 ? 0 : 9) + k] = tp[k];
  BAND_STAMP(1);
  band_grid_sync(bars + 2, (unsigned)(G + G1));
  BAND_STAMP(2);
  // ---- the half's middle block (its own space): history = tail of team 0's front + the block's earlier columns, and
  // the tail of the reverse front; C's rows: entry (nh + q, j), inside the band iff nh + q - j <= bw
  {
    const BandAug aug{H.fv.base + (long long)nh * H.fv.si, H.fv.si, H.fv.sj, nh, -1, H.lc, wc, H.qflip, Trev - m_begin};
    for (int k0 = m_begin; k0 < m_end; k0 += BS) {
      const int nb = min(BS, m_end - k0);
      const int i_end = min(m_end, k0 + nb + bw);
      const BandSegX s0{{H.fv, max(0, k0 - bw), k0, 0}, Trev - m_begin, m_begin};
      const BandSegX s1{{H.rv, max(0, (nh - i_end) - bw), Trev, 1}, 0, 0};
      band_step<true, 1>(M, H.fv, H.Linv_f, k0, nb, i_end, s0, s1, 2, aug, w, cw, has_rhs, nh, bw, zero, dflag, ep + 1, tpp, &t_prev);
      ep++;
      band_grid_sync(bars + 1, ep * (unsigned)G1);
    }
  }
  // ---- both halves (and their Schur accumulators, team 2) done
  BAND_STAMP(3);
  band_grid_sync(P.bars + 6, 2u * (unsigned)(G1 + nslab));
  BAND_STAMP(4);
  if (hid != 0 || wi >= GCC) return;
  // ---- the C team: Schur complement of its row slab (lower triangle; the first GS workgroups, 16 rows each), then the dense factorisation
  if (wi < nslab) {
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
  unsigned epc = 1;
  band_grid_sync(P.bars + 8, epc * (unsigned)GCC);
  BAND_STAMP(7);
  const BandView vc{P.SC, 1, (long long)wc, P.rhsC, 1};
  for (int k0 = 0; k0 < wc; k0 += BS) {
    const BandSeg s0{vc, 0, k0, 0};
    band_step<false, 1>(M, vc, P.LinvC, k0, BS, wc, s0, none, 1, noaug, wi, -1, wi == GCC - 1, wc, wc - 1, zero, dflag, ep + epc, tpp, &t_prev);
    epc++;
    band_grid_sync(P.bars + 8, epc * (unsigned)GCC);
  }
  BAND_STAMP(8);
#undef BAND_STAMP
}

// L^T x = y in place in rhs, in the elimination order reversed: first the middle block (forward view), then the two
// fronts' regions independently -- workgroup 0 walks the forward front's blocks back to the top, workgroup 1 (it
// repeats the middle block for itself) the reverse front's blocks back to the bottom.  Per block: t = (rows below)^T x
// with x from an LDS window, x_k = L_kk^-T (y_k - t) with the inverted diagonal block.  Everything a step reads from
// memory (its panel of L, L_kk^-1, y_k) is independent of x and is fetched one step ahead, all loads unconditional.
// With the nested elimination the same runs per half (4 workgroups), after C's unknowns have been solved and their
// contribution taken out of the halves' right-hand sides (band_sep_solve_kernel, band_sep_correct_kernel).
enum { BAND_WIN = 8192, BAND_PF = 24 };      // x window (a step touches <= 32 + 4096 consecutive rows); prefetched rows per thread
struct BandSolveLds { double* xw; double* z; double (*Li)[BS + 1]; double (*part)[BS]; };
// blocks kb_hi .. kb_lo (descending) of `view`; rows of the view at or beyond row_limit do not exist
__device__ __forceinline__ void band_backsolve_run(const BandSolveLds& M, const BandView& view, const double* Linv, int kb_hi, int kb_lo, int row_limit, int bw, const double* zero) {
  constexpr int WIN = BAND_WIN, PF = BAND_PF;
  const int tid = threadIdx.x, c = tid & 31, g = tid >> 5;
  double lv[PF], li[4], yv = 0.0;
  auto prefetch = [&](int kb) {
    const int k0 = kb * BS, nb = min(BS, row_limit - k0), i_end = min(row_limit, k0 + nb + bw);
    const double* col = view.base + (long long)(k0 + c) * view.sj;     // &L(i, k0 + c) = col + i * si
#pragma unroll
    for (int s = 0; s < PF; s++) {
      const int i = k0 + nb + g + 8 * s;
      const bool ok = c < nb && i < i_end && i - (k0 + c) <= bw;
      lv[s] = *(ok ? col + (long long)i * view.si : zero);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) li[u] = Linv[(size_t)kb * BS * BS + tid + 256 * u];
    yv = (tid < nb) ? view.rb[(long long)(k0 + tid) * view.sr] : 0.0;
  };
  if (kb_hi < kb_lo) return;
  prefetch(kb_hi);
  __syncthreads();
  for (int kb = kb_hi; kb >= kb_lo; kb--) {
    const int k0 = kb * BS, nb = min(BS, row_limit - k0);
    const int i_end = min(row_limit, k0 + nb + bw);
    double acc = 0;   // t[c] = sum over the rows below the block of L(i, k0 + c) x[i]; 8 row groups
#pragma unroll
    for (int s = 0; s < PF; s++) acc = fma(lv[s], M.xw[(k0 + nb + g + 8 * s) & (WIN - 1)], acc);
    if (c < nb)
      for (int i = k0 + nb + g + 8 * PF; i < i_end; i += 8)
        if (i - (k0 + c) <= bw) acc = fma(view.base[(long long)i * view.si + (long long)(k0 + c) * view.sj], M.xw[i & (WIN - 1)], acc);
    M.part[g][c] = acc;
#pragma unroll
    for (int u = 0; u < 4; u++) { int e = tid + 256 * u; M.Li[e >> 5][e & 31] = li[u]; }
    __syncthreads();
    if (tid < BS) {
      double tt = 0;
#pragma unroll
      for (int q = 0; q < 8; q++) tt += M.part[q][tid];
      M.z[tid] = (tid < nb) ? yv - tt : 0.0;
    }
    __syncthreads();
    if (kb > kb_lo) prefetch(kb - 1);
    if (tid < nb) {
      double a2 = 0;
#pragma unroll 8
      for (int r = 0; r < BS; r++) a2 = fma(M.Li[r][tid], M.z[r], a2);   // (L^-1)^T z
      band_pstore(&view.rb[(long long)(k0 + tid) * view.sr], a2);
      M.xw[(k0 + tid) & (WIN - 1)] = a2;
    }
    __syncthreads();
  }
}
struct BandSolve { BandHalf h[2]; int bw; const double* zero; };   // one half (the whole matrix) or two
__global__ __launch_bounds__(256) void band_backsolve_kernel(BandSolve P) {
  __shared__ double xw[BAND_WIN];
  __shared__ double z[BS];
  __shared__ double Li[BS][BS + 1];
  __shared__ double part[8][BS];
  const BandSolveLds M{xw, z, Li, part};
  const int tid = threadIdx.x;
  const BandHalf H = P.h[blockIdx.x >> 1];
  const int bw = P.bw, nh = H.nh, m_end = nh - BS * H.K2;
  for (int e = tid; e < BAND_WIN; e += 256) xw[e] = 0.0;
  // the middle block (the half's own view; rows end at m_end)
  band_backsolve_run(M, H.fv, H.Linv_f, (m_end - 1) / BS, H.K1, m_end, bw, P.zero);
  if ((blockIdx.x & 1) == 0) {
    band_backsolve_run(M, H.fv, H.Linv_f, H.K1 - 1, 0, nh, bw, P.zero);
  } else {
    // the same unknowns seen from the reverse front: x of the middle block into the window at its reverse indices
    __syncthreads();
    for (int e = tid; e < BAND_WIN; e += 256) xw[e] = 0.0;
    __syncthreads();
    for (int v = BS * H.K2 + tid; v < min(nh, BS * H.K2 + bw + BS); v += 256) xw[v & (BAND_WIN - 1)] = H.rv.rb[(long long)v * H.rv.sr];
    __syncthreads();
    band_backsolve_run(M, H.rv, H.Linv_r, H.K2 - 1, 0, nh, bw, P.zero);
  }
}
// C's unknowns: L_C^T x_C = y_C on the dense block (one workgroup)
__global__ __launch_bounds__(256) void band_sep_solve_kernel(BandNested P) {
  __shared__ double xw[BAND_WIN];
  __shared__ double z[BS];
  __shared__ double Li[BS][BS + 1];
  __shared__ double part[8][BS];
  const BandSolveLds M{xw, z, Li, part};
  for (int e = threadIdx.x; e < BAND_WIN; e += 256) xw[e] = 0.0;
  const BandView vc{P.SC, 1, (long long)P.wc, P.rhsC, 1};
  band_backsolve_run(M, vc, P.LinvC, P.wc / BS - 1, 0, P.wc, P.wc - 1, P.zero);
}
// y(t) -= sum_q L(q, t) x_C(q) for every elimination column t that carries C's rows (32 per workgroup, 8 lanes each);
// workgroup 0 also stores x_C at C's place in the solution vector
__global__ __launch_bounds__(256) void band_sep_correct_kernel(BandNested P) {
  __shared__ double xc[256];
  const int tid = threadIdx.x, wc = P.wc;
  for (int e = tid; e < wc; e += 256) xc[e] = P.rhsC[e];
  __syncthreads();
  if (blockIdx.x == 0) for (int e = tid; e < wc; e += 256) band_pstore(&P.rhs[P.c0 + e], xc[e]);
  const int T0 = P.h[0].nh - BS * P.h[0].K1, T1 = P.h[1].nh - BS * P.h[1].K1;
  const int t = blockIdx.x * 32 + (tid >> 3), l8 = tid & 7;
  double s = 0.0;
  if (t < T0 + T1) {
    const BandHalf& H = P.h[t < T0 ? 0 : 1];
    const double* row = H.lc + (long long)(t < T0 ? t : t - T0) * wc;
    for (int q = l8; q < wc; q += 8) s = fma(row[q], xc[q], s);
  }
  s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
  if (t < T0 + T1 && l8 == 0) {
    const double* y = band_half_y(P.h[t < T0 ? 0 : 1], t < T0 ? t : t - T0);
    band_pstore(y, *y - s);
  }
}

int ba_band_team(int LD, int* rw_out) {   // workgroups of the factorisation team and their rows per step
  const int bw = LD - 1;
  int G = (bw + BAND_RW - 1) / BAND_RW;
  if (G < 1) G = 1;
  *rw_out = BAND_RW;
  return G;
}

void ba_band_split(int n, int LD, int* K1, int* K2, bool force_one_sided = false) {   // column blocks of the two fronts; the middle block keeps >= bw columns
  const int bw = LD - 1;
  const int Kt = n > bw ? (n - bw) / BS : 0;
  static const bool one_sided = getenv("CS_BAND_ONE_SIDED") != nullptr;   // diagnostics: plain left-looking order
  *K2 = (Kt >= 8 && !one_sided && !force_one_sided) ? Kt / 2 : 0;
  *K1 = Kt - *K2;
}
// The nested (four-front) order pays when both halves still have two real fronts; it needs the separator's row
// workgroups co-resident with the four teams, which bounds the bandwidth it is used for.
bool ba_band_nested(int n, int LD, int* wc_out, int* c0_out, bool force_one_sided = false) {
  if (force_one_sided) return false;
  static const bool off = getenv("CS_BAND_TWO_FRONTS") != nullptr || getenv("CS_BAND_ONE_SIDED") != nullptr;   // diagnostics: the two-front order
  const int bw = LD - 1;
  const int wc = ((bw + BS - 1) / BS) * BS;
  const int nh = (n - wc) / 2;
  if (off || bw > 256 || bw < 1 || nh < bw + 16 * BS) return false;
  *wc_out = wc; *c0_out = nh;
  return true;
}
static size_t band_blocks(int n) { return (size_t)((n + BS - 1) / BS) * BS * BS; }
// doubles of workspace behind `work` (inverted diagonal blocks; nested order: + the separator's factor rows, partial
// Schur complements and dense block)
size_t ba_band_workspace_doubles(int n, int LD) {
  if (n <= 0) return 1;
  size_t two = 2 * band_blocks(n);
  int wc = 0, c0 = 0;
  if (!ba_band_nested(n, LD, &wc, &c0)) return two;
  const int nh0 = c0, nh1 = n - c0 - wc;
  size_t nest = 2 * band_blocks(nh0) + 2 * band_blocks(nh1) + band_blocks(wc) + (size_t)(nh0 + nh1) * wc + (size_t)2 * wc * (wc + 1) + (size_t)wc * wc + wc;
  return nest > two ? nest : two;
}

// Does the persistent factorisation of an n x n band with LD = bandwidth + 1 fit the current device?  Its workgroups wait for
// each other, so the whole grid must be resident at once: the grid may not exceed (workgroups the occupancy query admits per
// CU, capped at 1: the kernels are written for one workgroup per CU) x (CUs of the device or partition).  The caller takes the
// dense rocSOLVER path otherwise.
bool ba_band_fits_device(int n, int LD, bool one_sided) {   // one_sided: the order ba_launch_band_cholesky(..., one_sided = true) runs (the sharded solve's interiors)
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
  int rw = 0, K1 = 0, K2 = 0, wc = 0, c0 = 0, occ = 0, grid = 0;
  const int G = ba_band_team(LD, &rw), bw = LD - 1;
  if (ba_band_nested(n, LD, &wc, &c0, one_sided)) {
    const int Gn = (bw + BAND_RW_NESTED - 1) / BAND_RW_NESTED, GC = wc / BAND_RW_NESTED, GS = wc / 16;
    grid = 2 * (Gn + 1 + Gn + GC + 1 + GS);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, band_chol_nested_kernel, 256, 0) != hipSuccess) return false;
  } else {
    ba_band_split(n, LD, &K1, &K2, one_sided);
    grid = K2 > 0 ? 2 * (G + 1) : G + 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, band_chol_coop_kernel, 256, 0) != hipSuccess) return false;
  }
  return occ >= 1 && grid <= prop.multiProcessorCount;
}

// info (24 ints, zeroed by the caller): [0] first non-positive pivot (+1; INT_MAX: a wait timed out, see band_wait_ge), [1..3] barrier counters of the two-front
// order (forward team, reverse team, both), [4..5] a zero double (the target of masked loads), [6..14] barrier counters
// of the nested order, [15..18] published-block counters of the fronts' diagonal workgroups, [19] abort word.  work: ba_band_workspace_doubles(n, LD) doubles.
// one_sided: the plain left-looking order (the factor then lies in the band in the matrix's own index space: what the sharded
// solve's interior elimination needs, ba_launch_sep_*)
void ba_launch_band_cholesky(double* Sb, double* work, int n, int LD, double* rhs, int* info, bool solve, hipStream_t st, bool one_sided) {
  int rw = 0, K1 = 0, K2 = 0, wc = 0, c0 = 0;
  const int G = ba_band_team(LD, &rw);
  const int bw = LD - 1;
  const double* zero = reinterpret_cast<const double*>(info + 4);
  const BandView fwd{Sb, 1, (long long)bw, rhs, 1};
  const BandView rev{Sb + (size_t)(n - 1) * LD, -(long long)bw, -1, rhs + (n - 1), -1};
  if (ba_band_nested(n, LD, &wc, &c0, one_sided)) {
    BandNested P;
    const int c1 = c0 + wc, nh[2] = {c0, n - c1};
    double* wp = work;
    for (int h = 0; h < 2; h++) {
      BandHalf& H = P.h[h];
      H.nh = nh[h];
      ba_band_split(nh[h], LD, &H.K1, &H.K2);
      H.qflip = h;
      H.Linv_f = wp; wp += band_blocks(nh[h]);
      H.Linv_r = wp; wp += band_blocks(nh[h]);
    }
    // left half: the matrix's own index space, reverse front from c0 - 1 down; right half: the mirrored space, its
    // reverse front therefore walks forward from c1
    P.h[0].fv = fwd;
    P.h[0].rv = BandView{Sb + (size_t)(c0 - 1) * LD, -(long long)bw, -1, rhs + (c0 - 1), -1};
    P.h[1].fv = rev;
    P.h[1].rv = BandView{Sb + (size_t)c1 * LD, 1, (long long)bw, rhs + c1, 1};
    P.LinvC = wp; wp += band_blocks(wc);
    for (int h = 0; h < 2; h++) { P.h[h].lc = wp; wp += (size_t)nh[h] * wc; }
    P.part = wp; wp += (size_t)2 * wc * (wc + 1);
    P.SC = wp; wp += (size_t)wc * wc;
    P.rhsC = wp; wp += wc;
    P.n = n; P.bw = bw; P.wc = wc; P.c0 = c0; P.LD = LD; P.G = (bw + BAND_RW_NESTED - 1) / BAND_RW_NESTED; P.GC = wc / BAND_RW_NESTED; P.GS = wc / 16;
    const int G1 = P.G + P.GC;
    P.Sb = Sb; P.rhs = rhs; P.zero = zero; P.info = info; P.bars = reinterpret_cast<unsigned*>(info + 6);
    static const bool want_stamps = getenv("CS_BAND_PROF") != nullptr;
    static long long* stamps = nullptr;
    if (want_stamps && !stamps) (void)hipMalloc(&stamps, 40 * sizeof(long long));
    if (stamps) (void)hipMemsetAsync(stamps + 34, 0, 6 * sizeof(long long), st);
    P.prof = stamps;
    hipLaunchKernelGGL(band_chol_nested_kernel, dim3(2 * (P.G + 1 + G1 + 1 + P.GS)), dim3(256), 0, st, P);
    long long hs_d[6] = {0, 0, 0, 0, 0, 0};
    if (stamps) {
      long long h[40];
      (void)hipMemcpyAsync(h, stamps, sizeof(h), hipMemcpyDeviceToHost, st);
      (void)hipStreamSynchronize(st);
      static int shown = 0;
      if (shown++ < 3)
        fprintf(stderr, "[band nested] n=%d bw=%d C=%d+%d halves %d/%d fronts %d+%d / %d+%d us: reverse front %.0f  join wait %.0f  middle %.0f  halves wait %.0f  C build %.0f  C factor %.0f\n",
                n, bw, c0, wc, nh[0], nh[1], P.h[0].K1, P.h[0].K2, P.h[1].K1, P.h[1].K2, (h[1] - h[0]) * 0.01, (h[2] - h[1]) * 0.01, (h[3] - h[2]) * 0.01, (h[4] - h[3]) * 0.01,
                (h[7] - h[4]) * 0.01, (h[8] - h[7]) * 0.01);
      if (shown <= 3)
        for (int q = 0; q < 2; q++)
          fprintf(stderr, "[band nested] reverse front, %s workgroup us: fetch-issue %.0f  store+wait %.0f  gemm %.0f  reduce+U %.0f  wait for L^-1 %.0f  panel %.0f  barrier %.0f\n", q ? "separator-row" : "panel-row",
                  h[16 + 9 * q + 6] * 0.01, h[16 + 9 * q + 7] * 0.01, h[16 + 9 * q + 8] * 0.01, h[16 + 9 * q + 1] * 0.01, h[16 + 9 * q + 2] * 0.01, h[16 + 9 * q + 4] * 0.01, h[16 + 9 * q + 5] * 0.01);
      for (int q = 0; q < 6; q++) hs_d[q] = h[34 + q];
    }
    if (stamps) {
      static int shown_d = 0;
      if (shown_d++ < 3) fprintf(stderr, "[band nested] reverse front, diagonal workgroup us: wait for the team's barrier %.0f  newest block + U %.0f  POTF2 + inverse %.0f  publish %.0f  look-ahead accumulation %.0f\n",
                                 hs_d[0] * 0.01, hs_d[1] * 0.01, hs_d[2] * 0.01, hs_d[3] * 0.01, hs_d[4] * 0.01);
    }
    if (solve) {
      const int TT = (nh[0] - BS * P.h[0].K1) + (nh[1] - BS * P.h[1].K1);
      hipLaunchKernelGGL(band_sep_solve_kernel, dim3(1), dim3(256), 0, st, P);
      hipLaunchKernelGGL(band_sep_correct_kernel, dim3((TT + 31) / 32), dim3(256), 0, st, P);
      BandSolve Q; Q.h[0] = P.h[0]; Q.h[1] = P.h[1]; Q.bw = bw; Q.zero = zero;
      hipLaunchKernelGGL(band_backsolve_kernel, dim3(4), dim3(256), 0, st, Q);
    }
    return;
  }
  ba_band_split(n, LD, &K1, &K2, one_sided);
  unsigned* bars = reinterpret_cast<unsigned*>(info + 1);
  double* Linv_f = work;
  double* Linv_r = work + band_blocks(n);
  static const bool want_prof = getenv("CS_BAND_PROF") != nullptr;   // diagnostics: phase clock of the forward team's first / last workgroup
  static long long* prof = nullptr;
  if (want_prof && !prof) (void)hipMalloc(&prof, 20 * sizeof(long long));
  hipLaunchKernelGGL(band_chol_coop_kernel, dim3(K2 > 0 ? 2 * (G + 1) : G + 1), dim3(256), 0, st, Sb, Linv_f, Linv_r, rhs, zero, n, LD, K1, K2, info, bars, G, prof);
  if (prof) {
    long long h[20];
    (void)hipMemcpyAsync(h, prof, sizeof(h), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    static int shown = 0;
    if (shown++ < 3) {
      fprintf(stderr, "[band] shader clock %.0f MHz over %.0f us; fronts %d + %d blocks, middle %d columns\n", h[18] / (h[19] * 0.01), h[19] * 0.01, K1, K2, n - BS * (K1 + K2));
      for (int q = 0; q < 2; q++)
        fprintf(stderr, "[band] n=%d LD=%d G=%d wg%s us: fetch-issue %.0f  store+wait %.0f  gemm %.0f  reduce+U %.0f  potf2+inverse %.0f  panel %.0f  barrier %.0f\n", n, LD, G, q ? "last" : "0",
                h[9 * q + 6] * 0.01, h[9 * q + 7] * 0.01, h[9 * q + 8] * 0.01, h[9 * q + 1] * 0.01, h[9 * q + 2] * 0.01, h[9 * q + 4] * 0.01, h[9 * q + 5] * 0.01);
    }
  }
  if (solve) {
    BandSolve Q;
    Q.h[0] = BandHalf{fwd, rev, n, K1, K2, 0, Linv_f, Linv_r, nullptr}; Q.h[1] = Q.h[0]; Q.bw = bw; Q.zero = zero;
    hipLaunchKernelGGL(band_backsolve_kernel, dim3(K2 > 0 ? 2 : 1), dim3(256), 0, st, Q);
  }
}


// ------------------------------------------------------------- sharded reduced solve: interiors and separators --
// Sharded BA (ba_host.cpp, "separator mode"): rank r owns the columns [cut_r, cut_r+1) of the band; its first w >= bw columns are the
// separator Z_r (none on rank 0), the rest its interior I_r.  Interiors of different ranks are further apart than the bandwidth, so
// they are decoupled; an interior is coupled to the separator in front of it (Z_r, "left") and to the one behind it (Z_r+1, the next
// rank's, "right").  Per damped solve a rank factorises ITS interior only (band_chol_coop_kernel, one-sided order, right-hand side
// riding along), and forms what the separators need of it:
//     Y = B L^-T   (rows: the unknowns of Z_r and Z_r+1, columns: the interior)      ba_sep_trsm_kernel
//     T = C - Y Y^T,   t = b_Z - Y y                                                  ba_sep_schur_kernel, ba_sep_rhs_kernel
// with C the rank's own partial blocks S(Z_r, Z_r), S(Z_r+1, Z_r+1).  The ranks exchange (T, t) -- three w x w blocks and two
// w-vectors each, the only matrix data that crosses the links --, every rank assembles and solves the small block-tridiagonal
// separator system, then x_I = L^-T (y - Y^T x_Z) (ba_sep_correct_kernel + band_backsolve_kernel).
struct SepView {
  const double* S; int LD;          // the band (lower, LD doubles per column), the interior's factor in place
  const double* Linv;               // inverted diagonal blocks of the interior's factor (32 x 32 each, row-major)
  int ci, ni;                       // interior = columns [ci, ci + ni)
  int zl, wl, zr, wr;               // left separator [zl, zl + wl), right separator [zr, zr + wr); widths may be 0
  double* Y;                        // (wl + wr) x ni, row-major
  const double* rhs;                // the global right-hand side / solution vector (y of the interior at rhs + ci)
};
// A(q, j): coupling of separator row q (0 .. wl - 1 left, wl .. wl + wr - 1 right) with interior column j
__device__ __forceinline__ double sep_coupling(const SepView& V, int q, int j) {
  const int bw = V.LD - 1;
  if (q < V.wl) {           // S(row ci + j, col zl + q): stored in the separator's column
    const int d = (V.ci + j) - (V.zl + q);
    return d <= bw ? V.S[(size_t)(V.zl + q) * V.LD + d] : 0.0;
  }
  const int d = (V.zr + (q - V.wl)) - (V.ci + j);     // S(row zr + q', col ci + j): stored in the interior's column, below the interior
  return d <= bw ? V.S[(size_t)(V.ci + j) * V.LD + d] : 0.0;
}
enum { SEP_RW = 8 };
// One workgroup per SEP_RW separator rows, walking the interior's column blocks in order (the rows are independent of each other:
// no grid barrier).  Thread (rq, c): row rq of the group, column c of the block.
__global__ __launch_bounds__(256) void ba_sep_trsm_kernel(SepView V) {
  __shared__ double Lc[BS][BS + 1];      // Lc[c'][d] = L(k0 + c', s0 + d)
  __shared__ double Yc[SEP_RW][BS + 1];  // Yc[rq][d] = Y(q, s0 + d); afterwards U
  __shared__ double Li[BS][BS + 1];
  const int tid = threadIdx.x, c = tid & 31, rq = tid >> 5;
  const int nq = V.wl + V.wr, q = blockIdx.x * SEP_RW + rq, bw = V.LD - 1;
  const bool qok = q < nq;
  const double* Sint = V.S + (size_t)V.ci * V.LD;      // &L(i, j) = Sint + j LD + (i - j)
  // rows of the right separator are zero up to the first interior column inside their band
  int first = 0;
  {
    const int q_lo = blockIdx.x * SEP_RW;              // the group's first row decides (left rows: column 0)
    if (q_lo >= V.wl) first = max(0, (V.zr + (q_lo - V.wl)) - bw - V.ci) / BS;
  }
  double* Yrow = V.Y + (size_t)(qok ? q : 0) * V.ni;
  for (int j = tid; j < first * BS; j += 256)
    for (int r = 0; r < SEP_RW; r++) if (blockIdx.x * SEP_RW + r < nq) V.Y[(size_t)(blockIdx.x * SEP_RW + r) * V.ni + j] = 0.0;
  const int nblk = (V.ni + BS - 1) / BS;
  for (int kb = first; kb < nblk; kb++) {
    const int k0 = kb * BS, nb = min(BS, V.ni - k0);
    double acc = 0.0;
    const int jlo = max(first * BS, max(0, k0 - bw) & ~(BS - 1));
    for (int s0 = jlo; s0 < k0; s0 += BS) {
      // L(k0 + c', s0 + d): thread loads c' = c, d = rq + 8 u (consecutive c' are consecutive addresses)
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int d = rq + 8 * u, i = k0 + c, j = s0 + d;
        Lc[c][d] = (c < nb && i - j <= bw) ? Sint[(size_t)j * V.LD + (i - j)] : 0.0;
      }
      Yc[rq][c] = qok ? Yrow[s0 + c] : 0.0;
      __syncthreads();
#pragma unroll 8
      for (int d = 0; d < BS; d++) acc = fma(Yc[rq][d], Lc[c][d], acc);
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; u++) { const int e = tid + 256 * u; Li[e >> 5][e & 31] = V.Linv[(size_t)kb * BS * BS + e]; }
    Yc[rq][c] = (qok && c < nb) ? sep_coupling(V, q, k0 + c) - acc : 0.0;     // U
    __syncthreads();
    double y = 0.0;
    for (int cc = 0; cc <= c; cc++) y = fma(Yc[rq][cc], Li[c][cc], y);        // U L_kk^-T
    if (qok && c < nb) Yrow[k0 + c] = y;
    __syncthreads();   // (also orders this block's stores before the next block's loads of them)
  }
}
// T = C - Y Y^T on the index pairs a >= b of the combined separator index (left rows first): one 16 x 16 tile per workgroup.
// msg = [LL | RL | RR | tL | tR] with blocks of wm x wm (row-major, wm = the widest separator of the job) and vectors of wm.
__global__ __launch_bounds__(256) void ba_sep_schur_kernel(SepView V, double* msg, int wm) {
  __shared__ double Ya[16][BS + 1], Yb[16][BS + 1];
  const int nq = V.wl + V.wr, nt = (nq + 15) / 16;
  // tile (ta, tb), ta >= tb, from the linear block index
  int ta = 0, rem = blockIdx.x;
  while (rem > ta) { rem -= ta + 1; ta++; }
  const int tb = rem;
  if (ta >= nt) return;
  const int tid = threadIdx.x, la = tid >> 4, lb = tid & 15;
  const int a = ta * 16 + la, b = tb * 16 + lb;
  double acc = 0.0;
  for (int j0 = 0; j0 < V.ni; j0 += BS) {
    for (int e = tid; e < 16 * BS; e += 256) {
      const int r = e >> 5, d = e & 31, j = j0 + d;
      Ya[r][d] = (ta * 16 + r < nq && j < V.ni) ? V.Y[(size_t)(ta * 16 + r) * V.ni + j] : 0.0;
      Yb[r][d] = (tb * 16 + r < nq && j < V.ni) ? V.Y[(size_t)(tb * 16 + r) * V.ni + j] : 0.0;
    }
    __syncthreads();
#pragma unroll 8
    for (int d = 0; d < BS; d++) acc = fma(Ya[la][d], Yb[lb][d], acc);
    __syncthreads();
  }
  if (a >= nq || b >= nq || a < b) return;
  const int bw = V.LD - 1;
  const bool ar = a >= V.wl, br = b >= V.wl;
  double cval = 0.0;
  if (ar == br) {         // both in one separator: the rank's own partial block of S
    const int z = ar ? V.zr : V.zl, ia = ar ? a - V.wl : a, ib = br ? b - V.wl : b;
    if (ia - ib <= bw) cval = V.S[(size_t)(z + ib) * V.LD + (ia - ib)];
  }
  double* dst = (!ar) ? msg + (size_t)a * wm + b                                   // LL
              : (!br) ? msg + (size_t)wm * wm + (size_t)(a - V.wl) * wm + b        // RL (row: right, column: left)
                      : msg + 2 * (size_t)wm * wm + (size_t)(a - V.wl) * wm + (b - V.wl);   // RR
  *dst = cval - acc;
}
// t = b_Z - Y y: one wavefront per separator row
__global__ __launch_bounds__(64) void ba_sep_rhs_kernel(SepView V, double* msg, int wm) {
  const int q = blockIdx.x, lane = threadIdx.x;
  if (q >= V.wl + V.wr) return;
  double s = 0.0;
  for (int j = lane; j < V.ni; j += 64) s = fma(V.Y[(size_t)q * V.ni + j], V.rhs[V.ci + j], s);
  s = wave_sum(s);
  if (lane == 0) {
    const bool right = q >= V.wl;
    const int g = right ? V.zr + (q - V.wl) : V.zl + q;
    msg[3 * (size_t)wm * wm + (right ? wm + (q - V.wl) : q)] = V.rhs[g] - s;
  }
}
// The separator system from the ranks' messages.  Separator k (k = 1 .. R - 1, rows sep_off[k] .. sep_off[k + 1] of the system) is
// the left separator of rank k and the right separator of rank k - 1: its diagonal block is LL of rank k + RR of rank k - 1, its
// right-hand side tL of rank k + tR of rank k - 1; the block (Z_k+1, Z_k) is RL of rank k, whose interior lies between the two.
// Lower band, LDs = 2 * wm doubles per column (a block-tridiagonal matrix with blocks <= wm wide has bandwidth < 2 wm): element
// (r, c), r >= c, at Ssep[c * LDs + (r - c)] -- the layout band_chol_*_kernel factorises.
__global__ __launch_bounds__(256) void ba_sep_assemble_kernel(const double* msgs, size_t msg_doubles, int wm, int R, const int* sep_off, int n, int LDs, double* Ssep, double* rsep) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)n * (LDs + 1)) return;
  const int c = (int)(e / (LDs + 1)), dd = (int)(e % (LDs + 1));
  if (dd == LDs) {   // right-hand side of row c: tL of rank k + tR of rank k - 1
    int k = 1;
    while (k + 1 < R && sep_off[k + 1] <= c) k++;
    const int l = c - sep_off[k];
    rsep[c] = msgs[(size_t)k * msg_doubles + 3 * (size_t)wm * wm + l] + msgs[(size_t)(k - 1) * msg_doubles + 3 * (size_t)wm * wm + wm + l];
    return;
  }
  const int r = c + dd;
  double v = 0.0;
  if (r < n) {
    int kr = 1, kc = 1;
    while (kr + 1 < R && sep_off[kr + 1] <= r) kr++;
    while (kc + 1 < R && sep_off[kc + 1] <= c) kc++;
    const int lr = r - sep_off[kr], lc = c - sep_off[kc];
    if (kr == kc) v = msgs[(size_t)kr * msg_doubles + (size_t)lr * wm + lc] + msgs[(size_t)(kr - 1) * msg_doubles + 2 * (size_t)wm * wm + (size_t)lr * wm + lc];
    else if (kr == kc + 1) v = msgs[(size_t)kc * msg_doubles + (size_t)wm * wm + (size_t)lr * wm + lc];
  }
  Ssep[(size_t)c * LDs + dd] = v;
}
// x of the separators to their places in the solution vector (sep_col[k]: first column of Z_k)
__global__ __launch_bounds__(256) void ba_sep_scatter_kernel(const double* xsep, int n, int R, const int* sep_off, const int* sep_col, double* x) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  int k = 1;
  while (k + 1 < R && sep_off[k + 1] <= r) k++;
  x[sep_col[k] + (r - sep_off[k])] = xsep[r];
}
// y_I -= Y^T x_Z
__global__ __launch_bounds__(256) void ba_sep_correct_kernel(SepView V, double* y) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= V.ni) return;
  double s = 0.0;
  for (int q = 0; q < V.wl; q++) s = fma(V.Y[(size_t)q * V.ni + j], V.rhs[V.zl + q], s);
  for (int q = 0; q < V.wr; q++) s = fma(V.Y[(size_t)(V.wl + q) * V.ni + j], V.rhs[V.zr + q], s);
  y[V.ci + j] -= s;
}
// status words of a trial -> one double (0 = fine), so that the flag can ride in the scalar all-reduce of the trial
__global__ void ba_fail_flag_kernel(const int* a, const int* b, const int* c, double* out) {
  if (threadIdx.x == 0) *out = ((a && *a) || (b && *b) || (c && *c)) ? 1.0 : 0.0;
}

void ba_launch_sep_reduce(const double* S, int LD, const double* Linv, int ci, int ni, int zl, int wl, int zr, int wr, double* Y, const double* rhs, double* msg, int wm, hipStream_t st) {
  const SepView V{S, LD, Linv, ci, ni, zl, wl, zr, wr, Y, rhs};
  const int nq = wl + wr;
  if (nq <= 0) return;
  hipLaunchKernelGGL(ba_sep_trsm_kernel, dim3((nq + SEP_RW - 1) / SEP_RW), dim3(256), 0, st, V);
  const int nt = (nq + 15) / 16;
  hipLaunchKernelGGL(ba_sep_schur_kernel, dim3(nt * (nt + 1) / 2), dim3(256), 0, st, V, msg, wm);
  hipLaunchKernelGGL(ba_sep_rhs_kernel, dim3(nq), dim3(64), 0, st, V, msg, wm);
}
void ba_launch_sep_assemble(const double* msgs, size_t msg_doubles, int wm, int R, const int* sep_off, int n, int LDs, double* Ssep, double* rsep, hipStream_t st) {
  const long long total = (long long)n * (LDs + 1);
  if (total > 0) hipLaunchKernelGGL(ba_sep_assemble_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, msgs, msg_doubles, wm, R, sep_off, n, LDs, Ssep, rsep);
}
void ba_launch_sep_scatter(const double* xsep, int n, int R, const int* sep_off, const int* sep_col, double* x, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL(ba_sep_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, st, xsep, n, R, sep_off, sep_col, x);
}
// x_I = L^-T (y - Y^T x_Z): the correction, then the interior's back-substitution (one-sided order: one workgroup)
void ba_launch_sep_backsolve(double* S, int LD, double* work, int ci, int ni, int zl, int wl, int zr, int wr, double* Y, double* rhs, int* info, hipStream_t st) {
  if (ni <= 0) return;
  const SepView V{S, LD, work, ci, ni, zl, wl, zr, wr, Y, rhs};
  if (wl + wr > 0) hipLaunchKernelGGL(ba_sep_correct_kernel, dim3((ni + 255) / 256), dim3(256), 0, st, V, rhs);
  int K1 = 0, K2 = 0;
  ba_band_split(ni, LD, &K1, &K2, true);
  const int bw = LD - 1;
  double* Sb = S + (size_t)ci * LD;
  const BandView fwd{Sb, 1, (long long)bw, rhs + ci, 1};
  const BandView rev{Sb + (size_t)(ni - 1) * LD, -(long long)bw, -1, rhs + ci + (ni - 1), -1};
  BandSolve Q;
  Q.h[0] = BandHalf{fwd, rev, ni, K1, K2, 0, work, work + band_blocks(ni), nullptr}; Q.h[1] = Q.h[0]; Q.bw = bw; Q.zero = reinterpret_cast<const double*>(info + 4);
  hipLaunchKernelGGL(band_backsolve_kernel, dim3(1), dim3(256), 0, st, Q);
}
void ba_launch_fail_flag(const int* a, const int* b, const int* c, double* out, hipStream_t st) {
  hipLaunchKernelGGL(ba_fail_flag_kernel, dim3(1), dim3(64), 0, st, a, b, c, out);
}

// ---------------------------------------------------------------------------------------- launchers --
// dst[r][0..width) = src[idx[r]][0..width): the structure phase's edge tables are permuted on the device (point-major order from
// the caller's order, camera-major from point-major) instead of on the host + a second upload
__global__ __launch_bounds__(256) void ba_gather_rows_kernel(const double* __restrict__ src, const int* __restrict__ idx, long long total, int width, double* __restrict__ dst) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const long long r = t / width;
  const int c = (int)(t - r * width);
  dst[t] = src[(long long)idx[r] * width + c];
}
// dst[r][0..4) = the same four doubles for every row: the per-edge information / intrinsics records of a graph whose edges all shared one
// record, written out when an appended edge brings another (cs_ba_append_edges_proj) -- such a graph never uploads the records themselves
__global__ __launch_bounds__(256) void ba_fill_rows4_kernel(double* __restrict__ dst, double a, double b, double c, double d, long long total) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int q = (int)(t & 3);
  dst[t] = q == 0 ? a : (q == 1 ? b : (q == 2 ? c : d));
}
void ba_launch_fill_rows4(double* dst, const double* rec4, long long rows, hipStream_t st) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(ba_fill_rows4_kernel, dim3((unsigned)((4 * rows + 255) / 256)), dim3(256), 0, st, dst, rec4[0], rec4[1], rec4[2], rec4[3], 4 * rows);
}
void ba_launch_gather_rows(const double* src, const int* idx, int n, int width, double* dst, hipStream_t st) {
  const long long total = (long long)n * width;
  if (total <= 0) return;
  hipLaunchKernelGGL(ba_gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, idx, total, width, dst);
}

int ba_chi2_blocks(int n_proj) { int nb = (n_proj + 255) / 256; return nb < 1 ? 1 : (nb > 2048 ? 2048 : nb); }

void ba_launch_chi2(const BaView& v, int nb_proj, hipStream_t st) {
  const int ne = v.n_cub + v.n_odom, nb_pose = (ne + 63) / 64;
  hipLaunchKernelGGL(ba_chi2_kernel, dim3(nb_proj + (nb_pose + 3) / 4), dim3(256), 0, st, v, nb_proj, nb_pose);
}
// the numeric-Jacobian edges (cuboid, odometry: few edges, long dependent chains) run beside the projection edges (many edges, short
// chains) on a second stream; ba_accum_pose_kernel needs both
// Three streams: the camera kernel is one workgroup per camera (1000 at C4: a fraction of the CUs), so the landmark kernel runs
// beside it on st3 and the numeric-Jacobian edges (cuboid, odometry) on st2; all meet before the per-vertex accumulation.
// ev_pre (optional): an event ALREADY recorded on st at the point from which the state no longer changes -- the side streams then start from
// there instead of from this call (a speculated linearisation: the numeric-Jacobian edges run beside the chi2 kernels of the trial before it)
void ba_launch_linearize(const BaView& v, hipStream_t st, hipStream_t st2, hipEvent_t ev_fork, hipEvent_t ev_join, hipStream_t st3, hipEvent_t ev_join3, hipEvent_t ev_pre) {
  const bool side = st2 != nullptr && v.n_cub > 0;
  const bool lin_pt = v.np > 0 && !v.fuse_lin;       // (fuse_lin: the Schur kernels of this iteration's trials linearise the landmark side themselves)
  const bool side3 = st3 != nullptr && ev_join3 != nullptr && lin_pt && (v.n_proj > 0 || v.nc > 0);
  hipStream_t se = side ? st2 : st;
  if ((side || side3) && !ev_pre) (void)hipEventRecord(ev_fork, st);
  if (ev_pre) ev_fork = ev_pre;
  if (side) (void)hipStreamWaitEvent(st2, ev_fork, 0);
  static const bool odom_own_launch = getenv("CS_BA_ODOM_OWN_LAUNCH") != nullptr;      // (the former form: A / B)
  const int ob = (!odom_own_launch && v.n_cub3 > 0 && v.n_odom > 0) ? (v.n_odom + 7) / 8 : 0;      // the odometry edges as the cuboid launch's first workgroups
  if (v.n_cub3 > 0) hipLaunchKernelGGL(ba_cub_edge_kernel<true>, dim3((v.n_cub3 + 3) / 4 + ob), dim3(128), 0, se, v, ob);
  if (v.n_cub > v.n_cub3) hipLaunchKernelGGL(ba_cub_edge_kernel<false>, dim3((v.n_cub - v.n_cub3 + 3) / 4), dim3(128), 0, se, v, 0);
  // (the odometry edges: a single short wave per four edges, 38 us of latency -- beside the cuboid edges on the main stream, not behind
  // them on the side stream, whose 135 us are the phase's critical path)
  // (the odometry edges behind the cuboid edges on the side stream: ~250 short wavefronts, 35 us; in front of the camera sums on the main
  // stream -- rounds 3-4 -- they waited for wave slots beside the cuboid edges for 112 us, behind them they lengthen the main stream's chain)
  if (v.n_odom > 0 && ob == 0) hipLaunchKernelGGL(ba_odom_edge_kernel, dim3((v.n_odom + 3) / 4), dim3(64), 0, se, v);
  if (side) (void)hipEventRecord(ev_join, st2);
  if (side3) {
    (void)hipStreamWaitEvent(st3, ev_fork, 0);
    hipLaunchKernelGGL(ba_lin_pt_kernel, dim3((v.np + LIN_PT_GROUP - 1) / LIN_PT_GROUP), dim3(256), 0, st3, v);
    (void)hipEventRecord(ev_join3, st3);
  }
  if (v.n_proj > 0 || v.nc > 0) hipLaunchKernelGGL(ba_lin_cam_kernel, dim3(v.nc), dim3(256), 0, st, v);
  if (!side3 && lin_pt) hipLaunchKernelGGL(ba_lin_pt_kernel, dim3((v.np + LIN_PT_GROUP - 1) / LIN_PT_GROUP), dim3(256), 0, st, v);
  if (side) (void)hipStreamWaitEvent(st, ev_join, 0);
  if (side3) (void)hipStreamWaitEvent(st, ev_join3, 0);
  hipLaunchKernelGGL(ba_accum_pose_kernel, dim3(v.nc + v.no), dim3(128), 0, st, v, 0);
}
void ba_launch_trial_prologue(double* d_lam, double lam0, double lam1, int* info24, int* elim_fail, double* S, size_t n_clear, hipStream_t st);
void ba_launch_reduce(const BaView& v, const double* lambda, hipStream_t st, hipStream_t st2, hipEvent_t ev_fork, hipEvent_t ev_join, const BaSidePrologue* sp) {   // lambda: device, [lambda, lambda of the pose diagonals in the scale term]
  // sp (optional): the trial's prologue has NOT been launched yet -- it goes to the side stream when there is one and the landmark segments can take
  // lambda by value (the fused-linearisation kernels, no long-track segments), else in front of everything on the main stream
  const bool side0 = v.fused && v.elim && v.no > 0 && st2 != nullptr;
  const bool sp_side = sp && side0 && v.fuse_lin && v.n_seg <= max(max(max(v.seg_class[0], v.seg_class[1]), max(v.seg_class[2], v.seg_class[3])), v.seg_class[4]);
  if (sp && !sp_side) ba_launch_trial_prologue(sp->d_lam, sp->lam0, sp->lam1, sp->info24, sp->elim_fail, sp->S, sp->n_clear, st);
  const double* lin_lamp = sp_side ? nullptr : lambda;
  const double lin_lam = sp_side ? sp->lam0 : 0.0;
  if (v.fused) {
    // the cuboid elimination (one workgroup per cuboid, latency-bound) runs beside the landmark segments on a second stream; both
    // write disjoint ranges of the partial arrays and meet before the destination schedule reads them
    const bool side = v.elim && v.no > 0 && st2 != nullptr;
    if (side) {
      (void)hipEventRecord(ev_fork, st);
      (void)hipStreamWaitEvent(st2, ev_fork, 0);
      if (sp_side) ba_launch_trial_prologue(sp->d_lam, sp->lam0, sp->lam1, sp->info24, sp->elim_fail, sp->S, sp->n_clear, st2);
      hipLaunchKernelGGL(ba_cub_elim_kernel, dim3(v.no), dim3(256), 2 * 54 * sizeof(double) * (size_t)v.elim_max_slots, st2, v, lambda);
    }
    // (the cuboid elimination must be dispatched BEFORE the bulk of the segments fills the device: started 18 us ahead -- behind the short
    // kernel of the one- and two-camera segments below -- its 500 workgroups run 90 us beside the bulk; started together with it they
    // queue for slots and take 200 us, whatever the stream priority.  Measured in round 5 by moving that short kernel to this stream.)
    if (side) (void)hipEventRecord(ev_join, st2);
    // the diagonal blocks / right-hand side of the cameras need the segments' partial vectors; the gather of the blocks runs last
    // (it subtracts from entries the vertex kernels have written)
    {
      // (class boundaries are cumulative: a class without segments repeats the boundary before it)
      int c0 = v.seg_class[0], c1 = max(c0, v.seg_class[1]), c2 = max(c1, v.seg_class[2]), c3 = max(c2, v.seg_class[3]);
      const int c4 = max(c3, v.seg_class[4]);
      if (v.fuse_lin) {
        // (round 6: the one- and two-camera segments ride in the <2> launch -- the same products on the same operands, their second tile idle --
        // instead of a 17 us launch of their own in front of it: the two kernels never overlapped; C4 1 312-1 324 -> 1 336-1 342 LM it/s)
        if (c1 > c0) c0 = 0;
        if (c0 > 0) hipLaunchKernelGGL(ba_lin_schur_kernel<1>, dim3((c0 + 3) / 4), dim3(256), 0, st, v, lin_lamp, 0, c0, lin_lam);
        if (c1 > c0) hipLaunchKernelGGL(ba_lin_schur_kernel<2>, dim3((c1 - c0 + 3) / 4), dim3(256), 0, st, v, lin_lamp, c0, c1, lin_lam);
        if (c2 > c1) hipLaunchKernelGGL(ba_lin_schur_kernel<3>, dim3((c2 - c1 + 3) / 4), dim3(256), 0, st, v, lin_lamp, c1, c2, lin_lam);
        if (c3 > c2) hipLaunchKernelGGL(ba_lin_schur_kernel<4>, dim3((c3 - c2 + 3) / 4), dim3(256), 0, st, v, lin_lamp, c2, c3, lin_lam);
        if (c4 > c3) hipLaunchKernelGGL(ba_lin_schur_kernel<5>, dim3((c4 - c3 + 3) / 4), dim3(256), 0, st, v, lin_lamp, c3, c4, lin_lam);
      } else {
        if (c0 > 0) hipLaunchKernelGGL(ba_schur_fused_kernel<1>, dim3((c0 + 3) / 4), dim3(256), 0, st, v, lambda, 0, c0);
        if (c1 > c0) hipLaunchKernelGGL(ba_schur_fused_kernel<2>, dim3((c1 - c0 + 3) / 4), dim3(256), 0, st, v, lambda, c0, c1);
        if (c2 > c1) hipLaunchKernelGGL(ba_schur_fused_kernel<3>, dim3((c2 - c1 + 3) / 4), dim3(256), 0, st, v, lambda, c1, c2);
        if (c3 > c2) hipLaunchKernelGGL(ba_schur_fused_kernel<4>, dim3((c3 - c2 + 3) / 4), dim3(256), 0, st, v, lambda, c2, c3);
        if (c4 > c3) hipLaunchKernelGGL(ba_schur_fused_kernel<5>, dim3((c4 - c3 + 3) / 4), dim3(256), 0, st, v, lambda, c3, c4);
      }
      if (v.n_seg > c4) hipLaunchKernelGGL(ba_schur_long_kernel, dim3((v.n_seg - c4 + 3) / 4), dim3(256), 0, st, v, lambda, c4, v.n_seg);
    }
    if (side) (void)hipStreamWaitEvent(st, ev_join, 0);
    else if (v.elim && v.no > 0) hipLaunchKernelGGL(ba_cub_elim_kernel, dim3(v.no), dim3(256), 2 * 54 * sizeof(double) * (size_t)v.elim_max_slots, st, v, lambda);
    hipLaunchKernelGGL(ba_cam_rhs_offdiag_kernel, dim3(v.nc + v.n_cub + v.n_odom), dim3(64), 0, st, v, lambda);
    if (!v.elim && v.no > 0) hipLaunchKernelGGL(ba_cub_scatter_kernel, dim3(v.no), dim3(128), 0, st, v, lambda);
    if (v.n_gpairs > 0) hipLaunchKernelGGL(ba_schur_gather_kernel, dim3((v.n_gpairs + 3) / 4), dim3(256), 0, st, v);
    return;
  }
  if (v.np > 0) hipLaunchKernelGGL(ba_prep_kernel, dim3((v.np + 255) / 256), dim3(256), 0, st, v, lambda);
  if (v.n_proj > 0) hipLaunchKernelGGL(ba_wd_kernel, dim3((unsigned)((6 * (size_t)v.n_proj + 255) / 256)), dim3(256), 0, st, v);
  hipLaunchKernelGGL(ba_cam_rhs_kernel, dim3(v.nc), dim3(256), 0, st, v, lambda);
  if (v.no > 0) hipLaunchKernelGGL(ba_cub_scatter_kernel, dim3(v.no), dim3(128), 0, st, v, lambda);
  if (v.n_cub + v.n_odom > 0) hipLaunchKernelGGL(ba_offdiag_kernel, dim3(v.n_cub + v.n_odom), dim3(64), 0, st, v);
  if (v.n_pairs > 0) hipLaunchKernelGGL(ba_schur_kernel, dim3((v.n_pairs + 1) / 2), dim3(128), 0, st, v);
}
// ---- debug: NaN / Inf scan (cs_ba_check_finite, CS_BA_DEBUG_NAN) -------------------------------------------------------------------
// g2o's debug builds look for NaNs where they are born: in the errors (SparseOptimizer::computeActiveErrors, sparse_optimizer.cpp:78-86)
// and in the Jacobians of every edge (BlockSolver::buildSystem, block_solver.hpp:533-544).  Here the errors and Jacobians never reach
// memory, so the scan covers what they turn into -- the edges' squared errors, every block of the linear system, the increments and
// the estimates -- and names the first offending entry of each array.  out[2 t] = number of non-finite entries of array t,
// out[2 t + 1] = smallest offending index + 1.
__global__ __launch_bounds__(256) void ba_scan_finite_kernel(const double* p, long long n, int* out) {
  int bad = 0;
  long long first = -1;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double x = p[i];
    if (!(x - x == 0.0)) { bad++; if (first < 0) first = i; }     // NaN and +-Inf
  }
  if (bad) { atomicAdd(out, bad); atomicMin((unsigned*)(out + 1), (unsigned)min(first + 1, 0x7ffffffeLL)); }
}
void ba_launch_scan_finite(const double* p, long long n, int* out, hipStream_t st) {
  if (n <= 0 || !p) return;
  const long long nb = (n + 255) / 256;
  hipLaunchKernelGGL(ba_scan_finite_kernel, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, st, p, n, out);
}
// the edges' (un-robustified) squared errors e^T Omega e, one double per edge: projection edges (point-major order), then the
// camera-cuboid and the odometry edges -- what a NaN in an error vector turns into
__global__ __launch_bounds__(256) void ba_edge_chi_kernel(BaView v, double* out) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < v.n_proj) {
    Pose T = pose_load(v.cams + 7 * v.pm_cam[k]);
    double e[2], pc[3];
    proj_error(T, v.points + 3 * v.pm_pt[k], v.pm_uv + 2 * k, v.intr_u ? v.intr_u : v.pm_intr + 4 * k, e, pc);
    const double* info = v.info_u ? v.info_u : v.pm_info + 4 * k;
    out[k] = e[0] * (info[0] * e[0] + info[1] * e[1]) + e[1] * (info[2] * e[0] + info[3] * e[1]);
    return;
  }
  const int q = k - v.n_proj;
  if (q < v.n_cub3) {
    double e[9];
    cuboid_edge_error(pose_load(v.cams + 7 * v.ce_cam[q]), cube_load(v.cubes + 10 * v.ce_cub[q]), cube_load(v.ce_meas + 10 * q), e);
    out[k] = quad_form(e, v.ce_info + 81 * q, 9);
  } else if (q < v.n_cub) {
    const int r = q - v.n_cub3;
    double e[4];
    cuboid_proj_error(pose_load(v.cams + 7 * v.ce_cam[q]), cube_load(v.cubes + 10 * v.ce_cub[q]), v.pe_K + 9 * (size_t)r, v.pe_meas + 4 * (size_t)r, e);
    out[k] = quad_form(e, v.pe_info + 16 * (size_t)r, 4);
  } else if (q < v.n_cub + v.n_odom) {
    const int r = q - v.n_cub;
    double e[6];
    odom_edge_error(pose_load(v.cams + 7 * v.oe_i[r]), pose_load(v.cams + 7 * v.oe_j[r]), pose_load(v.oe_meas + 7 * r), e);
    out[k] = quad_form(e, v.oe_info + 36 * r, 6);
  }
}
void ba_launch_edge_chi(const BaView& v, double* out, hipStream_t st) {
  const int n = v.n_proj + v.n_cub + v.n_odom;
  if (n > 0) hipLaunchKernelGGL(ba_edge_chi_kernel, dim3((n + 255) / 256), dim3(256), 0, st, v, out);
}

// ---- external (host-evaluated) edges: cs_ba_set_external_edges / cs_ba_set_external_terms ---------------------------------------
// Edges of types the library does not evaluate go through their own linearizeOplus + constructQuadraticForm on the host
// (core/base_binary_edge.hpp:54-205, core/base_unary_edge.hpp:42-123); what those accumulate -- A_ii / b_i of the vertices, the
// off-diagonal block of a binary edge -- is added here to the blocks the device built from its own edges.
__global__ __launch_bounds__(256) void ba_ext_add_kernel(BaView v, const double* cam36, const double* cam6, const double* cub81, const double* cub9, const double* pt9, const double* pt3) {
  const long long nCam = (long long)v.nc * 42, nCub = (long long)v.no * 90, nPt = (long long)v.np * 12;
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < nCam) {
    const int c = (int)(t / 42), e = (int)(t % 42);
    if (!cam36 || v.cam_col[c] < 0) return;
    if (e < 36) v.Hcam[36 * (size_t)c + e] += cam36[36 * (size_t)c + e]; else v.bcam[6 * (size_t)c + e - 36] += cam6[6 * (size_t)c + e - 36];
    return;
  }
  t -= nCam;
  if (t < nCub) {
    const int o = (int)(t / 90), e = (int)(t % 90);
    if (!cub81 || v.cub_col[o] < 0) return;
    if (e < 81) v.Hcub[81 * (size_t)o + e] += cub81[81 * (size_t)o + e]; else v.bcub[9 * (size_t)o + e - 81] += cub9[9 * (size_t)o + e - 81];
    return;
  }
  t -= nCub;
  if (t < nPt) {
    const int p = (int)(t / 12), e = (int)(t % 12);
    if (!pt9 || !v.pt_free[p]) return;
    if (e < 9) v.Hll[9 * (size_t)p + e] += pt9[9 * (size_t)p + e]; else v.bl[3 * (size_t)p + e - 9] += pt3[3 * (size_t)p + e - 9];
  }
}
// e4: (class_i, idx_i, class_j, idx_j) per edge, class 0 = camera (6), 1 = cuboid (9); Hij: 81 per edge, row-major dim_i x dim_j.
// A binary edge between two free pose vertices adds its block to the reduced system like an odometry edge's (ba_offdiag_kernel).
// One workgroup per destination pair of vertices (gptr / order: the edges grouped by unordered pair on the host, in the caller's order inside
// a group): its edges are added one after the other with plain read-modify-writes -- several edges on one pair, in either orientation, sum in
// a fixed order, like every other sum of the reduced system (no atomics: a trial's S does not depend on scheduling).
__global__ __launch_bounds__(128) void ba_ext_offdiag_kernel(BaView v, int n_groups, const int* gptr, const int* order, const int* e4, const double* Hij) {
  const int g = blockIdx.x, t = threadIdx.x;
  if (g >= n_groups) return;
  for (int q = gptr[g]; q < gptr[g + 1]; q++) {
    const int k = order[q];
    const int ci = e4[4 * k], ii = e4[4 * k + 1], cj = e4[4 * k + 2], ij = e4[4 * k + 3];
    if (ij < 0 || ci > 1 || cj > 1) continue;               // unary edge: diagonal terms only   (uniform over the workgroup)
    const int ca = ci ? v.cub_col[ii] : v.cam_col[ii], cb = cj ? v.cub_col[ij] : v.cam_col[ij];
    const int di = ci ? 9 : 6, dj = cj ? 9 : 6;
    if (ca < 0 || cb < 0 || ca >= v.n_red || cb >= v.n_red) continue;
    if (t < di * dj) {
      const int i = t / dj, j = t % dj;
      double* dst = cb > ca ? ba_S_at(v, cb + j, ca + i) : ba_S_at(v, ca + i, cb + j);
      *dst += Hij[81 * (size_t)k + t];
    }
    __syncthreads();     // (the next edge of the pair may be stored the other way round: another thread's element)
  }
}
void ba_launch_ext_add(const BaView& v, const double* cam36, const double* cam6, const double* cub81, const double* cub9, const double* pt9, const double* pt3, hipStream_t st) {
  const long long total = (long long)v.nc * 42 + (long long)v.no * 90 + (long long)v.np * 12;
  if (total > 0) hipLaunchKernelGGL(ba_ext_add_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, v, cam36, cam6, cub81, cub9, pt9, pt3);
}
void ba_launch_ext_offdiag(const BaView& v, int n_groups, const int* gptr, const int* order, const int* e4, const double* Hij, hipStream_t st) {
  if (n_groups > 0) hipLaunchKernelGGL(ba_ext_offdiag_kernel, dim3(n_groups), dim3(128), 0, st, v, n_groups, gptr, order, e4, Hij);
}
int ba_scale_blocks() { return SCALE_BLOCKS; }
// out[0] = sum of a[0 .. na), out[1] = sum of b[0 .. nb): fixed-shape tree (one workgroup), so the value does not depend on
// scheduling; lets chi2 and the LM scale term stay on the device until the one read-back (and RCCL all-reduce) of a trial
__global__ __launch_bounds__(256) void ba_sum2_kernel(const double* a, int na, const double* b, int nb, double* out) {
  __shared__ double ws[2][4];
  double sa = 0, sb = 0;
  for (int i = threadIdx.x; i < na; i += 256) sa += a[i];
  for (int i = threadIdx.x; i < nb; i += 256) sb += b[i];
  sa = wave_sum(sa); sb = wave_sum(sb);
  if ((threadIdx.x & 63) == 0) { ws[0][threadIdx.x >> 6] = sa; ws[1][threadIdx.x >> 6] = sb; }
  __syncthreads();
  if (threadIdx.x == 0) { out[0] = (ws[0][0] + ws[0][1]) + (ws[0][2] + ws[0][3]); out[1] = (ws[1][0] + ws[1][1]) + (ws[1][2] + ws[1][3]); }
}
void ba_launch_sum2(const double* a, int na, const double* b, int nb, double* out, hipStream_t st) {
  hipLaunchKernelGGL(ba_sum2_kernel, dim3(1), dim3(256), 0, st, a, na, b, nb, out);
}
// the same with the trial's failure flag in out[2] (ba_fail_flag_kernel's job, one launch fewer at the end of a trial)
// host_out (optional): the pinned mirror of the trial's scalars -- the three values and, behind a system-scope fence, the trial's sequence
// number, which the host polls: no copy command, no event, no signal between the last kernel of a trial and the host's decision
__global__ __launch_bounds__(256) void ba_sum2_flag_kernel(const double* a, int na, const double* b, int nb, const int* f0, const int* f1, double* out, double* host_out, double seq) {
  __shared__ double ws[2][4];
  double sa = 0, sb = 0;
  for (int i = threadIdx.x; i < na; i += 256) sa += a[i];
  for (int i = threadIdx.x; i < nb; i += 256) sb += b[i];
  sa = wave_sum(sa); sb = wave_sum(sb);
  if ((threadIdx.x & 63) == 0) { ws[0][threadIdx.x >> 6] = sa; ws[1][threadIdx.x >> 6] = sb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = (ws[0][0] + ws[0][1]) + (ws[0][2] + ws[0][3]); out[1] = (ws[1][0] + ws[1][1]) + (ws[1][2] + ws[1][3]);
    out[2] = ((f0 && *f0) || (f1 && *f1)) ? 1.0 : 0.0;
    if (host_out) {
      host_out[0] = out[0]; host_out[1] = out[1]; host_out[2] = out[2];
      __threadfence_system();
      __hip_atomic_store(host_out + 3, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
void ba_launch_sum2_flag(const double* a, int na, const double* b, int nb, const int* f0, const int* f1, double* out, hipStream_t st, double* host_out, double seq) {
  hipLaunchKernelGGL(ba_sum2_flag_kernel, dim3(1), dim3(256), 0, st, a, na, b, nb, f0, f1, out, host_out, seq);
}
// computeLambdaInit (optimization_algorithm_levenberg.cpp:166-180): max |H_jj| over every non-fixed vertex -- cameras, cuboids, landmarks -- on the
// device (the maximum is exact whatever the order; non-negative doubles order like their bit patterns).  out[0] must be zero on entry.
__global__ __launch_bounds__(256) void ba_max_diag_kernel(BaView v, unsigned long long* out) {
  const long long nCam = 6ll * v.nc, nCub = 9ll * v.no, nPt = 3ll * v.np;
  double m = 0.0;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < nCam + nCub + nPt; t += (long long)gridDim.x * 256) {
    double x = 0.0;
    if (t < nCam) { const int i = (int)(t / 6), d = (int)(t % 6); if (v.cam_col[i] >= 0) x = v.Hcam[36 * (size_t)i + 7 * d]; }
    else if (t < nCam + nCub) { const long long q = t - nCam; const int i = (int)(q / 9), d = (int)(q % 9); if (v.cub_col[i] >= 0) x = v.Hcub[81 * (size_t)i + 10 * d]; }
    else { const long long q = t - nCam - nCub; const int i = (int)(q / 3), d = (int)(q % 3); if (v.pt_free[i]) x = v.Hll[9 * (size_t)i + 4 * d]; }
    x = fabs(x);
    if (x > m) m = x;      // (a NaN never wins, as in the reference's std::max(fabs(x), m) chain once a later entry follows it)
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const double y = __shfl_xor(m, o); if (y > m) m = y; }
  __shared__ double wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {      // one atomic per workgroup (4096 wavefronts on one address took 50 us)
    const double a = wm[0] > wm[1] ? wm[0] : wm[1], b = wm[2] > wm[3] ? wm[2] : wm[3], mm = a > b ? a : b;
    if (mm > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(mm));
  }
}
void ba_launch_max_diag(const BaView& v, double* out, hipStream_t st) {
  const long long n = 6ll * v.nc + 9ll * v.no + 3ll * v.np;
  if (n <= 0) return;
  const long long nb = (n + 255) / 256;
  hipLaunchKernelGGL(ba_max_diag_kernel, dim3((unsigned)(nb > 512 ? 512 : nb)), dim3(256), 0, st, v, reinterpret_cast<unsigned long long*>(out));
}
// up to 48 buffers zeroed by one launch (the structure phase's allocations): blockIdx.y = buffer, 4-byte words
struct BaZeroList { unsigned* p[48]; unsigned long long words[48]; };
__global__ __launch_bounds__(256) void ba_multi_zero_kernel(BaZeroList L) {
  unsigned* __restrict__ p = L.p[blockIdx.y];
  const unsigned long long n = L.words[blockIdx.y];
  // 16 bytes per thread and step where the buffer allows it (hipMalloc aligns to 256 bytes)
  const unsigned long long n4 = n / 4;
  uint4* __restrict__ p4 = reinterpret_cast<uint4*>(p);
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (unsigned long long)gridDim.x * 256) p4[i] = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3ull)) p[4 * n4 + threadIdx.x] = 0u;
}
void ba_launch_multi_zero(const std::pair<void*, size_t>* list, int n, hipStream_t st) {
  if (n <= 0) return;
  BaZeroList L;
  unsigned long long mx = 0;
  for (int i = 0; i < 48; i++) {
    const int q = i < n ? i : 0;
    L.p[i] = static_cast<unsigned*>(list[q].first);
    L.words[i] = i < n ? (list[q].second + 3) / 4 : 0;          // (every buffer's size is a multiple of 4 bytes: int and double elements)
    if (L.words[i] > mx) mx = L.words[i];
  }
  unsigned gx = (unsigned)std::min<unsigned long long>(512, (mx / 4 + 255) / 256 + 1);
  hipLaunchKernelGGL(ba_multi_zero_kernel, dim3(gx, n), dim3(256), 0, st, L);
}
// up to 48 small uploads by one launch: the sources sit in the structure phase's pinned staging arena (host memory the device reads over
// the link), the destinations are the handle's tables.  One hipMemcpyAsync per table was ~80 copies of a few KB in an appended frame's
// structure phase: 5 us of host time and a blit of its own on the stream apiece.  blockIdx.y = table, 4-byte words, 16-byte steps.
struct BaCopyList { unsigned* dst[48]; const unsigned* src[48]; unsigned long long words[48]; };
__global__ __launch_bounds__(256) void ba_multi_copy_kernel(BaCopyList L) {
  unsigned* __restrict__ d = L.dst[blockIdx.y];
  const unsigned* __restrict__ s = L.src[blockIdx.y];
  const unsigned long long n = L.words[blockIdx.y], n4 = n / 4;
  uint4* __restrict__ d4 = reinterpret_cast<uint4*>(d);
  const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(s);      // (arena offsets are multiples of 64 bytes, hipMalloc aligns to 256)
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (unsigned long long)gridDim.x * 256) d4[i] = s4[i];
  if (blockIdx.x == 0 && threadIdx.x < (n & 3ull)) d[4 * n4 + threadIdx.x] = s[4 * n4 + threadIdx.x];
}
void ba_launch_multi_copy(const BaCopyItem* list, int n, hipStream_t st) {
  if (n <= 0) return;
  BaCopyList L;
  unsigned long long mx = 0;
  for (int i = 0; i < 48; i++) {
    const int q = i < n ? i : 0;
    L.dst[i] = static_cast<unsigned*>(list[q].dst); L.src[i] = static_cast<const unsigned*>(list[q].src);
    L.words[i] = i < n ? list[q].bytes / 4 : 0;                   // (int and double elements: a multiple of 4 bytes)
    if (L.words[i] > mx) mx = L.words[i];
  }
  unsigned gx = (unsigned)std::min<unsigned long long>(64, (mx / 4 + 255) / 256 + 1);
  hipLaunchKernelGGL(ba_multi_copy_kernel, dim3(gx, n), dim3(256), 0, st, L);
}
// Head of a trial on the banded path: lambda into device memory (the kernels read it from there), the factorisation's status words,
// the cuboid elimination's failure word and the reduced system's right-hand side cleared -- one launch instead of a copy and three fills.
__global__ __launch_bounds__(256) void ba_trial_prologue_kernel(double* d_lam, double lam0, double lam1, int* info24, int* elim_fail, double* S, size_t n_clear) {
  // [S | rhs] cleared (16 bytes per thread and step: hipMalloc aligns to 256 bytes), the status words, lambda
  double2* __restrict__ S2 = reinterpret_cast<double2*>(S);
  const size_t n2 = n_clear / 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) S2[i] = make_double2(0.0, 0.0);
  if (blockIdx.x == 0) {
    if ((n_clear & 1) && threadIdx.x == 27) S[n_clear - 1] = 0.0;
    if (threadIdx.x < 24) info24[threadIdx.x] = 0;
    if (threadIdx.x == 24) *elim_fail = 0;
    if (threadIdx.x == 25) d_lam[0] = lam0;
    if (threadIdx.x == 26) d_lam[1] = lam1;
  }
}
// S: the reduced system's buffer, n_clear = its doubles + the right-hand side's behind them (one buffer)
void ba_launch_trial_prologue(double* d_lam, double lam0, double lam1, int* info24, int* elim_fail, double* S, size_t n_clear, hipStream_t st) {
  const unsigned blocks = (unsigned)std::min<size_t>(1024, (n_clear / 2 + 255) / 256 + 1);
  hipLaunchKernelGGL(ba_trial_prologue_kernel, dim3(blocks), dim3(256), 0, st, d_lam, lam0, lam1, info24, elim_fail, S, n_clear);
}
void ba_launch_scale(const BaView& v, const double* lambda, double* partial, hipStream_t st) {
  hipLaunchKernelGGL(ba_scale_kernel, dim3(SCALE_BLOCKS), dim3(256), 0, st, v, lambda, partial);
}
void ba_launch_backsub(const BaView& v, hipStream_t st) {
  const int ncb = (v.elim && v.no > 0) ? (v.no + 3) / 4 : 0, npb = (v.np + 255) / 256;
  if (ncb + npb <= 0) return;
  if (v.fuse_lin) hipLaunchKernelGGL(ba_backsub_all_kernel<true>, dim3(ncb + npb), dim3(256), 0, st, v, ncb);      // (the trial's Schur kernels did not write H_pl)
  else hipLaunchKernelGGL(ba_backsub_all_kernel<false>, dim3(ncb + npb), dim3(256), 0, st, v, ncb);
}
void ba_launch_scale_update(const BaView& v, const double* lambda, double* partial, hipStream_t st, double* bak_cams, double* bak_points, double* bak_cubes) {
  const int n = v.np + v.nc + v.no;
  hipLaunchKernelGGL(ba_scale_update_kernel, dim3(SCALE_BLOCKS + (n + 255) / 256), dim3(256), 0, st, v, lambda, partial, bak_cams, bak_points, bak_cubes);
}
void ba_launch_update(const BaView& v, hipStream_t st, double* bak_cams, double* bak_points, double* bak_cubes) {
  int n = v.np + v.nc + v.no;
  if (n > 0) hipLaunchKernelGGL(ba_update_kernel, dim3((n + 255) / 256), dim3(256), 0, st, v, bak_cams, bak_points, bak_cubes);
}

}  // namespace cs
