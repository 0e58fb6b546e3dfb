// oracle/ba_oracle.cpp -- CPU restatement of the reference's g2o bundle-adjustment iteration.
//
// TEST INFRASTRUCTURE ONLY (see oracle/detect_oracle.cpp for the rule).
//
// PARITY STATUS: *parity unpinned* at the bit level; pinned to millimetres by the reference's saved run (below).  The
// reference's g2o (object_slam/Thirdparty/g2o) needs Eigen and cannot be built here; it has no tests.  This file follows, function by function (paths relative to
// /root/reference/object_slam):
//   SE3Quat                 Thirdparty/g2o/g2o/types/se3quat.h:41-362, se3_ops.hpp:28-48
//   VertexSE3Expmap         Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:59-77   (oplus: exp(d) * T)
//   VertexSBAPointXYZ       Thirdparty/g2o/g2o/types/types_sba.h:40-57
//   g2o::cuboid, VertexCuboid, EdgeSE3Cuboid   include/object_slam/g2o_Object.h:23-259
//   EdgeSE3Expmap           Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:83-99    (numeric Jacobian)
//   EdgeSE3ProjectXYZ       types_six_dof_expmap.h:145-174, types_six_dof_expmap.cpp:148-192 (analytic)
//   numeric linearizeOplus  Thirdparty/g2o/g2o/core/base_binary_edge.hpp:130-205  (delta = 1e-9, central)
//   constructQuadraticForm  Thirdparty/g2o/g2o/core/base_binary_edge.hpp:54-120, Huber robust_kernel_impl.cpp:78-91
//   BlockSolver buildSystem/setLambda/solve(Schur)   Thirdparty/g2o/g2o/core/block_solver.hpp:353-604
//   Levenberg-Marquardt     Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-189
//   optimize loop           Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:354-419
// It is pinned by (a) tests/test_reference_frames.py: the reference's own ONLINE run over its 58 bundled TUM frames
// (images -> detections -> growing graph -> LM, driver restated from src/main_obj.cpp:479-841) -- this file, fed by the
// detector restatement, reproduces the object poses the reference saved (output_obj_poses.txt) to a millimetre in every
// frame and its camera trajectory (output_cam_poses.txt) to 3 cm in the mean (the cameras follow the single detections,
// which come from a different segment detector); (b) tests/test_ba_oracle.py: the offline sequence (object_slam/data/*.txt)
// lands inside the envelope of the same saved outputs; (c) an independent numpy/scipy restatement of the residuals and
// Jacobians.
// Eigen-internal arithmetic (quaternion product, Quaterniond(R), toRotationMatrix, 3x3 inverse, LDLT) is
// restated from the published algorithms; the block containers are replaced by equivalent flat arrays
// (dense Hpp, per-landmark 3x3 Hll, one 6x3 Hpl block per projection edge); the dense LDLT has no pivoting.
//
// Vertex order (hessian index): non-marginalised vertices sorted by id, then marginalised ones
// (sparse_optimizer.cpp:166-190).  Ids: see ba_set_vertices().

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

namespace {

typedef double V3[3];

struct Quat { double w, x, y, z; };
struct SE3 { Quat r; double t[3]; };

inline void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
// Eigen quaternion product (generic path)
inline Quat qmul(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
// Eigen QuaternionBase::_transformVector
inline void qrot(const Quat& q, const double v[3], double o[3]) {
  double qv[3] = {q.x, q.y, q.z}, uv[3], c2[3];
  cross3(qv, v, uv);
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  cross3(qv, uv, c2);
  for (int i = 0; i < 3; i++) o[i] = v[i] + q.w * uv[i] + c2[i];
}
inline void qnormalize(Quat& q) {
  double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}
// se3quat.h:346-351
inline void normalizeRotation(SE3& T) {
  if (T.r.w < 0) { T.r.x *= -1; T.r.y *= -1; T.r.z *= -1; T.r.w *= -1; }
  qnormalize(T.r);
}
// Eigen toRotationMatrix
inline void q2R(const Quat& q, double R[9]) {
  double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// Eigen Quaterniond(Matrix3d)
inline Quat R2q(const double R[9]) {
  Quat q;
  double t = R[0] + R[4] + R[8];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (R[7] - R[5]) * t; q.y = (R[2] - R[6]) * t; q.z = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (R[3 * k + j] - R[3 * j + k]) * t;
    v[j] = (R[3 * j + i] + R[3 * i + j]) * t;
    v[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}
inline SE3 se3_identity() { SE3 T; T.r = Quat{1, 0, 0, 0}; T.t[0] = T.t[1] = T.t[2] = 0; return T; }
// se3quat.h:110-116
inline SE3 se3_mul(const SE3& a, const SE3& b) {
  SE3 r = a;
  double rt[3];
  qrot(a.r, b.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] += rt[i];
  r.r = qmul(a.r, b.r);
  normalizeRotation(r);
  return r;
}
// se3quat.h:129-134
inline SE3 se3_inv(const SE3& a) {
  SE3 r;
  r.r = Quat{a.r.w, -a.r.x, -a.r.y, -a.r.z};
  double nt[3] = {a.t[0] * -1., a.t[1] * -1., a.t[2] * -1.};
  qrot(r.r, nt, r.t);
  return r;
}
inline void se3_map(const SE3& T, const double p[3], double o[3]) {
  double rp[3];
  qrot(T.r, p, rp);
  for (int i = 0; i < 3; i++) o[i] = rp[i] + T.t[i];
}
inline void skew(const double v[3], double m[9]) {
  m[0] = 0; m[1] = -v[2]; m[2] = v[1];
  m[3] = v[2]; m[4] = 0; m[5] = -v[0];
  m[6] = -v[1]; m[7] = v[0]; m[8] = 0;
}
inline void mm3(const double a[9], const double b[9], double o[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) o[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
inline void mv3(const double a[9], const double v[3], double o[3]) {
  for (int i = 0; i < 3; i++) o[i] = a[3 * i] * v[0] + a[3 * i + 1] * v[1] + a[3 * i + 2] * v[2];
}
// se3quat.h:230-272
inline void se3_log(const SE3& T, double res[6]) {
  double R[9];
  q2R(T.r, R);
  double d = 0.5 * (R[0] + R[4] + R[8] - 1);
  double omega[3], dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  double Vinv[9], Om[9], Om2[9];
  if (d > 0.99999) {
    for (int i = 0; i < 3; i++) omega[i] = 0.5 * dR[i];
    skew(omega, Om);
    mm3(Om, Om, Om2);
    for (int i = 0; i < 9; i++) Vinv[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + (1. / 12.) * Om2[i];
  } else {
    double theta = std::acos(d);
    double f = theta / (2 * std::sqrt(1 - d * d));
    for (int i = 0; i < 3; i++) omega[i] = f * dR[i];
    skew(omega, Om);
    mm3(Om, Om, Om2);
    double c = (1 - theta / (2 * std::tan(theta / 2))) / (theta * theta);
    for (int i = 0; i < 9; i++) Vinv[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + c * Om2[i];
  }
  double ups[3];
  mv3(Vinv, T.t, ups);
  for (int i = 0; i < 3; i++) { res[i] = omega[i]; res[i + 3] = ups[i]; }
}
// se3quat.h:280-323
inline SE3 se3_exp(const double u[6]) {
  double omega[3] = {u[0], u[1], u[2]}, ups[3] = {u[3], u[4], u[5]};
  double theta = std::sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  double Om[9], Om2[9], R[9], V[9];
  skew(omega, Om);
  mm3(Om, Om, Om2);
  if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) { R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + Om[i] + Om2[i]; V[i] = R[i]; }
  } else {
    double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta), c = (theta - std::sin(theta)) / (std::pow(theta, 3));
    for (int i = 0; i < 9; i++) {
      R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
      V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + b * Om[i] + c * Om2[i];
    }
  }
  SE3 T;
  T.r = R2q(R);
  mv3(V, ups, T.t);
  normalizeRotation(T);
  return T;
}
// fromVector (se3quat.h:166-169): x y z qx qy qz qw; constructors normalise (:68-71)
inline SE3 se3_from7(const double v[7], bool normalize) {
  SE3 T;
  T.r = Quat{v[6], v[3], v[4], v[5]};
  T.t[0] = v[0]; T.t[1] = v[1]; T.t[2] = v[2];
  if (normalize) normalizeRotation(T);
  return T;
}
inline void se3_to7(const SE3& T, double v[7]) {
  v[0] = T.t[0]; v[1] = T.t[1]; v[2] = T.t[2]; v[3] = T.r.x; v[4] = T.r.y; v[5] = T.r.z; v[6] = T.r.w;
}

// g2o::cuboid (g2o_Object.h:23-199)
struct Cuboid { SE3 pose; double scale[3]; };

inline Cuboid cub_exp_update(const Cuboid& c, const double u[9]) {  // :57-63
  Cuboid r;
  r.pose = se3_mul(c.pose, se3_exp(u));
  for (int i = 0; i < 3; i++) r.scale[i] = c.scale[i] + u[6 + i];
  return r;
}
inline void cube_log_error(const Cuboid& self, const Cuboid& newone, double res[9]) {  // :66-73
  SE3 diff = se3_mul(se3_inv(newone.pose), self.pose);
  se3_log(diff, res);
  for (int i = 0; i < 3; i++) res[6 + i] = self.scale[i] - newone.scale[i];
}
inline Cuboid rotate_cuboid(const Cuboid& c, double yaw_angle) {  // :104-114
  Cuboid r;
  SE3 rot;
  rot.r = Quat{std::cos(yaw_angle * 0.5), 0, 0, std::sin(yaw_angle * 0.5)};
  rot.t[0] = rot.t[1] = rot.t[2] = 0;
  normalizeRotation(rot);  // SE3Quat(Quaterniond, Vector3d) constructor
  r.pose = se3_mul(c.pose, rot);
  for (int i = 0; i < 3; i++) r.scale[i] = c.scale[i];
  if ((yaw_angle == M_PI / 2.0) || (yaw_angle == -M_PI / 2.0) || (yaw_angle == 3 * M_PI / 2.0)) std::swap(r.scale[0], r.scale[1]);
  return r;
}
inline void min_log_error(const Cuboid& self, const Cuboid& newone, double res[9]) {  // :76-101
  double norms[4], errs[4][9];
  const double angles[4] = {-1, 0, 1, 2};
  for (int i = 0; i < 4; i++) {
    Cuboid rc = rotate_cuboid(newone, angles[i] * M_PI / 2.0);
    cube_log_error(self, rc, errs[i]);
    double s = 0;
    for (int k = 0; k < 9; k++) s += errs[i][k] * errs[i][k];
    norms[i] = std::sqrt(s);
  }
  int m = 0;  // Eigen minCoeff(&idx): starts at element 0, strict <, so a NaN norm never wins (SURVEY note 14)
  for (int i = 1; i < 4; i++)
    if (norms[i] < norms[m]) m = i;
  std::memcpy(res, errs[m], sizeof(double) * 9);
}

// An edge's robust kernel (OptimizableGraph::Edge::robustKernel(), core/optimizable_graph.h:419-421): which of the classes of
// core/robust_kernel_impl.h:41-172 and its RobustKernel::_delta; kind 0 = no kernel.  Kinds: 1 Huber, 2 PseudoHuber, 3 Cauchy, 4 Saturated,
// 5 DCS, 6 Tukey.
struct Robust { int kind = 0; double delta = 0; };
struct EdgeProj { int pt, cam; double uv[2], info[4], intr[4]; Robust rk; };
struct EdgeCub { int cam, cub; Cuboid meas; double info[81]; Robust rk; };
struct EdgeOdom { int ci, cj; SE3 meas; double info[36]; Robust rk; };
struct EdgeCubProj { int cam, cub; double meas[4], info[16], K[9]; Robust rk; };  // EdgeSE3CuboidProj (g2o_Object.h:264-293): bbox centre, width, height

struct Problem {
  std::vector<SE3> cams; std::vector<int> cam_fixed;
  std::vector<Cuboid> cubs; std::vector<int> cub_fixed;
  std::vector<double> pts; std::vector<int> pt_fixed;  // 3 per point; points are marginalised
  int cuboids_first = 0;  // vertex-id order: 0 = cams, cuboids ; 1 = cuboids, cams (main_obj.cpp:741-757 uses cube id 0)
  int marginalize_points = 1;
  std::vector<EdgeProj> eproj; std::vector<EdgeCub> ecub; std::vector<EdgeCubProj> ecproj; std::vector<EdgeOdom> eodom;
  // index mapping
  std::vector<int> cam_col, cub_col, pt_col;  // column (scalar offset) in the pose / landmark part, -1 if fixed
  int size_pose = 0, size_lm = 0, n_lm = 0;
  std::vector<int> pt_lmidx;  // landmark block index, -1 if fixed
  // linear system
  std::vector<double> Hpp;    // dense size_pose^2 (both triangles kept in sync)
  std::vector<double> Hll;    // n_lm * 9
  std::vector<double> Hpl;    // one 6x3 per projection edge (row-major), valid if both ends free
  std::vector<double> b, x;   // size_pose + size_lm
  std::vector<double> diag_backup_p, diag_backup_l;
  // errors
  std::vector<double> err_proj, err_cub, err_cproj, err_odom;  // 2, 9, 4, 6 per edge
  // LM state
  double lambda = -1, ni = 2; int nBad = 0, levenberg_iterations = 0;
  double user_lambda_init = 0; int max_trials_after_failure = 10;      // the class's two properties (optimization_algorithm_levenberg.cpp:50-51, :191-199)
  // backup stack (depth 1 is all LM needs)
  std::vector<SE3> cams_bak; std::vector<Cuboid> cubs_bak; std::vector<double> pts_bak;
  // stats
  std::vector<double> chi_hist; std::vector<double> lambda_hist; std::vector<int> trials_hist;
  // wall time per stage, the split of G2OBatchStatistics (Thirdparty/g2o/g2o/core/batch_stats.h:48-62): [0] residuals
  // (timeResiduals), [1] linearisation + quadratic form (timeLinearization + timeQuadraticForm), [2] Schur complement
  // (timeSchurComplement), [3] linear solver incl. back-substitution (timeLinearSolver), [4] update (timeUpdate); milliseconds
  double stage_ms[5] = {0, 0, 0, 0, 0};
  // bench.py's cpu_baseline at C4 only: > 1 = time every ldlt_stride-th column of the dense LDL^T and scale up (the
  // factorisation of a 10 494 x 10 494 system is ~5e11 flop, minutes on one core); the increment is then NOT computed
  int ldlt_stride = 1;
  // tests / bench at C4 only: the dense solve of the reduced system handed to the caller (LAPACK through scipy, oracle/ba_parity.py) -- the
  // textbook LDL^T below streams the 881 MB matrix once per column at 10 494 unknowns (24 minutes); everything else of an LM iteration
  // stays this file's.  0 = solved, anything else = not positive definite (the trial is then rejected like a failed LDL^T).
  int (*dense_solver)(double* S, int n, const double* b, double* x) = nullptr;
};
inline double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void build_index(Problem& P) {
  int nc = (int)P.cams.size(), no = (int)P.cubs.size(), np = (int)(P.pts.size() / 3);
  P.cam_col.assign(nc, -1); P.cub_col.assign(no, -1); P.pt_col.assign(np, -1); P.pt_lmidx.assign(np, -1);
  int col = 0;
  auto do_cams = [&]() { for (int i = 0; i < nc; i++) if (!P.cam_fixed[i]) { P.cam_col[i] = col; col += 6; } };
  auto do_cubs = [&]() { for (int i = 0; i < no; i++) if (!P.cub_fixed[i]) { P.cub_col[i] = col; col += 9; } };
  if (P.cuboids_first) { do_cubs(); do_cams(); } else { do_cams(); do_cubs(); }
  int lcol = 0, nl = 0;
  if (P.marginalize_points) {
    P.size_pose = col;
    for (int i = 0; i < np; i++) if (!P.pt_fixed[i]) { P.pt_col[i] = lcol; P.pt_lmidx[i] = nl++; lcol += 3; }
    P.size_lm = lcol; P.n_lm = nl;
  } else {
    // points as ordinary (non-marginalised) vertices after the poses; no Schur complement
    for (int i = 0; i < np; i++) if (!P.pt_fixed[i]) { P.pt_col[i] = col; col += 3; }
    P.size_pose = col; P.size_lm = 0; P.n_lm = 0;
  }
}

// ---- errors -------------------------------------------------------------------------------------
inline void err_proj_fn(const SE3& Tcw, const double X[3], const EdgeProj& e, double r[2]) {  // types_six_dof_expmap.h:156-161
  double p[3];
  se3_map(Tcw, X, p);
  double px = p[0] / p[2], py = p[1] / p[2];
  r[0] = e.uv[0] - (px * e.intr[0] + e.intr[2]);
  r[1] = e.uv[1] - (py * e.intr[1] + e.intr[3]);
}
inline void err_cub_fn(const SE3& Tcw, const Cuboid& cube, const EdgeCub& e, double r[9]) {  // g2o_Object.h:250-259
  SE3 Twc = se3_inv(Tcw);
  Cuboid esti;
  esti.pose = se3_mul(Twc, e.meas.pose);
  for (int i = 0; i < 3; i++) esti.scale[i] = e.meas.scale[i];
  min_log_error(cube, esti, r);
}
// cuboid::projectOntoImageBbox (g2o_Object.h:181-197) through compute3D_BoxCorner (:165-178) and similarityTransform
// (:154-160): the 8 corners (+-1 pattern) * diag(scale), rotated / translated to the world, mapped by Tcw, projected
// by K; bounding rectangle as (centre x, centre y, width, height).  The homogeneous divisions by w = 1 are exact.
inline void cuboid_project_bbox(const SE3& Tcw, const Cuboid& cube, const double K[9], double out[4]) {
  static const double cb[3][8] = {{1, 1, -1, -1, 1, 1, -1, -1}, {1, -1, -1, 1, 1, -1, -1, 1}, {-1, -1, -1, -1, 1, 1, 1, 1}};
  double Ro[9], Rc[9];
  q2R(cube.pose.r, Ro);
  q2R(Tcw.r, Rc);
  double M[9];  // R * diag(scale)
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[3 * i + j] = Ro[3 * i + j] * cube.scale[j];
  double xmin = 0, xmax = 0, ymin = 0, ymax = 0;
  for (int c = 0; c < 8; c++) {
    double Xw[3], Xc[3], p[3];
    for (int i = 0; i < 3; i++) Xw[i] = ((M[3 * i] * cb[0][c] + M[3 * i + 1] * cb[1][c]) + M[3 * i + 2] * cb[2][c]) + cube.pose.t[i] * 1.0;
    for (int i = 0; i < 3; i++) Xc[i] = ((Rc[3 * i] * Xw[0] + Rc[3 * i + 1] * Xw[1]) + Rc[3 * i + 2] * Xw[2]) + Tcw.t[i] * 1.0;
    for (int i = 0; i < 3; i++) p[i] = (K[3 * i] * Xc[0] + K[3 * i + 1] * Xc[1]) + K[3 * i + 2] * Xc[2];
    double u = p[0] / p[2], v = p[1] / p[2];
    if (c == 0) { xmin = xmax = u; ymin = ymax = v; }
    else {  // Eigen maxCoeff / minCoeff: strict comparisons, first occurrence wins
      if (u > xmax) xmax = u;
      if (u < xmin) xmin = u;
      if (v > ymax) ymax = v;
      if (v < ymin) ymin = v;
    }
  }
  out[0] = (xmax + xmin) / 2; out[1] = (ymax + ymin) / 2; out[2] = xmax - xmin; out[3] = ymax - ymin;
}
inline void err_cproj_fn(const SE3& Tcw, const Cuboid& cube, const EdgeCubProj& e, double r[4]) {  // g2o_Object.h:279-290
  double rect[4];
  cuboid_project_bbox(Tcw, cube, e.K, rect);
  for (int i = 0; i < 4; i++) r[i] = rect[i] - e.meas[i];
}
inline void err_odom_fn(const SE3& T1, const SE3& T2, const EdgeOdom& e, double r[6]) {  // types_six_dof_expmap.h:90-99
  SE3 err = se3_mul(se3_mul(e.meas, T1), se3_inv(T2));
  se3_log(err, r);
}
inline SE3 cam_oplus(const SE3& T, const double d[6]) { return se3_mul(se3_exp(d), T); }  // :73-76

inline double quad(const double* e, const double* info, int n) {  // chi2 = e^T Omega e
  double s = 0;
  for (int i = 0; i < n; i++) {
    double t = 0;
    for (int j = 0; j < n; j++) t += info[n * i + j] * e[j];
    s += e[i] * t;
  }
  return s;
}

void compute_errors(Problem& P) {  // sparse_optimizer.cpp:61-76
  P.err_proj.resize(2 * P.eproj.size()); P.err_cub.resize(9 * P.ecub.size()); P.err_cproj.resize(4 * P.ecproj.size()); P.err_odom.resize(6 * P.eodom.size());
  for (size_t k = 0; k < P.eproj.size(); k++) err_proj_fn(P.cams[P.eproj[k].cam], &P.pts[3 * P.eproj[k].pt], P.eproj[k], &P.err_proj[2 * k]);
  for (size_t k = 0; k < P.ecub.size(); k++) err_cub_fn(P.cams[P.ecub[k].cam], P.cubs[P.ecub[k].cub], P.ecub[k], &P.err_cub[9 * k]);
  for (size_t k = 0; k < P.ecproj.size(); k++) err_cproj_fn(P.cams[P.ecproj[k].cam], P.cubs[P.ecproj[k].cub], P.ecproj[k], &P.err_cproj[4 * k]);
  for (size_t k = 0; k < P.eodom.size(); k++) err_odom_fn(P.cams[P.eodom[k].ci], P.cams[P.eodom[k].cj], P.eodom[k], &P.err_odom[6 * k]);
}

// RobustKernel::robustify of the vendored g2o (core/robust_kernel_impl.cpp), kernel by kernel.  Two members of this (ORB-SLAM2-derived)
// copy are single precision: RobustKernelHuber::dsqr (`float dsqr`, robust_kernel_impl.h:86; setDelta stores delta * delta into it,
// robust_kernel_impl.cpp:65-69) and RobustKernelTukey::_deltaSqr / _invDeltaSqr (`float`, robust_kernel_impl.h:107-108, set by
// setDeltaSqr(deltaSqr, inv), robust_kernel_impl.cpp:94-99 -- here from delta: deltaSqr = delta^2, inv = 1 / delta^2, the call a user of
// that class makes).  The comparisons and products below therefore see the float-rounded squares, as the reference's do.
inline void robustify(const Robust& k, double e, double rho[3]) {
  const double delta = k.delta;
  switch (k.kind) {
    case 1: {   // RobustKernelHuber::robustify :78-91
      const float dsqr = (float)(delta * delta);
      if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
      else { double sqrte = std::sqrt(e); rho[0] = 2 * sqrte * delta - dsqr; rho[1] = delta / sqrte; rho[2] = -0.5 * rho[1] / e; }
      break;
    }
    case 2: {   // RobustKernelPseudoHuber::robustify :119-128
      double dsqr = delta * delta, dsqrReci = 1. / dsqr, aux1 = dsqrReci * e + 1.0, aux2 = std::sqrt(aux1);
      rho[0] = 2 * dsqr * (aux2 - 1); rho[1] = 1. / aux2; rho[2] = -0.5 * dsqrReci * rho[1] / aux1;
      break;
    }
    case 3: {   // RobustKernelCauchy::robustify :130-138
      double dsqr = delta * delta, dsqrReci = 1. / dsqr, aux = dsqrReci * e + 1.0;
      rho[0] = dsqr * std::log(aux); rho[1] = 1. / aux; rho[2] = -dsqrReci * std::pow(rho[1], 2);
      break;
    }
    case 4: {   // RobustKernelSaturated::robustify :140-152
      double dsqr = delta * delta;
      if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; } else { rho[0] = dsqr; rho[1] = 0.; rho[2] = 0.; }
      break;
    }
    case 5: {   // RobustKernelDCS::robustify :155-165 (delta is phi)
      double scale = (2.0 * delta) / (delta + e);
      if (scale >= 1.0) scale = 1.0;
      rho[0] = scale * e * scale; rho[1] = scale * scale; rho[2] = 0;
      break;
    }
    case 6: {   // RobustKernelTukey::robustify :101-117
      const float deltaSqr = (float)(delta * delta), invDeltaSqr = (float)(1.0 / (delta * delta));
      if (e <= deltaSqr) { double factor = e * invDeltaSqr, d = 1 - factor, dd = d * d; rho[0] = deltaSqr * (1 - dd * d); rho[1] = 3 * dd; rho[2] = -6 * invDeltaSqr * d; }
      else { rho[0] = deltaSqr; rho[1] = 0.; rho[2] = 0.; }
      break;
    }
    default: rho[0] = e; rho[1] = 1.; rho[2] = 0.;   // no kernel: chi2 itself (sparse_optimizer.cpp:100-114), Omega unweighted
  }
}

double robust_chi2(const Problem& P) {  // sparse_optimizer.cpp:100-114 (edges in insertion order: proj, cuboid, cuboid-projection, odom)
  double chi = 0;
  for (size_t k = 0; k < P.eproj.size(); k++) {
    double rho[3]; robustify(P.eproj[k].rk, quad(&P.err_proj[2 * k], P.eproj[k].info, 2), rho); chi += rho[0];
  }
  for (size_t k = 0; k < P.ecub.size(); k++) { double rho[3]; robustify(P.ecub[k].rk, quad(&P.err_cub[9 * k], P.ecub[k].info, 9), rho); chi += rho[0]; }
  for (size_t k = 0; k < P.ecproj.size(); k++) { double rho[3]; robustify(P.ecproj[k].rk, quad(&P.err_cproj[4 * k], P.ecproj[k].info, 4), rho); chi += rho[0]; }
  for (size_t k = 0; k < P.eodom.size(); k++) { double rho[3]; robustify(P.eodom[k].rk, quad(&P.err_odom[6 * k], P.eodom[k].info, 6), rho); chi += rho[0]; }
  return chi;
}

// ---- quadratic form helpers: H[ra.., ca..] += A^T W B, b[ra..] += A^T r ---------------------------
// A: D x Da, B: D x Db (row-major), W: D x D
inline void add_AtWB(std::vector<double>& H, int ld, int r0, int c0, const double* A, int Da, const double* W, const double* B, int Db, int D) {
  std::vector<double> AtW(Da * D);
  for (int i = 0; i < Da; i++)
    for (int j = 0; j < D; j++) {
      double s = 0;
      for (int k = 0; k < D; k++) s += A[k * Da + i] * W[k * D + j];
      AtW[i * D + j] = s;
    }
  for (int i = 0; i < Da; i++)
    for (int j = 0; j < Db; j++) {
      double s = 0;
      for (int k = 0; k < D; k++) s += AtW[i * D + k] * B[k * Db + j];
      H[(size_t)(r0 + i) * ld + (c0 + j)] += s;
    }
}
inline void add_Atr(std::vector<double>& b, int r0, const double* A, int Da, const double* r, int D) {
  for (int i = 0; i < Da; i++) {
    double s = 0;
    for (int k = 0; k < D; k++) s += A[k * Da + i] * r[k];
    b[r0 + i] += s;
  }
}

// generic binary edge: vertices (a: dim Da, column ca in pose part or -1) and (b: Db, cb), error e (D), info, rho1
void quadratic_form_pp(Problem& P, int ca, int Da, const double* Ja, int cb, int Db, const double* Jb, const double* e, const double* info, int D, double rho1) {
  // base_binary_edge.hpp:54-120: omega_r = -Omega e (scaled by rho'), weighted Omega = rho' Omega
  std::vector<double> W(D * D), r(D);
  for (int i = 0; i < D * D; i++) W[i] = rho1 * info[i];
  for (int i = 0; i < D; i++) {
    double s = 0;
    for (int j = 0; j < D; j++) s += info[D * i + j] * e[j];
    r[i] = -s * rho1;
  }
  int n = P.size_pose;
  if (ca >= 0) {
    add_Atr(P.b, ca, Ja, Da, r.data(), D);
    add_AtWB(P.Hpp, n, ca, ca, Ja, Da, W.data(), Ja, Da, D);
    if (cb >= 0) {
      // upper-triangle block is stored; keep the mirror in sync so the dense matrix stays symmetric
      add_AtWB(P.Hpp, n, ca, cb, Ja, Da, W.data(), Jb, Db, D);
      add_AtWB(P.Hpp, n, cb, ca, Jb, Db, W.data(), Ja, Da, D);
    }
  }
  if (cb >= 0) {
    add_Atr(P.b, cb, Jb, Db, r.data(), D);
    add_AtWB(P.Hpp, n, cb, cb, Jb, Db, W.data(), Jb, Db, D);
  }
}

void build_system(Problem& P) {  // block_solver.hpp:501-560
  int n = P.size_pose;
  P.Hpp.assign((size_t)n * n, 0.0);
  P.Hll.assign((size_t)P.n_lm * 9, 0.0);
  P.Hpl.assign(P.eproj.size() * 18, 0.0);
  P.b.assign(n + P.size_lm, 0.0);
  const double delta = 1e-9, scalar = 1.0 / (2 * delta);
  // --- projection edges: vertex 0 = point, vertex 1 = camera (types_six_dof_expmap.cpp:148-184)
  for (size_t k = 0; k < P.eproj.size(); k++) {
    const EdgeProj& e = P.eproj[k];
    const SE3& T = P.cams[e.cam];
    double xyz_trans[3];
    se3_map(T, &P.pts[3 * e.pt], xyz_trans);
    double x = xyz_trans[0], y = xyz_trans[1], z = xyz_trans[2], z_2 = z * z;
    double fx = e.intr[0], fy = e.intr[1];
    double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
    double R[9];
    q2R(T.r, R);
    double Ji[6];  // 2x3: -1/z * tmp * R
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int q = 0; q < 3; q++) s += (-1. / z * tmp[3 * i + q]) * R[3 * q + j];
        Ji[3 * i + j] = s;
      }
    double Jj[12];
    Jj[0] = x * y / z_2 * fx; Jj[1] = -(1 + (x * x / z_2)) * fx; Jj[2] = y / z * fx; Jj[3] = -1. / z * fx; Jj[4] = 0; Jj[5] = x / z_2 * fx;
    Jj[6] = (1 + y * y / z_2) * fy; Jj[7] = -x * y / z_2 * fy; Jj[8] = -x / z * fy; Jj[9] = 0; Jj[10] = -1. / z * fy; Jj[11] = y / z_2 * fy;
    const double* err = &P.err_proj[2 * k];
    double rho1 = 1.0;
    if (e.rk.kind) { double rho[3]; robustify(e.rk, quad(err, e.info, 2), rho); rho1 = rho[1]; }
    double W[4], r[2];
    for (int i = 0; i < 4; i++) W[i] = rho1 * e.info[i];
    for (int i = 0; i < 2; i++) r[i] = -(e.info[2 * i] * err[0] + e.info[2 * i + 1] * err[1]) * rho1;
    int cc = P.cam_col[e.cam];
    if (P.marginalize_points) {
      int li = P.pt_lmidx[e.pt];
      if (li >= 0) {  // "from" = point
        for (int i = 0; i < 3; i++) {
          double s = 0;
          for (int q = 0; q < 2; q++) s += Ji[3 * q + i] * r[q];
          P.b[n + P.pt_col[e.pt] + i] += s;
        }
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int p = 0; p < 2; p++) for (int q = 0; q < 2; q++) s += Ji[3 * p + i] * W[2 * p + q] * Ji[3 * q + j];
            P.Hll[9 * li + 3 * i + j] += s;
          }
        if (cc >= 0)  // Hpl block (pose row, landmark col) = Jj^T W Ji, 6x3
          for (int i = 0; i < 6; i++)
            for (int j = 0; j < 3; j++) {
              double s = 0;
              for (int p = 0; p < 2; p++) for (int q = 0; q < 2; q++) s += Jj[6 * p + i] * W[2 * p + q] * Ji[3 * q + j];
              P.Hpl[18 * k + 3 * i + j] += s;
            }
      }
      if (cc >= 0) {
        add_Atr(P.b, cc, Jj, 6, r, 2);
        add_AtWB(P.Hpp, n, cc, cc, Jj, 6, W, Jj, 6, 2);
      }
    } else {
      quadratic_form_pp(P, P.pt_col[e.pt], 3, Ji, cc, 6, Jj, err, e.info, 2, rho1);
    }
  }
  // --- cuboid edges: vertex 0 = camera, vertex 1 = cuboid; numeric Jacobians (base_binary_edge.hpp:130-205)
  for (size_t k = 0; k < P.ecub.size(); k++) {
    const EdgeCub& e = P.ecub[k];
    int ca = P.cam_col[e.cam], cb = P.cub_col[e.cub];
    if (ca < 0 && cb < 0) continue;
    double Ji[9 * 6] = {0}, Jj[9 * 9] = {0};
    if (ca >= 0)
      for (int d = 0; d < 6; d++) {
        double add[6] = {0}, e1[9], e2[9];
        add[d] = delta;
        err_cub_fn(cam_oplus(P.cams[e.cam], add), P.cubs[e.cub], e, e1);
        add[d] = -delta;
        err_cub_fn(cam_oplus(P.cams[e.cam], add), P.cubs[e.cub], e, e2);
        for (int r = 0; r < 9; r++) Ji[r * 6 + d] = scalar * (e1[r] - e2[r]);
      }
    if (cb >= 0)
      for (int d = 0; d < 9; d++) {
        double add[9] = {0}, e1[9], e2[9];
        add[d] = delta;
        err_cub_fn(P.cams[e.cam], cub_exp_update(P.cubs[e.cub], add), e, e1);
        add[d] = -delta;
        err_cub_fn(P.cams[e.cam], cub_exp_update(P.cubs[e.cub], add), e, e2);
        for (int r = 0; r < 9; r++) Jj[r * 9 + d] = scalar * (e1[r] - e2[r]);
      }
    double rho1 = 1.0;   // base_binary_edge.hpp:88-92: with a kernel, rho' of the edge's chi2 weights Omega and omega_r
    if (e.rk.kind) { double rho[3]; robustify(e.rk, quad(&P.err_cub[9 * k], e.info, 9), rho); rho1 = rho[1]; }
    quadratic_form_pp(P, ca, 6, Ji, cb, 9, Jj, &P.err_cub[9 * k], e.info, 9, rho1);
  }
  // --- cuboid projection edges (EdgeSE3CuboidProj): vertex 0 = camera, vertex 1 = cuboid; numeric Jacobians, 4-dim error
  for (size_t k = 0; k < P.ecproj.size(); k++) {
    const EdgeCubProj& e = P.ecproj[k];
    int ca = P.cam_col[e.cam], cb = P.cub_col[e.cub];
    if (ca < 0 && cb < 0) continue;
    double Ji[4 * 6] = {0}, Jj[4 * 9] = {0};
    if (ca >= 0)
      for (int d = 0; d < 6; d++) {
        double add[6] = {0}, e1[4], e2[4];
        add[d] = delta;
        err_cproj_fn(cam_oplus(P.cams[e.cam], add), P.cubs[e.cub], e, e1);
        add[d] = -delta;
        err_cproj_fn(cam_oplus(P.cams[e.cam], add), P.cubs[e.cub], e, e2);
        for (int r = 0; r < 4; r++) Ji[r * 6 + d] = scalar * (e1[r] - e2[r]);
      }
    if (cb >= 0)
      for (int d = 0; d < 9; d++) {
        double add[9] = {0}, e1[4], e2[4];
        add[d] = delta;
        err_cproj_fn(P.cams[e.cam], cub_exp_update(P.cubs[e.cub], add), e, e1);
        add[d] = -delta;
        err_cproj_fn(P.cams[e.cam], cub_exp_update(P.cubs[e.cub], add), e, e2);
        for (int r = 0; r < 4; r++) Jj[r * 9 + d] = scalar * (e1[r] - e2[r]);
      }
    double rho1 = 1.0;
    if (e.rk.kind) { double rho[3]; robustify(e.rk, quad(&P.err_cproj[4 * k], e.info, 4), rho); rho1 = rho[1]; }
    quadratic_form_pp(P, ca, 6, Ji, cb, 9, Jj, &P.err_cproj[4 * k], e.info, 4, rho1);
  }
  // --- odometry edges
  for (size_t k = 0; k < P.eodom.size(); k++) {
    const EdgeOdom& e = P.eodom[k];
    int ca = P.cam_col[e.ci], cb = P.cam_col[e.cj];
    if (ca < 0 && cb < 0) continue;
    double Ji[36] = {0}, Jj[36] = {0};
    if (ca >= 0)
      for (int d = 0; d < 6; d++) {
        double add[6] = {0}, e1[6], e2[6];
        add[d] = delta;
        err_odom_fn(cam_oplus(P.cams[e.ci], add), P.cams[e.cj], e, e1);
        add[d] = -delta;
        err_odom_fn(cam_oplus(P.cams[e.ci], add), P.cams[e.cj], e, e2);
        for (int r = 0; r < 6; r++) Ji[r * 6 + d] = scalar * (e1[r] - e2[r]);
      }
    if (cb >= 0)
      for (int d = 0; d < 6; d++) {
        double add[6] = {0}, e1[6], e2[6];
        add[d] = delta;
        err_odom_fn(P.cams[e.ci], cam_oplus(P.cams[e.cj], add), e, e1);
        add[d] = -delta;
        err_odom_fn(P.cams[e.ci], cam_oplus(P.cams[e.cj], add), e, e2);
        for (int r = 0; r < 6; r++) Jj[r * 6 + d] = scalar * (e1[r] - e2[r]);
      }
    double rho1 = 1.0;
    if (e.rk.kind) { double rho[3]; robustify(e.rk, quad(&P.err_odom[6 * k], e.info, 6), rho); rho1 = rho[1]; }
    quadratic_form_pp(P, ca, 6, Ji, cb, 6, Jj, &P.err_odom[6 * k], e.info, 6, rho1);
  }
}

void set_lambda(Problem& P, double lambda) {  // block_solver.hpp:563-589 (with backup)
  int n = P.size_pose;
  P.diag_backup_p.resize(n);
  for (int i = 0; i < n; i++) { P.diag_backup_p[i] = P.Hpp[(size_t)i * n + i]; P.Hpp[(size_t)i * n + i] += lambda; }
  P.diag_backup_l.resize(3 * (size_t)P.n_lm);
  for (int j = 0; j < P.n_lm; j++)
    for (int d = 0; d < 3; d++) { P.diag_backup_l[3 * j + d] = P.Hll[9 * j + 4 * d]; P.Hll[9 * j + 4 * d] += lambda; }
}
void restore_diagonal(Problem& P) {  // :591-604
  int n = P.size_pose;
  for (int i = 0; i < n; i++) P.Hpp[(size_t)i * n + i] = P.diag_backup_p[i];
  for (int j = 0; j < P.n_lm; j++)
    for (int d = 0; d < 3; d++) P.Hll[9 * j + 4 * d] = P.diag_backup_l[3 * j + d];
}

// dense LDL^T without pivoting (LinearSolverDense uses Eigen::LDLT; solvers/linear_solver_dense.h:104-111)
bool ldlt_solve(std::vector<double> A, int n, const double* b, double* x) {
  std::vector<double> D(n);
  for (int j = 0; j < n; j++) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k] * D[k];
    if (!(d > 0)) return false;  // isPositive()
    D[j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[(size_t)i * n + j];
      for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k] * D[k];
      A[(size_t)i * n + j] = s / d;
    }
  }
  std::vector<double> y(n);
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * y[k];
    y[i] = s;
  }
  for (int i = 0; i < n; i++) y[i] /= D[i];
  for (int i = n - 1; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * x[k];
    x[i] = s;
  }
  return true;
}

// Timing only: the arithmetic of every stride-th column of the factorisation above (same loops, same operands), wall time scaled
// by the sampled share of the multiply-adds.  Leaves A untouched in the columns it skips; returns the estimated milliseconds.
double ldlt_sampled_ms(std::vector<double>& A, int n, int stride) {
  std::vector<double> D(n, 1.0);
  double work_all = 0, work_done = 0;
  for (int j = 0; j < n; j++) work_all += (double)j * (n - j);
  const double t0 = wall_ms();
  for (int j = stride / 2; j < n; j += stride) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k] * D[k];
    D[j] = (d != 0) ? d : 1.0;
    for (int i = j + 1; i < n; i++) {
      double s = A[(size_t)i * n + j];
      for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k] * D[k];
      A[(size_t)i * n + j] = s / D[j];
    }
    work_done += (double)j * (n - j);
  }
  const double t = wall_ms() - t0;
  return work_done > 0 ? t * work_all / work_done : 0.0;
}

inline void inv3(const double* a, double* r) {  // Eigen 3x3 inverse, cofactor form
  auto cof = [&](int i, int j) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return a[3 * i1 + j1] * a[3 * i2 + j2] - a[3 * i1 + j2] * a[3 * i2 + j1];
  };
  double c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
  double det = (c0 * a[0] + c1 * a[3]) + c2 * a[6];
  double id = 1.0 / det;
  r[0] = c0 * id; r[1] = c1 * id; r[2] = c2 * id;
  r[3] = cof(0, 1) * id; r[4] = cof(1, 1) * id; r[5] = cof(2, 1) * id;
  r[6] = cof(0, 2) * id; r[7] = cof(1, 2) * id; r[8] = cof(2, 2) * id;
}

// the Schur complement of the landmarks (block_solver.hpp:373-439): S = Hpp - sum_j W_j D_j^-1 W_j^T, bschur = b_p - sum_j W_j D_j^-1 b_j
void schur_complement(Problem& P, std::vector<double>& S, std::vector<double>& bschur, std::vector<double>& Dinv) {
  int n = P.size_pose;
  S = P.Hpp;  // Hschur = Hpp
  std::vector<double> coeff(n, 0.0);
  Dinv.assign((size_t)P.n_lm * 9, 0.0);
  // edges per landmark, sorted by pose row (the CCS column order, block_solver.hpp:398-431)
  std::vector<std::vector<int>> lm_edges(P.n_lm);
  for (size_t k = 0; k < P.eproj.size(); k++) {
    int li = P.pt_lmidx[P.eproj[k].pt];
    if (li >= 0 && P.cam_col[P.eproj[k].cam] >= 0) lm_edges[li].push_back((int)k);
  }
  for (int j = 0; j < P.n_lm; j++) {
    std::sort(lm_edges[j].begin(), lm_edges[j].end(), [&](int a, int c) { return P.cam_col[P.eproj[a].cam] < P.cam_col[P.eproj[c].cam]; });
    inv3(&P.Hll[9 * j], &Dinv[9 * j]);
    double db[3];
    mv3(&Dinv[9 * j], &P.b[n + 3 * j], db);
    for (size_t a = 0; a < lm_edges[j].size(); a++) {
      int ka = lm_edges[j][a];
      int i1 = P.cam_col[P.eproj[ka].cam];
      const double* Bi = &P.Hpl[18 * ka];
      double BDinv[18];
      for (int r = 0; r < 6; r++)
        for (int c = 0; c < 3; c++) BDinv[3 * r + c] = Bi[3 * r] * Dinv[9 * j + c] + Bi[3 * r + 1] * Dinv[9 * j + 3 + c] + Bi[3 * r + 2] * Dinv[9 * j + 6 + c];
      for (int r = 0; r < 6; r++) coeff[i1 + r] += Bi[3 * r] * db[0] + Bi[3 * r + 1] * db[1] + Bi[3 * r + 2] * db[2];
      for (size_t c2 = a; c2 < lm_edges[j].size(); c2++) {
        int kb = lm_edges[j][c2];
        int i2 = P.cam_col[P.eproj[kb].cam];
        const double* Bj = &P.Hpl[18 * kb];
        for (int r = 0; r < 6; r++)
          for (int c = 0; c < 6; c++) {
            double v = BDinv[3 * r] * Bj[3 * c] + BDinv[3 * r + 1] * Bj[3 * c + 1] + BDinv[3 * r + 2] * Bj[3 * c + 2];
            S[(size_t)(i1 + r) * n + (i2 + c)] -= v;
            if (i1 != i2) S[(size_t)(i2 + c) * n + (i1 + r)] -= v;
          }
      }
    }
  }
  bschur.resize(n);
  for (int i = 0; i < n; i++) bschur[i] = P.b[i] - coeff[i];
}
// x_l = D^-1 (b_l - W^T x_p) (block_solver.hpp:457-482); x holds x_p in its first size_pose entries
void back_substitute(Problem& P, const std::vector<double>& Dinv, std::vector<double>& x) {
  int n = P.size_pose;
  std::vector<double> cl(P.b.begin() + n, P.b.end());
  for (size_t k = 0; k < P.eproj.size(); k++) {
    int li = P.pt_lmidx[P.eproj[k].pt], cc = P.cam_col[P.eproj[k].cam];
    if (li < 0 || cc < 0) continue;
    const double* B = &P.Hpl[18 * k];
    for (int c = 0; c < 3; c++) {
      double s = 0;
      for (int r = 0; r < 6; r++) s += B[3 * r + c] * (-x[cc + r]);
      cl[3 * li + c] += s;
    }
  }
  for (int j = 0; j < P.n_lm; j++) mv3(&Dinv[9 * j], &cl[3 * j], &x[n + 3 * j]);
}
bool solve_system(Problem& P) {  // block_solver.hpp:353-486
  int n = P.size_pose;
  P.x.assign(n + P.size_lm, 0.0);
  if (P.n_lm == 0) {
    const double t0 = wall_ms();
    bool ok;
    if (P.dense_solver) { std::vector<double> H = P.Hpp; ok = P.dense_solver(H.data(), n, P.b.data(), P.x.data()) == 0; }
    else ok = ldlt_solve(P.Hpp, n, P.b.data(), P.x.data());
    P.stage_ms[3] += wall_ms() - t0;
    return ok;
  }
  double t_stage = wall_ms();
  std::vector<double> S, bschur, Dinv;
  schur_complement(P, S, bschur, Dinv);
  P.stage_ms[2] += wall_ms() - t_stage;
  t_stage = wall_ms();
  if (P.ldlt_stride > 1) { P.stage_ms[3] += ldlt_sampled_ms(S, n, P.ldlt_stride); return true; }   // timing run: no increment
  if (P.dense_solver ? P.dense_solver(S.data(), n, bschur.data(), P.x.data()) != 0 : !ldlt_solve(std::move(S), n, bschur.data(), P.x.data())) { P.stage_ms[3] += wall_ms() - t_stage; return false; }
  back_substitute(P, Dinv, P.x);   // landmarks: xl = Dinv (bl - Hpl^T xp)
  P.stage_ms[3] += wall_ms() - t_stage;
  return true;
}

void apply_update(Problem& P) {  // sparse_optimizer.cpp:422-435
  int n = P.size_pose;
  for (size_t i = 0; i < P.cams.size(); i++)
    if (P.cam_col[i] >= 0) P.cams[i] = cam_oplus(P.cams[i], &P.x[P.cam_col[i]]);
  for (size_t i = 0; i < P.cubs.size(); i++)
    if (P.cub_col[i] >= 0) P.cubs[i] = cub_exp_update(P.cubs[i], &P.x[P.cub_col[i]]);
  for (size_t i = 0; i < P.pts.size() / 3; i++)
    if (P.pt_col[i] >= 0) {
      int off = P.marginalize_points ? n + P.pt_col[i] : P.pt_col[i];
      for (int d = 0; d < 3; d++) P.pts[3 * i + d] += P.x[off + d];
    }
}

// optimization_algorithm_levenberg.cpp:61-163; returns 0 = OK, 1 = Terminate
int lm_solve(Problem& P, int iteration) {
  double t_stage = wall_ms();
  compute_errors(P);
  double currentChi = robust_chi2(P), tempChi = currentChi, iniChi = currentChi;
  P.stage_ms[0] += wall_ms() - t_stage;
  t_stage = wall_ms();
  build_system(P);
  P.stage_ms[1] += wall_ms() - t_stage;
  if (iteration == 0) {  // computeLambdaInit (:166-180): tau * max |H_jj| over all non-fixed vertices
    double maxDiagonal = 0;
    int n = P.size_pose;
    for (int i = 0; i < n; i++) maxDiagonal = std::max(std::fabs(P.Hpp[(size_t)i * n + i]), maxDiagonal);
    for (int j = 0; j < P.n_lm; j++) for (int d = 0; d < 3; d++) maxDiagonal = std::max(std::fabs(P.Hll[9 * j + 4 * d]), maxDiagonal);
    P.lambda = P.user_lambda_init > 0 ? P.user_lambda_init : 1e-5 * maxDiagonal;      // (:168-169)
    P.ni = 2; P.nBad = 0;
  }
  double rho = 0;
  int qmax = 0;
  do {
    P.cams_bak = P.cams; P.cubs_bak = P.cubs; P.pts_bak = P.pts;  // push
    set_lambda(P, P.lambda);
    bool ok2 = solve_system(P);
    t_stage = wall_ms();
    apply_update(P);
    P.stage_ms[4] += wall_ms() - t_stage;
    restore_diagonal(P);
    t_stage = wall_ms();
    compute_errors(P);
    tempChi = robust_chi2(P);
    P.stage_ms[0] += wall_ms() - t_stage;
    if (!ok2) tempChi = std::numeric_limits<double>::max();
    rho = (currentChi - tempChi);
    double scale = 0;
    for (size_t j = 0; j < P.x.size(); j++) scale += P.x[j] * (P.lambda * P.x[j] + P.b[j]);
    scale += 1e-3;
    rho /= scale;
    if (rho > 0 && std::isfinite(tempChi)) {
      double alpha = 1. - std::pow((2 * rho - 1), 3);
      alpha = std::min(alpha, 2. / 3.);
      double scaleFactor = std::max(1. / 3., alpha);
      P.lambda *= scaleFactor;
      P.ni = 2;
      currentChi = tempChi;
    } else {
      P.lambda *= P.ni;
      P.ni *= 2;
      P.cams = P.cams_bak; P.cubs = P.cubs_bak; P.pts = P.pts_bak;  // pop
    }
    qmax++;
  } while (rho < 0 && qmax < P.max_trials_after_failure);
  P.levenberg_iterations = qmax;
  P.chi_hist.push_back(currentChi); P.lambda_hist.push_back(P.lambda); P.trials_hist.push_back(qmax);
  if (qmax == P.max_trials_after_failure || rho == 0) return 1;
  if ((iniChi - currentChi) * 1e3 < iniChi) P.nBad++; else P.nBad = 0;
  if (P.nBad >= 3) return 1;
  return 0;
}

}  // namespace

extern "C" {

struct ba_oracle;  // opaque = Problem

void* ba_oracle_create() { return new Problem(); }
void ba_oracle_destroy(void* h) { delete (Problem*)h; }

// Vertices.  cams: Nc x 7 (x y z qx qy qz qw of the *world-to-camera* SE3Quat a VertexSE3Expmap stores);
// cuboids: No x 10 (pose 7 as above, object-to-world, + 3 half sizes, g2o::cuboid::toVector); points: Np x 3.
// Vertex ids: cuboids_first ? (cuboids 0.., cams No..) : (cams 0.., cuboids Nc..); points after both.
void ba_oracle_set_vertices(void* h, const double* cams, const int* cam_fixed, int nc, const double* cuboids, const int* cub_fixed, int no,
                            const double* points, const int* pt_fixed, int np, int cuboids_first, int marginalize_points) {
  Problem& P = *(Problem*)h;
  P.cams.resize(nc); P.cam_fixed.assign(cam_fixed, cam_fixed + nc);
  for (int i = 0; i < nc; i++) P.cams[i] = se3_from7(cams + 7 * i, true);
  P.cubs.resize(no); P.cub_fixed.assign(cub_fixed, cub_fixed + no);
  for (int i = 0; i < no; i++) { P.cubs[i].pose = se3_from7(cuboids + 10 * i, false); for (int d = 0; d < 3; d++) P.cubs[i].scale[d] = cuboids[10 * i + 7 + d]; }
  P.pts.assign(points, points + 3 * (size_t)np); P.pt_fixed.assign(pt_fixed, pt_fixed + np);
  P.cuboids_first = cuboids_first; P.marginalize_points = marginalize_points;
  build_index(P);
}
void ba_oracle_set_edges_proj(void* h, int n, const int* pt, const int* cam, const double* uv, const double* info4, const double* intr4, const double* huber) {
  Problem& P = *(Problem*)h;
  P.eproj.resize(n);
  for (int k = 0; k < n; k++) {
    EdgeProj& e = P.eproj[k];
    e.pt = pt[k]; e.cam = cam[k];
    std::memcpy(e.uv, uv + 2 * k, 16); std::memcpy(e.info, info4 + 4 * k, 32); std::memcpy(e.intr, intr4 + 4 * k, 32);
    e.rk = Robust{};
    if (huber && huber[k] > 0) { e.rk.kind = 1; e.rk.delta = huber[k]; }   // RobustKernelHuber with setDelta(huber[k])
  }
}
// Robust kernels of one edge class (0 projection, 1 EdgeSE3Cuboid, 2 EdgeSE3CuboidProj, 3 EdgeSE3Expmap): kind as in struct Robust,
// delta = RobustKernel::delta().  n must be the class's edge count; call after the class's edges are set.
int ba_oracle_set_robust_kernels(void* h, int edge_class, int n, const int* kind, const double* delta) {
  Problem& P = *(Problem*)h;
  auto put = [&](auto& edges) -> int {
    if ((size_t)n != edges.size()) return 1;
    for (int k = 0; k < n; k++) { edges[k].rk.kind = kind ? kind[k] : 0; edges[k].rk.delta = delta ? delta[k] : 0.0; }
    return 0;
  };
  switch (edge_class) { case 0: return put(P.eproj); case 1: return put(P.ecub); case 2: return put(P.ecproj); case 3: return put(P.eodom); }
  return 1;
}
// RobustKernel::robustify for a known-answer table (tests/test_ba_oracle.py)
void ba_oracle_robustify(int kind, double delta, double e, double rho3[3]) { robustify(Robust{kind, delta}, e, rho3); }
void ba_oracle_set_edges_cuboid(void* h, int n, const int* cam, const int* cub, const double* meas10, const double* info81) {
  Problem& P = *(Problem*)h;
  P.ecub.resize(n);
  for (int k = 0; k < n; k++) {
    EdgeCub& e = P.ecub[k];
    e.cam = cam[k]; e.cub = cub[k];
    e.meas.pose = se3_from7(meas10 + 10 * k, false);
    for (int d = 0; d < 3; d++) e.meas.scale[d] = meas10[10 * k + 7 + d];
    std::memcpy(e.info, info81 + 81 * k, 81 * 8);
  }
}
// EdgeSE3CuboidProj: measured bbox (centre x, centre y, width, height), 4x4 information, the edge's Kalib (row-major)
void ba_oracle_set_edges_cuboid_proj(void* h, int n, const int* cam, const int* cub, const double* meas4, const double* info16, const double* K9) {
  Problem& P = *(Problem*)h;
  P.ecproj.resize(n);
  for (int k = 0; k < n; k++) {
    EdgeCubProj& e = P.ecproj[k];
    e.cam = cam[k]; e.cub = cub[k];
    std::memcpy(e.meas, meas4 + 4 * k, 32); std::memcpy(e.info, info16 + 16 * k, 128); std::memcpy(e.K, K9 + 9 * k, 72);
  }
}
void ba_oracle_set_edges_odom(void* h, int n, const int* ci, const int* cj, const double* meas7, const double* info36) {
  Problem& P = *(Problem*)h;
  P.eodom.resize(n);
  for (int k = 0; k < n; k++) {
    EdgeOdom& e = P.eodom[k];
    e.ci = ci[k]; e.cj = cj[k];
    e.meas = se3_from7(meas7 + 7 * k, true);  // SE3Quat C(_measurement) copy of a normalised SE3Quat
    std::memcpy(e.info, info36 + 36 * k, 36 * 8);
  }
}
// graph.optimize(iterations) (sparse_optimizer.cpp:354-419); returns the number of LM iterations executed
int ba_oracle_optimize(void* h, int iterations) {
  Problem& P = *(Problem*)h;
  P.chi_hist.clear(); P.lambda_hist.clear(); P.trials_hist.clear();
  int done = 0;
  for (int i = 0; i < iterations; i++) {
    int res = lm_solve(P, i);
    ++done;
    if (res != 0) break;
  }
  return done;
}
int ba_oracle_history(void* h, double* chi, double* lambda, int* trials, int cap) {
  Problem& P = *(Problem*)h;
  int n = std::min(cap, (int)P.chi_hist.size());
  for (int i = 0; i < n; i++) { chi[i] = P.chi_hist[i]; lambda[i] = P.lambda_hist[i]; trials[i] = P.trials_hist[i]; }
  return (int)P.chi_hist.size();
}
void ba_oracle_get_state(void* h, double* cams7, double* cuboids10, double* points3) {
  Problem& P = *(Problem*)h;
  for (size_t i = 0; i < P.cams.size(); i++) se3_to7(P.cams[i], cams7 + 7 * i);
  for (size_t i = 0; i < P.cubs.size(); i++) { se3_to7(P.cubs[i].pose, cuboids10 + 10 * i); for (int d = 0; d < 3; d++) cuboids10[10 * i + 7 + d] = P.cubs[i].scale[d]; }
  if (points3) std::memcpy(points3, P.pts.data(), sizeof(double) * P.pts.size());
}
// wall time per stage accumulated by ba_oracle_optimize since the last reset (milliseconds; Problem::stage_ms)
void ba_oracle_stage_ms(void* h, double out5[5], int reset) {
  Problem& P = *(Problem*)h;
  for (int i = 0; i < 5; i++) { out5[i] = P.stage_ms[i]; if (reset) P.stage_ms[i] = 0; }
}
// bench.py's cpu_baseline at C4: time every stride-th column of the dense LDL^T (Problem::ldlt_stride); 1 = the real solve
void ba_oracle_set_ldlt_stride(void* h, int stride) { ((Problem*)h)->ldlt_stride = stride < 1 ? 1 : stride; }
// The reduced system's dense solve through a caller's routine (see Problem::dense_solver); nullptr restores the LDL^T of this file.
void ba_oracle_set_lm_params(void* h, double user_lambda_init, int max_trials) { ((Problem*)h)->user_lambda_init = user_lambda_init; ((Problem*)h)->max_trials_after_failure = max_trials; }
void ba_oracle_set_dense_solver(void* h, int (*fn)(double*, int, const double*, double*)) { ((Problem*)h)->dense_solver = fn; }
// stage-level access for parity tests
double ba_oracle_compute_errors(void* h) { Problem& P = *(Problem*)h; compute_errors(P); return robust_chi2(P); }
void ba_oracle_get_errors_cproj(void* h, double* cproj4) { Problem& P = *(Problem*)h; if (cproj4) std::memcpy(cproj4, P.err_cproj.data(), 8 * P.err_cproj.size()); }
void ba_oracle_get_errors(void* h, double* proj2, double* cub9, double* odom6) {
  Problem& P = *(Problem*)h;
  if (proj2) std::memcpy(proj2, P.err_proj.data(), 8 * P.err_proj.size());
  if (cub9) std::memcpy(cub9, P.err_cub.data(), 8 * P.err_cub.size());
  if (odom6) std::memcpy(odom6, P.err_odom.data(), 8 * P.err_odom.size());
}
void ba_oracle_build_system(void* h) { Problem& P = *(Problem*)h; compute_errors(P); build_system(P); }
int ba_oracle_sizes(void* h, int* size_pose, int* size_lm) { Problem& P = *(Problem*)h; *size_pose = P.size_pose; *size_lm = P.size_lm; return 0; }
void ba_oracle_get_system(void* h, double* Hpp_dense, double* Hll9, double* Hpl18, double* b) {
  Problem& P = *(Problem*)h;
  if (Hpp_dense) std::memcpy(Hpp_dense, P.Hpp.data(), 8 * P.Hpp.size());
  if (Hll9) std::memcpy(Hll9, P.Hll.data(), 8 * P.Hll.size());
  if (Hpl18) std::memcpy(Hpl18, P.Hpl.data(), 8 * P.Hpl.size());
  if (b) std::memcpy(b, P.b.data(), 8 * P.b.size());
}
// one damped solve on the current system: returns 1 if the reduced system was positive definite
int ba_oracle_solve(void* h, double lambda, double* x) {
  Problem& P = *(Problem*)h;
  set_lambda(P, lambda);
  bool ok = solve_system(P);
  restore_diagonal(P);
  if (x) std::memcpy(x, P.x.data(), 8 * P.x.size());
  return ok ? 1 : 0;
}

// Stage access at sizes where the oracle's own unblocked dense LDL^T is out of reach (C4: 10 494 unknowns): the damped reduced system
// [S | b_schur] of the current linearisation (dense size_pose^2, both triangles), and the landmark back-substitution for pose
// increments solved elsewhere (the tests factorise the oracle's S with LAPACK).  Both apply lambda as setLambda does (:563-589) and
// restore the diagonals afterwards.
void ba_oracle_schur(void* h, double lambda, double* S_dense, double* bschur) {
  Problem& P = *(Problem*)h;
  set_lambda(P, lambda);
  std::vector<double> S, bs, Dinv;
  if (P.n_lm == 0) { S = P.Hpp; bs.assign(P.b.begin(), P.b.begin() + P.size_pose); } else schur_complement(P, S, bs, Dinv);
  restore_diagonal(P);
  if (S_dense) std::memcpy(S_dense, S.data(), 8 * S.size());
  if (bschur) std::memcpy(bschur, bs.data(), 8 * bs.size());
}
void ba_oracle_backsub(void* h, double lambda, const double* xp, double* x_full) {
  Problem& P = *(Problem*)h;
  set_lambda(P, lambda);
  std::vector<double> Dinv((size_t)P.n_lm * 9), x(P.size_pose + P.size_lm, 0.0);
  for (int j = 0; j < P.n_lm; j++) inv3(&P.Hll[9 * j], &Dinv[9 * j]);
  std::memcpy(x.data(), xp, 8 * (size_t)P.size_pose);
  back_substitute(P, Dinv, x);
  restore_diagonal(P);
  std::memcpy(x_full, x.data(), 8 * x.size());
}

// SE3 / cuboid primitives for known-answer tests
void ba_oracle_se3_exp(const double u[6], double out7[7]) { se3_to7(se3_exp(u), out7); }
void ba_oracle_se3_log(const double v7[7], double out6[6]) { se3_log(se3_from7(v7, true), out6); }
void ba_oracle_se3_mul(const double a7[7], const double b7[7], double out7[7]) { se3_to7(se3_mul(se3_from7(a7, true), se3_from7(b7, true)), out7); }
void ba_oracle_se3_inv(const double a7[7], double out7[7]) { se3_to7(se3_inv(se3_from7(a7, true)), out7); }
// g2o::cuboid::fromMinimalVector (g2o_Object.h:37-42): xyz roll pitch yaw half-scale -> 10-vector
void ba_oracle_cuboid_from_minimal(const double v9[9], double out10[10]) {
  double roll = v9[3], pitch = v9[4], yaw = v9[5];
  double sy = std::sin(yaw * 0.5), cy = std::cos(yaw * 0.5), sp = std::sin(pitch * 0.5), cp = std::cos(pitch * 0.5), sr = std::sin(roll * 0.5), cr = std::cos(roll * 0.5);
  SE3 T;
  T.r.w = cr * cp * cy + sr * sp * sy;
  T.r.x = sr * cp * cy - cr * sp * sy;
  T.r.y = cr * sp * cy + sr * cp * sy;
  T.r.z = cr * cp * sy - sr * sp * cy;
  T.t[0] = v9[0]; T.t[1] = v9[1]; T.t[2] = v9[2];
  normalizeRotation(T);
  se3_to7(T, out10);
  for (int d = 0; d < 3; d++) out10[7 + d] = v9[6 + d];
}
// cuboid::projectOntoImageBbox(campose_cw, Kalib) (g2o_Object.h:190-197)
void ba_oracle_cuboid_project_bbox(const double cub10[10], const double Tcw7[7], const double K9[9], double out4[4]) {
  Cuboid c; c.pose = se3_from7(cub10, false); for (int d = 0; d < 3; d++) c.scale[d] = cub10[7 + d];
  cuboid_project_bbox(se3_from7(Tcw7, false), c, K9, out4);
}
// cuboid.transform_to(Twc) / transform_from(Twc) (g2o_Object.h:117-133)
void ba_oracle_cuboid_transform(const double cub10[10], const double Twc7[7], int to_local, double out10[10]) {
  SE3 pose = se3_from7(cub10, false), T = se3_from7(Twc7, true);
  SE3 r = to_local ? se3_mul(se3_inv(T), pose) : se3_mul(T, pose);
  se3_to7(r, out10);
  for (int d = 0; d < 3; d++) out10[7 + d] = cub10[7 + d];
}
// toMinimalVector (g2o_Object.h:137-143, se3quat.h:196-222): xyz roll pitch yaw scale
void ba_oracle_cuboid_to_minimal(const double cub10[10], double out9[9]) {
  double qx = cub10[3], qy = cub10[4], qz = cub10[5], qw = cub10[6];
  out9[0] = cub10[0]; out9[1] = cub10[1]; out9[2] = cub10[2];
  out9[3] = std::atan2(2 * (qw * qx + qy * qz), 1 - 2 * (qx * qx + qy * qy));
  out9[4] = std::asin(2 * (qw * qy - qz * qx));
  out9[5] = std::atan2(2 * (qw * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz));
  for (int d = 0; d < 3; d++) out9[6 + d] = cub10[7 + d];
}

}  // extern "C"
