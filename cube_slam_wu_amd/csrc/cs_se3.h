// cs_se3.h -- SE(3) / cuboid algebra of the g2o bundle-adjustment path, for host and device.
//
// Follows g2o's SE3Quat (object_slam/Thirdparty/g2o/g2o/types/se3quat.h:41-362: unit quaternion +
// translation, compose-then-normalise with w >= 0, exp/log with the theta < 1e-5 and d > 0.99999 branches)
// and g2o::cuboid (object_slam/include/object_slam/g2o_Object.h:23-133).  A pose is 7 doubles in g2o's
// vector order x y z qx qy qz qw (se3quat.h:151-163).  FP64 throughout; libm calls resolve to glibc on the
// host and to ocml on gfx950 (BA parity is 1e-5 relative, not bit-exact).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define CS_HD __host__ __device__ __forceinline__
#else
#ifndef CS_HD
#define CS_HD static inline
#endif
#endif

namespace cs {

struct Pose {  // SE3Quat
  double t[3];
  double qx, qy, qz, qw;
};
struct Cube {  // g2o::cuboid
  Pose pose;
  double scale[3];
};

CS_HD Pose pose_load(const double* v) { Pose p; p.t[0] = v[0]; p.t[1] = v[1]; p.t[2] = v[2]; p.qx = v[3]; p.qy = v[4]; p.qz = v[5]; p.qw = v[6]; return p; }
CS_HD void pose_store(const Pose& p, double* v) { v[0] = p.t[0]; v[1] = p.t[1]; v[2] = p.t[2]; v[3] = p.qx; v[4] = p.qy; v[5] = p.qz; v[6] = p.qw; }
CS_HD Cube cube_load(const double* v) { Cube c; c.pose = pose_load(v); c.scale[0] = v[7]; c.scale[1] = v[8]; c.scale[2] = v[9]; return c; }
CS_HD void cube_store(const Cube& c, double* v) { pose_store(c.pose, v); v[7] = c.scale[0]; v[8] = c.scale[1]; v[9] = c.scale[2]; }

CS_HD void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
// q * v (Eigen QuaternionBase::_transformVector)
CS_HD void pose_rotate(const Pose& p, const double* v, double* o) {
  double qv[3] = {p.qx, p.qy, p.qz}, uv[3], c2[3];
  cross3(qv, v, uv);
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  cross3(qv, uv, c2);
  for (int i = 0; i < 3; i++) o[i] = v[i] + p.qw * uv[i] + c2[i];
}
CS_HD void pose_map(const Pose& p, const double* x, double* o) {  // se3quat.h:274-277
  pose_rotate(p, x, o);
  o[0] += p.t[0]; o[1] += p.t[1]; o[2] += p.t[2];
}
CS_HD void pose_normalize(Pose& p) {  // se3quat.h:346-351
  if (p.qw < 0) { p.qx = -p.qx; p.qy = -p.qy; p.qz = -p.qz; p.qw = -p.qw; }
  double n = sqrt(p.qx * p.qx + p.qy * p.qy + p.qz * p.qz + p.qw * p.qw);
  p.qx /= n; p.qy /= n; p.qz /= n; p.qw /= n;
}
CS_HD Pose pose_mul(const Pose& a, const Pose& b) {  // se3quat.h:110-116
  Pose r;
  double rt[3];
  pose_rotate(a, b.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] = a.t[i] + rt[i];
  r.qw = a.qw * b.qw - a.qx * b.qx - a.qy * b.qy - a.qz * b.qz;
  r.qx = a.qw * b.qx + a.qx * b.qw + a.qy * b.qz - a.qz * b.qy;
  r.qy = a.qw * b.qy + a.qy * b.qw + a.qz * b.qx - a.qx * b.qz;
  r.qz = a.qw * b.qz + a.qz * b.qw + a.qx * b.qy - a.qy * b.qx;
  pose_normalize(r);
  return r;
}
CS_HD Pose pose_inv(const Pose& a) {  // se3quat.h:129-134
  Pose r;
  r.qx = -a.qx; r.qy = -a.qy; r.qz = -a.qz; r.qw = a.qw;
  double nt[3] = {-a.t[0], -a.t[1], -a.t[2]};
  pose_rotate(r, nt, r.t);
  return r;
}
CS_HD void pose_rotmat(const Pose& p, double* R) {  // Eigen toRotationMatrix
  double tx = 2 * p.qx, ty = 2 * p.qy, tz = 2 * p.qz;
  double twx = tx * p.qw, twy = ty * p.qw, twz = tz * p.qw;
  double txx = tx * p.qx, txy = ty * p.qx, txz = tz * p.qx, tyy = ty * p.qy, tyz = tz * p.qy, tzz = tz * p.qz;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
CS_HD void quat_from_rotmat(const double* R, Pose& p) {  // Eigen Quaterniond(Matrix3d)
  double t = R[0] + R[4] + R[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    p.qw = 0.5 * t;
    t = 0.5 / t;
    p.qx = (R[7] - R[5]) * t; p.qy = (R[2] - R[6]) * t; p.qz = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    p.qw = (R[3 * k + j] - R[3 * j + k]) * t;
    v[j] = (R[3 * j + i] + R[3 * i + j]) * t;
    v[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    p.qx = v[0]; p.qy = v[1]; p.qz = v[2];
  }
}
CS_HD void skew3(const double* v, double* m) {
  m[0] = 0; m[1] = -v[2]; m[2] = v[1];
  m[3] = v[2]; m[4] = 0; m[5] = -v[0];
  m[6] = -v[1]; m[7] = v[0]; m[8] = 0;
}
CS_HD void mat3_mul(const double* a, const double* b, double* o) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) o[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
CS_HD void mat3_vec(const double* a, const double* v, double* o) {
  for (int i = 0; i < 3; i++) o[i] = a[3 * i] * v[0] + a[3 * i + 1] * v[1] + a[3 * i + 2] * v[2];
}
// se3quat.h:230-272: res = [omega, upsilon]
CS_HD void pose_log(const Pose& T, double* res) {
  double R[9];
  pose_rotmat(T, R);
  double d = 0.5 * (R[0] + R[4] + R[8] - 1);
  double omega[3], dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  double Om[9], Om2[9], c;
  if (d > 0.99999) {
    for (int i = 0; i < 3; i++) omega[i] = 0.5 * dR[i];
    c = 1. / 12.;
  } else {
    double theta = acos(d);
    double f = theta / (2 * sqrt(1 - d * d));
    for (int i = 0; i < 3; i++) omega[i] = f * dR[i];
    c = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
  }
  skew3(omega, Om);
  mat3_mul(Om, Om, Om2);
  double Vinv[9];
  for (int i = 0; i < 9; i++) Vinv[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + c * Om2[i];
  double ups[3];
  mat3_vec(Vinv, T.t, ups);
  for (int i = 0; i < 3; i++) { res[i] = omega[i]; res[i + 3] = ups[i]; }
}
// se3quat.h:280-323: update = [omega, upsilon]
CS_HD Pose pose_exp(const double* u) {
  double theta = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  double Om[9], Om2[9], R[9], V[9];
  skew3(u, Om);
  mat3_mul(Om, Om, Om2);
  double a, b, c;
  if (theta < 0.00001) { a = 1; b = 1; c = 1; }  // R = I + Om + Om^2, V = R
  else {
    a = sin(theta) / theta;
    b = (1 - cos(theta)) / (theta * theta);
    c = (theta - sin(theta)) / (theta * theta * theta);
  }
  bool small = theta < 0.00001;
  for (int i = 0; i < 9; i++) {
    double I = (i % 4 == 0) ? 1.0 : 0.0;
    R[i] = I + a * Om[i] + b * Om2[i];
    V[i] = small ? R[i] : (I + b * Om[i] + c * Om2[i]);
  }
  Pose T;
  quat_from_rotmat(R, T);
  mat3_vec(V, u + 3, T.t);
  pose_normalize(T);
  return T;
}

// ---- vertices' oplus ------------------------------------------------------------------------------
CS_HD Pose cam_oplus(const Pose& T, const double* d) { return pose_mul(pose_exp(d), T); }  // types_six_dof_expmap.h:73-76
CS_HD Cube cube_oplus(const Cube& c, const double* d) {  // g2o_Object.h:57-63, :208-211
  Cube r;
  r.pose = pose_mul(c.pose, pose_exp(d));
  for (int i = 0; i < 3; i++) r.scale[i] = c.scale[i] + d[6 + i];
  return r;
}

// ---- edge errors ----------------------------------------------------------------------------------
// EdgeSE3ProjectXYZ::computeError (types_six_dof_expmap.h:156-161, .cpp:186-192)
CS_HD void proj_error(const Pose& Tcw, const double* X, const double* uv, const double* intr, double* r, double* pc) {
  pose_map(Tcw, X, pc);
  r[0] = uv[0] - (pc[0] / pc[2] * intr[0] + intr[2]);
  r[1] = uv[1] - (pc[1] / pc[2] * intr[1] + intr[3]);
}
// cuboid::cube_log_error (g2o_Object.h:66-73)
CS_HD void cube_log_error(const Cube& self, const Cube& other, double* res) {
  Pose diff = pose_mul(pose_inv(other.pose), self.pose);
  pose_log(diff, res);
  for (int i = 0; i < 3; i++) res[6 + i] = self.scale[i] - other.scale[i];
}
// cuboid::min_log_error over yaw rotations {-90, 0, 90, 180} deg (g2o_Object.h:76-114): first minimum,
// strict <, so a NaN norm (log of an exact 180 deg rotation) never wins unless it is candidate 0.
CS_HD void cube_min_log_error(const Cube& self, const Cube& other, double* res) {
  // sin / cos of half the yaw angles (i - 1) pi / 2: glibc's values of sin(ang * 0.5), cos(ang * 0.5) (the four candidates are a LOOP on
  // the device -- unrolled, their four independent SE3 chains cost the numeric-Jacobian kernels 197 registers and two wavefronts per SIMD)
  const double QZ[4] = {-0x1.6a09e667f3bccp-1, 0.0, 0x1.6a09e667f3bccp-1, 1.0}, QW[4] = {0x1.6a09e667f3bcdp-1, 1.0, 0x1.6a09e667f3bcdp-1, 0x1.1a62633145c07p-54};
  double best_n = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int i = 0; i < 4; i++) {
    Pose rot;
    rot.t[0] = rot.t[1] = rot.t[2] = 0;
    rot.qx = 0; rot.qy = 0; rot.qz = QZ[i]; rot.qw = QW[i];
    pose_normalize(rot);
    Cube rc;
    rc.pose = pose_mul(other.pose, rot);
    rc.scale[0] = other.scale[0]; rc.scale[1] = other.scale[1]; rc.scale[2] = other.scale[2];
    if (i == 0 || i == 2) { double t = rc.scale[0]; rc.scale[0] = rc.scale[1]; rc.scale[1] = t; }  // +-90 deg swap x/y
    double e[9];
    cube_log_error(self, rc, e);
    double s = 0;
    for (int k = 0; k < 9; k++) s += e[k] * e[k];
    double n = sqrt(s);
    if (i == 0 || n < best_n) {
      best_n = n;
      for (int k = 0; k < 9; k++) res[k] = e[k];
    }
  }
}
// one candidate of min_log_error: the error against `other` turned by yaw (i - 1) pi / 2, and its norm -- the loop body above, for callers
// that know which candidate wins (ba_cub_edge_kernel: the perturbed evaluations of a numeric Jacobian take the unperturbed winner when
// its lead over the runner-up is a hundred thousand times what a 1e-9 step can move a norm)
CS_HD double cube_log_error_candidate(const Cube& self, const Cube& other, int i, double* e) {
  // (the same constants as the loop above, by selects: a runtime index into a local array would live in scratch memory on the device)
  Pose rot;
  rot.t[0] = rot.t[1] = rot.t[2] = 0;
  rot.qx = 0; rot.qy = 0;
  rot.qz = i == 0 ? -0x1.6a09e667f3bccp-1 : (i == 1 ? 0.0 : (i == 2 ? 0x1.6a09e667f3bccp-1 : 1.0));
  rot.qw = i == 0 ? 0x1.6a09e667f3bcdp-1 : (i == 1 ? 1.0 : (i == 2 ? 0x1.6a09e667f3bcdp-1 : 0x1.1a62633145c07p-54));
  pose_normalize(rot);
  Cube rc;
  rc.pose = pose_mul(other.pose, rot);
  const bool swap = i == 0 || i == 2;
  rc.scale[0] = swap ? other.scale[1] : other.scale[0]; rc.scale[1] = swap ? other.scale[0] : other.scale[1]; rc.scale[2] = other.scale[2];
  cube_log_error(self, rc, e);
  double s = 0;
  for (int k = 0; k < 9; k++) s += e[k] * e[k];
  return sqrt(s);
}
// EdgeSE3Cuboid::computeError (g2o_Object.h:250-259)
CS_HD void cuboid_edge_error(const Pose& Tcw, const Cube& cube, const Cube& meas, double* r) {
  Cube esti;
  esti.pose = pose_mul(pose_inv(Tcw), meas.pose);
  esti.scale[0] = meas.scale[0]; esti.scale[1] = meas.scale[1]; esti.scale[2] = meas.scale[2];
  cube_min_log_error(cube, esti, r);
}
// EdgeSE3CuboidProj::computeError (g2o_Object.h:279-290): cuboid::projectOntoImageBbox (:181-197) of the cuboid's 8
// corners (compute3D_BoxCorner :165-178, similarityTransform :154-160) minus the measured (centre x, centre y, w, h)
CS_HD void cuboid_proj_error(const Pose& Tcw, const Cube& cube, const double* K, const double* meas4, double* r) {
  double Ro[9], Rc[9], M[9];
  pose_rotmat(cube.pose, Ro);
  pose_rotmat(Tcw, Rc);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[3 * i + j] = Ro[3 * i + j] * cube.scale[j];
  double xmin = 0, xmax = 0, ymin = 0, ymax = 0;
  for (int c = 0; c < 8; c++) {
    // corner pattern of compute3D_BoxCorner: x = + + - - + + - -, y = + - - + + - - +, z = - - - - + + + +
    const double sx = ((c >> 1) & 1) ? -1.0 : 1.0, sy = (((c + 1) >> 1) & 1) ? -1.0 : 1.0, sz = (c < 4) ? -1.0 : 1.0;
    double Xw[3], Xc[3], p[3];
    for (int i = 0; i < 3; i++) Xw[i] = ((M[3 * i] * sx + M[3 * i + 1] * sy) + M[3 * i + 2] * sz) + cube.pose.t[i] * 1.0;
    for (int i = 0; i < 3; i++) Xc[i] = ((Rc[3 * i] * Xw[0] + Rc[3 * i + 1] * Xw[1]) + Rc[3 * i + 2] * Xw[2]) + Tcw.t[i] * 1.0;
    for (int i = 0; i < 3; i++) p[i] = (K[3 * i] * Xc[0] + K[3 * i + 1] * Xc[1]) + K[3 * i + 2] * Xc[2];
    const double u = p[0] / p[2], v = p[1] / p[2];
    if (c == 0) { xmin = xmax = u; ymin = ymax = v; }
    else {
      if (u > xmax) xmax = u;
      if (u < xmin) xmin = u;
      if (v > ymax) ymax = v;
      if (v < ymin) ymin = v;
    }
  }
  r[0] = (xmax + xmin) / 2 - meas4[0]; r[1] = (ymax + ymin) / 2 - meas4[1]; r[2] = (xmax - xmin) - meas4[2]; r[3] = (ymax - ymin) - meas4[3];
}
// EdgeSE3Expmap::computeError (types_six_dof_expmap.h:90-99)
CS_HD void odom_edge_error(const Pose& T1, const Pose& T2, const Pose& meas, double* r) {
  pose_log(pose_mul(pose_mul(meas, T1), pose_inv(T2)), r);
}

CS_HD void inv3x3(const double* a, double* r) {  // cofactor form (Eigen Matrix3d::inverse)
  double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[7] * a[2] - a[8] * a[1], c2 = a[1] * a[5] - a[2] * a[4];
  double det = (c0 * a[0] + c1 * a[3]) + c2 * a[6];
  double id = 1.0 / det;
  r[0] = c0 * id; r[1] = c1 * id; r[2] = c2 * id;
  r[3] = (a[5] * a[6] - a[3] * a[8]) * id; r[4] = (a[8] * a[0] - a[6] * a[2]) * id; r[5] = (a[2] * a[3] - a[0] * a[5]) * id;
  r[6] = (a[3] * a[7] - a[4] * a[6]) * id; r[7] = (a[6] * a[1] - a[7] * a[0]) * id; r[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

}  // namespace cs
