// cs_geom.h -- 2D/3D geometry of the cuboid proposal sweep, shared by the HIP kernels and the host
// stages of libcubeslam_hip.  Arithmetic follows the reference expression by expression (operation
// order matters: proposal scores are compared bit for bit), the structure is this library's own.
// Build with -ffp-contract=off.
#pragma once
#include "cs_atan2.h"

namespace cs {

struct V2 {
  double x, y;
};

CS_HD V2 v2(double x, double y) { return V2{x, y}; }
CS_HD double v2_dist(V2 a, V2 b) {
  double dx = a.x - b.x, dy = a.y - b.y;
  return __builtin_sqrt(dx * dx + dy * dy);
}
// squared distance, the argument of v2_dist's square root
CS_HD double v2_dist2(V2 a, V2 b) {
  double dx = a.x - b.x, dy = a.y - b.y;
  return dx * dx + dy * dy;
}
// Thresholds on a length become thresholds on the squared length: the IEEE square root is correctly rounded and monotone, so
//   sqrt(x) <  t  <=>  x <  sqrt_lt_bound(t)   (the smallest double whose root reaches t)
//   sqrt(x) >  t  <=>  x >  sqrt_le_bound(t)   (the largest double whose root does not exceed t)
// for every x >= 0, inf and NaN included.  Host side, computed once per parameter set.
static inline double cs_next_up(double x) { long long b = cs_bits(x); return cs_from_bits(b + 1); }     // x >= 0, finite
static inline double cs_next_down(double x) { long long b = cs_bits(x); return cs_from_bits(b - 1); }   // x > 0
static inline double sqrt_lt_bound(double t) {
  if (!(t > 0)) return 0.0;                              // sqrt(x) < t never holds; neither does x < 0
  if (t == __builtin_huge_val()) return t;
  double c = t * t;
  if (c == __builtin_huge_val()) c = 1.7976931348623157e308;
  while (c > 0 && __builtin_sqrt(c) >= t) c = cs_next_down(c);
  while (__builtin_sqrt(c) < t) { if (c >= 1.7976931348623157e308) return __builtin_huge_val(); c = cs_next_up(c); }
  return c;
}
static inline double sqrt_le_bound(double t) {
  if (t != t) return __builtin_huge_val();               // nothing exceeds NaN; nothing exceeds inf
  if (t < 0) return -1.0;                                // sqrt(x) > t holds for every x that is a number; so does x > -1
  if (t == __builtin_huge_val()) return t;
  const double lt = sqrt_lt_bound(cs_next_up(t));        // the smallest x with sqrt(x) > t ...
  if (lt == __builtin_huge_val()) return 1.7976931348623157e308;
  return lt > 0 ? cs_next_down(lt) : 0.0;                // ... and its predecessor
}
CS_HD double dmin(double a, double b) { return (b < a) ? b : a; }  // std::min
CS_HD double dmax(double a, double b) { return (a < b) ? b : a; }  // std::max
CS_HD double dabs(double a) { return __builtin_fabs(a); }

#define CS_PI 3.14159265358979323846

// matrix_utils.cpp:344-353
CS_HD double normalize_to_pi(double a) {
  if (a > CS_PI / 2) return a - CS_PI;
  else if (a < -CS_PI / 2) return a + CS_PI;
  return a;
}

// object_3d_util.cpp:239-242 (inclusive on all four sides)
CS_HD bool inside_box(V2 p, double l, double t, double r, double b) { return l <= p.x && p.x <= r && t <= p.y && p.y <= b; }

// Ray start->end against a vertical boundary segment x = bx, y in [by0, by1] (object_3d_util.cpp:339-351).
// Returns false when the ray misses; the reference encodes a miss as (-1,-1) and callers test one
// coordinate against -1, which `hit_is_minus_one` reproduces.
CS_HD V2 ray_hit_vertical(V2 s, V2 e, double bx, double by0, double by1) {
  V2 hit = v2(-1, -1);
  double dx = e.x - s.x, dy = e.y - s.y;
  if (by0 == by1) {  // degenerate: the reference would first treat it as horizontal (:322-336)
    double lambd = (by0 - s.y) / dy;
    if (lambd >= 0) {
      V2 t = v2(s.x + lambd * dx, s.y + lambd * dy);
      if ((bx <= t.x) && (t.x <= bx)) { hit = t; hit.y = by0; }
    }
  }
  double lambd = (bx - s.x) / dx;
  if (lambd >= 0) {
    V2 t = v2(s.x + lambd * dx, s.y + lambd * dy);
    if ((by0 <= t.y) && (t.y <= by1)) { hit = t; hit.x = bx; }
  }
  return hit;
}

// Ray against a horizontal boundary segment y = by, x in [bx0, bx1] (object_3d_util.cpp:322-336).
CS_HD V2 ray_hit_horizontal(V2 s, V2 e, double by, double bx0, double bx1) {
  V2 hit = v2(-1, -1);
  double dx = e.x - s.x, dy = e.y - s.y;
  double lambd = (by - s.y) / dy;
  if (lambd >= 0) {
    V2 t = v2(s.x + lambd * dx, s.y + lambd * dy);
    if ((bx0 <= t.x) && (t.x <= bx1)) { hit = t; hit.y = by; }
  }
  if (bx0 == bx1) {  // degenerate: also a vertical edge (:339-351)
    double l2 = (bx0 - s.x) / dx;
    if (l2 >= 0) {
      V2 t = v2(s.x + l2 * dx, s.y + l2 * dy);
      if ((by <= t.y) && (t.y <= by)) { hit = t; hit.x = bx0; }
    }
  }
  return hit;
}

// Intersection of the infinite lines (a1,a2) and (b1,b2) (object_3d_util.cpp:357-382, infinite_line=true).
CS_HD V2 line_intersect(V2 a1, V2 a2, V2 b1, V2 b2) {
  double X2_X1 = a2.x - a1.x, Y2_Y1 = a2.y - a1.y;
  double X4_X3 = b2.x - b1.x, Y4_Y3 = b2.y - b1.y;
  double X1_X3 = a1.x - b1.x, Y1_Y3 = a1.y - b1.y;
  double u_a = (X4_X3 * Y1_Y3 - Y4_Y3 * X1_X3) / (Y4_Y3 * X2_X1 - X4_X3 * Y2_Y1);
  return v2((a1.x + X2_X1 * u_a) * 1.0, (a1.y + Y2_Y1 * u_a) * 1.0);
}

// Integer geometry of one (box, height sample) job.
struct BoxGeom {
  int left, top, right, down;          // raw box, bottom already expanded by the height sample
  int el, et, er, eb;                  // expanded ROI, inclusive bounds
};

// The eight 2D corners of one proposal (box_proposal_detail.cpp:413-625).
// Returns 0 when the proposal is rejected, else vp_1_position (1 = left, 2 = right).
// The reference rejects edges with sqrt(dx^2 + dy^2) < shorted_edge_thre.  The IEEE square root is monotone, so that test is
// d2 < short_sq_bound with short_sq_bound = the smallest double whose rounded square root reaches the threshold (computed once
// on the host, sqrt_lt_bound() in detect_host.cpp): the same decision for every d2 including NaN / inf, without the 13 roots.
CS_HD int build_corners(const BoxGeom& g, V2 vp1, V2 vp2, V2 vp3, double top_x, int config_id, double short_sq_bound, V2 c[8]) {
  V2 c1 = v2(top_x, (double)g.top);
  int vp1_pos = 0;
  V2 c2 = ray_hit_vertical(vp1, c1, (double)g.right, (double)g.top, (double)g.down);
  if (c2.x == -1) {
    c2 = ray_hit_vertical(vp1, c1, (double)g.left, (double)g.top, (double)g.down);
    if (c2.x != -1) vp1_pos = 2;
  } else {
    vp1_pos = 1;
  }
  if (!(vp1_pos > 0)) return 0;
  if (v2_dist2(c1, c2) < short_sq_bound) return 0;
  V2 c3, c4;
  if (config_id == 1) {
    if (vp1_pos == 1) c4 = ray_hit_vertical(vp2, c1, (double)g.left, (double)g.top, (double)g.down);
    else c4 = ray_hit_vertical(vp2, c1, (double)g.right, (double)g.top, (double)g.down);
    if (c4.y == -1) return 0;
    if (v2_dist2(c1, c4) < short_sq_bound) return 0;
    c3 = line_intersect(vp2, c2, vp1, c4);
    if (!inside_box(c3, g.left, g.top, g.right, g.down)) return 0;
    if ((v2_dist2(c3, c4) < short_sq_bound) || (v2_dist2(c3, c2) < short_sq_bound)) return 0;
  } else {
    if (vp1_pos == 1) c3 = ray_hit_vertical(vp2, c2, (double)g.left, (double)g.top, (double)g.down);
    else c3 = ray_hit_vertical(vp2, c2, (double)g.right, (double)g.top, (double)g.down);
    if (c3.y == -1) return 0;
    if (v2_dist2(c2, c3) < short_sq_bound) return 0;
    c4 = line_intersect(vp1, c3, vp2, c1);
    if (!inside_box(c4, g.left, g.et, g.right, g.eb)) return 0;  // raw x bounds, expanded y bounds (:558)
    if ((v2_dist2(c3, c4) < short_sq_bound) || (v2_dist2(c4, c1) < short_sq_bound)) return 0;
  }
  V2 c5 = ray_hit_horizontal(vp3, c3, (double)g.down, (double)g.left, (double)g.right);
  if (c5.y == -1) return 0;
  if (v2_dist2(c3, c5) < short_sq_bound) return 0;
  V2 c6 = line_intersect(vp2, c5, vp3, c2);
  if (!inside_box(c6, g.el, g.et, g.er, g.eb)) return 0;
  if ((v2_dist2(c6, c2) < short_sq_bound) || (v2_dist2(c6, c5) < short_sq_bound)) return 0;
  V2 c7 = line_intersect(vp1, c6, vp3, c1);
  if (!inside_box(c7, g.el, g.et, g.er, g.eb)) return 0;
  if ((v2_dist2(c7, c1) < short_sq_bound) || (v2_dist2(c7, c6) < short_sq_bound)) return 0;
  V2 c8 = line_intersect(vp1, c5, vp2, c7);
  if (!inside_box(c8, g.el, g.et, g.er, g.eb)) return 0;
  if ((v2_dist2(c8, c4) < short_sq_bound) || (v2_dist2(c8, c5) < short_sq_bound) || (v2_dist2(c8, c7) < short_sq_bound)) return 0;
  c[0] = c1; c[1] = c2; c[2] = c3; c[3] = c4; c[4] = c5; c[5] = c6; c[6] = c7; c[7] = c8;
  return vp1_pos;
}

// The corners of a proposal that build_corners ACCEPTED, values only.  vp1_pos = build_corners' return value (1 / 2: which side the
// top edge runs into).  Every corner is the same expression on the same operands as in build_corners -- an accepted ray hit is
// (bx, s.y + ((bx - s.x) / dx) * dy) resp. (s.x + ((by - s.y) / dy) * dx, by), the intersections are line_intersect itself -- so the
// values carry the same bits; what is left out is everything that only DECIDES (the second attempt of the first ray, the miss
// sentinels, the inside-box tests, the 13 edge-length tests): a proposal that reaches the scorer has passed them.  (An accepted
// proposal's box is never degenerate -- top == down would put corner 5 on corner 3 and fail the 20-pixel test -- so the degenerate
// branches of the ray hits do not contribute.)  Both configurations share the code: configuration 1 hits the far side from corner 1
// and intersects for corner 3, configuration 2 hits it from corner 2 and intersects for corner 4 (box_proposal_detail.cpp:468-572).
CS_HD void rebuild_accepted_corners(const BoxGeom& g, V2 vp1, V2 vp2, V2 vp3, double top_x, int config_id, int vp1_pos, V2 c[8]) {
  const V2 c1 = v2(top_x, (double)g.top);
  const bool right_first = vp1_pos == 1;
  const double side1 = (double)(right_first ? g.right : g.left), side2 = (double)(right_first ? g.left : g.right);
  const bool k1 = config_id == 1;
  V2 c2;
  { const double dx = c1.x - vp1.x, dy = c1.y - vp1.y, lambd = (side1 - vp1.x) / dx; c2 = v2(side1, vp1.y + lambd * dy); }
  const V2 P = k1 ? c1 : c2;
  V2 H;
  { const double dx = P.x - vp2.x, dy = P.y - vp2.y, lambd = (side2 - vp2.x) / dx; H = v2(side2, vp2.y + lambd * dy); }
  const V2 L = line_intersect(k1 ? vp2 : vp1, k1 ? c2 : H, k1 ? vp1 : vp2, k1 ? H : c1);
  const V2 c3 = k1 ? L : H, c4 = k1 ? H : L;
  V2 c5;
  { const double dx = c3.x - vp3.x, dy = c3.y - vp3.y, lambd = ((double)g.down - vp3.y) / dy; c5 = v2(vp3.x + lambd * dx, (double)g.down); }
  const V2 c6 = line_intersect(vp2, c5, vp3, c2);
  const V2 c7 = line_intersect(vp1, c6, vp3, c1);
  const V2 c8 = line_intersect(vp1, c5, vp2, c7);
  c[0] = c1; c[1] = c2; c[2] = c3; c[3] = c4; c[4] = c5; c[5] = c6; c[6] = c7; c[7] = c8;
}

// Camera data of one (roll, pitch) sample: what set_cam_pose() leaves behind for the sweep and the
// 3D lift (box_proposal_detail.cpp:45-56, :131/:376).
struct RpPose {
  double KinvR[9];       // Kalib * invR
  double R[9];           // rotationToWorld
  double t[3];           // camera position in the world
  double plane[4];       // ground plane in the sensor frame = T_wc^T (0,0,1,0)
  double roll, pitch;    // the sample's angles (row columns 7, 8)
};

// pixel -> ray -> plane hit -> world point (object_3d_util.cpp:841-876), T_wc = [R t; 0 0 0 1].
CS_HD void plane_hit_world(const double* R, const double* t, const double* invK, const double plane[4], double px, double py, double out[3]) {
  double ray[3];
  for (int i = 0; i < 3; i++) ray[i] = (invK[3 * i + 0] * px + invK[3 * i + 1] * py) + invK[3 * i + 2] * 1.0;
  double den = (plane[0] * ray[0] + plane[1] * ray[1]) + plane[2] * ray[2];
  double frac = -plane[3] / den;
  double p[3] = {frac * ray[0], frac * ray[1], frac * ray[2]};
  // homogeneous row (0 0 0 1): w = ((0*x + 0*y) + 0*z) + 1*1 is exactly 1 for a finite point and NaN otherwise (0 * inf), and the
  // reference divides by it: v / 1 == v bit for bit, v / NaN == NaN -- so the three divisions reduce to a select
  double w = ((0.0 * p[0] + 0.0 * p[1]) + 0.0 * p[2]) + 1.0 * 1.0;
  for (int i = 0; i < 3; i++) {
    double v = ((R[3 * i + 0] * p[0] + R[3 * i + 1] * p[1]) + R[3 * i + 2] * p[2]) + t[i] * 1.0;
    out[i] = (w != w) ? w : v;
  }
}

// Position and half sizes of the cuboid whose image is `c` (object_3d_util.cpp:941-990).
CS_HD void lift_to_3d(const V2 c[8], const double* R, const double* t, const double* invK, const double ground_plane[4], double pos[3], double scale[3]) {
  double g[4][3];
  for (int k = 0; k < 4; k++) plane_hit_world(R, t, invK, ground_plane, c[4 + k].x, c[4 + k].y, g[k]);
  double d03[3] = {g[0][0] - g[3][0], g[0][1] - g[3][1], g[0][2] - g[3][2]};
  double d01[3] = {g[0][0] - g[1][0], g[0][1] - g[1][1], g[0][2] - g[1][2]};
  double length_half = __builtin_sqrt((d03[0] * d03[0] + d03[1] * d03[1]) + d03[2] * d03[2]) / 2;
  double width_half = __builtin_sqrt((d01[0] * d01[0] + d01[1] * d01[1]) + d01[2] * d01[2]) / 2;
  // wall plane through corners 5-6 (get_wall_plane_equation, object_3d_util.cpp:909-925)
  double n[3] = {d01[1] * 1.0 - d01[2] * 0.0, d01[2] * 0.0 - d01[0] * 1.0, d01[0] * 0.0 - d01[1] * 0.0};
  double nn = __builtin_sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
  for (int i = 0; i < 3; i++) n[i] /= nn;
  double dist = -((n[0] * g[0][0] + n[1] * g[0][1]) + n[2] * g[0][2]);
  double pw[4] = {n[0], n[1], n[2], dist};
  if (dist < 0) for (int i = 0; i < 4; i++) pw[i] = -pw[i];
  // plane in the sensor frame: T_wc^T * pw
  double ps[4];
  for (int i = 0; i < 3; i++) ps[i] = ((R[0 + i] * pw[0] + R[3 + i] * pw[1]) + R[6 + i] * pw[2]) + 0.0 * pw[3];
  ps[3] = ((t[0] * pw[0] + t[1] * pw[1]) + t[2] * pw[2]) + 1.0 * pw[3];
  double top[3];
  plane_hit_world(R, t, invK, ps, c[1].x, c[1].y, top);
  double height_half = top[2] / 2;
  pos[0] = (((g[0][0] + g[1][0]) + g[2][0]) + g[3][0]) / 4.0;
  pos[1] = (((g[0][1] + g[1][1]) + g[2][1]) + g[3][1]) / 4.0;
  pos[2] = height_half;
  scale[0] = length_half; scale[1] = width_half; scale[2] = height_half;
}

}  // namespace cs
