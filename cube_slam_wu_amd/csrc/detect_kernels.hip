// detect_kernels.hip -- HIP kernels of the cuboid proposal sweep for gfx950 (MI355X).
//
// What runs here (reference loops L2..L5 and the ranking, detect_3d_cuboid/src/box_proposal_detail.cpp:200-838):
//   line_setup_kernel   one wave per job: ROI segment filter (:271-283), merge_break_lines (object_3d_util.cpp:431-543),
//                       segment angles / mid points (:309-315);
//   vp_points_kernel    one lane per (job, roll/pitch, yaw): the three vanishing points (object_3d_util.cpp:928);
//   vp3_support_kernel / vp_support_kernel   their supporting segment angles (object_3d_util.cpp:548-619);
//   candidate_kernel    one lane per proposal slot (job, roll/pitch, yaw, top sample, config): the eight corners (:413-625),
//                       a flag per slot, corners of the valid ones;
//   scan_jobs / compact ordered stream compaction of the valid proposals per job (the reference appends rows in loop
//                       order, :677-702, and that order defines tie-breaking);
//   score_kernel        one lane per valid proposal: distance-map edge score (object_3d_util.cpp:622), VP angle alignment
//                       (:670), half sizes of the lifted cuboid (:941-990);
//   rank_kernel         one workgroup per box: fuse_normalize_scores_v2 (object_3d_util.cpp:726-837) + the skew-weighted
//                       final ranking (box_proposal_detail.cpp:804-838) as order statistics and arg-mins; boxes whose ties
//                       could reach the output are flagged for the exact host ranking;
//   gather_*            columns / corners of those boxes.
//
// Numerics: values that leave a kernel -- corners, errors, merged angles -- are computed with the reference's
// operations in its order (FP64 geometry, float32 gathers with a sequential float sum per proposal, cs_atan2 for libm's atan2);
// decisions (inlier tests, extreme selection, merge tests, length thresholds) take a cheaper exact-equivalent form with the
// reference's evaluation inside a margin.  Compiled with -ffp-contract=off: every result is bit-identical to the CPU oracle.
#include <hip/hip_runtime.h>
#include "cs_hip_util.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/cubeslam_hip.h"
#include "detect_types.h"

namespace cs {

// job of element `elem` of a prefix table, for a wave whose lanes hold consecutive elements: one binary search per wave on
// the first lane's element (uniform operands: scalar loads), then every lane walks forward from there (jobs are thousands
// of elements long, so the walk is 0 or 1 step) -- instead of a 13-step dependent chain of vector loads in every lane
__device__ __forceinline__ int wave_uniform_i32(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ long long wave_uniform_i64(long long x) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x & 0xffffffffll)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(x >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
template <typename T>
__device__ __forceinline__ int find_job_wave(const T* __restrict__ prefix, int n, T elem) {
  const T first = sizeof(T) == 8 ? (T)wave_uniform_i64((long long)elem) : (T)wave_uniform_i32((int)elem);   // lane 0 holds the smallest
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (prefix[mid] <= first) lo = mid; else hi = mid;
  }
  while (lo + 1 < n && prefix[lo + 1] <= elem) lo++;
  return lo;
}

// Blocks with the same (blockIdx % 8) run on the same XCD (private 4 MiB L2).  Remap so that each XCD
// walks a contiguous range of virtual blocks: all blocks of a job then share one L2, which keeps the
// job's distance map and line arrays L2-resident.  grid must be a multiple of 8.
__device__ __forceinline__ long long xcd_virtual_block() {
  long long nb = gridDim.x, b = blockIdx.x;
  return (b & 7) * (nb >> 3) + (b >> 3);
}

// VP_support_edge_infos (object_3d_util.cpp:548-619) for ONE vanishing point: sequential over the job's merged lines.
//
// The reference evaluates atan2 for every segment, keeps those whose direction is within `thre` of the segment's own angle,
// unwraps the kept angles against the first one (smooth_jump_angles, :278-302) and returns the segment angles at the arg-max and
// arg-min.  Only DECISIONS leave this function (which segments are inliers, which one is the extreme), so the angles are
// evaluated in float (|error| < 2.5e-6 rad over all quadrants, tests/test_device_decisions.py) and every decision that falls
// within a margin of its threshold -- a few per million -- is re-decided with the exact cs_atan2 values of the segments
// involved.  Error budget of a float angle: conversion of the double differences 1e-7, quotient (__fdividef) 3e-7, polynomial
// 2e-6, pi constants and the +-2 pi shift 7e-7; of a difference of two of them twice that.  Margins: 1e-5 on the inlier test,
// 2e-5 on the unwrap decision and on the comparisons against the running extremes.  Each decision therefore equals the
// reference's, including ties (equal angles keep the first occurrence, :609-614 maxCoeff / minCoeff).
// out[0], out[1] = the two bounding segment angles (NaN = none): (max, min) for vp 1, swapped for vp 2, 3 (:609-614).
__device__ __attribute__((noinline)) double vp_exact_raw(double dy, double dx) { return cs_atan2(dy, dx); }

// exact unwrapped angle of segment i against the first inlier `ib` (both re-evaluated: this runs a few times per million segments)
__device__ __forceinline__ double vp_exact_unwrapped(const double* __restrict__ mx, const double* __restrict__ my, int i, int ib, double vpxk, double vpyk) {
  const double raw = vp_exact_raw(my[i] - vpyk, mx[i] - vpxk);
  if (i == ib) return raw;
  const double base = vp_exact_raw(my[ib] - vpyk, mx[ib] - vpxk);
  if ((raw - base) < -CS_PI) return raw + 2 * CS_PI;
  if ((raw - base) > CS_PI) return raw - 2 * CS_PI;
  return raw;
}
// the rarely taken exact decisions, out of line so that the sweep's loop body stays small
__device__ __attribute__((noinline)) float vp_rare_unwrapped_f(const double* mx, const double* my, int i, int ib, double vpxk, double vpyk) {
  return (float)vp_exact_unwrapped(mx, my, i, ib, vpxk, vpyk);
}
__device__ __attribute__((noinline)) bool vp_rare_greater(const double* mx, const double* my, int i, int i_ref, int ib, double vpxk, double vpyk) {
  return vp_exact_unwrapped(mx, my, i, ib, vpxk, vpyk) > vp_exact_unwrapped(mx, my, i_ref, ib, vpxk, vpyk);
}
__device__ __attribute__((noinline)) bool vp_rare_less(const double* mx, const double* my, int i, int i_ref, int ib, double vpxk, double vpyk) {
  return vp_exact_unwrapped(mx, my, i, ib, vpxk, vpyk) < vp_exact_unwrapped(mx, my, i_ref, ib, vpxk, vpyk);
}
// exact inlier test of one segment, with its angle as a float (returned by value: a callee that stores through a pointer would
// make the compiler assume the segment arrays may change, and their loads could no longer be scalar)
struct VpExactInlier { float at; int inl; };
__device__ __attribute__((noinline)) VpExactInlier vp_rare_inlier(double dyd, double dxd, double lai, double thre) {
  const double raw = cs_atan2(dyd, dxd);
  const double nrm = normalize_to_pi(raw);
  double d = dabs(lai - nrm);
  d = dmin(d, CS_PI - d);
  return VpExactInlier{(float)raw, d < thre ? 1 : 0};
}

struct VpRun {               // running state of one vanishing point's sweep over the segments
  bool have;
  int ib, i_hi, i_lo;        // first inlier, running arg-max / arg-min of the unwrapped angle
  float base_f, hi_f, lo_f;  // their float angles
  double hx, hy, lx, ly;     // mid point - vanishing point of the two running extremes
};

// atan2 in float, all quadrants: |error| < 2.5e-6 rad (tests/test_device_decisions.py; budget in the comment above).
// *usable = false for a zero or non-finite argument pair: the caller then decides exactly.
__device__ __forceinline__ float atan2_float(float fy, float fx, bool* usable) {
  const float PI_F = 3.14159274f, HPI_F = 1.57079637f;
  const float ay = fabsf(fy), ax = fabsf(fx);
  const float hi = fmaxf(ax, ay), lo = fminf(ax, ay);
  const float q = lo * __builtin_amdgcn_rcpf(hi), q2 = q * q;
  float at = -0.01172120f;
  at = __builtin_fmaf(at, q2, 0.05265332f);
  at = __builtin_fmaf(at, q2, -0.11643287f);
  at = __builtin_fmaf(at, q2, 0.19354346f);
  at = __builtin_fmaf(at, q2, -0.33262347f);
  at = __builtin_fmaf(at, q2, 0.99997726f);
  at = at * q;
  if (ay > ax) at = HPI_F - at;
  if (fx < 0.0f) at = PI_F - at;
  if (fy < 0.0f) at = -at;
  *usable = hi > 0.0f && hi < 3.0e38f;
  return at;
}

// one segment (mid point mxi, myi; angle lai) against one vanishing point
__device__ __forceinline__ void vp_step(const double* __restrict__ mx, const double* __restrict__ my, int i, double mxi, double myi, double lai,
                                        double vpxk, double vpyk, double thre, float thre_f, VpRun& R) {
  const float PI_F = 3.14159274f, HPI_F = 1.57079637f, M_IN = 1.0e-5f, M_ORD = 2.0e-5f;
  const double dyd = myi - vpyk, dxd = mxi - vpxk;
  bool usable;
  float at = atan2_float((float)dyd, (float)dxd, &usable);
  float nr = at;                           // normalize_to_pi: the distance below is circular, so which side of the fold a float lands on does not matter
  if (at > HPI_F) nr = at - PI_F; else if (at < -HPI_F) nr = at + PI_F;
  float df = fabsf((float)lai - nr);
  df = fminf(df, PI_F - df);
  bool inl = usable && df < thre_f - M_IN;              // (zero or non-finite differences are decided exactly)
  if (!(inl || (usable && df > thre_f + M_IN))) { const VpExactInlier x = vp_rare_inlier(dyd, dxd, lai, thre); inl = x.inl != 0; at = x.at; }
  if (inl) {
    if (!R.have) {  // first inlier: base of smooth_jump_angles (:278-302), and initial arg-max / arg-min
      R.have = true; R.ib = i; R.i_hi = i; R.i_lo = i; R.base_f = at; R.hi_f = at; R.lo_f = at; R.hx = dxd; R.hy = dyd; R.lx = dxd; R.ly = dyd;
    } else {
      const float d = at - R.base_f;
      float sh = d < -PI_F ? at + 2.0f * PI_F : d > PI_F ? at - 2.0f * PI_F : at;
      if (fabsf(fabsf(d) - PI_F) < M_ORD) sh = vp_rare_unwrapped_f(mx, my, i, R.ib, vpxk, vpyk);
      // Against the running extremes.  Far vanishing points see all segments within a fraction of a milliradian, closer than the
      // float angles resolve; inside the margin the order comes from the cross product of the two (double) difference vectors:
      // its sign is the order of the true angles (both unwrapped angles lie within the margin of each other, far from the
      // +-pi fold), reliable when it exceeds its rounding error -- 1e-12 of the product of the 1-norms, i.e. an angle
      // difference above 1e-12 rad, thousands of ulps of the rounded atan2 values.  Below that (equal or nearly equal
      // directions): the exact values decide, including the tie rule.
      if (sh > R.hi_f + M_ORD) { R.hi_f = sh; R.i_hi = i; R.hx = dxd; R.hy = dyd; }                 // maxCoeff: first occurrence, strict
      else if (sh > R.hi_f - M_ORD) {
        const double c = R.hx * dyd - R.hy * dxd, lim = 1.0e-12 * ((dabs(dxd) + dabs(dyd)) * (dabs(R.hx) + dabs(R.hy)));
        const bool greater = c > lim ? true : c < -lim ? false : vp_rare_greater(mx, my, i, R.i_hi, R.ib, vpxk, vpyk);
        if (greater) { R.hi_f = sh; R.i_hi = i; R.hx = dxd; R.hy = dyd; }
      }
      if (sh < R.lo_f - M_ORD) { R.lo_f = sh; R.i_lo = i; R.lx = dxd; R.ly = dyd; }                 // minCoeff
      else if (sh < R.lo_f + M_ORD) {
        const double c = R.lx * dyd - R.ly * dxd, lim = 1.0e-12 * ((dabs(dxd) + dabs(dyd)) * (dabs(R.lx) + dabs(R.ly)));
        const bool less = c < -lim ? true : c > lim ? false : vp_rare_less(mx, my, i, R.i_lo, R.ib, vpxk, vpyk);
        if (less) { R.lo_f = sh; R.i_lo = i; R.lx = dxd; R.ly = dyd; }
      }
    }
  }
}

// One wave's staging area for a chunk of 64 segments (the wave sits inside one job: every lane sweeps the same segments).
struct VpChunk {
  double2 xy[64];   // mid point
  double ang[64];   // segment angle
  float2 dir[64];   // unit direction (cosf, sinf: 2 ulp)
};

// K vanishing points swept together over the job's segments, 64 segments per chunk.
//   staging: lane u loads segment u of the chunk (one coalesced load per array) and its direction into LDS; both passes read
//     from there (pass 1: every lane the same address = broadcast; pass 2: a gather), not from global memory.
//   pass 1 (every segment, branch-free, ~10 instructions per vanishing point): drop what is certainly not an inlier.  With u the
//     segment's unit direction and d = mid point - vanishing point, the circular angle between them is below thre exactly
//     when |u x d| < tan(thre) |u . d|; in float, with thre widened by 1e-4 rad and 2e-6 (|dx| + |dy|) of slack for the
//     roundings (error budget: direction 3e-7 rad, each product / sum 6e-8 relative), no true inlier fails the test.
//     The survivors (about a fifth of the segments) are a bit mask per lane.
//   pass 2 (survivors, in order): vp_step -- float angle, the inlier decision, the running extremes, exact where a margin is hit.
// UNIFORM = false (a wave that spans jobs; only the round-based path produces those): no staging, every segment goes to vp_step.
// out[2k], out[2k+1] = the two bounding segment angles of vanishing point k (NaN = none): (max, min) when !swapped[k], else (min, max) (:609-614)
template <int K, bool UNIFORM>
__device__ __forceinline__ void vp_support_multi(const double* __restrict__ mx, const double* __restrict__ my, const double* __restrict__ la, int m, VpChunk* ch,
                                                 const double* vpx, const double* vpy, const double* thre, const bool* swapped, bool lane_on, double* out) {
  VpRun R[K];
  float thre_f[K], tan_f[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    R[k] = VpRun{false, 0, 0, 0, 0.f, 0.f, 0.f, 0., 0., 0., 0.}; thre_f[k] = (float)thre[k];
    tan_f[k] = thre_f[k] < 1.5f ? tanf(thre_f[k] + 1.0e-4f) : __builtin_huge_valf();      // (a threshold near 90 degrees keeps everything)
  }
  if (UNIFORM) {
    const int lane = threadIdx.x & 63;
    for (int c0 = 0; c0 < m; c0 += 64) {
      const int n = min(64, m - c0);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the previous chunk's readers are done (one wave, program order)
      if (lane < n) {
        const double a = la[c0 + lane];
        ch->xy[lane] = make_double2(mx[c0 + lane], my[c0 + lane]);
        ch->ang[lane] = a;
        ch->dir[lane] = make_float2(cosf((float)a), sinf((float)a));
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      unsigned long long mask[K];
#pragma unroll
      for (int k = 0; k < K; k++) mask[k] = 0;
#pragma unroll 4
      for (int u = 0; u < n; u++) {
        const double2 P = ch->xy[u];
        const float2 ud = ch->dir[u];
#pragma unroll
        for (int k = 0; k < K; k++) {
          const float fx = (float)(P.x - vpx[k]), fy = (float)(P.y - vpy[k]);
          const float cr = fabsf(__builtin_fmaf(fx, ud.y, -(fy * ud.x))), dt = fabsf(__builtin_fmaf(fx, ud.x, fy * ud.y));
          const bool out_for_sure = cr > __builtin_fmaf(tan_f[k], dt, 2.0e-6f * (fabsf(fx) + fabsf(fy)));      // (false for NaN / inf: those go on)
          mask[k] |= (unsigned long long)(out_for_sure ? 0 : 1) << u;
        }
      }
#pragma unroll
      for (int k = 0; k < K; k++) {
        unsigned long long mk = lane_on ? mask[k] : 0ull;
        while (__any(mk != 0)) {
          if (mk != 0) {
            const int u = __ffsll((long long)mk) - 1;
            mk &= mk - 1;
            const double2 P = ch->xy[u];
            vp_step(mx, my, c0 + u, P.x, P.y, ch->ang[u], vpx[k], vpy[k], thre[k], thre_f[k], R[k]);
          }
        }
      }
    }
  } else {
    const int mm = lane_on ? m : 0;
    for (int i = 0; i < mm; i++) {
      const double X = mx[i], Y = my[i], A = la[i];
#pragma unroll
      for (int k = 0; k < K; k++) vp_step(mx, my, i, X, Y, A, vpx[k], vpy[k], thre[k], thre_f[k], R[k]);
    }
  }
  const double NaN = __builtin_nan("");
#pragma unroll
  for (int k = 0; k < K; k++) {
    const double ang_hi = R[k].have ? la[R[k].i_hi] : NaN, ang_lo = R[k].have ? la[R[k].i_lo] : NaN;
    out[2 * k] = swapped[k] ? ang_lo : ang_hi;
    out[2 * k + 1] = swapped[k] ? ang_hi : ang_lo;
  }
}

enum { VP3_RPCAP = 32 };   // slots of the per-(job, roll/pitch sample) table of the third vanishing point's support angles

// MODE 0: everything in one kernel (vanishing points written, three VP supports).
// MODE 1: the lean path -- the vanishing points are produced by vp_points_kernel on another stream (they do not depend on
// the segments, so the corner construction runs beside line setup + VP support), and the support of the third
// (vertical) vanishing point, which does not depend on the yaw sample, comes from vp3_support_kernel's table.
template <int MODE, bool UNIFORM>
__device__ __forceinline__ void vp_support_body(const DetectDeviceView& v, const SweepParams& sp, const JobDesc& jd, int j, long long e, bool on, VpChunk* ch) {
  int rp = 0, y = 0;
  if (on) {
    int local = (int)e - jd.vp_off;
    on = local < jd.RP * jd.Y;             // (the lean path pads a job's entries to whole waves)
    rp = local / jd.Y; y = local - rp * jd.Y;
  }
  double vpx[3] = {0, 0, 0}, vpy[3] = {0, 0, 0};
  if (on) {
    const RpPose* pose = v.rp + jd.rp_off + rp;
    double cy = v.yaw_cos[jd.yaw_off + y], sy = v.yaw_sin[jd.yaw_off + y];
    const double* A = pose->KinvR;
    // getVanishingPoints (object_3d_util.cpp:928-937)
    double d[3][3] = {{cy, sy, 0.0}, {-sy, cy, 0.0}, {0.0, 0.0, 1.0}};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      double h0 = (A[0] * d[k][0] + A[1] * d[k][1]) + A[2] * d[k][2];
      double h1 = (A[3] * d[k][0] + A[4] * d[k][1]) + A[5] * d[k][2];
      double h2 = (A[6] * d[k][0] + A[7] * d[k][1]) + A[8] * d[k][2];
      vpx[k] = h0 / h2;
      vpy[k] = h1 / h2;
    }
    if (MODE == 0) {
      double* vout = v.vp + 6 * e;
      vout[0] = vpx[0]; vout[1] = vpy[0]; vout[2] = vpx[1]; vout[3] = vpy[1]; vout[4] = vpx[2]; vout[5] = vpy[2];
    }
  }
  const double* mx = v.mid_x + jd.line_off;
  const double* my = v.mid_y + jd.line_off;
  const double* la = v.line_angle + jd.line_off;
  double b6[6];
  const double thre[3] = {sp.vp12_thre_rad, sp.vp12_thre_rad, sp.vp3_thre_rad};
  const bool swapped[3] = {false, true, true};
  vp_support_multi<(MODE == 0 ? 3 : 2), UNIFORM>(mx, my, la, jd.m, ch, vpx, vpy, thre, swapped, on, b6);
  if (!on) return;
  if (MODE == 1) { const double* t3 = v.bound3 + 2 * ((size_t)j * VP3_RPCAP + rp); b6[4] = t3[0]; b6[5] = t3[1]; }
  double* bout = v.bound + 6 * e;
#pragma unroll
  for (int q = 0; q < 6; q++) bout[q] = b6[q];
}

template <int MODE>
__global__ __launch_bounds__(256) void vp_support_kernel(DetectDeviceView v, SweepParams sp, int vp_total) {
  long long e = xcd_virtual_block() * blockDim.x + threadIdx.x;
  const bool on = e < vp_total;
  const int j = find_job_wave<int>(v.vp_prefix, v.n_jobs, on ? (int)e : vp_total - 1);
  const int ju = wave_uniform_i32(j);
  __shared__ VpChunk chunk[4];
  if (__all(j == ju)) vp_support_body<MODE, true>(v, sp, v.jobs[ju], ju, e, on, &chunk[threadIdx.x >> 6]);
  else vp_support_body<MODE, false>(v, sp, v.jobs[j], j, e, on, nullptr);
}

// support angles of the third vanishing point, K R^-1 (0, 0, 1): one lane per (job, roll/pitch sample)
// rp_stride: lanes per job = the largest roll/pitch sample count of the launch's jobs (1 without roll/pitch sampling: a lane per job, 125 wavefronts for
// 8 000 jobs -- with VP3_RPCAP lanes per job it was 4 000 wavefronts of two live lanes each); the table keeps its VP3_RPCAP slots per job
__global__ __launch_bounds__(64) void vp3_support_kernel(DetectDeviceView v, SweepParams sp, int rp_stride) {
  const long long e = (long long)blockIdx.x * 64 + threadIdx.x;
  const int j = (int)(e / rp_stride), rp = (int)(e % rp_stride);
  bool on = j < v.n_jobs;
  JobDesc jd{};
  if (on) { jd = v.jobs[j]; on = rp < jd.RP && jd.Y > 0; }
  double vx = 0, vy = 0;
  if (on) {
    const double* A = (v.rp + jd.rp_off + rp)->KinvR;
    double h0 = (A[0] * 0.0 + A[1] * 0.0) + A[2] * 1.0;
    double h1 = (A[3] * 0.0 + A[4] * 0.0) + A[5] * 1.0;
    double h2 = (A[6] * 0.0 + A[7] * 0.0) + A[8] * 1.0;
    vx = h0 / h2; vy = h1 / h2;
  }
  double o2[2];
  const bool swapped3 = true;
  vp_support_multi<1, false>(v.mid_x + jd.line_off, v.mid_y + jd.line_off, v.line_angle + jd.line_off, on ? jd.m : 0, nullptr, &vx, &vy, &sp.vp3_thre_rad, &swapped3, on, o2);
  if (on) { double* t3 = v.bound3 + 2 * ((size_t)j * VP3_RPCAP + rp); t3[0] = o2[0]; t3[1] = o2[1]; }
}

// getVanishingPoints (object_3d_util.cpp:928-937) alone: the same arithmetic as the head of vp_support_kernel
__global__ __launch_bounds__(256) void vp_points_kernel(DetectDeviceView v, int vp_total) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int j = find_job_wave<int>(v.vp_prefix, v.n_jobs, e < vp_total ? (int)e : vp_total - 1);
  if (e >= vp_total) return;
  const JobDesc jd = v.jobs[j];
  int local = (int)e - jd.vp_off;
  if (local >= jd.RP * jd.Y) return;       // padding of the job's entries to whole waves
  int rp = local / jd.Y, y = local - rp * jd.Y;
  const RpPose* pose = v.rp + jd.rp_off + rp;
  double cy = v.yaw_cos[jd.yaw_off + y], sy = v.yaw_sin[jd.yaw_off + y];
  const double* A = pose->KinvR;
  double d[3][3] = {{cy, sy, 0.0}, {-sy, cy, 0.0}, {0.0, 0.0, 1.0}};
  double* vout = v.vp + 6 * e;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    double h0 = (A[0] * d[k][0] + A[1] * d[k][1]) + A[2] * d[k][2];
    double h1 = (A[3] * d[k][0] + A[4] * d[k][1]) + A[5] * d[k][2];
    double h2 = (A[6] * d[k][0] + A[7] * d[k][1]) + A[8] * d[k][2];
    vout[2 * k] = h0 / h2;
    vout[2 * k + 1] = h1 / h2;
  }
}

// ---- proposal geometry: one lane per slot -----------------------------------------------------------
// The eight corners (box_proposal_detail.cpp:413-625).  ~70 % of the slots are rejected here, so scoring runs as a
// second kernel over the compacted survivors (dense wavefronts) instead of inside this one.
__device__ __forceinline__ int candidate_one(const DetectDeviceView& v, const SweepParams& sp, const JobDesc& jd, long long tid) {
  // lane -> proposal: config-major inside the job so that a wave runs one configuration
  // (a job's slot count fits 32 bits: checked when the batch is laid out)
  const unsigned k = (unsigned)(tid - jd.slot_off);
  const unsigned half = (unsigned)jd.RP * (unsigned)jd.Y * (unsigned)jd.T;
  // (the lean roll/pitch path lays a job's slots out for the longest yaw list the box can get; the list in force may be a sample
  // shorter, and the slots behind the job's 2 * half proposals are empty)
  if (k >= 2 * half) { v.flag[tid] = 0; return 0; }
  const int cfg = (k >= half) ? 2 : 1;
  const unsigned rest = (k >= half) ? k - half : k;
  const unsigned ryu = rest / (unsigned)jd.T;      // rp * Y + yaw
  const int t = (int)(rest - ryu * (unsigned)jd.T);
  const long long slot = jd.slot_off + (long long)rest * 2 + (cfg - 1);
  const bool enabled = (cfg == 1) ? (sp.consider_config_1 != 0) : (sp.consider_config_2 != 0);
  int flag = 0;
  if (enabled) {
    const double* vp = v.vp + 6 * (long long)(jd.vp_off + (int)ryu);
    V2 c[8];
    // only the decision leaves this kernel (4 bytes per slot): whoever needs the corners of a valid proposal -- the scorer, the
    // winners' records, the tie boxes' gather -- rebuilds them from the same inputs with the same build_corners (slot_corners below:
    // the same instructions on the same operands, hence the same bits), instead of 128 bytes per valid slot going to memory and back
    flag = build_corners(jd.g, v2(vp[0], vp[1]), v2(vp[2], vp[3]), v2(vp[4], vp[5]), (double)v.top_x[jd.top_off + t], cfg, sp.short_sq_bound, c);
  }
  v.flag[slot] = flag;
  return flag;
}

__global__ __launch_bounds__(256) void candidate_kernel(DetectDeviceView v, SweepParams sp, long long slot_total) {
  long long tid = xcd_virtual_block() * blockDim.x + threadIdx.x;
  bool active = tid < slot_total;
  int flag = 0;
  int j = find_job_wave<long long>(v.slot_prefix, v.n_jobs, active ? tid : slot_total - 1);
  const int ju = wave_uniform_i32(j);
  if (__all(j == ju)) {          // the usual case: the job record is read once per wave (scalar loads), its fields live in SGPRs
    if (active) flag = candidate_one(v, sp, v.jobs[ju], tid);
  } else {
    if (active) flag = candidate_one(v, sp, v.jobs[j], tid);
  }
  // per-job valid count: one atomic per (wave, job) run
  unsigned long long valid = __ballot(flag != 0);
  if (valid) {
    int lane = threadIdx.x & 63;
    int j0 = __shfl(j, __ffsll((long long)valid) - 1);
    unsigned long long same = __ballot(flag != 0 && j == j0);
    if (same == valid) {
      if (lane == __ffsll((long long)valid) - 1) atomicAdd(&v.job_valid[j0], __popcll(valid));
    } else if (flag != 0) {
      atomicAdd(&v.job_valid[j], 1);
    }
  }
}

// Round 6: vanishing points + corner construction + ordered compaction of ONE job per workgroup (the lean path; replaces vp_points_kernel,
// candidate_kernel, scan_jobs_kernel and compact_kernel there).  The job record is uniform (scalar registers), the decisions of a trip of
// 4096 slots stay in LDS (a byte each) instead of going through a 4-byte flag per slot in memory, the valid count needs no atomics, and --
// with the compacted rows of job j starting at slot_prefix[j] (capacity layout, DetectDeviceView::blk_info) -- nothing waits for a scan over
// all jobs.  Lane -> proposal as in candidate_kernel: a wave runs one configuration (eight passes of 256 top-edge/yaw samples per
// configuration and trip); the compaction then walks the trip in slot order (configuration fastest: the reference's row order, :677-702).
enum { COMPACT_PER = 16 };     // groups of 64 slots a wave owns per compaction trip (compact_kernel below, candidate_compact_kernel)
enum { CC_TRIP = 4096 };
__global__ __launch_bounds__(256) void candidate_compact_kernel(DetectDeviceView v, SweepParams sp) {
  const int j = blockIdx.x;
  if (j >= v.n_jobs) return;
  __shared__ unsigned char s_flag[CC_TRIP];
  __shared__ int wsum[4];
  const JobDesc& jd = v.jobs[j];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // ---- getVanishingPoints (object_3d_util.cpp:928-937) of the job's (roll/pitch, yaw) samples: vp_points_kernel's arithmetic
  const int n_vp = jd.RP * jd.Y;
  for (int e = tid; e < n_vp; e += 256) {
    const int rp = e / jd.Y, y = e - rp * jd.Y;
    const RpPose* pose = v.rp + jd.rp_off + rp;
    const double cy = v.yaw_cos[jd.yaw_off + y], sy = v.yaw_sin[jd.yaw_off + y];
    const double* A = pose->KinvR;
    const double d[3][3] = {{cy, sy, 0.0}, {-sy, cy, 0.0}, {0.0, 0.0, 1.0}};
    double* vout = v.vp + 6 * (long long)(jd.vp_off + e);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double h0 = (A[0] * d[k][0] + A[1] * d[k][1]) + A[2] * d[k][2];
      const double h1 = (A[3] * d[k][0] + A[4] * d[k][1]) + A[5] * d[k][2];
      const double h2 = (A[6] * d[k][0] + A[7] * d[k][1]) + A[8] * d[k][2];
      vout[2 * k] = h0 / h2;
      vout[2 * k + 1] = h1 / h2;
    }
  }
  __syncthreads();        // (one workgroup, one CU: the rows written above are read back below)
  const unsigned half = (unsigned)n_vp * (unsigned)jd.T;            // proposals per configuration
  // rest / T by one multiplication (T is uniform; the generic 32-bit division is ~20 instructions per slot): m = floor(2^32 / T) + 1 gives
  // floor(rest m / 2^32) = floor(rest / T) whenever rest T < 2^32 (the error term rest (m T - 2^32) / (2^32 T) stays below 1 / T);
  // rest < half, so half T < 2^32 decides -- otherwise (never at any image size: n_vp T^2 in the millions) the division itself
  const unsigned Tu = (unsigned)jd.T;
  const bool fast_div = Tu > 1 && (unsigned long long)half * Tu < (1ull << 32);
  const unsigned inv_T = fast_div ? 0xffffffffu / Tu + 1u : 0u;
  const long long base = v.slot_prefix[j];
  long long run = base;
  const unsigned long long below = (1ull << lane) - 1ull;
  const bool en1 = sp.consider_config_1 != 0, en2 = sp.consider_config_2 != 0;
  for (unsigned r0 = 0; r0 < half; r0 += CC_TRIP / 2) {
    // ---- decisions of the trip's proposals
#pragma unroll 1
    for (int q = 0; q < 16; q++) {
      const int cfg = 1 + (q >> 3);
      const unsigned rl = (unsigned)(q & 7) * 256u + (unsigned)tid, rest = r0 + rl;
      int flag = 0;
      if (rest < half && (cfg == 1 ? en1 : en2)) {
        const unsigned ryu = fast_div ? __umulhi(rest, inv_T) : rest / Tu;
        const int t = (int)(rest - ryu * Tu);
        const double* vp = v.vp + 6 * (long long)(jd.vp_off + (int)ryu);
        V2 c[8];
        flag = build_corners(jd.g, v2(vp[0], vp[1]), v2(vp[2], vp[3]), v2(vp[4], vp[5]), (double)v.top_x[jd.top_off + t], cfg, sp.short_sq_bound, c);
      }
      s_flag[2 * rl + (unsigned)(cfg - 1)] = (unsigned char)flag;
    }
    __syncthreads();
    // ---- ordered compaction of the trip (compact_kernel's scheme on the LDS bytes): a wave owns 1024 consecutive slots
    int f[COMPACT_PER];
    int wcount = 0;
#pragma unroll
    for (int q = 0; q < COMPACT_PER; q++) {
      f[q] = s_flag[wid * (64 * COMPACT_PER) + q * 64 + lane];
      wcount += __popcll(__ballot(f[q] != 0));
    }
    if (lane == 0) wsum[wid] = wcount;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const int x = wsum[w]; if (w < wid) woff += x; total += x; }
    long long pos0 = run + woff;
    const long long sl0 = base + 2ll * r0 + wid * (64 * COMPACT_PER) + lane;
#pragma unroll
    for (int q = 0; q < COMPACT_PER; q++) {
      const unsigned long long bal = __ballot(f[q] != 0);
      if (f[q] != 0) {
        const long long pos = pos0 + __popcll(bal & below);
        v.c_slot[pos] = sl0 + q * 64;
        v.c_flag[pos] = f[q] | (j << CAND_JOB_SHIFT);
      }
      pos0 += __popcll(bal);
    }
    run += total;
    __syncthreads();                                    // s_flag and wsum are rewritten by the next trip
  }
  const int n_valid = (int)(run - base);
  if (tid == 0) { v.job_valid[j] = n_valid; v.job_cbase[j] = base; }
  // the scorer's work list: this job's blocks of 256 rows, appended as the job finishes (the list's order is roughly the jobs' order:
  // neighbouring entries read the same distance maps; it does not matter for any result)
  const int nblk = (n_valid + 255) >> 8;
  if (nblk > 0) {
    __shared__ int s_at;
    if (tid == 0) s_at = atomicAdd(&v.blk_info[0], nblk);
    __syncthreads();
    const int at = s_at, b0 = (int)(base >> 8);
    for (int k = tid; k < nblk; k += 256) {
      const int left = n_valid - 256 * k;
      v.blk_info[2 + 2 * (long long)(at + k)] = b0 + k;
      v.blk_info[3 + 2 * (long long)(at + k)] = left > 256 ? 256 : left;
    }
  }
}

// The eight corners of the proposal in `slot` of job jd (box_proposal_detail.cpp:413-625), rebuilt: slot = slot_off + ((rp Y + yaw) T
// + top) 2 + (config - 1).  Returns build_corners' flag.
__device__ __forceinline__ int slot_corners(const DetectDeviceView& v, const JobDesc& jd, long long slot, double short_sq_bound, V2 c[8]) {
  const unsigned local = (unsigned)(slot - jd.slot_off);
  const unsigned rest = local >> 1, ryu = rest / (unsigned)jd.T;
  const int t = (int)(rest - ryu * (unsigned)jd.T);
  const double* vp = v.vp + 6 * (long long)(jd.vp_off + (int)ryu);
  return build_corners(jd.g, v2(vp[0], vp[1]), v2(vp[2], vp[3]), v2(vp[4], vp[5]), (double)v.top_x[jd.top_off + t], (int)(local & 1) + 1, short_sq_bound, c);
}
// ... of a slot whose job is not at hand (winners, tie boxes: a handful per box)
__device__ __forceinline__ void slot_corners16(const DetectDeviceView& v, long long slot, double short_sq_bound, double out16[16]) {
  int lo = 0, hi = v.n_jobs;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (v.slot_prefix[mid] <= slot) lo = mid; else hi = mid; }
  V2 c[8];
#pragma unroll
  for (int q = 0; q < 8; q++) c[q] = v2(0.0, 0.0);
  slot_corners(v, v.jobs[lo], slot, short_sq_bound, c);
#pragma unroll
  for (int q = 0; q < 8; q++) { out16[q] = c[q].x; out16[8 + q] = c[q].y; }
}

// ---- proposal scoring: one lane per VALID proposal (compacted, in the reference's row order) ------------------
// box_edge_sum_dists (object_3d_util.cpp:622-667: 11 samples per visible edge, float gathers from the distance map,
// sequential float running sum), box_edge_alignment_angle_error (:670-723) and the 3D half sizes (:941-990).
// Both configurations share one code path: the edge / VP tables are indexed by configuration and the corners sit in
// LDS so that they can be indexed dynamically.
// edge tables as compile-time constants, selected per lane by configuration with register selects (no table loads)
//   config 1 (0): visible edges 1-2 2-3 3-4 4-1 2-6 3-5 4-8 5-8 5-6 (:646); VP edges 1-2,8-5 / 4-1,5-6 / 4-8,2-6 (:651)
//   config 2 (1): visible edges 1-2 2-3 3-4 4-1 2-6 3-5 5-6         (:663); VP edges 1-2,3-4 / 4-1,5-6 / 3-5,2-6 (:665)
__device__ __forceinline__ int sel(int cfg, int a, int b) { return cfg ? b : a; }

enum { SCORE_JOBS = 4, SCORE_SUB = 128, SCORE_BINS = SCORE_JOBS * SCORE_SUB };   // sort keys of score_kernel: (job within the block, configuration x top sample)
// CAP (round 6): the capacity layout -- a workgroup's proposals are all of ONE job, so the job record is read through a uniform index (scalar
// loads: its 34 words per lane were vector loads of one address) and the three integer divisions by T and Y per proposal are one
// multiplication each (reciprocals formed once; exact while n d < 2^32, the division itself otherwise -- as in candidate_compact_kernel).
__device__ __forceinline__ unsigned score_udiv(unsigned n, unsigned d, unsigned inv) { return inv ? __umulhi(n, inv) : n / d; }
template <bool CAP>
__global__ __launch_bounds__(256) void score_kernel(DetectDeviceView v, long long slot_total, double short_sq_bound) {
  // [coordinate: x0..x7, y0..y7][lane]: every lane keeps its proposal's corners in its own column (LDS because the edge tables index
  // them dynamically); lanes of a wave mostly ask for the same corner (sorted by configuration)
  __shared__ double C16[16][260];
  double (*CXt)[260] = C16, (*CYt)[260] = C16 + 8;
  // the grid is sized for the worst case (every slot valid) because the exact count lives on the device; spread the
  // ACTIVE blocks over the 8 XCDs (contiguous range per XCD), the surplus blocks exit immediately
  // (capacity layout, candidate_compact_kernel: the work list names the blocks of 256 rows that hold valid rows -- all of one job, leading the block)
  const bool cap = CAP;
  long long n_valid = cap ? 256ll * v.blk_info[0] : v.job_cbase[v.n_jobs];
  const long long per_xcd = ((n_valid + 255) / 256 + 7) / 8;
  const long long kx = blockIdx.x >> 3;
  if (kx >= per_xcd) return;
  long long base = ((long long)(blockIdx.x & 7) * per_xcd + kx) * blockDim.x;
  if (base >= n_valid) return;
  if (cap) {
    const int* entry = v.blk_info + 2 + 2 * (base >> 8);
    base = 256ll * entry[0];
    n_valid = base + entry[1];
  }
  // ---- who scores what.  The 256 proposals of this block are consecutive in the reference's order: configuration
  // fastest, then top-edge sample, then yaw.  Neighbouring lanes would gather from unrelated places of the distance map,
  // and a wave load that touches 64 different cache lines occupies the L1 tag pipeline for 64 cycles -- that, not the
  // arithmetic, used to bound this kernel.  Proposals that differ only in yaw (0.5 degrees: corners 1-2 px apart) sample
  // almost the same pixels, so the block re-sorts its proposals by (job, configuration, top-edge sample) with a counting
  // sort in LDS and every lane takes the proposal at its sorted position; results go back to the proposal's own index.
  __shared__ int hist[SCORE_BINS + 1];
  __shared__ int s_src[256], s_job[256], s_flag[256];
  __shared__ long long s_slot[256];
  {
    const int t0 = threadIdx.x;
    for (int b = t0; b <= SCORE_BINS; b += 256) hist[b] = 0;
    __syncthreads();
    const long long i0 = base + t0;
    long long slot0 = 0;
    int j0 = 0;
    int flag0 = 0;
    if (i0 < n_valid) { slot0 = v.c_slot[i0]; flag0 = v.c_flag[i0]; j0 = flag0 >> CAND_JOB_SHIFT; }
    s_slot[t0] = slot0; s_job[t0] = j0; s_flag[t0] = flag0 & CAND_VP_MASK;
    __syncthreads();
    int key = SCORE_BINS;                       // beyond the list: sorted last, skipped below
    if (i0 < n_valid) {
      const int jq = CAP ? __builtin_amdgcn_readfirstlane(s_job[0]) : j0;
      const int T0 = v.jobs[jq].T;
      const unsigned loc = (unsigned)(slot0 - v.jobs[jq].slot_off);
      const int jrel = j0 - s_job[0];                                                // jobs in this block, in order
      const unsigned hq = loc >> 1;
      const unsigned qT = CAP ? score_udiv(hq, (unsigned)T0, ((unsigned long long)v.jobs[jq].RP * v.jobs[jq].Y * T0 * T0 < (1ull << 32) && T0 > 1) ? 0xffffffffu / (unsigned)T0 + 1u : 0u) : hq / (unsigned)T0;
      const int sub = (int)(loc & 1) * T0 + (int)(hq - qT * (unsigned)T0);           // configuration-major, then top-edge sample
      key = (jrel < SCORE_JOBS && sub < SCORE_SUB) ? jrel * SCORE_SUB + sub : SCORE_BINS - 1;
    }
    const int rank = atomicAdd(&hist[key], 1);
    __syncthreads();
    if (t0 < 64) {                              // exclusive prefix over the bins: 64 lanes x (SCORE_BINS + 1) / 64 bins each
      constexpr int PER = (SCORE_BINS + 1 + 63) / 64;
      int loc_sum = 0, vals[PER];
#pragma unroll
      for (int q = 0; q < PER; q++) { const int b = t0 * PER + q; vals[q] = b <= SCORE_BINS ? hist[b] : 0; loc_sum += vals[q]; }
      int incl = loc_sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (t0 >= o) incl += y; }
      int run = incl - loc_sum;
#pragma unroll
      for (int q = 0; q < PER; q++) { const int b = t0 * PER + q; if (b <= SCORE_BINS) hist[b] = run; run += vals[q]; }
    }
    __syncthreads();
    s_src[hist[key] + rank] = t0;
    __syncthreads();
  }
  const int mine = s_src[threadIdx.x];
  const JobDesc& jd = v.jobs[CAP ? __builtin_amdgcn_readfirstlane(s_job[0]) : s_job[mine]];
  const long long i = base + mine;
  if (i >= n_valid) return;                     // (no barrier below this point)
  const long long slot = s_slot[mine];
  const unsigned local = (unsigned)(slot - jd.slot_off);
  const int cfg = (int)(local & 1);          // 0 = configuration 1
  const unsigned inv_T = (CAP && jd.T > 1 && (unsigned long long)jd.RP * jd.Y * jd.T * jd.T < (1ull << 32)) ? 0xffffffffu / (unsigned)jd.T + 1u : 0u;
  const unsigned inv_Y = (CAP && jd.Y > 1 && (unsigned long long)jd.RP * jd.Y * jd.Y < (1ull << 32)) ? 0xffffffffu / (unsigned)jd.Y + 1u : 0u;
  const int ry = (int)score_udiv(local >> 1, (unsigned)jd.T, inv_T);
  const int rp = (int)score_udiv((unsigned)ry, (unsigned)jd.Y, inv_Y);
  const int tx = threadIdx.x;
  const double ox = (double)jd.g.el, oy = (double)jd.g.et;
  const float* __restrict__ map = v.maps + jd.map_off;
  // the proposal's corners, rebuilt (candidate_kernel kept only the decision): into this lane's own LDS column, no barrier needed.
  // Round 4: values only (rebuild_accepted_corners, cs_geom.h: build_corners' value expressions without its decisions -- the second
  // attempt of the first ray, the miss sentinels, the inside-box tests and the 13 edge-length tests), the side from the kept flag
  {
    V2 cb[8];
    const unsigned rest = local >> 1;
    const double* vp = v.vp + 6 * (long long)(jd.vp_off + ry);
    rebuild_accepted_corners(jd.g, v2(vp[0], vp[1]), v2(vp[2], vp[3]), v2(vp[4], vp[5]), (double)v.top_x[jd.top_off + (int)(rest - (unsigned)ry * (unsigned)jd.T)], cfg + 1,
                             s_flag[mine], cb);
#pragma unroll
    for (int q = 0; q < 8; q++) { CXt[q][tx] = cb[q].x; CYt[q][tx] = cb[q].y; }
  }
  // (the six VP-support angles of the angle term are requested here, ahead of the gathers: one round trip less on the block's path)
  double bnd[6];
  {
    const double* bound = v.bound + 6 * (long long)(jd.vp_off + ry);
#pragma unroll
    for (int q = 0; q < 6; q++) bnd[q] = bound[q];
  }
  // ---- distance error: all gathers of an edge are issued before its (sequential, float) accumulation
  float sum_dist = 0;
  // corner ids of the 9 edges, one nibble each (edge 0 lowest): {0,1,2,3,1,2,3,4,4}-{1,2,3,0,5,4,7,7,5} / {0,1,2,3,1,2,4,0,0}-{1,2,3,0,5,4,5,0,0}
  const unsigned long long EA = cfg ? 0x004213210ull : 0x443213210ull, EB = cfg ? 0x005450321ull : 0x577450321ull;
  // config 2 reweights edges 4, 5 by 3/2 and edge 6 by 2 (:655-661), one weight nibble per edge in halves.  The reference
  // computes float(double(d) * 3.0 / 2.0) and float(double(d) * 2.0): both products are exact in double, so the one rounding
  // to float is the rounding of the float product d * 1.5f (d * 2.0f), and d * 1.0f is d.
  const unsigned long long EW = cfg ? 0x004332222ull : 0x222222222ull;
  const int n_edges = cfg ? 7 : 9;
  const int map_w = jd.map_w;
  constexpr int EU = 3;   // edges per trip: their 11 * EU gathers are in flight together
  const bool wave_has_cfg1 = __any(cfg == 0);  // round 4: a wavefront of configuration-2 proposals has no edges 7 and 8 (it mostly is one
                                               // configuration after the re-sort), so its last trip gathers one edge instead of three
#pragma unroll 1
  for (int e0 = 0; e0 < n_edges; e0 += EU) {
    float dv[EU][11];
#pragma unroll
    for (int u = 0; u < EU; u++) {
      const int e = e0 + u;                    // (beyond the list: nibble 0 = corner 1, a valid address; the sum skips it)
      if (u > 0 && e0 == 6 && !wave_has_cfg1) {
#pragma unroll
        for (int s = 0; s < 11; s++) dv[u][s] = 0.0f;
        continue;
      }
      const int a = (int)((EA >> (4 * e)) & 7), b = (int)((EB >> (4 * e)) & 7);
      const double x1 = CXt[a][tx] - ox, y1 = CYt[a][tx] - oy, x2 = CXt[b][tx] - ox, y2 = CYt[b][tx] - oy;
#pragma unroll
      for (int s = 0; s < 11; s++) {
        // s / 10 * p1 + (1 - s / 10) * p2 (object_3d_util.cpp:645-652).  Round 4: s = 0 and s = 10 ARE the end points (0 * a + 1 * b: the
        // product with 0 is +-0, the product with 1 exact, and the sum with +-0 leaves b unless b is itself a zero, whose sign the
        // integer cast drops); s = 5 is (a + b) * 0.5 (both halves are exact, the single rounding happens in the sum either way)
        double sx, sy;
        if (s == 0) { sx = x2; sy = y2; }
        else if (s == 10) { sx = x1; sy = y1; }
        else if (s == 5) { sx = (x1 + x2) * 0.5; sy = (y1 + y2) * 0.5; }
        else { const double w = (double)s / 10.0; sx = w * x1 + (1 - w) * x2; sy = w * y1 + (1 - w) * y2; }
        // samples lie inside the ROI the map covers (corners were tested against it): row * width + column fits 24 x 24 -> 32 bits
        dv[u][s] = map[(unsigned)(__mul24((int)sy, map_w) + (int)sx)];
      }
    }
#pragma unroll
    for (int u = 0; u < EU; u++) {
      const int e = e0 + u;
      const float wt = 0.5f * (float)(int)((EW >> (4 * e)) & 7);
      const bool on = e < n_edges;
#pragma unroll
      for (int s = 0; s < 11; s++) { const float nx = sum_dist + dv[u][s] * wt; sum_dist = on ? nx : sum_dist; }
    }
  }
  // ---- angle alignment error
  double total = 0;
  const double not_found_penalty = 30.0 / 180.0 * CS_PI * 2;
  const int ID1[3][4] = {{0, 1, 7, 4}, {3, 0, 4, 5}, {3, 7, 1, 5}}, ID2[3][4] = {{0, 1, 2, 3}, {3, 0, 4, 5}, {2, 4, 1, 5}};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    double b0 = bnd[2 * k], b1 = bnd[2 * k + 1];
    bool v0 = !(b0 != b0), v1 = !(b1 != b1);
    if (v0 || v1) {
#pragma unroll
      for (int ee = 0; ee < 2; ee++) {
        int pa = sel(cfg, ID1[k][2 * ee], ID2[k][2 * ee]), pb = sel(cfg, ID1[k][2 * ee + 1], ID2[k][2 * ee + 1]);
        double ang = normalize_to_pi(cs_atan2(CYt[pb][tx] - CYt[pa][tx], CXt[pb][tx] - CXt[pa][tx]));
        double best = 100;
        if (v0) { double t = dabs(ang - b0); t = dmin(t, CS_PI - t); if (t < best) best = t; }
        if (v1) { double t = dabs(ang - b1); t = dmin(t, CS_PI - t); if (t < best) best = t; }
        total = total + best;
      }
    } else {
      total = total + not_found_penalty;
    }
  }
  // ---- half sizes of the lifted cuboid -> skew ratio, negative-scale flag
  V2 c[8];
#pragma unroll
  for (int q = 0; q < 8; q++) c[q] = v2(CXt[q][tx], CYt[q][tx]);
  const RpPose* pose = v.rp + jd.rp_off + rp;
  double p3[3], s3[3];
  lift_to_3d(c, pose->R, pose->t, v.invK + 9 * jd.frame, pose->plane, p3, s3);
  int flag = s_flag[mine];
  if (s3[0] < 0 || s3[1] < 0 || s3[2] < 0) flag |= CAND_NEG_SCALE;
  v.c_flag[i] = flag;
  v.c_dist[i] = (double)sum_dist / jd.diag;
  v.c_angle[i] = total;
  v.c_skew[i] = dmax(s3[0], s3[1]) / dmin(s3[0], s3[1]);
}

// Exclusive scan of job_valid -> job_cbase (n_jobs + 1).  Single block.
__global__ __launch_bounds__(1024) void scan_jobs_kernel(const int* job_valid, long long* job_cbase, int n_jobs) {
  __shared__ long long wsum[16];
  __shared__ long long carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int base = 0; base < n_jobs; base += 1024) {
    int i = base + threadIdx.x;
    long long x = (i < n_jobs) ? job_valid[i] : 0;
    long long incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      long long y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    long long woff = 0;
    for (int w = 0; w < wid; w++) woff += wsum[w];
    long long carry = carry_s;
    if (i < n_jobs) job_cbase[i] = carry + woff + incl - x;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) job_cbase[n_jobs] = carry_s;
}

// Ordered compaction: one block per job walks the job's slots in slot order, 4096 per trip (a C2 job's 3 943 slots are one trip, a
// roll/pitch job's 79 k twenty).  Each wave owns 1024 consecutive slots as COMPACT_PER groups of 64 (lane = slot inside the
// group: coalesced loads, ballot for the order inside a group, a running count from group to group); one barrier per trip gives
// the waves their offsets.
__global__ __launch_bounds__(256) void compact_kernel(DetectDeviceView v) {
  int j = blockIdx.x;
  if (j >= v.n_jobs) return;
  __shared__ int wsum[4];
  const long long s0 = v.slot_prefix[j], s1 = v.slot_prefix[j + 1];
  long long run = v.job_cbase[j];                       // (every thread carries the same running base)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (long long base = s0; base < s1; base += 256 * COMPACT_PER) {
    const long long w0 = base + (long long)wid * (64 * COMPACT_PER) + lane;
    int f[COMPACT_PER];
    int wcount = 0;
#pragma unroll
    for (int q = 0; q < COMPACT_PER; q++) {
      const long long sl = w0 + q * 64;
      f[q] = (sl < s1) ? v.flag[sl] : 0;
      wcount += __popcll(__ballot(f[q] != 0));
    }
    if (lane == 0) wsum[wid] = wcount;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const int x = wsum[w]; if (w < wid) woff += x; total += x; }
    long long pos0 = run + woff;
#pragma unroll
    for (int q = 0; q < COMPACT_PER; q++) {
      const unsigned long long bal = __ballot(f[q] != 0);
      if (f[q] != 0) {
        const long long pos = pos0 + __popcll(bal & below);
        v.c_slot[pos] = w0 + q * 64;
        v.c_flag[pos] = f[q] | (j << CAND_JOB_SHIFT);
      }
      pos0 += __popcll(bal);
    }
    run += total;
    __syncthreads();                                    // wsum is rewritten by the next trip
  }
}

// The same compaction for jobs of many trips (roll/pitch sampling: 25 poses, ~79 k slots, twenty trips that the one workgroup per job
// above walks one after the other -- 0.13 ms per round on ~100 jobs): a workgroup per (job, trip).  compact_count_kernel counts the
// valid slots of every trip, compact_chunk_kernel starts a trip at the job's base + the counts of the job's earlier trips.
__global__ __launch_bounds__(256) void compact_count_kernel(DetectDeviceView v, int* __restrict__ cnt, int maxc) {
  const int j = blockIdx.x, c = blockIdx.y;
  __shared__ int wsum[4];
  const long long end = v.slot_prefix[j + 1], s0 = v.slot_prefix[j] + (long long)c * (256 * COMPACT_PER), s1 = (s0 + 256 * COMPACT_PER < end) ? s0 + 256 * COMPACT_PER : end;
  int n = 0;
  for (long long sl = s0 + threadIdx.x; sl < s1; sl += 256) n += v.flag[sl] != 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) cnt[(size_t)j * maxc + c] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(256) void compact_chunk_kernel(DetectDeviceView v, const int* __restrict__ cnt, int maxc) {
  const int j = blockIdx.x, c = blockIdx.y;
  __shared__ int wsum[4];
  const long long s1 = v.slot_prefix[j + 1], base = v.slot_prefix[j] + (long long)c * (256 * COMPACT_PER);
  if (base >= s1) return;
  long long run = v.job_cbase[j];
  for (int q = 0; q < c; q++) run += cnt[(size_t)j * maxc + q];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  const long long w0 = base + (long long)wid * (64 * COMPACT_PER) + lane;
  int f[COMPACT_PER];
  int wcount = 0;
#pragma unroll
  for (int q = 0; q < COMPACT_PER; q++) {
    const long long sl = w0 + q * 64;
    f[q] = (sl < s1) ? v.flag[sl] : 0;
    wcount += __popcll(__ballot(f[q] != 0));
  }
  if (lane == 0) wsum[wid] = wcount;
  __syncthreads();
  int woff = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) if (w < wid) woff += wsum[w];
  long long pos0 = run + woff;
#pragma unroll
  for (int q = 0; q < COMPACT_PER; q++) {
    const unsigned long long bal = __ballot(f[q] != 0);
    if (f[q] != 0) {
      const long long pos = pos0 + __popcll(bal & below);
      v.c_slot[pos] = w0 + q * 64;
      v.c_flag[pos] = f[q] | (j << CAND_JOB_SHIFT);
    }
    pos0 += __popcll(bal);
  }
}

__global__ __launch_bounds__(64) void gather_corners_kernel(DetectDeviceView v, double short_sq_bound, const long long* slots, int n, double* out) {
  int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n) return;
  long long s = slots[w];
  double c16[16];
#pragma unroll
  for (int q = 0; q < 16; q++) c16[q] = 0.0;
  if (s >= 0) slot_corners16(v, s, short_sq_bound, c16);
#pragma unroll
  for (int q = 0; q < 16; q++) out[16 * (size_t)w + q] = c16[q];
}


// ------------------------------------------------------------------ ranking on the device --------
// fuse_normalize_scores_v2 (object_3d_util.cpp:726-837) keeps the best 2/3 by distance error, intersects with the
// best 2/3 by angle error when the angle errors do not saturate at the cut, min-max normalises both over the kept
// set and fuses them; box_proposal_detail.cpp:804-838 adds the skew penalty and takes the max_cuboid_num smallest.
// The reference does this with std::partial_sort, whose order among *equal* keys is an artefact of the heap.  The
// results here are order statistics and arg-mins, which are identical to the reference's whenever no tie straddles
// a cut or the final top-k; when one does, the box is flagged and the host stage (exact std::partial_sort) redoes it.
__device__ __forceinline__ unsigned long long order_key(double x) {
  unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// radix selection, 8 passes of 8 bits over order_key(): all NT threads of the block must call it (NT = 64, 128 or 256).
// Two columns at once (their passes share the barriers): keys of rank r (0-based, ascending) among a[0..n) and among b[0..n)
template <int NT>
__device__ void block_radix_select2(const double* __restrict__ a, const double* __restrict__ b, int n, int r, unsigned* hist /* LDS, 512 */, unsigned long long* bcast /* LDS, 4 */,
                                    unsigned long long* key_a, unsigned long long* key_b) {
  unsigned long long prefix[2] = {0, 0}, mask = 0;
  int rr2[2] = {r, r};
  for (int pass = 0; pass < 8; pass++) {
    const int shift = 56 - 8 * pass;
    for (int q = threadIdx.x; q < 512; q += NT) hist[q] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += NT) {
      const unsigned long long ka = order_key(a[i]), kb = order_key(b[i]);
      if ((ka & mask) == prefix[0]) atomicAdd(&hist[(unsigned)(ka >> shift) & 255u], 1u);
      if ((kb & mask) == prefix[1]) atomicAdd(&hist[256 + ((unsigned)(kb >> shift) & 255u)], 1u);
    }
    __syncthreads();
    // one wave per column (waves 0 and 1; a single-wave block does both in turn), every lane owns 4 consecutive bins
    for (int col = 0; col < 2; col++) {
      if ((int)(threadIdx.x >> 6) != (NT >= 128 ? col : 0)) continue;
      const int lane = threadIdx.x & 63;
      const unsigned* h = hist + 256 * col;
      unsigned c0 = h[4 * lane], c1 = h[4 * lane + 1], c2 = h[4 * lane + 2], c3 = h[4 * lane + 3];
      unsigned tot = c0 + c1 + c2 + c3, incl = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { unsigned y = __shfl_up(incl, o); if (lane >= o) incl += y; }
      unsigned excl = incl - tot;
      const unsigned rk = (unsigned)rr2[col];
      if (rk >= excl && rk < incl) {
        unsigned rr = rk - excl, d;
        if (rr < c0) { d = 0; }
        else if (rr < c0 + c1) { d = 1; rr -= c0; }
        else if (rr < c0 + c1 + c2) { d = 2; rr -= c0 + c1; }
        else { d = 3; rr -= c0 + c1 + c2; }
        bcast[2 * col] = (unsigned long long)(4 * lane + d);
        bcast[2 * col + 1] = rr;
      }
    }
    __syncthreads();
    prefix[0] |= bcast[0] << shift; prefix[1] |= bcast[2] << shift;
    mask |= 255ull << shift;
    rr2[0] = (int)bcast[1]; rr2[1] = (int)bcast[3];
    __syncthreads();
  }
  *key_a = prefix[0]; *key_b = prefix[1];
}

template <int NT>
__device__ __forceinline__ double block_reduce_min(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { double y = __shfl_down(v, o); v = (y < v) ? y : v; }
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = sh[0];
  for (int w = 1; w < NT / 64; w++) r = (sh[w] < r) ? sh[w] : r;
  __syncthreads();
  return r;
}
template <int NT>
__device__ __forceinline__ double block_reduce_max(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { double y = __shfl_down(v, o); v = (y > v) ? y : v; }
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = sh[0];
  for (int w = 1; w < NT / 64; w++) r = (sh[w] > r) ? sh[w] : r;
  __syncthreads();
  return r;
}
template <int NT>
__device__ __forceinline__ int block_reduce_sum_i(int v, int* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  int r = sh[0];
  for (int w = 1; w < NT / 64; w++) r += sh[w];
  __syncthreads();
  return r;
}

struct JobCut {      // per height sample of the box
  double vd, va;     // keep a proposal iff dist <= vd (and angle <= va when use_angle)
  double dmin, dmax, amin, amax;
  int use_angle, n_keep, V;
  int tie;           // several proposals share the cut value vd, and which of them the reference keeps depends on its heap order
};

// (rank_wave_kernel, below rank_kernel: boxes whose height samples hold <= RW_CAP valid proposals each)
enum { RW_K = 28, RW_CAP = 64 * RW_K, RW_BATCH = 8 };      // (a lane's candidate bits are one 32-bit word)
// does the box go to the wavefront kernel?  (uniform; both kernels ask)
__device__ __forceinline__ bool rank_box_fits_wave(const DetectDeviceView& v, int j0, int nj) {
  bool fits = true;
  for (int h = 0; h < nj; h++) fits = fits && v.job_valid[j0 + h] <= (int)RW_CAP;
  return fits;
}
// the score of a kept proposal (fuse_normalize_scores_v2's normalisation + the skew penalty): ONE definition for both ranking kernels
__device__ __forceinline__ double rank_combined_score(const JobCut& c, const RankParams& rp, double d, double a, double sk, double* score_out) {
  double score;
  if (c.n_keep > 1) {
    double dn = (d - c.dmin) / (c.dmax - c.dmin);
    double an = ((c.amax - c.amin) > 0) ? (a - c.amin) / (c.amax - c.amin) : a;
    score = (dn + rp.w_angle * an) / (1 + rp.w_angle);
  } else {
    score = (d + rp.w_angle * a) / (1 + rp.w_angle);
  }
  double skew_error = rp.w_skew * dmax(sk - rp.nominal_skew, 0.0);
  if (sk > rp.max_cut_skew) skew_error = 100;
  *score_out = score;
  return score + rp.w_skew * skew_error;
}
enum { RANK_STAGE = 1536, RANK_THREADS_DEFAULT = 256 };   // proposals of one height sample staged in LDS (2 x 12 KB)
// DYN: the staging columns in dynamic LDS, stage_cap proposals each -- the roll/pitch-sampling rounds, where a box holds ~9 000 valid
// proposals (25 camera poses) and only ~100 boxes are ranked per launch: a workgroup then owns a CU's LDS and 16 waves, and the ~20
// selection passes run from LDS instead of L2
enum { RANK_STAGE_BIG = 9216, RANK_THREADS_BIG = 1024 };
// (the body: rank_kernel's workgroups, and -- NT = 64, no staging columns -- rank_wave_kernel's wavefront for a box with a height sample
// too large for its registers)
template <int NT, bool DYN>
__device__ __forceinline__ void rank_block_body(const DetectDeviceView& v, const RankView& rv, const RankParams& rp, int stage_cap, int box) {
  __shared__ unsigned hist[512];
  __shared__ unsigned long long bcast[4];
  __shared__ double shd[NT / 64 > 4 ? NT / 64 : 4];
  __shared__ int shi[NT / 64 > 4 ? NT / 64 : 4];
  __shared__ JobCut cuts[3];
  __shared__ int s_fallback;
  __shared__ double sD_fixed[DYN ? 1 : RANK_STAGE], sA_fixed[DYN ? 1 : RANK_STAGE];
  extern __shared__ double rank_dyn_lds[];
  double* const sD = DYN ? rank_dyn_lds : sD_fixed;
  double* const sA = DYN ? rank_dyn_lds + stage_cap : sA_fixed;
  const int STAGE = DYN ? stage_cap : (int)RANK_STAGE;
  const int j0 = rv.box_job0[box], nj = rv.box_njobs[box];
  if (threadIdx.x == 0) s_fallback = 0;
  __syncthreads();
  const double INF = __builtin_huge_val();
  // ---- per height sample: the cuts of fuse_normalize_scores_v2 and the min/max over the kept set
  for (int h = 0; h < nj; h++) {
    int j = j0 + h;
    long long c0 = v.job_cbase[j];
    int V = v.job_valid[j];
    const double* D = v.c_dist + c0;
    const double* A = v.c_angle + c0;
    // the selection below makes ~20 passes over the two columns: from LDS when they fit (they do unless nearly every slot of a
    // large box is valid), each pass then costs LDS latency instead of a round trip to L2
    if (V <= STAGE) {
      __syncthreads();
      for (int i = threadIdx.x; i < V; i += NT) { sD[i] = D[i]; sA[i] = A[i]; }
      __syncthreads();
      D = sD; A = sA;
    }
    double vd = INF, va = INF;
    int use_angle = 0, tie = 0, bn_keep = 0;
    if (V > 4) {
      int bn = (int)round((double)((float)V) / 3.0 * 2.0);
      bn_keep = bn - 1;
      unsigned long long kd, ka;
      block_radix_select2<NT>(D, A, V, bn - 2, hist, bcast, &kd, &ka);
      int cd = 0, ca = 0, nan = 0;
      for (int i = threadIdx.x; i < V; i += NT) {
        double d = D[i], a = A[i];
        cd += order_key(d) <= kd; ca += order_key(a) <= ka;
        nan += (d != d) || (a != a);
      }
      cd = block_reduce_sum_i<NT>(cd, shi); ca = block_reduce_sum_i<NT>(ca, shi); nan = block_reduce_sum_i<NT>(nan, shi);
      // the (bn-1)-th and bn-th smallest distance errors may coincide (float sums: 4 % of the boxes): then which of the tied
      // proposals is kept depends on the heap order of std::partial_sort.  That is decided below -- the box only goes to the
      // host when the choice can change the output
      if (nan && threadIdx.x == 0) s_fallback = 1;
      tie = (cd != bn - 1);
      use_angle = (ca == bn - 1);  // angle[sorted[bn-1]] > angle[sorted[bn-2]] (:766)
      // thresholds as doubles: the largest kept value
      double md = -INF, ma = -INF;
      for (int i = threadIdx.x; i < V; i += NT) {
        if (order_key(D[i]) <= kd) md = (D[i] > md) ? D[i] : md;
        if (order_key(A[i]) <= ka) ma = (A[i] > ma) ? A[i] : ma;
      }
      vd = block_reduce_max<NT>(md, shd);
      va = block_reduce_max<NT>(ma, shd);
    }
    double dmin = 1e6, dmax = -1, amin = 1e6, amax = -1;  // :798-801
    int nk = 0;
    for (int i = threadIdx.x; i < V; i += NT) {
      double d = D[i], a = A[i];
      bool keep = (d <= vd) && (!use_angle || a <= va);
      if (keep) {
        nk++;
        dmin = (d < dmin) ? d : dmin; dmax = (dmax < d) ? d : dmax;
        amin = (a < amin) ? a : amin; amax = (amax < a) ? a : amax;
      }
    }
    dmin = block_reduce_min<NT>(dmin, shd); dmax = block_reduce_max<NT>(dmax, shd);
    amin = block_reduce_min<NT>(amin, shd); amax = block_reduce_max<NT>(amax, shd);
    nk = block_reduce_sum_i<NT>(nk, shi);
    if (tie) {
      // The reference keeps all proposals below the cut value ("sure") and r >= 1 of the ct proposals at it.  The constants
      // above were taken over sure + ALL tied proposals (that pass the angle cut).  They equal the reference's whatever it
      // picks iff: two or more sure proposals are kept (same normalisation formula, dmin from a sure one), the tied ones do
      // not extend the angle range, and -- with the angle cut on -- either none of the tied ones passes it or the reference
      // cannot avoid keeping one that does (dmax = the cut value either way).  Then a tied proposal only matters if it wins.
      double amin_s = 1e6, amax_s = -1;
      int nk_s = 0, nt = 0, ct = 0, cl = 0;
      for (int i = threadIdx.x; i < V; i += NT) {
        double d = D[i], a = A[i];
        const bool pass = !use_angle || a <= va;
        if (d < vd) { cl++; if (pass) { nk_s++; amin_s = (a < amin_s) ? a : amin_s; amax_s = (amax_s < a) ? a : amax_s; } }
        else if (d == vd) { ct++; nt += pass; }
      }
      amin_s = block_reduce_min<NT>(amin_s, shd); amax_s = block_reduce_max<NT>(amax_s, shd);
      nk_s = block_reduce_sum_i<NT>(nk_s, shi); nt = block_reduce_sum_i<NT>(nt, shi); ct = block_reduce_sum_i<NT>(ct, shi); cl = block_reduce_sum_i<NT>(cl, shi);
      const int r = bn_keep - cl;                  // tied proposals the reference keeps
      const bool safe = nk_s >= 2 && amin_s == amin && amax_s == amax && (nt == 0 || r - (ct - nt) >= 1);
      if (!safe && threadIdx.x == 0) s_fallback = 1;
    }
    if (rv.last_slot && h == nj - 1) {
      // The last element of the kept list (object_3d_util.cpp:760-781).  V <= 4: every proposal is kept, in order.  With the
      // angle cut the list is a sorted-id intersection: its last element is the kept proposal with the largest index.
      // Without it the list is the first bn - 1 of the distance order: its last element is THE proposal whose distance is the
      // cut value -- when two share that value, or the cut itself is tied, the reference's heap order decides: host.
      int last = -1, n_at_cut = 0;
      if (V > 4) {
        for (int i = threadIdx.x; i < V; i += NT) {
          const double d = D[i], a = A[i];
          if (use_angle) { if (d <= vd && a <= va) last = i; }
          else if (d == vd) { last = i; n_at_cut++; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_down(last, o); last = y > last ? y : last; }
        if ((threadIdx.x & 63) == 0) shi[threadIdx.x >> 6] = last;
        __syncthreads();
        { int mx = shi[0]; for (int w = 1; w < NT / 64; w++) mx = max(mx, shi[w]); last = mx; }
        __syncthreads();
        n_at_cut = block_reduce_sum_i<NT>(n_at_cut, shi);
        if ((tie || (!use_angle && n_at_cut != 1)) && threadIdx.x == 0) s_fallback = 1;
      } else {
        last = V - 1;
      }
      if (threadIdx.x == 0) rv.last_slot[box] = last >= 0 ? v.c_slot[c0 + last] : -1;
    }
    if (threadIdx.x == 0) {
      JobCut c; c.vd = vd; c.va = va; c.dmin = dmin; c.dmax = dmax; c.amin = amin; c.amax = amax; c.use_angle = use_angle; c.n_keep = nk; c.V = V; c.tie = tie;
      cuts[h] = c;
    }
    __syncthreads();
  }
  // ---- final ranking over the proposals of all height samples: kmax rounds of arg-min, one pass over the proposals per round
  // (every thread keeps its own minimum, how many of its proposals attain it, and that proposal's record)
  const bool staged = (nj == 1) && (cuts[0].V <= STAGE);   // the columns of the only height sample are still in LDS
  double prev = -INF;
  int n_win = 0;
  for (int round = 0; round < rp.kmax; round++) {
    double best = INF, w_score = 0, w_d = 0, w_a = 0;
    long long w_at = 0;
    int bad = 0, cnt_local = 0;
    for (int h = 0; h < nj; h++) {
      const JobCut c = cuts[h];
      long long c0 = v.job_cbase[j0 + h];
      for (int i = threadIdx.x; i < c.V; i += NT) {
        double d = staged ? sD[i] : v.c_dist[c0 + i], a = staged ? sA[i] : v.c_angle[c0 + i];
        bool keep = (d <= c.vd) && (!c.use_angle || a <= c.va);
        if (!keep || (v.c_flag[c0 + i] & CAND_NEG_SCALE)) continue;
        double score;
        const double comb = rank_combined_score(c, rp, d, a, v.c_skew[c0 + i], &score);
        if (comb != comb || comb == INF || comb == -INF) { bad = 1; continue; }  // NaN / inf: let the host decide
        if (!(comb > prev)) continue;
        const int weight = (c.tie && d == c.vd) ? 3 : 1;   // a winner among the tied proposals: the reference may not have kept it -> host
        if (comb < best) { best = comb; cnt_local = weight; w_score = score; w_d = d; w_a = a; w_at = c0 + i; }
        else if (comb == best) cnt_local += weight;
      }
    }
    bad = block_reduce_sum_i<NT>(bad, shi);
    double gbest = block_reduce_min<NT>(best, shd);
    if (bad && threadIdx.x == 0) s_fallback = 1;
    if (!(gbest < INF)) break;  // no proposal left
    // how many proposals attain the minimum?  (> 1: the reference's pick depends on the heap order)
    const bool mine = best == gbest;
    int cnt = block_reduce_sum_i<NT>(mine ? cnt_local : 0, shi);
    if (mine) {   // the (unique, unless the box goes to the host anyway) owner writes the winner record
      RankWinner* w = rv.winners + (size_t)box * rp.kmax + round;
      long long slot = v.c_slot[w_at];
      w->slot = slot; w->normalized_error = w_score; w->dist_err = w_d; w->angle_err = w_a; w->flag = v.c_flag[w_at] & CAND_VP_MASK; w->pad = 0;
      // (its corners: winner_corners_kernel, behind the ranking -- one lane per winner instead of one lane of this workgroup)
    }
    if (cnt != 1 && threadIdx.x == 0) s_fallback = 1;
    prev = gbest;
    n_win++;
  }
  __syncthreads();
  if (threadIdx.x == 0) { rv.win_count[box] = n_win; rv.fallback[box] = s_fallback; }
}
template <int NT, bool DYN>
__global__ __launch_bounds__(NT) void rank_kernel(DetectDeviceView v, RankView rv, RankParams rp, int stage_cap) {
  if ((int)blockIdx.x < rv.n_boxes) rank_block_body<NT, DYN>(v, rv, rp, stage_cap, blockIdx.x);
}


// ------------------------------------------------------------------ line setup on the device ------
// Per (box, height sample): the ROI line filter (box_proposal_detail.cpp:271-283), merge_break_lines
// (object_3d_util.cpp:431-543) and the angle / midpoint tables (:309-315), one wavefront per job.
//
// merge_break_lines is sequential -- merge the lexicographically first pair (a, b), a < b, that passes three tests
// (angle difference, end-point gap, angle of the merged segment), overwrite row a, move the last row into b,
// restart -- and its result depends on that order, so the order is reproduced exactly.  Rescanning all pairs every
// round (what the reference does) costs O(m^2) pair tests per merge.  Round 5: the wave keeps the whole PAIR-TEST MATRIX as
// bits -- PM[a] = the set of partners b > a of row a -- built once (every lane owns the rows lane, lane + 64, ... and tests them
// against row b, broadcast from LDS, for b = 1 .. m-1; the two cheap tests first, the third only for their few survivors), and a
// round is: the first row with a partner (one ballot per 64 rows), its first partner (a find-first-set), the merge, and a repair
// of the matrix -- only pairs that involve the two rewritten rows can change: every row tests itself against the merged row (the
// merged row's own partner set is those tests' ballot), the row that moved into b inherits the bits of the old last row from the
// rows before it and tests the rows behind it.  No lane ever scans for a first partner again.  (Rounds 1-4 kept F[a] = the first
// partner only: every lane walked its row's partners until the first hit -- the wave as long as its unluckiest lane -- and a
// rewritten first partner meant a cooperative re-scan: 23 k wave instructions per job, now ~10 k.)
//
// Angles.  Only DECISIONS need a row's angle during the merges, so rows carry it as a float (atan2_float, 2.5e-6 rad) and the
// two angle tests are taken on floats with the exact evaluation (cs_atan2 of the row's end points -- the value the reference
// holds, whether the row is an input segment or a merge product) inside a margin; the exact angles of the rows that survive are
// computed once at the end, in parallel, where they leave the kernel.
enum { LS_CAP = 512, LS_THREADS = 64 };

struct LineSetupParams {
  double dist_thre, angle_thre_rad, len_thre;  // 20 px, 5 deg, 30 px (:288-290)
  double dist_sq_bound;                        // sqrt(x) < dist_thre <=> x < dist_sq_bound   (cs_geom.h sqrt_lt_bound)
  double len_sq_bound;                         // sqrt(x) > len_thre  <=> x > len_sq_bound    (sqrt_le_bound)
};

struct LsRow { float af; double x1, y1, x2, y2; };   // af: the float angle, NaN = unusable (zero-length / non-finite): decided exactly

__device__ __forceinline__ float ls_angf(double x1, double y1, double x2, double y2) {
  bool usable;
  const float a = atan2_float((float)(y2 - y1), (float)(x2 - x1), &usable);
  return usable ? a : __builtin_nanf("");
}
// first test of a pair, exactly as the reference evaluates it (rare: only inside the float test's margin)
__device__ __attribute__((noinline)) bool ls_rare_angle_diff(double ax1, double ay1, double ax2, double ay2, double bx1, double by1, double bx2, double by2, double thre) {
  const double diff = dabs(cs_atan2(ay2 - ay1, ax2 - ax1) - cs_atan2(by2 - by1, bx2 - bx1));
  return dmin(diff, CS_PI - diff) < thre;
}
// third test of a pair, exactly as the reference evaluates it (rare)
__device__ __attribute__((noinline)) bool ls_rare_angle_close(double ax1, double ay1, double ax2, double ay2, double dy, double dx, double thre) {
  const double t = dabs(cs_atan2(ay2 - ay1, ax2 - ax1) - cs_atan2(dy, dx));
  return dmin(t, CS_PI - t) < thre;
}
// The three tests of object_3d_util.cpp:464-497 for the ordered pair (A = the row with the lower index, B).  Only decisions leave
// these functions: float angles (each within 2.5e-6 rad of the row's exact angle, so a float difference is within 5.4e-6 of the exact
// one: margin 1.5e-5, the exact evaluation inside it), end-point gaps compared squared (sqrt_lt_bound).  Tests 1 + 2 are symmetric
// in A and B; test 3 is not (it compares the merged segment's angle with A's and breaks end-point ties towards B).
#define LS_MARGIN 1.5e-5f
__device__ __forceinline__ bool ls_t12(const LsRow& A, const LsRow& B, const LineSetupParams& lp, float thf) {
  float df = fabsf(A.af - B.af);
  df = fminf(df, 3.14159274f - df);
  bool p1 = df < thf - LS_MARGIN;
  if (!p1 && !(df > thf + LS_MARGIN)) p1 = ls_rare_angle_diff(A.x1, A.y1, A.x2, A.y2, B.x1, B.y1, B.x2, B.y2, lp.angle_thre_rad);   // (a NaN lands here)
  if (!p1) return false;
  const double d_ab = v2_dist2(v2(A.x2, A.y2), v2(B.x1, B.y1));
  const double d_ba = v2_dist2(v2(B.x2, B.y2), v2(A.x1, A.y1));
  return (d_ab < lp.dist_sq_bound) || (d_ba < lp.dist_sq_bound);
}
// test 3 of the pair {P, Q}; p_is_a: P is the row with the lower index (A).  The merged segment starts at A's start if A.x1 < B.x1,
// else at B's (a tie goes to B), ends at A's end if A.x2 > B.x2, else at B's, and its angle is compared with A's (rows hold finite
// coordinates: a NaN end point is never inside the ROI).
__device__ __forceinline__ bool ls_t3(const LsRow& P, const LsRow& Q, bool p_is_a, const LineSetupParams& lp, float thf) {
  const bool sp = p_is_a ? (P.x1 < Q.x1) : !(Q.x1 < P.x1), ep = p_is_a ? (P.x2 > Q.x2) : !(Q.x2 > P.x2);
  const double sx = sp ? P.x1 : Q.x1, sy = sp ? P.y1 : Q.y1, ex = ep ? P.x2 : Q.x2, ey = ep ? P.y2 : Q.y2;
  const double dy = ey - sy, dx = ex - sx;
  bool usable;
  const float at = atan2_float((float)dy, (float)dx, &usable);
  const float tf = fabsf((p_is_a ? P.af : Q.af) - at);
  const float mf = fminf(tf, 3.14159274f - tf);
  if (usable && mf < thf - LS_MARGIN) return true;
  if (usable && mf > thf + LS_MARGIN) return false;
  return p_is_a ? ls_rare_angle_close(P.x1, P.y1, P.x2, P.y2, dy, dx, lp.angle_thre_rad) : ls_rare_angle_close(Q.x1, Q.y1, Q.x2, Q.y2, dy, dx, lp.angle_thre_rad);
}

// Two instances share the jobs.  A typical ROI holds a few dozen segments: CAP = LS_SMALL rows, one wavefront, 6.6 KB of LDS -- the
// kernel is bound by instruction issue, and a CU holds every wave slot's worth of such jobs.  The few crowded ROIs (6 % at C2) go to the
// CAP = LS_CAP instance: what counts there is the latency of the longest job (the batch's chain waits for it), so a job gets four
// wavefronts, rows tid and tid + 256 per thread -- each wavefront's ballot is one 64-bit word of a partner set.  Each instance counts
// the ROI's segments first and leaves the jobs of the other size class alone.
enum { LS_SMALL = 128, LS_MID = 256, LS_CROWDED_THREADS = 64 };
// (one job: `listed` = the job comes from line_classify_kernel's list of crowded ROIs -- no counting pass, no size-class test)
template <int CAP, int NT>
__device__ __forceinline__ void line_setup_job(int j, bool listed, JobDesc* jobs, int n_jobs, const double* __restrict__ frame_lines, const int* __restrict__ frame_line_ptr,
                                               double* mid_x, double* mid_y, double* line_angle, const LineSetupParams& lp) {
  constexpr int SL = CAP / NT;                  // rows per thread: tid + NT s
  constexpr int W = CAP / 64;                   // 64-bit words per partner set; row a's own bit sits in word a >> 6 = wave + NW s
  constexpr int NW = NT / 64;
  __shared__ double X1[CAP], Y1[CAP], X2[CAP], Y2[CAP];
  __shared__ float AF[CAP];
  __shared__ unsigned long long PM[CAP * W];    // PM[r * W + w]: the partners b in [64 w, 64 w + 64) of row r (bits b > r only)
  __shared__ int CAND[NW > 1 ? NW : 1];
  __shared__ int TOT;
  if (j >= n_jobs) return;
  const JobDesc jd = jobs[j];
  if (jd.Y == 0 || jd.T == 0) return;   // a box the sweep skips (no yaw / top-edge samples): m stays 0
  const double* FL = frame_lines + 4 * (size_t)frame_line_ptr[jd.frame];
  const int M = frame_line_ptr[jd.frame + 1] - frame_line_ptr[jd.frame];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float thf = (float)lp.angle_thre_rad;
  const int NONE = 0x7fffffff;
  if (!listed) {  // size class of this job: segments with both end points inside the expanded ROI (every wavefront counts for itself)
    int n_in = 0;
    for (int base = 0; base < M; base += 64) {
      const int i = base + lane;
      bool in = false;
      if (i < M) in = inside_box(v2(FL[4 * i], FL[4 * i + 1]), jd.g.el, jd.g.et, jd.g.er, jd.g.eb) && inside_box(v2(FL[4 * i + 2], FL[4 * i + 3]), jd.g.el, jd.g.et, jd.g.er, jd.g.eb);
      n_in += __popcll(__ballot(in));
    }
    if ((n_in <= LS_SMALL) != (CAP == LS_SMALL)) return;
  }
  auto row = [&](int i) { return LsRow{AF[i], X1[i], Y1[i], X2[i], Y2[i]}; };

  // ---- 1. segments with both end points inside the expanded ROI, in input order (the first wavefront)
  if (wave == 0) {
    int total = 0;
    for (int base = 0; base < M; base += 64) {
      int i = base + lane;
      double x1 = 0, y1 = 0, x2 = 0, y2 = 0;
      bool in = false;
      if (i < M) {
        x1 = FL[4 * i]; y1 = FL[4 * i + 1]; x2 = FL[4 * i + 2]; y2 = FL[4 * i + 3];
        in = inside_box(v2(x1, y1), jd.g.el, jd.g.et, jd.g.er, jd.g.eb) && inside_box(v2(x2, y2), jd.g.el, jd.g.et, jd.g.er, jd.g.eb);
      }
      unsigned long long bal = __ballot(in);
      int off = total + __popcll(bal & ((1ull << lane) - 1ull));
      if (in && off < CAP) { X1[off] = x1; Y1[off] = y1; X2[off] = x2; Y2[off] = y2; AF[off] = ls_angf(x1, y1, x2, y2); }
      total = min(CAP, total + __popcll(bal));
    }
    if (lane == 0) TOT = total;
  }
  for (int e = tid; e < CAP * W; e += NT) PM[e] = 0ull;
  __syncthreads();
  int total = TOT;
  // ---- 2. the pair-test matrix.  Own rows in registers; tests 1 + 2 against every later row b (broadcast), then test 3 for the survivors
  LsRow own[SL];
  int cnt[SL];
#pragma unroll
  for (int s = 0; s < SL; s++) {
    const int a = tid + NT * s;
    own[s] = (a < total) ? row(a) : LsRow{0.f, 0., 0., 0., 0.};
    cnt[s] = 0;
  }
  for (int b = 1; b < total; b++) {
    const LsRow Rb = row(b);
#pragma unroll
    for (int s = 0; s < SL; s++) {
      if (NT * s >= b) break;
      const int a = tid + NT * s;
      if (a < b && ls_t12(own[s], Rb, lp, thf)) PM[a * W + (b >> 6)] |= 1ull << (b & 63);
    }
  }
#pragma unroll
  for (int s = 0; s < SL; s++) {
    if (NT * s >= total) break;
    const int a = tid + NT * s;
    if (a < total) {
      for (int w = a >> 6; 64 * w < total; w++) {
        unsigned long long v = PM[a * W + w], keep = v;
        while (v) {
          const int b = 64 * w + __ffsll((long long)v) - 1;
          v &= v - 1;
          if (!ls_t3(own[s], row(b), true, lp, thf)) keep &= ~(1ull << (b & 63));
        }
        PM[a * W + w] = keep;
        cnt[s] += __popcll(keep);
      }
    }
  }
  __syncthreads();
  // ---- 3. merge rounds
  for (int rounds = 0; rounds < CAP; rounds++) {
    // the first row with a partner
    int as = NONE;
#pragma unroll
    for (int s = 0; s < SL; s++) {
      if (NT * s >= total) break;
      const unsigned long long bal = __ballot(cnt[s] > 0 && tid + NT * s < total);
      if (bal && as == NONE) as = 64 * (wave + NW * s) + __ffsll((long long)bal) - 1;
    }
    if (NW > 1) {
      if (lane == 0) CAND[wave] = as;
      __syncthreads();
#pragma unroll
      for (int q = 0; q < NW; q++) as = min(as, CAND[q]);
    }
    if (as == NONE) break;
    int bs = -1;
    for (int w = as >> 6; 64 * w < total; w++) {
      const unsigned long long v = PM[as * W + w];
      if (v) { bs = 64 * w + __ffsll((long long)v) - 1; break; }
    }
    const int last = total - 1;
    const LsRow Ra = row(as), Rb = row(bs), RL = row(last);
    LsRow Mg;                                   // the merged row (:499-523): the wider start / end of the two
    {
      const bool sa = Ra.x1 < Rb.x1, ea = Ra.x2 > Rb.x2;
      Mg.x1 = sa ? Ra.x1 : Rb.x1; Mg.y1 = sa ? Ra.y1 : Rb.y1; Mg.x2 = ea ? Ra.x2 : Rb.x2; Mg.y2 = ea ? Ra.y2 : Rb.y2;
      Mg.af = ls_angf(Mg.x1, Mg.y1, Mg.x2, Mg.y2);
    }
    const bool moved = (bs != last);            // row bs takes what used to be the last row (fast_RemoveRow, matrix_utils.cpp:183)
    __syncthreads();                            // (everybody holds the three rows: they may be rewritten)
    if (tid == 0) {
      X1[as] = Mg.x1; Y1[as] = Mg.y1; X2[as] = Mg.x2; Y2[as] = Mg.y2; AF[as] = Mg.af;
      if (moved) { X1[bs] = RL.x1; Y1[bs] = RL.y1; X2[bs] = RL.x2; Y2[bs] = RL.y2; AF[bs] = RL.af; }
    }
    total = last;
#pragma unroll
    for (int s = 0; s < SL; s++) {
      const int wd = wave + NW * s;             // this wavefront's word of a partner set, for its rows of slot s
      if (NT * s >= total) {                    // (no rows here any more: the two rebuilt rows hold nothing in these words)
        if (lane == 0) { PM[as * W + wd] = 0ull; if (moved) PM[bs * W + wd] = 0ull; }
        continue;
      }
      const int a = tid + NT * s;
      if (a == as) own[s] = Mg;
      else if (moved && a == bs) own[s] = RL;
      const bool valid = a < total;
      // tests 1 + 2 of this row against the merged row and -- behind bs -- against the row that moved there
      bool c_as = valid && a != as && ls_t12(own[s], Mg, lp, thf);
      bool c_bs = moved && valid && a > bs && ls_t12(RL, own[s], lp, thf);
      // test 3 for the survivors through one call site: every lane takes its pending test against the merged row first, then the one
      // against the moved row (a second trip only if some lane has both)
      bool p_as = false, p_bs = false;
      while (__ballot(c_as | c_bs)) {
        const bool vs_merged = c_as, pending = c_as | c_bs;
        const LsRow Q = vs_merged ? Mg : RL;
        const bool r = pending && ls_t3(own[s], Q, vs_merged && a < as, lp, thf);
        if (vs_merged) { p_as = r; c_as = false; } else if (pending) { p_bs = r; c_bs = false; }
      }
      // the two rebuilt rows' partner sets are these tests' ballots
      const unsigned long long bal_as = __ballot(p_as && a > as), bal_bs = __ballot(p_bs);
      if (lane == 0) { PM[as * W + wd] = bal_as; if (moved) PM[bs * W + wd] = bal_bs; }
      // every other row: its bit for the old last row goes; its bit for `as` is the fresh test; its bit for `bs` is what its bit for the old last row was
      if (valid && a != as && !(moved && a == bs)) {
        unsigned long long* pm = PM + a * W;
        int d = 0;
        bool old_last = false;
        { const unsigned long long v = pm[last >> 6]; old_last = (v >> (last & 63)) & 1ull; if (old_last) { pm[last >> 6] = v & ~(1ull << (last & 63)); d--; } }
        if (a < as) { const unsigned long long v = pm[as >> 6]; const bool old = (v >> (as & 63)) & 1ull; if (old != p_as) { pm[as >> 6] = v ^ (1ull << (as & 63)); d += p_as ? 1 : -1; } }
        if (moved && a < bs) { const unsigned long long v = pm[bs >> 6]; const bool old = (v >> (bs & 63)) & 1ull; if (old != old_last) { pm[bs >> 6] = v ^ (1ull << (bs & 63)); d += old_last ? 1 : -1; } }
        cnt[s] += d;
      }
    }
    __syncthreads();
    // the owners of the two rebuilt rows count their new partner sets
#pragma unroll
    for (int s = 0; s < SL; s++) {
      const int a = tid + NT * s;
      if (a == as || (moved && a == bs)) {
        int c = 0;
        for (int w = a >> 6; w < W; w++) c += __popcll(PM[a * W + w]);
        cnt[s] = c;
      }
    }
  }
  // ---- 4. keep segments longer than the threshold (in order) and emit the angle (exact: cs_atan2 of the row's end points) / midpoint tables
  if (wave == 0) {
    int kept = 0;
    for (int base = 0; base < total; base += 64) {
      int i = base + lane;
      bool keep = false;
      if (i < total) {
        double len2 = v2_dist2(v2(X2[i], Y2[i]), v2(X1[i], Y1[i]));
        keep = (lp.len_thre > 0) ? (len2 > lp.len_sq_bound) : true;
      }
      unsigned long long bal = __ballot(keep);
      int off = kept + __popcll(bal & ((1ull << lane) - 1ull));
      if (keep) {
        line_angle[jd.line_off + off] = cs_atan2(Y2[i] - Y1[i], X2[i] - X1[i]);
        mid_x[jd.line_off + off] = (X1[i] + X2[i]) / 2;
        mid_y[jd.line_off + off] = (Y1[i] + Y2[i]) / 2;
      }
      kept += __popcll(bal);
    }
    if (lane == 0) jobs[j].m = kept;
  }
}
template <int CAP, int NT>
__global__ __launch_bounds__(NT) void line_setup_kernel(JobDesc* jobs, int n_jobs, const double* __restrict__ frame_lines, const int* __restrict__ frame_line_ptr,
                                                       double* mid_x, double* mid_y, double* line_angle, LineSetupParams lp, const int* __restrict__ order) {
  const int j = order ? order[blockIdx.x] : blockIdx.x;   // longest jobs first (the kernel ends with its slowest workgroup)
  line_setup_job<CAP, NT>(j, false, jobs, n_jobs, frame_lines, frame_line_ptr, mid_x, mid_y, line_angle, lp);
}
// Round 6, last.  line_setup_kernel above is launched with a workgroup per job of the batch, and 94 % of them count their ROI's segments and leave
// again -- each holding the instance's 50 KB of LDS while it counts: ~7 500 such workgroups per batch, three at a time on a CU, beside the other
// batches' kernels (what that costs: a second, 17 KB instance launched the same way for the ROIs of <= 256 rows took 5.5 % OFF the sweep's rate;
// doubling this instance's LDS took 5-7 %).  The lean path asks first: line_classify_kernel (a wavefront per job, no LDS) appends the crowded jobs to
// a list, and this kernel's workgroups -- a fixed, modest grid -- take their jobs from it.
// crowded[0] = how many, crowded[1 ..] = the jobs (in the order the atomics fell: the jobs are independent).
__global__ __launch_bounds__(256) void line_classify_kernel(const JobDesc* __restrict__ jobs, int n_jobs, const double* __restrict__ frame_lines, const int* __restrict__ frame_line_ptr,
                                                            int* __restrict__ crowded, int* __restrict__ crowded_big) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= n_jobs) return;
  const JobDesc& jd = jobs[j];
  if (jd.Y == 0 || jd.T == 0) return;
  const int f = jd.frame, el = jd.g.el, et = jd.g.et, er = jd.g.er, eb = jd.g.eb;
  const double* FL = frame_lines + 4 * (size_t)frame_line_ptr[f];
  const int M = frame_line_ptr[f + 1] - frame_line_ptr[f];
  int n_in = 0;
  for (int base = 0; base < M; base += 64) {
    const int i = base + lane;
    bool in = false;
    if (i < M) in = inside_box(v2(FL[4 * i], FL[4 * i + 1]), el, et, er, eb) && inside_box(v2(FL[4 * i + 2], FL[4 * i + 3]), el, et, er, eb);
    n_in += __popcll(__ballot(in));
  }
  // (two lists: ROIs of up to LS_MID rows -- their instance holds 17 KB of LDS -- and the rest, 50 KB; `crowded_big` null: one list for both)
  if (n_in > LS_SMALL && lane == 0) {
    int* list = (crowded_big && n_in > LS_MID) ? crowded_big : crowded;
    list[1 + atomicAdd(list, 1)] = j;
  }
}
template <int CAP, int NT>
__global__ __launch_bounds__(NT) void line_setup_listed_kernel(JobDesc* jobs, int n_jobs, const double* __restrict__ frame_lines, const int* __restrict__ frame_line_ptr,
                                                              double* mid_x, double* mid_y, double* line_angle, LineSetupParams lp, const int* __restrict__ crowded) {
  const int n = crowded[0];
  for (int k = blockIdx.x; k < n; k += gridDim.x) {
    line_setup_job<CAP, NT>(crowded[1 + k], true, jobs, n_jobs, frame_lines, frame_line_ptr, mid_x, mid_y, line_angle, lp);
    __syncthreads();      // (the next job rewrites the rows and the matrix)
  }
}

// The typical job -- at most 128 segments inside the ROI -- on one wavefront with the pair-test matrix in REGISTERS: lane l owns the rows
// l and l + 64; row l's partner set is two words (partners below / from 64 on), row l + 64's one.  Same procedure as line_setup_kernel
// above (which keeps the matrix in LDS and serves the crowded ROIs), written without data-dependent branches: both cheap tests are
// evaluated for every pair, the exact evaluations sit behind wave-uniform ballots, bit updates select their word with a uniform branch.
__device__ __forceinline__ bool ls_t12_flat(const LsRow& A, const LsRow& B, const LineSetupParams& lp, float thf) {
  float df = fabsf(A.af - B.af);
  df = fminf(df, 3.14159274f - df);
  bool p1 = df < thf - LS_MARGIN;
  const bool unsure = !p1 && !(df > thf + LS_MARGIN);             // (a NaN lands here)
  const double d_ab = v2_dist2(v2(A.x2, A.y2), v2(B.x1, B.y1));
  const double d_ba = v2_dist2(v2(B.x2, B.y2), v2(A.x1, A.y1));
  const bool p2 = (d_ab < lp.dist_sq_bound) || (d_ba < lp.dist_sq_bound);
  if (__ballot(unsure && p2)) { if (unsure && p2) p1 = ls_rare_angle_diff(A.x1, A.y1, A.x2, A.y2, B.x1, B.y1, B.x2, B.y2, lp.angle_thre_rad); }
  return p1 && p2;
}
__device__ __forceinline__ void ls_setbit(unsigned long long& lo, unsigned long long& hi, int pos, bool val) {   // pos: wave-uniform
  const unsigned long long bit = 1ull << (pos & 63);
  if (pos < 64) lo = val ? (lo | bit) : (lo & ~bit); else hi = val ? (hi | bit) : (hi & ~bit);
}
__device__ __forceinline__ bool ls_getbit(unsigned long long lo, unsigned long long hi, int pos) {
  return (((pos < 64) ? lo : hi) >> (pos & 63)) & 1ull;
}
__global__ __launch_bounds__(64) void line_setup_small_kernel(JobDesc* jobs, int n_jobs, const double* __restrict__ frame_lines, const int* __restrict__ frame_line_ptr,
                                                              double* mid_x, double* mid_y, double* line_angle, LineSetupParams lp, const int* __restrict__ order) {
  constexpr int CAP = LS_SMALL;
  __shared__ double X1[CAP], Y1[CAP], X2[CAP], Y2[CAP];
  __shared__ float AF[CAP];
  int j = order ? order[blockIdx.x] : blockIdx.x;   // longest jobs first (the kernel ends with its slowest workgroup)
  if (j >= n_jobs) return;
  const JobDesc jd = jobs[j];
  if (jd.Y == 0 || jd.T == 0) return;   // a box the sweep skips (no yaw / top-edge samples): m stays 0
  const double* FL = frame_lines + 4 * (size_t)frame_line_ptr[jd.frame];
  const int M = frame_line_ptr[jd.frame + 1] - frame_line_ptr[jd.frame];
  const int lane = threadIdx.x;
  const float thf = (float)lp.angle_thre_rad;
  const int NONE = 0x7fffffff;
  auto row = [&](int i) { return LsRow{AF[i], X1[i], Y1[i], X2[i], Y2[i]}; };
  // ---- 1. segments with both end points inside the expanded ROI, in input order; more than LS_SMALL: the other instance's job
  int total = 0;
  for (int base = 0; base < M; base += 64) {
    int i = base + lane;
    double x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    bool in = false;
    if (i < M) {
      x1 = FL[4 * i]; y1 = FL[4 * i + 1]; x2 = FL[4 * i + 2]; y2 = FL[4 * i + 3];
      in = inside_box(v2(x1, y1), jd.g.el, jd.g.et, jd.g.er, jd.g.eb) && inside_box(v2(x2, y2), jd.g.el, jd.g.et, jd.g.er, jd.g.eb);
    }
    unsigned long long bal = __ballot(in);
    int off = total + __popcll(bal & ((1ull << lane) - 1ull));
    if (in && off < CAP) { X1[off] = x1; Y1[off] = y1; X2[off] = x2; Y2[off] = y2; AF[off] = ls_angf(x1, y1, x2, y2); }
    total += __popcll(bal);
  }
  if (total > CAP) return;
  __syncthreads();
  // ---- 2. the pair-test matrix: tests 1 + 2 of the own rows against every later row b (broadcast), then test 3 for the survivors
  LsRow own0 = (lane < total) ? row(lane) : LsRow{0.f, 0., 0., 0., 0.};
  LsRow own1 = (lane + 64 < total) ? row(lane + 64) : LsRow{0.f, 0., 0., 0., 0.};
  unsigned long long m00 = 0ull, m01 = 0ull, m11 = 0ull;    // row lane: partners < 64 / >= 64; row lane + 64: partners >= 64
  for (int b = 1; b < total; b++) {
    const LsRow Rb = row(b);
    const bool t0 = ls_t12_flat(own0, Rb, lp, thf) && lane < b;
    const unsigned long long bit = 1ull << (b & 63);
    if (b < 64) m00 |= t0 ? bit : 0ull;
    else {
      m01 |= t0 ? bit : 0ull;
      if (b > 64) { const bool t1 = ls_t12_flat(own1, Rb, lp, thf) && lane + 64 < b; m11 |= t1 ? bit : 0ull; }
    }
  }
  {
    // every lane walks the survivors of its rows (lowest partner first; three words)
#pragma unroll
    for (int q = 0; q < 3; q++) {
      unsigned long long v = q == 0 ? m00 : (q == 1 ? m01 : m11), keep = v;
      const int wbase = q == 0 ? 0 : 64;
      while (__ballot(v != 0ull)) {
        const bool have = v != 0ull;
        const int b = have ? wbase + __ffsll((long long)v) - 1 : 0;
        const unsigned long long lowbit = v & (0ull - v);
        v ^= lowbit;
        const bool ok = have && ls_t3(q == 2 ? own1 : own0, row(b), true, lp, thf);
        if (have && !ok) keep ^= lowbit;
      }
      if (q == 0) m00 = keep; else if (q == 1) m01 = keep; else m11 = keep;
    }
  }
  // ---- 3. merge rounds
  for (int rounds = 0; rounds < CAP; rounds++) {
    const int first0 = m00 ? __ffsll((long long)m00) - 1 : (m01 ? 64 + __ffsll((long long)m01) - 1 : NONE);
    const int first1 = m11 ? 64 + __ffsll((long long)m11) - 1 : NONE;
    int as, bs;
    const unsigned long long bal0 = __ballot(first0 != NONE);
    if (bal0) {
      as = __ffsll((long long)bal0) - 1;
      bs = __builtin_amdgcn_readlane(first0, as);
    } else {
      const unsigned long long bal1 = __ballot(first1 != NONE);
      if (!bal1) break;
      const int l = __ffsll((long long)bal1) - 1;
      as = 64 + l;
      bs = __builtin_amdgcn_readlane(first1, l);
    }
    const int last = total - 1;
    const LsRow Ra = row(as), Rb = row(bs), RL = row(last);
    LsRow Mg;                                   // the merged row (:499-523): the wider start / end of the two
    {
      const bool sa = Ra.x1 < Rb.x1, ea = Ra.x2 > Rb.x2;
      Mg.x1 = sa ? Ra.x1 : Rb.x1; Mg.y1 = sa ? Ra.y1 : Rb.y1; Mg.x2 = ea ? Ra.x2 : Rb.x2; Mg.y2 = ea ? Ra.y2 : Rb.y2;
      Mg.af = ls_angf(Mg.x1, Mg.y1, Mg.x2, Mg.y2);
    }
    const bool moved = (bs != last);            // row bs takes what used to be the last row (fast_RemoveRow, matrix_utils.cpp:183)
    __syncthreads();
    if (lane == 0) {
      X1[as] = Mg.x1; Y1[as] = Mg.y1; X2[as] = Mg.x2; Y2[as] = Mg.y2; AF[as] = Mg.af;
      if (moved) { X1[bs] = RL.x1; Y1[bs] = RL.y1; X2[bs] = RL.x2; Y2[bs] = RL.y2; AF[bs] = RL.af; }
    }
    __syncthreads();
    total = last;
    // the removed row's partner set goes; the two rewritten rows are read back by their owners
    if (lane == (last & 63)) { if (last < 64) { m00 = 0ull; m01 = 0ull; } else m11 = 0ull; }
    if (lane < total) own0 = row(lane);
    const bool two = total > 64;
    if (two && lane + 64 < total) own1 = row(lane + 64);
    // this round's tests: every row against the merged row, the rows behind bs against the row that moved there
    unsigned long long bal_as0 = 0ull, bal_as1 = 0ull, bal_bs0 = 0ull, bal_bs1 = 0ull;
    bool p_as0 = false, p_as1 = false;
    {
      const int a = lane;
      const bool valid = a < total;
      bool c_as = valid && a != as && ls_t12_flat(own0, Mg, lp, thf);
      bool c_bs = false;
      if (moved && bs < 63) c_bs = valid && a > bs && ls_t12_flat(RL, own0, lp, thf);
      bool p_bs = false;
      while (__ballot(c_as | c_bs)) {
        const bool vs_merged = c_as, pending = c_as | c_bs;
        const LsRow Q = vs_merged ? Mg : RL;
        const bool r = pending && ls_t3(own0, Q, vs_merged && a < as, lp, thf);
        if (vs_merged) { p_as0 = r; c_as = false; } else if (pending) { p_bs = r; c_bs = false; }
      }
      bal_as0 = __ballot(p_as0 && a > as); bal_bs0 = __ballot(p_bs);
    }
    if (two) {
      const int a = lane + 64;
      const bool valid = a < total;
      bool c_as = valid && a != as && ls_t12_flat(own1, Mg, lp, thf);
      bool c_bs = false;
      if (moved) c_bs = valid && a > bs && ls_t12_flat(RL, own1, lp, thf);
      bool p_bs = false;
      while (__ballot(c_as | c_bs)) {
        const bool vs_merged = c_as, pending = c_as | c_bs;
        const LsRow Q = vs_merged ? Mg : RL;
        const bool r = pending && ls_t3(own1, Q, vs_merged && a < as, lp, thf);
        if (vs_merged) { p_as1 = r; c_as = false; } else if (pending) { p_bs = r; c_bs = false; }
      }
      bal_as1 = __ballot(p_as1 && a > as); bal_bs1 = __ballot(p_bs);
    }
    // every other row: its bit for the old last row goes; its bit for `as` is the fresh test; its bit for `bs` is what its bit for the old last row was
    {
      const int a = lane;
      if (a < total && a != as && !(moved && a == bs)) {
        const bool old_last = ls_getbit(m00, m01, last);
        ls_setbit(m00, m01, last, false);
        if (a < as) ls_setbit(m00, m01, as, p_as0);
        if (moved && a < bs) ls_setbit(m00, m01, bs, old_last);
      }
    }
    if (two) {
      const int a = lane + 64;
      if (a < total && a != as && !(moved && a == bs)) {
        unsigned long long none = 0ull;
        const bool old_last = ls_getbit(none, m11, last);
        ls_setbit(none, m11, last, false);
        if (a < as) ls_setbit(none, m11, as, p_as1);
        if (moved && a < bs) ls_setbit(none, m11, bs, old_last);
      }
    }
    // the two rebuilt rows' partner sets are the tests' ballots
    if (as < 64) { if (lane == as) { m00 = bal_as0; m01 = bal_as1; } } else if (lane == as - 64) m11 = bal_as1;
    if (moved) { if (bs < 64) { if (lane == bs) { m00 = bal_bs0; m01 = bal_bs1; } } else if (lane == bs - 64) m11 = bal_bs1; }
  }
  __syncthreads();
  // ---- 4. keep segments longer than the threshold (in order) and emit the angle (exact: cs_atan2 of the row's end points) / midpoint tables
  int kept = 0;
  for (int base = 0; base < total; base += 64) {
    int i = base + lane;
    bool keep = false;
    if (i < total) {
      double len2 = v2_dist2(v2(X2[i], Y2[i]), v2(X1[i], Y1[i]));
      keep = (lp.len_thre > 0) ? (len2 > lp.len_sq_bound) : true;
    }
    unsigned long long bal = __ballot(keep);
    int off = kept + __popcll(bal & ((1ull << lane) - 1ull));
    if (keep) {
      line_angle[jd.line_off + off] = cs_atan2(Y2[i] - Y1[i], X2[i] - X1[i]);
      mid_x[jd.line_off + off] = (X1[i] + X2[i]) / 2;
      mid_y[jd.line_off + off] = (Y1[i] + Y2[i]) / 2;
    }
    kept += __popcll(bal);
  }
  if (lane == 0) jobs[j].m = kept;
}

// st_crowded (with the two events): the few crowded ROIs -- long, low-occupancy workgroups -- run on their own stream beside the
// others; `fork` must already be recorded on st, `join` is recorded here and st waits for it.
// diagnostics: CS_DETECT_SKIP=<names> leaves the named kernels out of the sweep, to measure their marginal cost in the saturated
// pipeline (the results are then meaningless; never set outside a timing experiment)
// Compiled in only with -DCS_DIAG (make DIAG=1): the shipped library has no switch that changes results, cs_diag_build() says which
// build is loaded and bench.py refuses a diagnostic one.
#ifdef CS_DIAG
static bool skip_kernel(const char* name) {
  static const char* e = getenv("CS_DETECT_SKIP");
  return e && strstr(e, name) != nullptr;
}
#else
static constexpr bool skip_kernel(const char*) { return false; }
#endif
void launch_line_setup(JobDesc* jobs, int n_jobs, const double* frame_lines, const int* frame_line_ptr, double* mid_x, double* mid_y, double* line_angle,
                       double dist_thre, double angle_thre_deg, double len_thre, hipStream_t st, const int* order, hipStream_t st_crowded, hipEvent_t fork, hipEvent_t join) {
  if (skip_kernel("line_setup")) return;
  if (n_jobs <= 0) return;
  LineSetupParams lp{dist_thre, angle_thre_deg / 180.0 * CS_PI, len_thre, sqrt_lt_bound(dist_thre), sqrt_le_bound(len_thre)};
  const bool beside = st_crowded && fork && join;
  hipStream_t sc = beside ? st_crowded : st;
  if (beside) (void)hipStreamWaitEvent(sc, fork, 0);
  hipLaunchKernelGGL((line_setup_kernel<LS_CAP, LS_CROWDED_THREADS>), dim3(n_jobs), dim3(LS_CROWDED_THREADS), 0, sc, jobs, n_jobs, frame_lines, frame_line_ptr, mid_x, mid_y, line_angle, lp, order);
  if (beside) (void)hipEventRecord(join, sc);
  hipLaunchKernelGGL(line_setup_small_kernel, dim3(n_jobs), dim3(64), 0, st, jobs, n_jobs, frame_lines, frame_line_ptr, mid_x, mid_y, line_angle, lp, order);
  if (beside) (void)hipStreamWaitEvent(st, join, 0);
}
// the lean path's form: the crowded ROIs from line_classify_kernel's list (crowded: n_jobs + 1 ints of scratch), the others as above
void launch_line_setup_listed(JobDesc* jobs, int n_jobs, const double* frame_lines, const int* frame_line_ptr, double* mid_x, double* mid_y, double* line_angle,
                              double dist_thre, double angle_thre_deg, double len_thre, hipStream_t st, const int* order, hipStream_t st_crowded, hipEvent_t fork, hipEvent_t join, int* crowded,
                              hipStream_t st_mid, hipEvent_t join_mid) {
  if (skip_kernel("line_setup")) return;
  if (n_jobs <= 0) return;
  LineSetupParams lp{dist_thre, angle_thre_deg / 180.0 * CS_PI, len_thre, sqrt_lt_bound(dist_thre), sqrt_le_bound(len_thre)};
  // crowded: 2 (n_jobs + 1) ints -- the list of the ROIs of up to LS_MID rows, behind it the list of the larger ones (one list when there is no second stream)
  static const bool one_list = getenv("CS_DETECT_LS_ONE_LIST") != nullptr;
  const bool two = st_mid && join_mid && !one_list;
  int* big = two ? crowded + n_jobs + 1 : nullptr;
  (void)hipMemsetAsync(crowded, 0, sizeof(int), st);
  if (two) (void)hipMemsetAsync(big, 0, sizeof(int), st);
  hipLaunchKernelGGL(line_classify_kernel, dim3((n_jobs + 3) / 4), dim3(256), 0, st, jobs, n_jobs, frame_lines, frame_line_ptr, crowded, big);
  (void)hipEventRecord(fork, st);
  const int grid = n_jobs < 1024 ? n_jobs : 1024;      // (more crowded ROIs than workgroups: a workgroup takes several)
  (void)hipStreamWaitEvent(st_crowded, fork, 0);
  hipLaunchKernelGGL((line_setup_listed_kernel<LS_CAP, LS_CROWDED_THREADS>), dim3(grid), dim3(LS_CROWDED_THREADS), 0, st_crowded, jobs, n_jobs, frame_lines, frame_line_ptr, mid_x, mid_y, line_angle, lp, two ? big : crowded);
  (void)hipEventRecord(join, st_crowded);
  if (two) {
    static const bool mid_own_stream = getenv("CS_DETECT_LS_MID_STREAM") != nullptr;
    hipStream_t sm = mid_own_stream ? st_mid : st_crowded;       // (behind the large ROIs' instance on ITS stream: a fourth stream per detector cost 7 % of the sweep's rate)
    if (mid_own_stream) (void)hipStreamWaitEvent(st_mid, fork, 0);
    hipLaunchKernelGGL((line_setup_listed_kernel<LS_MID, LS_CROWDED_THREADS>), dim3(grid), dim3(LS_CROWDED_THREADS), 0, sm, jobs, n_jobs, frame_lines, frame_line_ptr, mid_x, mid_y, line_angle, lp, crowded);
    (void)hipEventRecord(join_mid, sm);
  }
  hipLaunchKernelGGL(line_setup_small_kernel, dim3(n_jobs), dim3(64), 0, st, jobs, n_jobs, frame_lines, frame_line_ptr, mid_x, mid_y, line_angle, lp, order);
  (void)hipStreamWaitEvent(st, join, 0);
  if (two) (void)hipStreamWaitEvent(st, join_mid, 0);
}
int line_setup_capacity() { return LS_CAP; }


// ---- the winners' records, written on the device ----------------------------------------------------------------------------
// One cs_cuboid per (box, winner) of the boxes the device ranked (the tie boxes go through the host's exact ranking and are written
// there): box_proposal_detail.cpp:740-798 -- the row's columns, change_2d_corner_to_3d_object (object_3d_util.cpp:941-1011) and
// compute3D_BoxCorner (:59-73) with similarityTransformation (:15-44).  cos / sin of the yaw come from the sample tables the host
// filled with glibc's values (the same calls the host-side record writer makes), so every field carries the host writer's bits.
// rect_detect_2d is the caller's box and is filled in by the host when it copies the record out.
// REBUILD: the winner's corners are rebuilt here from its slot (slot_corners16) instead of read from the winner record -- the paths whose
// host never reads the device's winners launch no winner_corners_kernel
template <bool REBUILD>
__global__ __launch_bounds__(64) void record_kernel(DetectDeviceView v, RankView rv, int kmax, cs_cuboid* __restrict__ out, const double* __restrict__ raw_euler, double short_sq_bound) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= rv.n_boxes * kmax) return;
  const int q = e / kmax, r = e - q * kmax;
  if (rv.fallback[q] || r >= rv.win_count[q]) return;
  const RankWinner& w = rv.winners[e];
  double wc[16];
  if (REBUILD) slot_corners16(v, w.slot, short_sq_bound, wc);
  else {
#pragma unroll
    for (int i = 0; i < 16; i++) wc[i] = w.corners[i];
  }
  const int j0 = rv.box_job0[q], nh = rv.box_njobs[q];
  int h = 0;
  while (h + 1 < nh && w.slot >= v.jobs[j0 + h + 1].slot_off) h++;
  const JobDesc& jd = v.jobs[j0 + h];
  const long long local = w.slot - jd.slot_off, rest = local >> 1;
  const int ry = (int)(rest / jd.T), rp = ry / jd.Y, y = ry - rp * jd.Y;   // (roll/pitch sample, yaw sample)
  const RpPose& pose = v.rp[jd.rp_off + rp];
  cs_cuboid o;
  {
    double* z = reinterpret_cast<double*>(&o);
    static_assert(sizeof(cs_cuboid) % 8 == 0, "cs_cuboid is a whole number of doubles");
#pragma unroll
    for (int i = 0; i < (int)(sizeof(cs_cuboid) / 8); i++) z[i] = 0.0;
  }
  V2 c[8];
#pragma unroll
  for (int i = 0; i < 8; i++) c[i] = v2(wc[i], wc[8 + i]);
  lift_to_3d(c, pose.R, pose.t, v.invK + 9 * jd.frame, pose.plane, o.pos, o.scale);
  o.rotY = v.yaw[jd.yaw_off + y];
  const int vp1_pos = w.flag & CAND_VP_MASK;
  o.box_config_type[0] = (double)((local & 1) + 1); o.box_config_type[1] = (double)vp1_pos;
  const int left_ids[8] = {6, 5, 8, 7, 2, 3, 4, 1}, right_ids[8] = {5, 6, 7, 8, 3, 2, 1, 4};
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int id = (vp1_pos == 1) ? left_ids[i] : right_ids[i];
    o.box_corners_2d[i] = (int)wc[id - 1];
    o.box_corners_2d[8 + i] = (int)wc[8 + id - 1];
  }
  const double body[3][8] = {{1, 1, -1, -1, 1, 1, -1, -1}, {1, -1, -1, 1, 1, -1, -1, 1}, {-1, -1, -1, -1, 1, 1, 1, 1}};
  const double cr = v.yaw_cos[jd.yaw_off + y], sr = v.yaw_sin[jd.yaw_off + y];
  const double rot[3][3] = {{cr, -sr, 0}, {sr, cr, 0}, {0, 0, 1}};
  double S[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) S[i][j] = 0.0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) S[i][j] = rot[i][j] * o.scale[j];
    S[i][3] = o.pos[i];
  }
  S[3][3] = 1;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const double p[4] = {body[0][k], body[1][k], body[2][k], 1.0};
    double wv[4];
#pragma unroll
    for (int i = 0; i < 4; i++) wv[i] = ((S[i][0] * p[0] + S[i][1] * p[1]) + S[i][2] * p[2]) + S[i][3] * p[3];
#pragma unroll
    for (int i = 0; i < 3; i++) o.box_corners_3d_world[8 * i + k] = wv[i] / wv[3];
  }
  o.edge_distance_error = w.dist_err;
  o.edge_angle_error = w.angle_err;
  o.normalized_error = w.normalized_error;
  o.skew_ratio = ((o.scale[0] < o.scale[1]) ? o.scale[1] : o.scale[0]) / ((o.scale[1] < o.scale[0]) ? o.scale[1] : o.scale[0]);   // std::max / std::min
  o.down_expand_height = (double)jd.down_expand;
  if (raw_euler) {   // roll/pitch sampling: the sample's angles against the frame's own camera pose (box_proposal_detail.cpp:686, :790-791)
    o.camera_roll_delta = pose.roll - raw_euler[3 * jd.frame];
    o.camera_pitch_delta = pose.pitch - raw_euler[3 * jd.frame + 1];
  }
  out[e] = o;
}

// ------------------------------------------------------------------ launchers (host side) -----
static inline unsigned grid8(long long n, int bs) {
  long long nb = (n + bs - 1) / bs;
  nb = (nb + 7) / 8 * 8;
  return (unsigned)(nb < 8 ? 8 : nb);
}

void launch_vp_support(const DetectDeviceView& v, const SweepParams& sp, int vp_total, hipStream_t st) {
  if (vp_total <= 0) return;
  vp_support_kernel<0><<<dim3(grid8(vp_total, 256)), dim3(256), 0, st>>>(v, sp, vp_total);
}
void launch_vp_support_only(const DetectDeviceView& v, const SweepParams& sp, int vp_total, hipStream_t st, int rp_max) {
  if (skip_kernel("vp_support")) return;
  if (vp_total <= 0) return;
  static const bool sparse = getenv("CS_DETECT_VP3_SPARSE") != nullptr;      // (VP3_RPCAP lanes per job whatever the jobs hold: the form until round 6's end -- A / B)
  const int stride = (!sparse && rp_max >= 1 && rp_max <= (int)VP3_RPCAP) ? rp_max : (int)VP3_RPCAP;
  hipLaunchKernelGGL(vp3_support_kernel, dim3((unsigned)(((long long)v.n_jobs * stride + 63) / 64)), dim3(64), 0, st, v, sp, stride);
  vp_support_kernel<1><<<dim3(grid8(vp_total, 256)), dim3(256), 0, st>>>(v, sp, vp_total);
}
int vp3_table_doubles_per_job() { return 2 * VP3_RPCAP; }
void launch_vp_points(const DetectDeviceView& v, int vp_total, hipStream_t st) {
  if (vp_total <= 0) return;
  hipLaunchKernelGGL(vp_points_kernel, dim3((vp_total + 255) / 256), dim3(256), 0, st, v, vp_total);
}
void launch_candidates(const DetectDeviceView& v, const SweepParams& sp, long long slot_total, hipStream_t st) {
  if (skip_kernel("candidate")) return;
  if (slot_total <= 0) return;
  hipLaunchKernelGGL(candidate_kernel, dim3(grid8(slot_total, 256)), dim3(256), 0, st, v, sp, slot_total);
}
void launch_candidate_compact(const DetectDeviceView& v, const SweepParams& sp, hipStream_t st) {     // (v.blk_info set; slot_prefix in multiples of 256)
  if (skip_kernel("candidate")) return;
  if (v.n_jobs <= 0) return;
  hipLaunchKernelGGL(candidate_compact_kernel, dim3(v.n_jobs), dim3(256), 0, st, v, sp);
}
void launch_scan_compact(const DetectDeviceView& v, hipStream_t st) {
  if (v.n_jobs <= 0) return;
  hipLaunchKernelGGL(scan_jobs_kernel, dim3(1), dim3(1024), 0, st, v.job_valid, v.job_cbase, v.n_jobs);
  hipLaunchKernelGGL(compact_kernel, dim3(v.n_jobs), dim3(256), 0, st, v);
}
// ... with a workgroup per (job, trip): cnt = n_jobs x max_trips ints of scratch, max_trips = the largest job's slots / 4096, rounded up
void launch_scan_compact_trips(const DetectDeviceView& v, int* cnt, int max_trips, hipStream_t st) {
  if (v.n_jobs <= 0) return;
  if (max_trips <= 2 || max_trips > 65535) { launch_scan_compact(v, st); return; }
  hipLaunchKernelGGL(scan_jobs_kernel, dim3(1), dim3(1024), 0, st, v.job_valid, v.job_cbase, v.n_jobs);
  hipLaunchKernelGGL(compact_count_kernel, dim3(v.n_jobs, max_trips), dim3(256), 0, st, v, cnt, max_trips);
  hipLaunchKernelGGL(compact_chunk_kernel, dim3(v.n_jobs, max_trips), dim3(256), 0, st, v, cnt, max_trips);
}
// scoring over the compacted proposals; n_valid_bound is an upper bound known on the host (the exact count stays on the device)
void launch_score(const DetectDeviceView& v, const SweepParams& sp, long long n_valid_bound, long long slot_total, hipStream_t st) {
  if (skip_kernel("score")) return;
  if (n_valid_bound <= 0) return;
  if (v.blk_info) hipLaunchKernelGGL(score_kernel<true>, dim3(grid8(n_valid_bound, 256)), dim3(256), 0, st, v, slot_total, sp.short_sq_bound);
  else hipLaunchKernelGGL(score_kernel<false>, dim3(grid8(n_valid_bound, 256)), dim3(256), 0, st, v, slot_total, sp.short_sq_bound);
}
// copy [src_off, src_off + count) ranges of the compacted columns into packed buffers (fallback boxes)
__global__ __launch_bounds__(256) void gather_ranges_kernel(DetectDeviceView v, const long long* src_off, const int* count, const long long* dst_off, int n_ranges,
                                                            double* o_dist, double* o_angle, double* o_skew, int* o_flag, long long* o_slot) {
  int r = blockIdx.x;
  if (r >= n_ranges) return;
  long long s0 = src_off[r], d0 = dst_off[r];
  for (int i = threadIdx.x; i < count[r]; i += 256) {
    o_dist[d0 + i] = v.c_dist[s0 + i]; o_angle[d0 + i] = v.c_angle[s0 + i]; o_skew[d0 + i] = v.c_skew[s0 + i];
    o_flag[d0 + i] = v.c_flag[s0 + i]; o_slot[d0 + i] = v.c_slot[s0 + i];
  }
}
void launch_gather_ranges(const DetectDeviceView& v, const long long* src_off, const int* count, const long long* dst_off, int n_ranges,
                          double* o_dist, double* o_angle, double* o_skew, int* o_flag, long long* o_slot, hipStream_t st) {
  if (n_ranges <= 0) return;
  hipLaunchKernelGGL(gather_ranges_kernel, dim3(n_ranges), dim3(256), 0, st, v, src_off, count, dst_off, n_ranges, o_dist, o_angle, o_skew, o_flag, o_slot);
}
// ---- the ranking with ONE WAVEFRONT per box (round 6) ------------------------------------------------------------------------
// rank_kernel above is a chain of ~60 workgroup barriers per box (eight radix passes of four, twenty reductions of two) over ~900
// proposals: 66 us per box, nearly all of it waiting.  A box whose height samples hold <= RW_CAP valid proposals each is ranked by one
// wavefront instead, without LDS and without a barrier:
//   * an order statistic is found by walking the bits of order_key() from the highest bit in which the proposals differ (sign, exponent
//     and whatever else they share are skipped): lane l keeps the upper key words of proposals l, l + 64, ... in registers and ONE word
//     of candidate bits (bit q = proposal 64 q + l is still a candidate); a bit of the walk is two instructions per register (extract,
//     insert), a population count, a row reduction in DPP and four v_readlane -- and the walk stops when one candidate is left
//     (~11 bits for ~900 distinct doubles).  The lower key words are only fetched when candidates tie in the upper ones.
//   * the walk also yields how many keys are <= the statistic (the tie / angle-cut decisions), and the statistic's double IS the cut value
//     (order_key is a bijection): rank_kernel's counting and maximum passes have no counterpart;
//   * the remaining passes (extremes of the kept set, the final arg-min rounds) stream the columns from L2, coalesced.
// Every DECISION is the one rank_kernel takes, in the same arithmetic (selection and counts on order_key(), thresholds / normalisation /
// scores on the doubles, rank_combined_score): the two kernels agree bit for bit (CS_RANK_WAVE=0 in the tests), boxes with a larger
// height sample stay with rank_kernel (launched behind this one; its workgroups of the boxes ranked here return at once).
template <int CTRL> __device__ __forceinline__ int dpp_i32(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }
// sums / and / or / min / max over the wavefront, result in every lane: four DPP exchanges inside the rows of 16 (xor 1, xor 2, half-row
// mirror, row mirror), then the four rows' values through v_readlane
__device__ __forceinline__ int wave_sum_i32(int x) {
  x += dpp_i32<0xB1>(x); x += dpp_i32<0x4E>(x); x += dpp_i32<0x141>(x); x += dpp_i32<0x140>(x);
  return __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) + __builtin_amdgcn_readlane(x, 48);
}
__device__ __forceinline__ unsigned wave_and_u32(unsigned u) {
  int x = (int)u;
  x &= dpp_i32<0xB1>(x); x &= dpp_i32<0x4E>(x); x &= dpp_i32<0x141>(x); x &= dpp_i32<0x140>(x);
  return (unsigned)(__builtin_amdgcn_readlane(x, 0) & __builtin_amdgcn_readlane(x, 16) & __builtin_amdgcn_readlane(x, 32) & __builtin_amdgcn_readlane(x, 48));
}
__device__ __forceinline__ unsigned wave_or_u32(unsigned u) {
  int x = (int)u;
  x |= dpp_i32<0xB1>(x); x |= dpp_i32<0x4E>(x); x |= dpp_i32<0x141>(x); x |= dpp_i32<0x140>(x);
  return (unsigned)(__builtin_amdgcn_readlane(x, 0) | __builtin_amdgcn_readlane(x, 16) | __builtin_amdgcn_readlane(x, 32) | __builtin_amdgcn_readlane(x, 48));
}
__device__ __forceinline__ int wave_max_i32(int x) {
  x = max(x, dpp_i32<0xB1>(x)); x = max(x, dpp_i32<0x4E>(x)); x = max(x, dpp_i32<0x141>(x)); x = max(x, dpp_i32<0x140>(x));
  return max(max(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 16)), max(__builtin_amdgcn_readlane(x, 32), __builtin_amdgcn_readlane(x, 48)));
}
template <int CTRL> __device__ __forceinline__ double dpp_f64(double x) {
  return __hiloint2double(dpp_i32<CTRL>(__double2hiint(x)), dpp_i32<CTRL>(__double2loint(x)));
}
__device__ __forceinline__ double readlane_f64(double x, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
// (no NaN reaches these: a NaN in a column sends the box to the host before the extremes matter)
__device__ __forceinline__ double wave_min_f64(double x) {
  double y;
  y = dpp_f64<0xB1>(x); x = (y < x) ? y : x; y = dpp_f64<0x4E>(x); x = (y < x) ? y : x;
  y = dpp_f64<0x141>(x); x = (y < x) ? y : x; y = dpp_f64<0x140>(x); x = (y < x) ? y : x;
  double r = readlane_f64(x, 0);
  y = readlane_f64(x, 16); r = (y < r) ? y : r; y = readlane_f64(x, 32); r = (y < r) ? y : r; y = readlane_f64(x, 48); r = (y < r) ? y : r;
  return r;
}
__device__ __forceinline__ double wave_max_f64(double x) {
  double y;
  y = dpp_f64<0xB1>(x); x = (y > x) ? y : x; y = dpp_f64<0x4E>(x); x = (y > x) ? y : x;
  y = dpp_f64<0x141>(x); x = (y > x) ? y : x; y = dpp_f64<0x140>(x); x = (y > x) ? y : x;
  double r = readlane_f64(x, 0);
  y = readlane_f64(x, 16); r = (y > r) ? y : r; y = readlane_f64(x, 32); r = (y > r) ? y : r; y = readlane_f64(x, 48); r = (y > r) ? y : r;
  return r;
}
__device__ __forceinline__ double order_key_to_double(unsigned long long k) {      // order_key()'s inverse
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}
// one 32-bit word of the selection, KQ registers (proposals behind the column's end carry no candidate bit): the candidates are narrowed
// bit by bit from start_bit down until one is left; r = the rank wanted among the candidates, nact = their number
template <int KQ>
__device__ __forceinline__ void wave_select_walk(const unsigned (&w)[RW_K], unsigned& act, int start_bit, int& r, int& nact) {
  for (int bit = start_bit; bit >= 0 && nact > 1; bit--) {
    unsigned ones = 0;
#pragma unroll
    for (int q = 0; q < KQ; q++) ones |= ((w[q] >> bit) & 1u) << q;
    const unsigned zer = act & ~ones;
    const int c0 = wave_sum_i32(__popc(zer));
    if (r < c0) { act = zer; nact = c0; }
    else { act &= ones; r -= c0; nact -= c0; }
  }
}
__device__ __forceinline__ void wave_select_phase(const unsigned (&w)[RW_K], unsigned& act, int Q, int& r, int& nact) {
  // the highest bit in which the candidates' words differ; none: they all carry the same word
  unsigned a = 0xffffffffu, o = 0u;
#pragma unroll
  for (int q = 0; q < RW_K; q++) { if (q >= Q) continue; if ((act >> q) & 1u) { a &= w[q]; o |= w[q]; } }
  const unsigned diff = wave_and_u32(a) ^ wave_or_u32(o);
  if (!diff) return;
  const int start = 31 - __clz((int)diff);
  if (Q <= 8) wave_select_walk<8>(w, act, start, r, nact);
  else if (Q <= 12) wave_select_walk<12>(w, act, start, r, nact);
  else if (Q <= 16) wave_select_walk<16>(w, act, start, r, nact);
  else if (Q <= 20) wave_select_walk<20>(w, act, start, r, nact);
  else if (Q <= 24) wave_select_walk<24>(w, act, start, r, nact);
  else wave_select_walk<RW_K>(w, act, start, r, nact);
}
// order_key() of rank r (0-based, ascending) among X[0 .. V) (global memory, V <= RW_CAP); n_le = how many keys are <= it.
// (NOT inlined: with its two instances inside the kernel the compiler's schedule needs 512 registers and spills 4 800)
struct WaveSelected { unsigned long long key; int n_le; };
__device__ __attribute__((noinline)) WaveSelected wave_select_key(const double* __restrict__ X, int V, int Q, int r) {
  const int lane = threadIdx.x & 63;
  const unsigned* __restrict__ X32 = reinterpret_cast<const unsigned*>(X);
  unsigned w[RW_K];
  unsigned act = 0;
  // (every load asked for before the first is used -- indices behind the column's end re-read its last proposal --: one round trip to L2)
#pragma unroll
  for (int q = 0; q < RW_K; q++) { w[q] = 0u; if (q >= Q) continue; w[q] = X32[2 * (size_t)min(64 * q + lane, V - 1) + 1]; }
#pragma unroll
  for (int q = 0; q < RW_K; q++) {
    if (q >= Q) continue;
    const unsigned h = w[q];
    const bool in = 64 * q + lane < V;
    w[q] = in ? ((h >> 31) ? ~h : (h | 0x80000000u)) : 0u;      // order_key()'s upper word
    act |= in ? 1u << q : 0u;
  }
  const int r0 = r;
  int nact = V;
  wave_select_phase(w, act, Q, r, nact);
  if (nact > 1) {      // candidates tie in the upper word: the lower words (of a negative double: complemented)
#pragma unroll
    for (int q = 0; q < RW_K; q++) {
      if (q >= Q) continue;
      const long long x = __double_as_longlong(X[min(64 * q + lane, V - 1)]);
      w[q] = (x < 0) ? ~(unsigned)x : (unsigned)x;
    }
    wave_select_phase(w, act, Q, r, nact);
  }
  // the candidates left all carry the key (one of them, or equal keys): the first one's lane fetches it
  const int owner = __ffsll((long long)__ballot(act != 0u)) - 1;
  unsigned long long key = 0;
  if (lane == owner) key = order_key(X[64 * (__ffs((int)act) - 1) + lane]);
  key = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(key >> 32), owner) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)key, owner);
  WaveSelected out;
  out.key = key;
  out.n_le = (r0 - r) + nact;      // the keys below the candidates + the candidates
  return out;
}
__global__ __launch_bounds__(64) void rank_wave_kernel(DetectDeviceView v, RankView rv, RankParams rp) {
  const int box = blockIdx.x;
  if (box >= rv.n_boxes) return;
  const int j0 = rv.box_job0[box], nj = rv.box_njobs[box];
  // (a height sample too large for the registers: rank_kernel's procedure on this one wavefront, its columns read from L2 -- rare outside
  // the roll/pitch rounds, which have their own instance; no second launch in the batch's chain)
  if (!rank_box_fits_wave(v, j0, nj)) { rank_block_body<64, true>(v, rv, rp, 0, box); return; }
  const int lane = threadIdx.x;
  const double INF = __builtin_huge_val();
  int fallback = 0;
  __shared__ JobCut cuts[3];      // (written and read by this one wavefront)
  for (int h = 0; h < nj && h < 3; h++) {
    const int j = j0 + h;
    const long long c0 = v.job_cbase[j];
    const int V = v.job_valid[j];
    const int Q = (V + 63) >> 6;
    const double* __restrict__ D = v.c_dist + c0;
    const double* __restrict__ A = v.c_angle + c0;
    double vd = INF, va = INF;
    int use_angle = 0, tie = 0, bn_keep = 0;
    if (V > 4) {
      const int bn = (int)round((double)((float)V) / 3.0 * 2.0);
      bn_keep = bn - 1;
      const WaveSelected sd = wave_select_key(D, V, Q, bn - 2), sa = wave_select_key(A, V, Q, bn - 2);
      tie = (sd.n_le != bn - 1);
      use_angle = (sa.n_le == bn - 1);
      vd = order_key_to_double(sd.key);      // the largest kept value: the statistic itself
      va = order_key_to_double(sa.key);
    }
    double dmin = 1e6, dmax = -1, amin = 1e6, amax = -1, amin_s = 1e6, amax_s = -1;
    int nk = 0, nan = 0, nk_s = 0, nt = 0, ct = 0, cl = 0, last = -1, n_at_cut = 0;
    const bool want_last = rv.last_slot && h == nj - 1 && V > 4;
    for (int q0 = 0; q0 < Q; q0 += RW_BATCH) {
      double db[RW_BATCH], ab[RW_BATCH];
#pragma unroll
      for (int u = 0; u < RW_BATCH; u++) { const int ic = min(64 * (q0 + u) + lane, V - 1); db[u] = D[ic]; ab[u] = A[ic]; }
#pragma unroll
      for (int u = 0; u < RW_BATCH; u++) {
      const int i = 64 * (q0 + u) + lane;
      if (i < V) {
        const double d = db[u], a = ab[u];
        nan += (d != d) || (a != a);
        const bool pass = !use_angle || a <= va;
        if (d <= vd && pass) {
          nk++;
          dmin = (d < dmin) ? d : dmin; dmax = (dmax < d) ? d : dmax;
          amin = (a < amin) ? a : amin; amax = (amax < a) ? a : amax;
        }
        if (tie) {      // (rank_kernel: when do the tied proposals at the cut not matter?)
          if (d < vd) { cl++; if (pass) { nk_s++; amin_s = (a < amin_s) ? a : amin_s; amax_s = (amax_s < a) ? a : amax_s; } }
          else if (d == vd) { ct++; nt += pass; }
        }
        if (want_last) {
          if (use_angle) { if (d <= vd && a <= va) last = i; }
          else if (d == vd) { last = i; n_at_cut++; }
        }
      }
      }
    }
    dmin = wave_min_f64(dmin); dmax = wave_max_f64(dmax);
    amin = wave_min_f64(amin); amax = wave_max_f64(amax);
    nk = wave_sum_i32(nk);
    if (V > 4 && wave_sum_i32(nan)) fallback = 1;
    if (tie) {
      amin_s = wave_min_f64(amin_s); amax_s = wave_max_f64(amax_s);
      nk_s = wave_sum_i32(nk_s); nt = wave_sum_i32(nt); ct = wave_sum_i32(ct); cl = wave_sum_i32(cl);
      const int r = bn_keep - cl;
      const bool safe = nk_s >= 2 && amin_s == amin && amax_s == amax && (nt == 0 || r - (ct - nt) >= 1);
      if (!safe) fallback = 1;
    }
    if (rv.last_slot && h == nj - 1) {
      if (V > 4) {
        last = wave_max_i32(last);
        n_at_cut = wave_sum_i32(n_at_cut);
        if (tie || (!use_angle && n_at_cut != 1)) fallback = 1;
      } else {
        last = V - 1;
      }
      if (lane == 0) rv.last_slot[box] = last >= 0 ? v.c_slot[c0 + last] : -1;
    }
    if (lane == 0) {
      JobCut c; c.vd = vd; c.va = va; c.dmin = dmin; c.dmax = dmax; c.amin = amin; c.amax = amax; c.use_angle = use_angle; c.n_keep = nk; c.V = V; c.tie = tie;
      cuts[h] = c;
    }
  }
  __syncthreads();
  // ---- final ranking: kmax rounds of arg-min, one pass over the proposals per round
  double prev = -INF;
  int n_win = 0;
  for (int round = 0; round < rp.kmax; round++) {
    double best = INF, w_score = 0, w_d = 0, w_a = 0;
    long long w_at = 0;
    int bad = 0, cnt_local = 0;
    for (int h = 0; h < nj && h < 3; h++) {
      const JobCut c = cuts[h];
      const long long c0 = v.job_cbase[j0 + h];
      const int Q = (c.V + 63) >> 6;
      for (int q0 = 0; q0 < Q; q0 += RW_BATCH) {
        double db[RW_BATCH], ab[RW_BATCH], sb[RW_BATCH];
        int fb[RW_BATCH];
#pragma unroll
        for (int u = 0; u < RW_BATCH; u++) {
          const long long at = c0 + min(64 * (q0 + u) + lane, c.V - 1);
          db[u] = v.c_dist[at]; ab[u] = v.c_angle[at]; sb[u] = v.c_skew[at]; fb[u] = v.c_flag[at];
        }
#pragma unroll
        for (int u = 0; u < RW_BATCH; u++) {
        const int i = 64 * (q0 + u) + lane;
        if (i < c.V) {
          const double d = db[u], a = ab[u], sk = sb[u];
          const int fl = fb[u];
          const bool keep = (d <= c.vd) && (!c.use_angle || a <= c.va);
          if (keep && !(fl & CAND_NEG_SCALE)) {
            double score;
            const double comb = rank_combined_score(c, rp, d, a, sk, &score);
            if (comb != comb || comb == INF || comb == -INF) bad = 1;
            else if (comb > prev) {
              const int weight = (c.tie && d == c.vd) ? 3 : 1;
              if (comb < best) { best = comb; cnt_local = weight; w_score = score; w_d = d; w_a = a; w_at = c0 + i; }
              else if (comb == best) cnt_local += weight;
            }
          }
        }
        }
      }
    }
    if (__ballot(bad != 0)) fallback = 1;
    const double gbest = wave_min_f64(best);
    if (!(gbest < INF)) break;
    const bool mine = best == gbest;
    const int cnt = wave_sum_i32(mine ? cnt_local : 0);
    // the owner of the minimum (unique unless the box goes to the host anyway) writes the winner's scores (its corners: winner_corners_kernel)
    if (mine) {
      RankWinner* w = rv.winners + (size_t)box * rp.kmax + round;
      w->slot = v.c_slot[w_at]; w->normalized_error = w_score; w->dist_err = w_d; w->angle_err = w_a; w->flag = v.c_flag[w_at] & CAND_VP_MASK; w->pad = 0;
    }
    if (cnt != 1) fallback = 1;
    prev = gbest;
    n_win++;
  }
  if (lane == 0) { rv.win_count[box] = n_win; rv.fallback[box] = fallback; }
}
// the winners' corners, rebuilt from their slots (slot_corners16: the values the scorer saw): one lane per winner, behind the ranking kernels
// (round 6; a winner's corners used to be one lane's work inside its box's ranking workgroup -- ~800 instructions with the others idle)
__global__ __launch_bounds__(64) void winner_corners_kernel(DetectDeviceView v, RankView rv, int kmax, double short_sq_bound) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= rv.n_boxes * kmax) return;
  const int q = e / kmax, r = e - q * kmax;
  if (r >= rv.win_count[q]) return;
  RankWinner* w = rv.winners + e;
  slot_corners16(v, w->slot, short_sq_bound, w->corners);
}
// with_corners: the winners' corners are written into the winner records (winner_corners_kernel) -- the paths whose host reads the device's
// winners; the others let record_kernel rebuild them (launch_records' rebuild_short_sq_bound)
void launch_rank(const DetectDeviceView& v, const RankView& rv, const RankParams& rp, hipStream_t st, long long max_slots_per_box, bool with_corners) {
  if (skip_kernel("rank")) return;
  if (rv.n_boxes <= 0) return;
  static const int nt = [] { const char* e = getenv("CS_RANK_THREADS"); const int q = e ? atoi(e) : 0; return (q == 64 || q == 128 || q == 256) ? q : RANK_THREADS_DEFAULT; }();
  static const bool no_big = getenv("CS_RANK_NO_BIG") != nullptr;     // diagnostics / tests: the ordinary instance for every launch
  // one wavefront per box (boxes with a height sample above RW_CAP valid proposals: rank_kernel's procedure on that wavefront); CS_RANK_WAVE=0:
  // a workgroup per box, rank_kernel -- the A / B switch, and the tests that hold the two kernels to each other
  static const int wave = [] { const char* e = getenv("CS_RANK_WAVE"); return (e && e[0] == '0') ? 0 : 1; }();
  bool done = false;
  // boxes that can hold far more valid proposals than the fixed staging columns (the caller's bound on a box's slots): the big instance
  if (!no_big && max_slots_per_box > 8 * (long long)RANK_STAGE) {
    static DynLdsOnce big_lds;      // per device (cs_hip_util.h)
    const size_t lds = 2 * (size_t)RANK_STAGE_BIG * sizeof(double);
    if (big_lds.set(reinterpret_cast<const void*>(rank_kernel<RANK_THREADS_BIG, true>), (int)lds)) {
      hipLaunchKernelGGL((rank_kernel<RANK_THREADS_BIG, true>), dim3(rv.n_boxes), dim3(RANK_THREADS_BIG), lds, st, v, rv, rp, (int)RANK_STAGE_BIG);
      done = true;
    }
    // (refused: the ordinary instances below rank the same boxes in more passes)
  }
  if (done) { }
  else if (wave) hipLaunchKernelGGL(rank_wave_kernel, dim3(rv.n_boxes), dim3(64), 0, st, v, rv, rp);
  else if (nt == 64) hipLaunchKernelGGL((rank_kernel<64, false>), dim3(rv.n_boxes), dim3(64), 0, st, v, rv, rp, (int)RANK_STAGE);
  else if (nt == 128) hipLaunchKernelGGL((rank_kernel<128, false>), dim3(rv.n_boxes), dim3(128), 0, st, v, rv, rp, (int)RANK_STAGE);
  else hipLaunchKernelGGL((rank_kernel<256, false>), dim3(rv.n_boxes), dim3(256), 0, st, v, rv, rp, (int)RANK_STAGE);
  if (with_corners) hipLaunchKernelGGL(winner_corners_kernel, dim3((rv.n_boxes * rp.kmax + 63) / 64), dim3(64), 0, st, v, rv, rp.kmax, rp.short_sq_bound);
}
void launch_records(const DetectDeviceView& v, const RankView& rv, int kmax, cs_cuboid* out, hipStream_t st, const double* raw_euler, double rebuild_short_sq_bound) {
  if (skip_kernel("records")) return;
  const int n = rv.n_boxes * kmax;
  if (n <= 0) return;
  if (rebuild_short_sq_bound >= 0) hipLaunchKernelGGL(record_kernel<true>, dim3((n + 63) / 64), dim3(64), 0, st, v, rv, kmax, out, raw_euler, rebuild_short_sq_bound);
  else hipLaunchKernelGGL(record_kernel<false>, dim3((n + 63) / 64), dim3(64), 0, st, v, rv, kmax, out, raw_euler, 0.0);
}
// ---- roll/pitch sampling on the device: the camera yaw carried from box to box ----------------------------------------------
// With whether_sample_cam_roll_pitch the yaw list of a box starts from cam_pose.camera_yaw as the previous box of the frame left it
// (box_proposal_detail.cpp:180 reads what :374 / :734 wrote): the camera yaw of the roll/pitch sample of the last proposal that
// box's ranking kept, or -- nothing kept -- of the sweep's last sample.  That value is one of 1 + RP numbers known up front (the raw
// camera yaw, the RP samples' yaws), so the host lays down all 1 + RP yaw lists of a frame (with glibc's cos / sin) and the device
// only picks: one lane per frame reads the previous round's last kept proposal of the frame's box, moves the frame's table index,
// and points this round's jobs of the frame at that list.  The rounds then follow each other on the stream without a host decision.
__global__ __launch_bounds__(64) void rp_carry_kernel(RpCarryView c, JobDesc* jobs) {
  const int f = blockIdx.x * 64 + threadIdx.x;
  if (f >= c.n_frames) return;
  int idx = c.cur_idx[f];
  const int q = c.prev_box_of_frame ? c.prev_box_of_frame[f] : -1;
  if (q >= 0) {
    const JobDesc& jl = c.prev_jobs[c.prev_box_job0[q] + c.prev_box_njobs[q] - 1];
    const long long last = c.prev_last_slot[q];
    idx = last < 0 ? jl.RP : 1 + (int)((((last - jl.slot_off) >> 1) / jl.T) / jl.Y);
    c.cur_idx[f] = idx;
  }
  const int j0 = c.job0_of_frame[f];
  if (j0 < 0) return;
  for (int h = 0; h < c.njobs_of_frame[f]; h++) {
    jobs[j0 + h].yaw_off = (f * c.NT + idx) * c.YCAP;
    jobs[j0 + h].Y = c.tab_count[f * c.NT + idx];
  }
}
__global__ __launch_bounds__(256) void rp_save_fallback_kernel(DetectDeviceView v, RpSaveView s) {
  const int q = blockIdx.x;
  if (q >= s.n_boxes) return;
  __shared__ long long base_s;
  if (!s.fallback[q]) { if (threadIdx.x == 0) s.box_base[q] = -1; return; }
  const int j0 = s.box_job0[q], nh = s.box_njobs[q];
  if (threadIdx.x == 0) {
    long long V = 0;
    for (int h = 0; h < nh; h++) V += v.job_valid[j0 + h];
    const long long base = (long long)atomicAdd(s.pool_used, (unsigned long long)V);
    base_s = (base + V <= s.pool_cap) ? base : -1;
    s.box_base[q] = base_s;
  }
  __syncthreads();
  long long dst = base_s;
  if (dst < 0) return;
  for (int h = 0; h < nh; h++) {
    const long long c0 = v.job_cbase[j0 + h];
    const int V = v.job_valid[j0 + h];
    for (int i = threadIdx.x; i < V; i += 256) {
      s.p_dist[dst + i] = v.c_dist[c0 + i]; s.p_angle[dst + i] = v.c_angle[c0 + i]; s.p_skew[dst + i] = v.c_skew[c0 + i];
      s.p_flag[dst + i] = v.c_flag[c0 + i]; s.p_slot[dst + i] = v.c_slot[c0 + i];
    }
    dst += V;
  }
}
void launch_rp_save_fallback(const DetectDeviceView& v, const RpSaveView& s, hipStream_t st) {
  if (s.n_boxes > 0) hipLaunchKernelGGL(rp_save_fallback_kernel, dim3(s.n_boxes), dim3(256), 0, st, v, s);
}
void launch_rp_carry(const RpCarryView& c, JobDesc* jobs, hipStream_t st) {
  if (c.n_frames > 0) hipLaunchKernelGGL(rp_carry_kernel, dim3((c.n_frames + 63) / 64), dim3(64), 0, st, c, jobs);
}
void launch_gather_corners(const DetectDeviceView& v, const SweepParams& sp, const long long* slots, int n, double* out, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gather_corners_kernel, dim3((n + 63) / 64), dim3(64), 0, st, v, sp.short_sq_bound, slots, n, out);
}

}  // namespace cs
