// detect_kernels.hip -- HIP kernels of the cuboid proposal sweep for gfx950 (MI355X).
//
// What runs here (reference loops L3..L5, detect_3d_cuboid/src/box_proposal_detail.cpp:360-705):
//   vp_support_kernel   one lane per (job, roll/pitch, yaw): vanishing points (object_3d_util.cpp:928)
//                       and their supporting line angles (object_3d_util.cpp:548-619);
//   candidate_kernel    one lane per proposal slot (job, roll/pitch, yaw, top sample, config): the
//                       eight corners (:413-625), the distance-map edge score (object_3d_util.cpp:622),
//                       the VP angle alignment (object_3d_util.cpp:670) and the 3D half sizes;
//   scan/compact        ordered stream compaction of the valid proposals per job (the reference
//                       appends rows in loop order, :677-702, and that order defines tie-breaking);
//   gather_corners      corners of the selected proposals for the final records.
//
// Numerics: FP64 geometry, float32 gathers with a *sequential* float running sum per proposal (one lane
// owns one proposal; no shuffles/tree reductions on scores), cs_atan2 (double-double) instead of libm.
// Compiled with -ffp-contract=off: every result is bit-identical to the CPU oracle.
#include <hip/hip_runtime.h>

#include "detect_types.h"

namespace cs {

__device__ __forceinline__ int find_job_i32(const int* prefix, int n, int v) {
  int lo = 0, hi = n;  // prefix[lo] <= v < prefix[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (prefix[mid] <= v) lo = mid; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ int find_job_i64(const long long* prefix, int n, long long v) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (prefix[mid] <= v) lo = mid; else hi = mid;
  }
  return lo;
}

// Blocks with the same (blockIdx % 8) run on the same XCD (private 4 MiB L2).  Remap so that each XCD
// walks a contiguous range of virtual blocks: all blocks of a job then share one L2, which keeps the
// job's distance map and line arrays L2-resident.  grid must be a multiple of 8.
__device__ __forceinline__ long long xcd_virtual_block() {
  long long nb = gridDim.x, b = blockIdx.x;
  return (b & 7) * (nb >> 3) + (b >> 3);
}

__global__ __launch_bounds__(256) void vp_support_kernel(DetectDeviceView v, SweepParams sp, int vp_total) {
  long long e = xcd_virtual_block() * blockDim.x + threadIdx.x;
  if (e >= vp_total) return;
  int j = find_job_i32(v.vp_prefix, v.n_jobs, (int)e);
  const JobDesc jd = v.jobs[j];
  int local = (int)e - jd.vp_off;
  int rp = local / jd.Y, y = local - rp * jd.Y;
  const RpPose* pose = v.rp + jd.rp_off + rp;
  double cy = v.yaw_cos[jd.yaw_off + y], sy = v.yaw_sin[jd.yaw_off + y];
  const double* A = pose->KinvR;
  // getVanishingPoints (object_3d_util.cpp:928-937)
  double d[3][3] = {{cy, sy, 0.0}, {-sy, cy, 0.0}, {0.0, 0.0, 1.0}};
  double vpx[3], vpy[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    double h0 = (A[0] * d[k][0] + A[1] * d[k][1]) + A[2] * d[k][2];
    double h1 = (A[3] * d[k][0] + A[4] * d[k][1]) + A[5] * d[k][2];
    double h2 = (A[6] * d[k][0] + A[7] * d[k][1]) + A[8] * d[k][2];
    vpx[k] = h0 / h2;
    vpy[k] = h1 / h2;
  }
  double* vout = v.vp + 6 * e;
  vout[0] = vpx[0]; vout[1] = vpy[0]; vout[2] = vpx[1]; vout[3] = vpy[1]; vout[4] = vpx[2]; vout[5] = vpy[2];

  // VP_support_edge_infos (object_3d_util.cpp:548-619): sequential over the job's merged lines.
  const double* mx = v.mid_x + jd.line_off;
  const double* my = v.mid_y + jd.line_off;
  const double* la = v.line_angle + jd.line_off;
  double* bout = v.bound + 6 * e;
  const double NaN = __builtin_nan("");
#pragma unroll
  for (int k = 0; k < 3; k++) {
    double thre = (k != 2) ? sp.vp12_thre_rad : sp.vp3_thre_rad;
    bool have = false;
    double base = 0, best_hi = 0, best_lo = 0, ang_hi = NaN, ang_lo = NaN;
    for (int i = 0; i < jd.m; i++) {
      double raw = cs_atan2(my[i] - vpy[k], mx[i] - vpx[k]);
      double nrm = normalize_to_pi(raw);
      double df = dabs(la[i] - nrm);
      df = dmin(df, CS_PI - df);
      if (df < thre) {
        if (!have) {  // first inlier: base of smooth_jump_angles (:278-302), and initial arg-max / arg-min
          have = true; base = raw; best_hi = raw; best_lo = raw; ang_hi = la[i]; ang_lo = la[i];
        } else {
          double sh = raw;
          if ((raw - base) < -CS_PI) sh = raw + 2 * CS_PI;
          else if ((raw - base) > CS_PI) sh = raw - 2 * CS_PI;
          if (sh > best_hi) { best_hi = sh; ang_hi = la[i]; }  // maxCoeff: first occurrence, strict
          if (sh < best_lo) { best_lo = sh; ang_lo = la[i]; }  // minCoeff
        }
      }
    }
    // vp 1: (max, min); vp 2,3: swapped (:609-614)
    bout[2 * k + 0] = (k > 0) ? ang_lo : ang_hi;
    bout[2 * k + 1] = (k > 0) ? ang_hi : ang_lo;
  }
}

// box_edge_sum_dists (object_3d_util.cpp:622-667): 11 samples per edge, float gathers, float running sum.
template <int NE, bool REWEIGHT>
__device__ __forceinline__ double edge_sum_dists(const float* __restrict__ map, int map_w, const V2 c[8], const int (&ea)[NE], const int (&eb)[NE], double ox, double oy) {
  float sum_dist = 0;
#pragma unroll
  for (int e = 0; e < NE; e++) {
    double x1 = c[ea[e]].x - ox, y1 = c[ea[e]].y - oy, x2 = c[eb[e]].x - ox, y2 = c[eb[e]].y - oy;
    float dv[11];
#pragma unroll
    for (int s = 0; s < 11; s++) {
      double w = (double)s / 10.0;
      double sx = w * x1 + (1 - w) * x2;
      double sy = w * y1 + (1 - w) * y2;
      dv[s] = map[(long long)(int)sy * map_w + (int)sx];
    }
#pragma unroll
    for (int s = 0; s < 11; s++) {
      float d1 = dv[s];
      if (REWEIGHT) {
        if (e == 4 || e == 5) d1 = (float)((double)d1 * 3.0 / 2.0);
        if (e == 6) d1 = (float)((double)d1 * 2.0);
      }
      sum_dist = sum_dist + d1;
    }
  }
  return (double)sum_dist;
}

// box_edge_alignment_angle_error (object_3d_util.cpp:670-723)
__device__ __forceinline__ double angle_alignment_error(const double* bound, const int (&ids)[3][4], const V2 c[8]) {
  double total = 0;
  const double not_found_penalty = 30.0 / 180.0 * CS_PI * 2;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    double b0 = bound[2 * k], b1 = bound[2 * k + 1];
    bool v0 = !(b0 != b0), v1 = !(b1 != b1);
    if (v0 || v1) {
#pragma unroll
      for (int ee = 0; ee < 2; ee++) {
        V2 p1 = c[ids[k][2 * ee]], p2 = c[ids[k][2 * ee + 1]];
        double ang = normalize_to_pi(cs_atan2(p2.y - p1.y, p2.x - p1.x));
        double best = 100;
        if (v0) { double t = dabs(ang - b0); t = dmin(t, CS_PI - t); if (t < best) best = t; }
        if (v1) { double t = dabs(ang - b1); t = dmin(t, CS_PI - t); if (t < best) best = t; }
        total = total + best;
      }
    } else {
      total = total + not_found_penalty;
    }
  }
  return total;
}

__global__ __launch_bounds__(256) void candidate_kernel(DetectDeviceView v, SweepParams sp, long long slot_total) {
  long long tid = xcd_virtual_block() * blockDim.x + threadIdx.x;
  bool active = tid < slot_total;
  int flag = 0;
  int j = 0;
  long long slot = 0;
  if (active) {
    j = find_job_i64(v.slot_prefix, v.n_jobs, tid);
    const JobDesc jd = v.jobs[j];
    // lane -> proposal: config-major inside the job so that a wave runs one configuration
    long long k = tid - jd.slot_off;
    long long half = (long long)jd.RP * jd.Y * jd.T;
    int cfg = (k >= half) ? 2 : 1;
    long long rest = (k >= half) ? k - half : k;
    int t = (int)(rest % jd.T);
    int ry = (int)(rest / jd.T);  // rp*Y + yaw
    int rp = ry / jd.Y;
    slot = jd.slot_off + rest * 2 + (cfg - 1);
    bool enabled = (cfg == 1) ? (sp.consider_config_1 != 0) : (sp.consider_config_2 != 0);
    if (enabled) {
      const double* vp = v.vp + 6 * (long long)(jd.vp_off + ry);
      V2 vp1 = v2(vp[0], vp[1]), vp2 = v2(vp[2], vp[3]), vp3 = v2(vp[4], vp[5]);
      V2 c[8];
      int pos = build_corners(jd.g, vp1, vp2, vp3, (double)v.top_x[jd.top_off + t], cfg, sp.short_thre, c);
      if (pos) {
        const float* map = v.maps + jd.map_off;
        const double* bound = v.bound + 6 * (long long)(jd.vp_off + ry);
        double sum_dist, ang;
        if (cfg == 1) {
          const int ea[9] = {0, 1, 2, 3, 1, 2, 3, 4, 4}, eb[9] = {1, 2, 3, 0, 5, 4, 7, 7, 5};  // :646
          const int ids[3][4] = {{0, 1, 7, 4}, {3, 0, 4, 5}, {3, 7, 1, 5}};                     // :651
          sum_dist = edge_sum_dists<9, false>(map, jd.map_w, c, ea, eb, (double)jd.g.el, (double)jd.g.et);
          ang = angle_alignment_error(bound, ids, c);
        } else {
          const int ea[7] = {0, 1, 2, 3, 1, 2, 4}, eb[7] = {1, 2, 3, 0, 5, 4, 5};               // :663
          const int ids[3][4] = {{0, 1, 2, 3}, {3, 0, 4, 5}, {2, 4, 1, 5}};                     // :665
          sum_dist = edge_sum_dists<7, true>(map, jd.map_w, c, ea, eb, (double)jd.g.el, (double)jd.g.et);
          ang = angle_alignment_error(bound, ids, c);
        }
        const RpPose* pose = v.rp + jd.rp_off + rp;
        double p3[3], s3[3];
        lift_to_3d(c, pose->R, pose->t, v.invK + 9 * jd.frame, pose->plane, p3, s3);
        flag = pos;
        if (s3[0] < 0 || s3[1] < 0 || s3[2] < 0) flag |= CAND_NEG_SCALE;
        v.dist_err[slot] = sum_dist / jd.diag;
        v.angle_err[slot] = ang;
        v.skew[slot] = dmax(s3[0], s3[1]) / dmin(s3[0], s3[1]);
        double* co = v.corners + 16 * slot;
#pragma unroll
        for (int i = 0; i < 8; i++) { co[i] = c[i].x; co[8 + i] = c[i].y; }
      }
    }
    v.flag[slot] = flag;
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

// Ordered compaction: one block per job walks the job's slots in slot order.
__global__ __launch_bounds__(256) void compact_kernel(DetectDeviceView v) {
  int j = blockIdx.x;
  if (j >= v.n_jobs) return;
  __shared__ int wcnt[4];
  __shared__ long long run_s;
  long long s0 = v.slot_prefix[j], s1 = v.slot_prefix[j + 1];
  if (threadIdx.x == 0) run_s = v.job_cbase[j];
  __syncthreads();
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (long long base = s0; base < s1; base += 256) {
    long long s = base + threadIdx.x;
    int f = (s < s1) ? v.flag[s] : 0;
    unsigned long long b = __ballot(f != 0);
    if (lane == 0) wcnt[wid] = __popcll(b);
    __syncthreads();
    long long off = run_s;
    for (int w = 0; w < wid; w++) off += wcnt[w];
    if (f != 0) {
      long long pos = off + __popcll(b & ((1ull << lane) - 1ull));
      v.c_slot[pos] = s;
      v.c_flag[pos] = f;
      v.c_dist[pos] = v.dist_err[s];
      v.c_angle[pos] = v.angle_err[s];
      v.c_skew[pos] = v.skew[s];
    }
    __syncthreads();
    if (threadIdx.x == 0) run_s = off + wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void gather_corners_kernel(const double* corners, const long long* slots, int n, double* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 16) return;
  int w = i >> 4, k = i & 15;
  long long s = slots[w];
  out[i] = (s >= 0) ? corners[16 * s + k] : 0.0;
}

// ------------------------------------------------------------------ launchers (host side) -----
static inline unsigned grid8(long long n, int bs) {
  long long nb = (n + bs - 1) / bs;
  nb = (nb + 7) / 8 * 8;
  return (unsigned)(nb < 8 ? 8 : nb);
}

void launch_vp_support(const DetectDeviceView& v, const SweepParams& sp, int vp_total, hipStream_t st) {
  if (vp_total <= 0) return;
  hipLaunchKernelGGL(vp_support_kernel, dim3(grid8(vp_total, 256)), dim3(256), 0, st, v, sp, vp_total);
}
void launch_candidates(const DetectDeviceView& v, const SweepParams& sp, long long slot_total, hipStream_t st) {
  if (slot_total <= 0) return;
  hipLaunchKernelGGL(candidate_kernel, dim3(grid8(slot_total, 256)), dim3(256), 0, st, v, sp, slot_total);
}
void launch_scan_compact(const DetectDeviceView& v, hipStream_t st) {
  if (v.n_jobs <= 0) return;
  hipLaunchKernelGGL(scan_jobs_kernel, dim3(1), dim3(1024), 0, st, v.job_valid, v.job_cbase, v.n_jobs);
  hipLaunchKernelGGL(compact_kernel, dim3(v.n_jobs), dim3(256), 0, st, v);
}
void launch_gather_corners(const double* corners, const long long* slots, int n, double* out, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gather_corners_kernel, dim3((n * 16 + 255) / 256), dim3(256), 0, st, corners, slots, n, out);
}

}  // namespace cs
