// detect_host.cpp -- host side of path A (detect_cuboid) behind the C ABI of include/cubeslam_hip.h.
//
// Stages of one cs_batch_run() (production path: pipe_launch / pipe_finish):
//   setup   (host, threaded)  per frame: camera cache (box_proposal_detail.cpp:45-56); per box and height sample: integer
//                             box/ROI geometry (:143-256), yaw / top-edge / roll-pitch sample lists (:180-184, :212-219,
//                             :344-355)  ->  JobDesc + pooled SoA arrays in pinned memory, one H2D copy;
//   sweep   (HIP, 3 streams)  line setup (ROI filter + merge_break_lines) and VP support on one stream, vanishing points +
//                             corner construction + ordered compaction on a second, the crowded ROIs' line setup on a
//                             third; the scorer joins them (detect_kernels.hip);
//   rank    (HIP)             fuse_normalize_scores_v2 (object_3d_util.cpp:726-837) and the final skew-weighted ranking
//                             (box_proposal_detail.cpp:766-838) on the compacted (dist, angle, skew) columns; only the
//                             winners come back;
//   finish  (host, threaded)  cs_cuboid records of the winners (:740-798, object_3d_util.cpp:941-1011); the few boxes whose
//                             ties could reach the output are fetched and ranked with the exact std::partial_sort.
// The same stages exist as a general round-based path (debug getters, host ranking / host line setup on request).
// With whether_sample_cam_roll_pitch the reference carries cam_pose.camera_yaw from one box to the next
// (:180 reads what :374/:734 left behind), so boxes of a frame are processed in rounds (round r = box r of
// every frame), each ranked on the device; without it all boxes of all frames go through the sweep in one launch.
//
// There is no CPU fallback for the sweep: without a HIP device cs_detector_create() fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <chrono>
#include <cmath>
#include <limits>
#include <cstdio>
#include <cstring>
#include <functional>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cubeslam_hip.h"
#include "detect_types.h"

namespace cs {
void launch_vp_support(const DetectDeviceView& v, const SweepParams& sp, int vp_total, hipStream_t st);
void launch_vp_support_only(const DetectDeviceView& v, const SweepParams& sp, int vp_total, hipStream_t st, int rp_max = 0);      // rp_max: the jobs' largest roll/pitch sample count (0: unknown)
void launch_vp_points(const DetectDeviceView& v, int vp_total, hipStream_t st);
int vp3_table_doubles_per_job();
void launch_candidates(const DetectDeviceView& v, const SweepParams& sp, long long slot_total, hipStream_t st);
void launch_candidate_compact(const DetectDeviceView& v, const SweepParams& sp, hipStream_t st);
void launch_scan_compact(const DetectDeviceView& v, hipStream_t st);
void launch_scan_compact_trips(const DetectDeviceView& v, int* cnt, int max_trips, hipStream_t st);
void launch_score(const DetectDeviceView& v, const SweepParams& sp, long long n_valid_bound, long long slot_total, hipStream_t st);
void launch_gather_corners(const DetectDeviceView& v, const SweepParams& sp, const long long* slots, int n, double* out, hipStream_t st);
void launch_rank(const DetectDeviceView& v, const RankView& rv, const RankParams& rp, hipStream_t st, long long max_slots_per_box = 0, bool with_corners = true);
void launch_records(const DetectDeviceView& v, const RankView& rv, int kmax, cs_cuboid* out, hipStream_t st, const double* raw_euler = nullptr, double rebuild_short_sq_bound = -1.0);
void launch_rp_carry(const RpCarryView& c, JobDesc* jobs, hipStream_t st);
void launch_rp_save_fallback(const DetectDeviceView& v, const RpSaveView& s, hipStream_t st);
struct EdgeRoi { int l, t, w, h; long long img_off, cls_off, map_off; };
struct CopySeg { const void* src; void* dst; unsigned long long bytes; };
struct CopySegs { CopySeg s[16]; int n; };
void launch_multi_copy(const CopySegs& segs, hipStream_t st);
void launch_edge_maps(const unsigned char* gray, int W, int H, const EdgeRoi* rois, int n_rois, unsigned char* cls_pool, float* map_pool, int max_w, long long max_px, int low, int high,
                      hipStream_t st);
void launch_line_setup(JobDesc* jobs, int n_jobs, const double* frame_lines, const int* frame_line_ptr, double* mid_x, double* mid_y, double* line_angle,
                       double dist_thre, double angle_thre_deg, double len_thre, hipStream_t st, const int* order = nullptr, hipStream_t st_crowded = nullptr,
                       hipEvent_t fork = nullptr, hipEvent_t join = nullptr);
void launch_line_setup_listed(JobDesc* jobs, int n_jobs, const double* frame_lines, const int* frame_line_ptr, double* mid_x, double* mid_y, double* line_angle,
                              double dist_thre, double angle_thre_deg, double len_thre, hipStream_t st, const int* order, hipStream_t st_crowded, hipEvent_t fork, hipEvent_t join, int* crowded,
                              hipStream_t st_mid = nullptr, hipEvent_t join_mid = nullptr);
int line_setup_capacity();
void launch_gather_ranges(const DetectDeviceView& v, const long long* src_off, const int* count, const long long* dst_off, int n_ranges,
                          double* o_dist, double* o_angle, double* o_skew, int* o_flag, long long* o_slot, hipStream_t st);
}  // namespace cs

thread_local std::string g_cs_err;  // shared by both paths (ba_host.cpp reports through cs_set_error_ba)
void cs_set_error_ba(const std::string& s) { g_cs_err = s; }

namespace {

void set_err(const std::string& s) { g_cs_err = s; }

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      set_err(std::string(#expr) + ": " + hipGetErrorString(_e));                             \
      return CS_ERR_HIP;                                                                      \
    }                                                                                         \
  } while (0)

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Persistent worker pool for the host stages (one per detector; the calling thread takes part).
class WorkerPool {
 public:
  explicit WorkerPool(int n_workers) {
    for (int i = 0; i < n_workers; i++) th_.emplace_back([this]() { loop(); });
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; gen_++; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  // max_threads: at most this many threads (the caller included) take items -- for items that run for milliseconds, where more
  // runnable threads than the CPU quota tolerates only get the process throttled
  void run(int n, const std::function<void(int)>& fn, int max_threads = 0x7fffffff) {
    if (n <= 0) return;
    if (th_.empty() || n == 1) { for (int i = 0; i < n; i++) fn(i); return; }
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn; n_ = n; next_.store(0); joined_.store(0); limit_ = std::max(1, max_threads); pending_ = (int)th_.size(); gen_++;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this]() { return pending_ == 0; });
    fn_ = nullptr;
  }
 private:
  void work() {
    if (joined_.fetch_add(1) >= limit_) return;
    for (;;) {
      int i = next_.fetch_add(1);
      if (i >= n_) break;
      (*fn_)(i);
    }
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&]() { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
      }
      work();
      {
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  std::atomic<int> next_{0}, joined_{0};
  int n_ = 0, pending_ = 0, limit_ = 0x7fffffff;
  unsigned long long gen_ = 0;
  bool stop_ = false;
};

// Grow-only device / pinned buffers.
template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) return CS_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = n + n / 8 + 64;
    HIP_TRY(hipMalloc((void**)&p, want * sizeof(T)));
    cap = want;
    return CS_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
template <class T>
struct PinBuf {
  T* p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) return CS_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    size_t want = n + n / 8 + 64;
    HIP_TRY(hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocDefault));
    cap = want;
    return CS_OK;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// sin and cos stay two libm calls (the reference's default Debug build makes two calls; glibc's fused
// sincos() rounds differently in ~1.4e-3 of arguments).  The volatile copy keeps compilers from merging.
inline double h_sin(double x) { volatile double v = x; return std::sin(v); }
inline double h_cos(double x) { volatile double v = x; return std::cos(v); }

// ---------------------------------------------------------------- camera cache (set_cam_pose) ---
struct M3 { double m[9]; };

inline double cof3(const double* a, int i, int j) {
  int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return a[3 * i1 + j1] * a[3 * i2 + j2] - a[3 * i1 + j2] * a[3 * i2 + j1];
}
// 3x3 inverse in cofactor form, the algorithm Eigen's Matrix3d::inverse() uses.
void inv3(const double* a, double* r) {
  double c0 = cof3(a, 0, 0), c1 = cof3(a, 1, 0), c2 = cof3(a, 2, 0);
  double det = (c0 * a[0] + c1 * a[3]) + c2 * a[6];
  double id = 1.0 / det;
  r[0] = c0 * id; r[1] = c1 * id; r[2] = c2 * id;
  r[3] = cof3(a, 0, 1) * id; r[4] = cof3(a, 1, 1) * id; r[5] = cof3(a, 2, 1) * id;
  r[6] = cof3(a, 0, 2) * id; r[7] = cof3(a, 1, 2) * id; r[8] = cof3(a, 2, 2) * id;
}
void mul3(const double* a, const double* b, double* r) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
}
// Rotation matrix -> quaternion (Shoemake, as Eigen::Quaterniond(Matrix3d)) -> ZYX Euler angles
// (matrix_utils.cpp:38-49).
void rot_to_euler(const double* R, double e[3]) {
  double w, q[3];
  double t = R[0] + R[4] + R[8];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    w = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    w = (R[3 * k + j] - R[3 * j + k]) * t;
    q[j] = (R[3 * j + i] + R[3 * i + j]) * t;
    q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
  }
  double qx = q[0], qy = q[1], qz = q[2];
  e[0] = std::atan2(2 * (w * qx + qy * qz), 1 - 2 * (qx * qx + qy * qy));
  e[1] = std::asin(2 * (w * qy - qz * qx));
  e[2] = std::atan2(2 * (w * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz));
}
// matrix_utils.cpp:81-96
void euler_to_rot(double roll, double pitch, double yaw, double* R) {
  double cp = h_cos(pitch), sp = h_sin(pitch), sr = h_sin(roll), cr = h_cos(roll), sy = h_sin(yaw), cy = h_cos(yaw);
  R[0] = cp * cy; R[1] = (sr * sp * cy) - (cr * sy); R[2] = (cr * sp * cy) + (sr * sy);
  R[3] = cp * sy; R[4] = (sr * sp * sy) + (cr * cy); R[5] = (cr * sp * sy) - (sr * cy);
  R[6] = -sp; R[7] = sr * cp; R[8] = cr * cp;
}

struct CamCache {
  cs::RpPose pose;  // KinvR, R, t, ground plane, roll, pitch
  double euler[3];
  double cam_yaw;
};
// set_cam_pose (box_proposal_detail.cpp:45-56) for rotation R (row-major 3x3) and position t.
void make_cam(const double* K, const double* R, const double* t, CamCache& c) {
  std::memcpy(c.pose.R, R, 9 * sizeof(double));
  std::memcpy(c.pose.t, t, 3 * sizeof(double));
  rot_to_euler(R, c.euler);
  double invR[9];
  inv3(R, invR);
  mul3(K, invR, c.pose.KinvR);
  c.cam_yaw = c.euler[2];
  // ground plane in the sensor frame: T_wc^T (0,0,1,0)  (:130-131)
  for (int i = 0; i < 3; i++) c.pose.plane[i] = ((R[0 + i] * 0.0 + R[3 + i] * 0.0) + R[6 + i] * 1.0) + 0.0 * 0.0;
  c.pose.plane[3] = ((t[0] * 0.0 + t[1] * 0.0) + t[2] * 1.0) + 1.0 * 0.0;
  c.pose.roll = c.euler[0];
  c.pose.pitch = c.euler[1];
}

template <class T>
void linespace(T a, T b, T step, std::vector<T>& out) {  // matrix_utils.cpp:368-380
  while (a <= b) {
    out.push_back(a);
    a += step;
    if (out.size() > 1000) break;
  }
}

// merge_break_lines (object_3d_util.cpp:431-543) on a row-major n x 4 array.
void merge_lines(std::vector<double>& L, double dist_thre, double angle_thre_deg, double len_thre) {
  int total = (int)(L.size() / 4);
  const double athre = angle_thre_deg / 180.0 * CS_PI;
  // The reference recomputes every segment's angle at the top of each round (:451-456).  Only two rows change per
  // merge (row a is overwritten, row b receives the last row), so the cached angles below are the same doubles.
  std::vector<double> ang(total);
  for (int i = 0; i < total; i++) ang[i] = cs::cs_atan2(L[4 * i + 3] - L[4 * i + 1], L[4 * i + 2] - L[4 * i]);
  bool merged = true;
  int rounds = 0;
  while (merged && rounds < 500) {
    rounds++;
    merged = false;
    for (int a = 0; a < total - 1 && !merged; a++) {
      for (int b = a + 1; b < total; b++) {
        double diff = std::abs(ang[a] - ang[b]);
        if (std::min(diff, CS_PI - diff) >= athre) continue;
        double d_ab = cs::v2_dist(cs::v2(L[4 * a + 2], L[4 * a + 3]), cs::v2(L[4 * b], L[4 * b + 1]));
        double d_ba = cs::v2_dist(cs::v2(L[4 * b + 2], L[4 * b + 3]), cs::v2(L[4 * a], L[4 * a + 1]));
        if (!((d_ab < dist_thre) || (d_ba < dist_thre))) continue;
        const double* s = (L[4 * a] < L[4 * b]) ? &L[4 * a] : &L[4 * b];
        const double* e = (L[4 * a + 2] > L[4 * b + 2]) ? &L[4 * a + 2] : &L[4 * b + 2];
        double sx = s[0], sy = s[1], ex = e[0], ey = e[1];
        double ma = cs::cs_atan2(ey - sy, ex - sx);
        double t = std::abs(ang[a] - ma);
        if (std::min(t, CS_PI - t) < athre) {
          L[4 * a] = sx; L[4 * a + 1] = sy; L[4 * a + 2] = ex; L[4 * a + 3] = ey;
          for (int c = 0; c < 4; c++) L[4 * b + c] = L[4 * (total - 1) + c];  // swap-remove (matrix_utils.cpp:183)
          ang[a] = ma;               // atan2 of the merged row == merged_angle (:490)
          ang[b] = ang[total - 1];
          total--;
          merged = true;
          break;
        }
      }
    }
  }
  if (len_thre > 0) {
    int k = 0;
    for (int i = 0; i < total; i++) {
      double len = cs::v2_dist(cs::v2(L[4 * i + 2], L[4 * i + 3]), cs::v2(L[4 * i], L[4 * i + 1]));
      if (len > len_thre) {
        if (k != i) for (int c = 0; c < 4; c++) L[4 * k + c] = L[4 * i + c];
        k++;
      }
    }
    total = k;
  }
  L.resize(4 * (size_t)total);
}

// fuse_normalize_scores_v2 (object_3d_util.cpp:726-837)
void fuse_scores(const double* dist, const double* angle, int n, double w_angle, std::vector<int>& keep, std::vector<double>& score) {
  keep.clear();
  if (n > 4) {
    int bn = (int)std::round(float(n) / 3.0 * 2.0);
    // std::partial_sort of the ids by value (sort_indexes, matrix_utils.cpp:327-335).  The keys travel with the ids: the
    // algorithm makes the same comparisons and moves as on a bare id array (so ties resolve identically), without the
    // indirect loads
    struct KI { double key; int id; };
    std::vector<KI> ds(n), as(n);
    for (int i = 0; i < n; i++) { ds[i] = KI{dist[i], i}; as[i] = KI{angle[i], i}; }
    auto by_key = [](const KI& a, const KI& b) { return a.key < b.key; };
    std::partial_sort(ds.begin(), ds.begin() + bn, ds.end(), by_key);
    std::partial_sort(as.begin(), as.begin() + bn, as.end(), by_key);
    std::vector<int> di(bn), ai(bn);
    for (int i = 0; i < bn; i++) { di[i] = ds[i].id; ai[i] = as[i].id; }
    std::vector<int> dk(di.begin(), di.begin() + bn - 1);
    if (angle[ai[bn - 1]] > angle[ai[bn - 2]]) {
      // sort both id lists + set_intersection (:771-776) = the ids present in both, ascending: one pass over a marker array
      std::vector<unsigned char> in(n, 0);
      for (int i = 0; i < bn - 1; i++) { in[di[i]] |= 1; in[ai[i]] |= 2; }
      for (int id = 0; id < n; id++) if (in[id] == 3) keep.push_back(id);
    } else {
      keep = dk;
    }
  } else {
    keep.resize(n);
    std::iota(keep.begin(), keep.end(), 0);
  }
  int k = (int)keep.size();
  double dmin = 1e6, dmax = -1, amin = 1e6, amax = -1;
  for (int i = 0; i < k; i++) {
    double d = dist[keep[i]], a = angle[keep[i]];
    dmin = std::min(dmin, d); dmax = std::max(dmax, d);
    amin = std::min(amin, a); amax = std::max(amax, a);
  }
  score.resize(k);
  if (k > 1) {
    bool na = (amax - amin) > 0;
    for (int i = 0; i < k; i++) {
      double d = (dist[keep[i]] - dmin) / (dmax - dmin);
      double a = angle[keep[i]];
      if (na) a = (a - amin) / (amax - amin);
      score[i] = (d + w_angle * a) / (1 + w_angle);
    }
  } else {
    for (int i = 0; i < k; i++) score[i] = (dist[keep[i]] + w_angle * angle[keep[i]]) / (1 + w_angle);
  }
}

}  // namespace

// ================================================================== handles ======================
struct cs_detector {
  cs_detect_params prm;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // tie-break re-ranking fetches of the previous chunk, concurrent with the next chunk's sweep
  hipStream_t stream3 = nullptr;   // line setup of the crowded ROIs, beside the line setup of all the others
  hipStream_t stream4 = nullptr;   // ... of the ROIs of 129 .. 256 segments (their own, smaller instance)
  hipStream_t stream_hi = nullptr; // high priority: the small fetches of the tie boxes, which the host waits for while another batch's sweep owns the device
  hipEvent_t ev[12] = {};
  int n_threads = 1;
  int cpu_grant = 1;          // CPUs this process may use at once (cgroup quota, else the hardware threads)
  std::unique_ptr<WorkerPool> pool;
  // cs_detect_cuboids / cs_detect_cuboids_gray: one resident single-frame batch whose device and pinned buffers are reused from
  // call to call (a batch built and torn down per frame spent most of the call in hipMalloc / hipFree)
  cs_batch* single = nullptr;
  std::mutex single_mu;
  // the line-segment producer's resident scratch (lines_host.cpp owns its type; freed through lines_free)
  void* lines_scratch = nullptr;
  void (*lines_free)(void*) = nullptr;
  void* lsd_scratch = nullptr;          // the same for the LSD branch (lsd_host.cpp)
  void (*lsd_free)(void*) = nullptr;
  std::mutex lines_mu;
};

struct FrameIn {
  double K[9], invK[9], R[9], t[3];
  int img_w, img_h, n_boxes, n_lines;
  std::vector<double> boxes;     // n x 5
  std::vector<double> lines;     // M x 4, left-to-right aligned (object_3d_util.cpp:246-258)
  std::vector<long long> map_offs;  // per (box, k): float offset into the device map pool (-1 = absent)
  std::vector<cs_roi> rois;      // n x 3
  std::vector<int> n_heights;    // n
};

struct JobHost {          // host-side companion of a JobDesc of the current round
  int frame, box, hid;
  int yaw_off_local;
  std::vector<double> mids_x, mids_y, angs;
  std::vector<int> tops;
  int line_cap = 0;          // entries reserved in the pooled line tables (device setup: the frame's segment count)
};

struct FrameRound {       // setup products of one frame in one round
  std::vector<cs::JobDesc> jobs;
  std::vector<JobHost> jh;
  std::vector<double> yaw, yaw_c, yaw_s;   // concatenated yaw lists of the boxes in this round
};

struct JobResult {        // rank-stage products retained for cs_batch_debug_*
  int frame, box, hid, n_valid;
  std::vector<double> rows9;     // V x 9
  std::vector<long long> slots;  // V
  std::vector<int> rp_idx;       // V: roll/pitch sample of each candidate
  std::vector<int> keep;
  std::vector<double> score;
  std::vector<double> corners;   // V x 16 when debug is on
};

// One of the two in-flight chunks of the pipelined production path (run_pipelined).
struct PipeSlot {
  DevBuf<cs::JobDesc> jobs;
  DevBuf<unsigned char> tab_arena;       // a small call's tables in one block: one upload instead of ten (single-frame latency)
  PinBuf<unsigned char> h_tab_arena;
  bool merged_io = false;                // this slot's call went through the block: its results are unpacked from h_tab_arena in pipe_finish
  size_t o_jobs = 0, o_rec = 0, o_wc = 0, o_fb = 0, o_jv = 0, o_cb = 0, o_end = 0;
  DevBuf<long long> slot_prefix, job_cbase, c_slot, fb_src, fb_dst, fb_slot, win_slots;
  DevBuf<int> vp_prefix, top_x, flag, job_valid, c_flag, box_job0, box_njobs, win_count, fallback, fb_cnt, fb_flag;
  DevBuf<double> bound3;
  DevBuf<int> ls_order, blk_info, ls_crowded;
  PinBuf<int> h_ls_order;
  DevBuf<double> mid_x, mid_y, ang, yaw, yaw_c, yaw_s, vp, bound, corners, c_dist, c_angle, c_skew, fb_dist, fb_angle, fb_skew, win_corners;
  DevBuf<cs::RankWinner> winners;
  DevBuf<cs_cuboid> records;        // the winners' records of the device-ranked boxes (record_kernel)
  PinBuf<cs_cuboid> h_records;
  // lean roll/pitch path (rp_launch / rp_finish): all rounds of the batch queued at once
  struct RpRound { size_t j0 = 0, nj = 0, b0 = 0, nb = 0; long long slot_cap = 0; int vp_cap = 0; };
  std::vector<RpRound> rp_rounds;
  std::vector<int> rp_box_frame, rp_box_index;      // per box (all rounds): frame, box of the frame
  DevBuf<int> rp_trip_cnt;                           // valid slots per (job of a round, compaction trip)
  DevBuf<int> rp_cur_idx, rp_tab_count, rp_maps;    // per frame: list in force; (frame, list) -> samples; per round: box / first job / jobs of a frame
  DevBuf<long long> rp_last_slot, rp_box_base;
  DevBuf<unsigned long long> rp_pool_used;
  DevBuf<double> rp_raw_euler;
  PinBuf<int> h_rp_tab_count, h_rp_maps;
  PinBuf<double> h_rp_raw_euler;
  PinBuf<long long> h_rp_last_slot, h_rp_box_base;
  PinBuf<unsigned long long> h_rp_pool_used;
  long long rp_pool_cap = 0;
  int rp_NT = 0, rp_YCAP = 0;
  std::vector<hipEvent_t> rp_ev;     // 5 per round: start, corners + compaction, VP support, scorer, ranking + records
  bool rp_mode = false;
  PinBuf<cs::JobDesc> h_jobs_in, h_jobs_out;
  PinBuf<long long> h_slot_prefix, h_job_cbase;
  PinBuf<int> h_vp_prefix, h_top_x, h_box_job0, h_box_njobs, h_win_count, h_fallback, h_job_valid;
  PinBuf<double> h_yaw, h_yaw_c, h_yaw_s;
  PinBuf<cs::RankWinner> h_winners;
  PinBuf<long long> h_fb_src, h_fb_dst, h_fb_slot, h_win_slots;
  PinBuf<int> h_fb_cnt, h_fb_flag;
  PinBuf<double> h_fb_dist, h_fb_angle, h_fb_skew, h_win_corners;
  hipEvent_t done = nullptr, ev[14] = {};   // 0-6: phase marks on the main stream; 7: inputs resident; 8-11: second stream (corner construction); 12: line setup of the crowded ROIs done (third stream)
  cs::DetectDeviceView view{};
  int f0 = 0, f1 = 0, vp_total = 0;
  size_t nj = 0, nb = 0;
  long long slot_total = 0;
  bool in_flight = false;
  void release() {
    jobs.release(); slot_prefix.release(); job_cbase.release(); c_slot.release(); fb_src.release(); fb_dst.release(); fb_slot.release(); win_slots.release();
    vp_prefix.release(); top_x.release(); flag.release(); job_valid.release(); c_flag.release(); box_job0.release(); box_njobs.release(); win_count.release();
    fallback.release(); fb_cnt.release(); fb_flag.release(); mid_x.release(); mid_y.release(); ang.release(); yaw.release(); yaw_c.release(); yaw_s.release();
    vp.release(); bound.release(); bound3.release(); ls_order.release(); blk_info.release(); ls_crowded.release(); h_ls_order.release(); corners.release(); c_dist.release(); c_angle.release(); c_skew.release(); fb_dist.release(); fb_angle.release(); fb_skew.release();
    win_corners.release(); winners.release(); records.release(); h_records.release(); rp_trip_cnt.release(); rp_cur_idx.release(); rp_tab_count.release(); rp_maps.release(); rp_last_slot.release(); rp_box_base.release(); rp_pool_used.release(); rp_raw_euler.release(); h_rp_tab_count.release(); h_rp_maps.release(); h_rp_raw_euler.release(); h_rp_last_slot.release(); h_rp_box_base.release(); h_rp_pool_used.release(); h_jobs_in.release(); h_jobs_out.release(); h_slot_prefix.release(); h_job_cbase.release(); h_vp_prefix.release();
    h_top_x.release(); h_box_job0.release(); h_box_njobs.release(); h_win_count.release(); h_fallback.release(); h_job_valid.release(); h_yaw.release();
    h_yaw_c.release(); h_yaw_s.release(); h_winners.release();
    h_fb_src.release(); h_fb_dst.release(); h_fb_slot.release(); h_win_slots.release(); h_fb_cnt.release(); h_fb_flag.release(); h_fb_dist.release(); h_fb_angle.release();
    h_fb_skew.release(); h_win_corners.release();
    if (done) (void)hipEventDestroy(done);
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : rp_ev) if (e) (void)hipEventDestroy(e);
    rp_ev.clear();
    done = nullptr; for (auto& e : ev) e = nullptr;
  }
};

struct BatchRunState;   // what a submitted (not yet collected) sweep keeps alive: camera caches, timing, the caller's output arrays
struct cs_batch {
  cs_detector* det = nullptr;
  BatchRunState* run_state = nullptr;
  PinBuf<cs::RpPose> h_rp;
  DevBuf<unsigned char> d_gray, d_cls;      // image input: gray images and Canny's class bytes (kept: a refilled batch reuses them)
  DevBuf<cs::EdgeRoi> d_edge_rois;
  // cs_batch_refill_gray: a second image buffer (the upload of the next images runs on copy_stream beside the sweep over the current maps),
  // what launch_edge_maps needs again, and the events that order copy -> front end -> the buffer's next upload
  DevBuf<unsigned char> d_gray2;
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_edge_done[2] = {nullptr, nullptr};
  int gray_cur = 0, gray_w = 0, gray_h = 0, edge_n_rois = 0, edge_max_w = 1;
  long long edge_max_px = 1;
  bool gray_batch = false;
  int refill_q[2] = {-1, -1}, n_refill = 0;   // image buffers whose upload is on its way, oldest first; the next submit queues the oldest one's front end
  PinBuf<float> h_maps_stage;               // map upload staging
  PipeSlot pipe[2];
  // capacity layout of the lean path's staging pools (from the inputs alone): first job / first box of a frame, first
  // merged-segment row of a frame's jobs, first top-edge sample of a box
  std::vector<int> job_base, box_base, top_base;
  std::vector<long long> line_base;
  bool force_no_pipeline = false;
  int pipe_chunks = 1;   // chunks of the two-slot pipeline (cs_batch_set_pipeline_chunks)
  int n_frames = 0, max_boxes = 0;
  std::vector<FrameIn> frames;
  DevBuf<float> d_maps;
  DevBuf<double> d_invK, d_frame_lines;
  DevBuf<int> d_frame_line_ptr;
  bool device_setup = false;   // every frame's segment count fits the line-setup kernel's LDS table
  bool force_host_setup = false;
  // per-round device pools
  DevBuf<cs::JobDesc> d_jobs;
  DevBuf<long long> d_slot_prefix, d_job_cbase, d_c_slot, d_win_slots;
  DevBuf<int> d_vp_prefix, d_top_x, d_flag, d_job_valid, d_c_flag;
  DevBuf<double> d_mid_x, d_mid_y, d_ang, d_yaw, d_yaw_c, d_yaw_s, d_vp, d_bound, d_dist, d_angle, d_skew, d_corners, d_c_dist, d_c_angle, d_c_skew, d_win_corners;
  DevBuf<cs::RpPose> d_rp;
  PinBuf<char> h_stage;
  PinBuf<long long> h_c_slot, h_job_cbase;
  PinBuf<int> h_c_flag, h_job_valid;
  PinBuf<double> h_c_dist, h_c_angle, h_c_skew, h_win_corners;
  DevBuf<int> d_box_job0, d_box_njobs, d_win_count, d_fallback;
  DevBuf<long long> d_last_slot;
  PinBuf<long long> h_last_slot;
  DevBuf<cs::RankWinner> d_winners;
  DevBuf<long long> d_fb_src, d_fb_dst, d_fb_slot;
  DevBuf<int> d_fb_cnt, d_fb_flag;
  DevBuf<double> d_fb_dist, d_fb_angle, d_fb_skew;
  PinBuf<cs::RankWinner> h_winners;
  PinBuf<int> h_win_count, h_fallback;
  PinBuf<cs::JobDesc> h_jobs;
  PinBuf<unsigned char> h_tables, h_tables2;   // the job tables of one round, staged in pinned memory: their uploads are asynchronous and back to back
  bool force_host_rank = false;
  bool force_round_path = false;   // roll/pitch sampling through the round-by-round path (the lean path's redo of frames with tie boxes)
  // state
  bool debug = false, ran = false;
  std::vector<JobResult> results;             // all jobs of the last run (index via job_index)
  std::vector<int> job_index;                 // (frame*max_boxes + box)*3 + k -> results idx or -1
  cs_detect_timing timing{};
};


// The C ABI is the trust boundary: a 2D box must lie inside the image (its far edges are pixel coordinates the sweep
// intersects rays with and samples the distance map at).  The reference indexes the map unchecked (object_3d_util.cpp:651), so a
// box outside the image is undefined behaviour there; here it is CS_ERR_INVALID_ARG.
static bool box_inside_image(const double* box5, int img_w, int img_h) {
  if (img_w <= 0 || img_h <= 0) return false;
  for (int q = 0; q < 4; q++) if (!(box5[q] > -1e9 && box5[q] < 1e9)) return false;   // NaN / inf / absurd
  const int left = (int)box5[0], top = (int)box5[1], w = (int)box5[2], h = (int)box5[3];
  const int right = (int)(left + box5[2]), bottom = top + h;
  return left >= 0 && top >= 0 && w > 0 && h > 0 && right <= img_w - 1 && bottom <= img_h - 1;
}

// No C++ exception may cross the C boundary (std::bad_alloc / std::length_error from a host buffer would terminate the caller).
#define CS_GUARD_BEGIN try {
#define CS_GUARD_END(fn_name)                                                                      \
  } catch (const std::bad_alloc&) { set_err(std::string(fn_name) + ": out of host memory"); return CS_ERR_CAPACITY; } \
    catch (const std::exception& ex) { set_err(std::string(fn_name) + ": " + ex.what()); return CS_ERR_CAPACITY; }

extern "C" {

const char* cs_last_error(void) { return g_cs_err.c_str(); }

int cs_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int cs_diag_build(void) {
#ifdef CS_DIAG
  return 1;
#else
  return 0;
#endif
}

void cs_detect_default_params(cs_detect_params* p) {
  if (!p) return;
  p->consider_config_1 = 1; p->consider_config_2 = 1;
  p->whether_sample_cam_roll_pitch = 1; p->whether_sample_bbox_height = 0;
  p->max_cuboid_num = 1; p->nominal_skew_ratio = 1; p->max_cut_skew = 3;
  p->yaw_range_deg = 45; p->yaw_step_deg = 6;
  p->vp12_edge_angle_thre = 15; p->vp3_edge_angle_thre = 10; p->shorted_edge_thre = 20;
  p->weight_vp_angle = 0.8; p->weight_skew_error = 1.5;
  p->pre_merge_dist_thre = 20; p->pre_merge_angle_thre = 5; p->edge_length_threshold = 30;
  p->host_threads = 0;
}

int cs_box_rois(const double box5[5], int img_w, int img_h, int sample_height, cs_roi out[3]) {
  if (!box5 || !out) return CS_ERR_INVALID_ARG;
  int left = box5[0], top = box5[1], w = box5[2], h = box5[3];
  int right = left + box5[2];
  int downs[3], nd = 0;
  downs[nd++] = 0;
  if (sample_height) {
    int r = std::max(std::min(20, h - 90), 20);
    r = std::min(r, img_h - top - h - 1);
    if (r > 10) downs[nd++] = (int)std::round(r / 2);
    downs[nd++] = r;
  }
  for (int k = 0; k < nd; k++) {
    int he = h + downs[k];
    int dy = top + he;
    int e = std::min(std::max(std::min(20, w - 100), 10), std::max(std::min(20, he - 100), 10));
    int l = std::max(0, left - e), r = std::min(img_w - 1, right + e);
    int t = std::max(0, top - e), b = std::min(img_h - 1, dy + e);
    out[k].left = l; out[k].top = t; out[k].width = r - l; out[k].height = b - t; out[k].down_expand = downs[k];
  }
  return nd;
}

int cs_cam_euler_zyx(const double T_wc[16], double euler3[3]) {
  if (!T_wc || !euler3) return CS_ERR_INVALID_ARG;
  double R[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = T_wc[4 * i + j];
  rot_to_euler(R, euler3);
  return CS_OK;
}

// internal (not in the public header): the line-segment producer (lines_host.cpp) runs on the detector's stream
void* cs_internal_detector_stream(cs_detector* d) { return (void*)d->stream; }
// lines_host.cpp: the producer's scratch slot, its lock, and the detector's worker pool
void** cs_internal_detector_lines_slot(cs_detector* d, void (*deleter)(void*)) { d->lines_free = deleter; return &d->lines_scratch; }
void** cs_internal_detector_lsd_slot(cs_detector* d, void (*deleter)(void*)) { d->lsd_free = deleter; return &d->lsd_scratch; }
void* cs_internal_detector_lines_mutex(cs_detector* d) { return (void*)&d->lines_mu; }
void cs_internal_detector_parallel(cs_detector* d, int n, void (*fn)(int, void*), void* ctx) { d->pool->run(n, [&](int i) { fn(i, ctx); }); }
// (for the segment producers' image-long items: two threads per granted CPU at most, see cs_detector_create)
void cs_internal_detector_parallel_long(cs_detector* d, int n, void (*fn)(int, void*), void* ctx) { d->pool->run(n, [&](int i) { fn(i, ctx); }, 2 * d->cpu_grant + 1); }
int cs_internal_detector_device(cs_detector* d) { return d->device; }

int cs_detector_create(const cs_detect_params* params, int device, cs_detector** out) {
  if (!out) return CS_ERR_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_err("no HIP device visible; libcubeslam_hip has no CPU fallback");
    return CS_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) { set_err("device index out of range"); return CS_ERR_INVALID_ARG; }
  CS_GUARD_BEGIN
  struct Guard { cs_detector* d; ~Guard() { if (d) cs_detector_destroy(d); } } g{new cs_detector()};   // freed on every early return
  cs_detector* d = g.d;
  if (params) d->prm = *params; else cs_detect_default_params(&d->prm);
  if (d->prm.max_cuboid_num < 1 || !(d->prm.yaw_step_deg > 0) || !(d->prm.yaw_range_deg >= 0)) { set_err("bad params"); return CS_ERR_INVALID_ARG; }
  d->device = device;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&d->stream2, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&d->stream3, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&d->stream4, hipStreamNonBlocking));
  {
    int prio_low = 0, prio_high = 0;   // (numerically lowest = greatest priority)
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    HIP_TRY(hipStreamCreateWithPriority(&d->stream_hi, hipStreamNonBlocking, prio_high));
  }
  for (auto& e : d->ev) HIP_TRY(hipEventCreate(&e));
  int hc = (int)std::thread::hardware_concurrency();
  // default worker count: 64 on an unrestricted 256-thread EPYC (beats 32 and 128); under a cgroup CPU quota three threads
  // per granted CPU (measured under a 16-CPU quota: 48 threads 187 k frames/s, 32: 184 k, 64: 171-181 k, 16: 166-172 k --
  // the stages are short bursts, more runnable threads than the quota tolerates get the process throttled)
  int dflt = std::max(1, std::min(hc, 64));
  d->cpu_grant = std::max(1, hc);
  if (FILE* fq = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    long long quota = 0, period = 0;
    if (std::fscanf(fq, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
      dflt = std::max(1, std::min(dflt, std::max(8, (int)(3 * quota / period))));
      d->cpu_grant = std::max(1, std::min(d->cpu_grant, (int)(quota / period)));
    }
    std::fclose(fq);
  }
  d->n_threads = d->prm.host_threads > 0 ? d->prm.host_threads : dflt;
  d->pool.reset(new WorkerPool(d->n_threads - 1));
  *out = d;
  g.d = nullptr;
  return CS_OK;
  CS_GUARD_END("cs_detector_create")
}

void cs_detector_destroy(cs_detector* d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->single) { cs_batch_destroy(d->single); d->single = nullptr; }
  if (d->lines_scratch && d->lines_free) { d->lines_free(d->lines_scratch); d->lines_scratch = nullptr; }
  if (d->lsd_scratch && d->lsd_free) { d->lsd_free(d->lsd_scratch); d->lsd_scratch = nullptr; }
  for (auto& e : d->ev) if (e) (void)hipEventDestroy(e);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  if (d->stream2) (void)hipStreamDestroy(d->stream2);
  if (d->stream3) (void)hipStreamDestroy(d->stream3);
  if (d->stream4) (void)hipStreamDestroy(d->stream4);
  if (d->stream_hi) (void)hipStreamDestroy(d->stream_hi);
  delete d;
}

static int batch_fill(cs_detector* d, cs_batch* b, const cs_frame_desc* fr, const unsigned char* const* grays, int n_frames);
// Launch order of the distance-map front end's workgroups (one per ROI, its time grows with the ROI's area): largest first, so that the
// kernels do not end on a late-started large ROI.  The entries carry their own offsets: their order in the table means nothing else.
static void edge_rois_largest_first(std::vector<cs::EdgeRoi>& er) {
  if (er.size() <= 1024) return;
  std::stable_sort(er.begin(), er.end(), [](const cs::EdgeRoi& a, const cs::EdgeRoi& c) { return (long long)a.w * a.h > (long long)c.w * c.h; });
}
static int batch_layout(cs_detector* d, cs_batch* b);
static int batch_run_impl(cs_detector* d, cs_batch* b, cs_cuboid* out, int* out_counts, bool defer);
static int batch_flush_refill(cs_detector* d, cs_batch* b);
static void release_batch_buffers(cs_batch* b);
static int batch_create_impl(cs_detector* d, const cs_frame_desc* fr, const unsigned char* const* grays, int n_frames, cs_batch** out) {
  if (!d || !out || (!fr && n_frames) || n_frames < 0) return CS_ERR_INVALID_ARG;
  *out = nullptr;
  CS_GUARD_BEGIN
  HIP_TRY(hipSetDevice(d->device));
  struct Guard { cs_batch* b; ~Guard() { if (b) cs_batch_destroy(b); } } g{new cs_batch()};   // freed on every early return
  g.b->det = d;
  int rc = batch_fill(d, g.b, fr, grays, n_frames);
  if (rc) return rc;
  *out = g.b;
  g.b = nullptr;
  return CS_OK;
  CS_GUARD_END("cs_batch_create")
}
// (Re)describe the frames of a batch: host copies, ROIs, the map pool (uploaded, or produced in place from gray images), the
// segment pool, the capacity layout.  Device and pinned buffers only ever grow, so refilling a batch with frames of a similar size --
// the resident single-frame batch of cs_detect_cuboids -- allocates nothing.
static int batch_fill(cs_detector* d, cs_batch* b, const cs_frame_desc* fr, const unsigned char* const* grays, int n_frames) {
  b->n_frames = n_frames;
  b->max_boxes = 0;
  b->ran = false;
  b->frames.resize(n_frames);
  size_t map_floats = 0;
  for (int f = 0; f < n_frames; f++) {
    const cs_frame_desc& s = fr[f];
    FrameIn& F = b->frames[f];
    if (!s.K || !s.T_wc || s.n_boxes < 0 || s.n_lines < 0 || (s.n_boxes && (!s.boxes || (!grays && !s.dist_maps))) || (s.n_lines && !s.lines) || (grays && !grays[f])) {
      set_err("bad frame descriptor"); return CS_ERR_INVALID_ARG;
    }
    for (int i = 0; i < s.n_boxes; i++)
      if (!box_inside_image(s.boxes + 5 * (size_t)i, s.img_w, s.img_h)) { set_err("2D box outside the image (need 0 <= x, 0 <= y, x + w <= img_w - 1, y + h <= img_h - 1)"); return CS_ERR_INVALID_ARG; }
    if (s.T_wc[12] != 0 || s.T_wc[13] != 0 || s.T_wc[14] != 0 || s.T_wc[15] != 1) {
      set_err("T_wc must have last row 0 0 0 1"); return CS_ERR_INVALID_ARG;
    }
    std::memcpy(F.K, s.K, sizeof(F.K));
    inv3(F.K, F.invK);  // set_calibration (box_proposal_detail.cpp:38-42)
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) F.R[3 * i + j] = s.T_wc[4 * i + j];
      F.t[i] = s.T_wc[4 * i + 3];
    }
    F.img_w = s.img_w; F.img_h = s.img_h; F.n_boxes = s.n_boxes; F.n_lines = s.n_lines;
    F.boxes.assign(s.boxes, s.boxes + 5 * (size_t)s.n_boxes);
    F.lines.assign(s.lines, s.lines + 4 * (size_t)s.n_lines);
    for (int i = 0; i < s.n_lines; i++)  // align_left_right_edges
      if (F.lines[4 * i + 2] < F.lines[4 * i]) {
        std::swap(F.lines[4 * i], F.lines[4 * i + 2]);
        std::swap(F.lines[4 * i + 1], F.lines[4 * i + 3]);
      }
    F.rois.resize(3 * (size_t)s.n_boxes);
    F.n_heights.resize(s.n_boxes);
    F.map_offs.assign(3 * (size_t)s.n_boxes, -1);
    for (int i = 0; i < s.n_boxes; i++) {
      int nh = cs_box_rois(&F.boxes[5 * i], F.img_w, F.img_h, d->prm.whether_sample_bbox_height, &F.rois[3 * i]);
      F.n_heights[i] = nh;
      for (int k = 0; k < nh; k++) {
        const cs_roi& r = F.rois[3 * i + k];
        if (r.width <= 0 || r.height <= 0 || (!grays && !s.dist_maps[3 * i + k])) { set_err("missing distance map / empty ROI"); return CS_ERR_INVALID_ARG; }
        F.map_offs[3 * i + k] = (long long)map_floats;
        map_floats += (size_t)r.width * r.height + r.width + 1;  // + one row + one float of zero padding
      }
    }
    b->max_boxes = std::max(b->max_boxes, s.n_boxes);
  }
  // upload the distance maps (zero padded) and the per-frame invK
  int rc = b->d_maps.ensure(map_floats + 1);
  if (rc) { return rc; }
  {
    if (!grays) {
      // small pools (a frame, a few frames) go through pinned staging kept by the batch and a queued copy; a large batch's pool is
      // staged once in pageable memory and copied synchronously -- pinning gigabytes for a one-off upload costs more than it saves
      const bool pinned = map_floats + 1 <= ((size_t)16 << 20);
      std::vector<float> big;
      float* stage = nullptr;
      if (pinned) { rc = b->h_maps_stage.ensure(map_floats + 1); if (rc) return rc; stage = b->h_maps_stage.p; std::memset(stage, 0, sizeof(float) * (map_floats + 1)); }
      else { big.assign(map_floats + 1, 0.0f); stage = big.data(); }
      for (int f = 0; f < n_frames; f++) {
        FrameIn& F = b->frames[f];
        for (int i = 0; i < F.n_boxes; i++)
          for (int k = 0; k < F.n_heights[i]; k++) {
            const cs_roi& r = F.rois[3 * i + k];
            std::memcpy(&stage[F.map_offs[3 * i + k]], fr[f].dist_maps[3 * i + k], sizeof(float) * (size_t)r.width * r.height);
          }
      }
      if (pinned) HIP_TRY(hipMemcpyAsync(b->d_maps.p, stage, sizeof(float) * (map_floats + 1), hipMemcpyHostToDevice, d->stream));   // the sweep follows on the same stream
      else HIP_TRY(hipMemcpy(b->d_maps.p, stage, sizeof(float) * (map_floats + 1), hipMemcpyHostToDevice));
    } else {
      // image in: upload the gray images, produce every job's map in place in the pool (Canny + distance transform on the
      // device, box_proposal_detail.cpp:320-327); the padding between the maps stays zero.  All frames share one size.
      const int W = n_frames ? fr[0].img_w : 0, H = n_frames ? fr[0].img_h : 0;
      std::vector<cs::EdgeRoi> er;
      long long cls_tot = 0, max_px = 1;
      int max_w = 1;
      for (int f = 0; f < n_frames; f++) {
        FrameIn& F = b->frames[f];
        if (F.img_w != W || F.img_h != H) { set_err("cs_batch_create_gray: all frames must have the same image size"); return CS_ERR_INVALID_ARG; }
        for (int i = 0; i < F.n_boxes; i++)
          for (int k = 0; k < F.n_heights[i]; k++) {
            const cs_roi& r = F.rois[3 * i + k];
            if (r.left < 0 || r.top < 0 || r.left + r.width > W || r.top + r.height > H) { set_err("ROI outside the image"); return CS_ERR_INVALID_ARG; }
            er.push_back(cs::EdgeRoi{r.left, r.top, r.width, r.height, (long long)f * W * H, cls_tot, F.map_offs[3 * i + k]});
            cls_tot += (long long)r.width * r.height;
            max_w = std::max(max_w, r.width); max_px = std::max(max_px, 4LL * ((r.width + 5) / 4) * (r.height + 2));
          }
      }
      DevBuf<unsigned char>& d_gray = b->d_gray; DevBuf<unsigned char>& d_cls = b->d_cls;
      DevBuf<cs::EdgeRoi>& d_rois = b->d_edge_rois;
      if ((rc = d_gray.ensure((size_t)W * H * std::max(1, n_frames))) || (rc = d_cls.ensure((size_t)cls_tot + 8)) || (rc = d_rois.ensure(er.size() + 1))) { return rc; }
      hipStream_t st = d->stream;
      b->gray_batch = true; b->gray_w = W; b->gray_h = H; b->edge_n_rois = (int)er.size(); b->edge_max_w = max_w; b->edge_max_px = max_px; b->gray_cur = 0;
      HIP_TRY(hipMemsetAsync(b->d_maps.p, 0, sizeof(float) * (map_floats + 1), st));
      for (int f = 0; f < n_frames; f++) HIP_TRY(hipMemcpyAsync(d_gray.p + (size_t)f * W * H, grays[f], (size_t)W * H, hipMemcpyHostToDevice, st));
      if (!er.empty()) {
        edge_rois_largest_first(er);
        HIP_TRY(hipMemcpyAsync(d_rois.p, er.data(), sizeof(cs::EdgeRoi) * er.size(), hipMemcpyHostToDevice, st));
        cs::launch_edge_maps(d_gray.p, W, H, d_rois.p, (int)er.size(), d_cls.p, b->d_maps.p, max_w, max_px, 80, 200, st);
        HIP_TRY(hipGetLastError());
      }
      HIP_TRY(hipStreamSynchronize(st));      // (the host arrays `grays` and `er` are read by the copies above)
    }
  }
  return batch_layout(d, b);
}
// Second half of describing a batch: per-frame invK, the pooled line segments, the capacity layout of the staging pools.
static int batch_layout(cs_detector* d, cs_batch* b) {
  const int n_frames = b->n_frames;
  int rc;
  {
    std::vector<double> ik(9 * (size_t)std::max(1, n_frames));
    for (int f = 0; f < n_frames; f++) std::memcpy(&ik[9 * f], b->frames[f].invK, 9 * sizeof(double));
    rc = b->d_invK.ensure(ik.size());
    if (rc) { return rc; }
    HIP_TRY(hipMemcpy(b->d_invK.p, ik.data(), sizeof(double) * ik.size(), hipMemcpyHostToDevice));
    // line segments (already left-to-right aligned), pooled, for the device-side line setup
    std::vector<int> lp(n_frames + 1, 0);
    b->device_setup = true;
    for (int f = 0; f < n_frames; f++) { lp[f + 1] = lp[f] + b->frames[f].n_lines; if (b->frames[f].n_lines > cs::line_setup_capacity()) b->device_setup = false; }
    std::vector<double> fl(4 * (size_t)std::max(1, lp[n_frames]));
    for (int f = 0; f < n_frames; f++) if (b->frames[f].n_lines) std::memcpy(&fl[4 * (size_t)lp[f]], b->frames[f].lines.data(), 32 * (size_t)b->frames[f].n_lines);
    rc = b->d_frame_lines.ensure(fl.size()); if (rc) { return rc; }
    rc = b->d_frame_line_ptr.ensure(lp.size()); if (rc) { return rc; }
    HIP_TRY(hipMemcpy(b->d_frame_lines.p, fl.data(), 8 * fl.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_frame_line_ptr.p, lp.data(), 4 * lp.size(), hipMemcpyHostToDevice));
  }
  {
    b->job_base.assign(n_frames + 1, 0); b->box_base.assign(n_frames + 1, 0); b->line_base.assign(n_frames + 1, 0);
    b->top_base.assign(1, 0);
    for (int f = 0; f < n_frames; f++) {
      const FrameIn& F = b->frames[f];
      int nj = 0;
      for (int bi = 0; bi < F.n_boxes; bi++) {
        nj += F.n_heights[bi];
        const double* bb = &F.boxes[5 * bi];
        int left = bb[0], w = bb[2], right = left + bb[2];
        int res = (int)std::round(std::min(20, w / 10));
        std::vector<int> tops;
        if (res >= 1) linespace<int>(left + 5, right - 5, res, tops);   // :215-219: the count depends on the box alone
        b->top_base.push_back(b->top_base.back() + (int)tops.size());
      }
      b->job_base[f + 1] = b->job_base[f] + nj;
      b->box_base[f + 1] = b->box_base[f] + F.n_boxes;
      b->line_base[f + 1] = b->line_base[f] + (long long)nj * F.n_lines;
    }
  }
  return CS_OK;
}

int cs_batch_create(cs_detector* d, const cs_frame_desc* fr, int n_frames, cs_batch** out) { return batch_create_impl(d, fr, nullptr, n_frames, out); }

int cs_batch_create_gray(cs_detector* d, const cs_frame_desc* fr, const unsigned char* const* grays, int n_frames, cs_batch** out) {
  if (n_frames > 0 && !grays) return CS_ERR_INVALID_ARG;
  return batch_create_impl(d, fr, grays, n_frames, out);
}

// New images for the frames of an image-input batch (same frame descriptions: boxes, segments, cameras): the upload goes to the batch's OTHER
// image buffer on a copy stream of its own -- beside whatever sweep is running --, the front end (Canny + distance transform of every ROI)
// is queued by the next cs_batch_submit in front of its sweep.  Up to two uploads may be queued (one per image buffer): a caller that
// queues the upload AFTER next before each submit keeps the copy engine busy without a gap, and the whole device side of a step (front end
// + sweep) runs beside an upload.  Returns at once.
int cs_batch_refill_gray(cs_detector* d, cs_batch* b, const unsigned char* const* grays) {
  if (!d || !b || !grays || b->det != d) return CS_ERR_INVALID_ARG;
  if (!b->gray_batch) { set_err("cs_batch_refill_gray: the batch was not created by cs_batch_create_gray"); return CS_ERR_INVALID_ARG; }
  CS_GUARD_BEGIN
  HIP_TRY(hipSetDevice(d->device));
  const size_t px = (size_t)b->gray_w * b->gray_h;
  const int n = b->n_frames;
  if (n <= 0 || px == 0) return CS_OK;
  int rc;
  if ((rc = b->d_gray2.ensure(px * (size_t)n))) return rc;
  if (b->n_refill == 2) { set_err("cs_batch_refill_gray: two uploads are queued already (one per image buffer); cs_batch_submit / cs_batch_run takes the older one"); return CS_ERR_INVALID_ARG; }
  if (!b->copy_stream) {
    // (lowest priority: bulk traffic, and a hardware queue of its own)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    HIP_TRY(hipStreamCreateWithPriority(&b->copy_stream, hipStreamNonBlocking, prio_least));
    for (auto& e : b->ev_copied) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : b->ev_edge_done) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(b->ev_edge_done[0], d->stream));      // buffer 0's reader so far: the front end of cs_batch_create_gray (finished)
    HIP_TRY(hipEventRecord(b->ev_edge_done[1], d->stream));
  }
  // Which buffer: the one that is neither being read (gray_cur, or -- with an upload already queued -- that upload's target, which the next
  // submit's front end reads) ...  With one upload queued the new one goes BEHIND it on the copy stream into the buffer of the current maps:
  // the stream never idles between two batches of images, whatever the host does in between.
  const int nxt = b->n_refill == 1 ? 1 - b->refill_q[0] : 1 - b->gray_cur;
  unsigned char* dst = nxt ? b->d_gray2.p : b->d_gray.p;
  // the buffer's last reader -- the front end queued when it became current -- must be through before it is overwritten
  HIP_TRY(hipStreamWaitEvent(b->copy_stream, b->ev_edge_done[nxt], 0));
  // images that follow each other in host memory go up as one copy (a batch decoded into one pinned block: a single DMA)
  for (int f = 0; f < n;) {
    int g = f + 1;
    while (g < n && grays[g] == grays[g - 1] + px) g++;
    // The copy engine, not the shader cores: a kernel that reads pinned host memory keeps hundreds of PCIe reads in flight through the same
    // request queues the other kernels' HBM traffic takes -- with it beside them the distance transform ran 0.8 -> 7.7 ms and Canny
    // 2.4 -> 7.5 ms (profiles/r6_image_in_timeline.txt); the engine's 8.3 ms for 467 MB is the same and costs the kernels nothing.
    HIP_TRY(hipMemcpyAsync(dst + (size_t)f * px, grays[f], px * (size_t)(g - f), hipMemcpyHostToDevice, b->copy_stream));
    f = g;
  }
  HIP_TRY(hipEventRecord(b->ev_copied[nxt], b->copy_stream));
  b->refill_q[b->n_refill++] = nxt;
  return CS_OK;
  CS_GUARD_END("cs_batch_refill_gray")
}
// The front end over freshly uploaded images, queued by the next cs_batch_submit / cs_batch_run: the HOST waits for the upload and then queues
// the kernels.  (First form: the detector's stream waited for the copy's event on the device.  A wait that sits in a hardware queue for the
// 8 ms of a 467 MB upload blocks every stream the runtime maps to that queue -- the tie boxes' gather on the detector's second stream
// waited behind it and a step took 16.7 ms instead of 8.3.)
static int batch_flush_refill(cs_detector* d, cs_batch* b) {
  if (b->n_refill == 0) return CS_OK;
  const int nxt = b->refill_q[0];
  HIP_TRY(hipEventSynchronize(b->ev_copied[nxt]));
  hipStream_t st = d->stream;
  if (b->edge_n_rois > 0) {
    cs::launch_edge_maps(nxt ? b->d_gray2.p : b->d_gray.p, b->gray_w, b->gray_h, b->d_edge_rois.p, b->edge_n_rois, b->d_cls.p, b->d_maps.p, b->edge_max_w, b->edge_max_px, 80, 200, st);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(b->ev_edge_done[nxt], st));
  b->gray_cur = nxt;
  b->refill_q[0] = b->refill_q[1]; b->refill_q[1] = -1; b->n_refill--;
  return CS_OK;
}
// ... and wait for the upload only (the host images may be reused / freed from here; the front end may still be queued)
int cs_batch_refill_wait(cs_batch* b) {
  if (!b) return CS_ERR_INVALID_ARG;
  if (!b->copy_stream) return CS_OK;
  HIP_TRY(hipSetDevice(b->det->device));
  HIP_TRY(hipStreamSynchronize(b->copy_stream));
  return CS_OK;
}

int cs_batch_max_boxes(const cs_batch* b) { return b ? b->max_boxes : CS_ERR_INVALID_ARG; }

static void batch_drop_run_state(cs_batch* b);
void cs_batch_destroy(cs_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->det->device);
  if (b->run_state) { (void)hipDeviceSynchronize(); batch_drop_run_state(b); }   // a submitted sweep that was never collected
  release_batch_buffers(b);
  delete b;
}
static void release_batch_buffers(cs_batch* b) {
  b->pipe[0].release(); b->pipe[1].release();
  b->d_maps.release(); b->d_invK.release(); b->d_frame_lines.release(); b->d_frame_line_ptr.release(); b->d_jobs.release(); b->d_slot_prefix.release(); b->d_job_cbase.release();
  b->d_c_slot.release(); b->d_win_slots.release(); b->d_vp_prefix.release(); b->d_top_x.release(); b->d_flag.release();
  b->d_job_valid.release(); b->d_c_flag.release(); b->d_mid_x.release(); b->d_mid_y.release(); b->d_ang.release();
  b->d_yaw.release(); b->d_yaw_c.release(); b->d_yaw_s.release(); b->d_vp.release(); b->d_bound.release(); b->d_dist.release();
  b->d_angle.release(); b->d_skew.release(); b->d_corners.release(); b->d_c_dist.release(); b->d_c_angle.release();
  b->d_c_skew.release(); b->d_win_corners.release(); b->d_rp.release(); b->h_rp.release();
  b->d_gray.release(); b->d_cls.release(); b->d_edge_rois.release(); b->h_maps_stage.release(); b->d_gray2.release();
  if (b->copy_stream) { (void)hipStreamSynchronize(b->copy_stream); (void)hipStreamDestroy(b->copy_stream); b->copy_stream = nullptr; }
  for (auto& e : b->ev_copied) if (e) { (void)hipEventDestroy(e); e = nullptr; }
  for (auto& e : b->ev_edge_done) if (e) { (void)hipEventDestroy(e); e = nullptr; }
  b->h_stage.release(); b->h_c_slot.release(); b->h_job_cbase.release(); b->h_c_flag.release(); b->h_job_valid.release();
  b->h_c_dist.release(); b->h_c_angle.release(); b->h_c_skew.release(); b->h_win_corners.release();
  b->d_box_job0.release(); b->d_box_njobs.release(); b->d_win_count.release(); b->d_fallback.release(); b->d_winners.release(); b->d_last_slot.release(); b->h_last_slot.release();
  b->h_winners.release(); b->h_win_count.release(); b->h_fallback.release(); b->h_jobs.release(); b->h_tables.release(); b->h_tables2.release();
  b->d_fb_src.release(); b->d_fb_dst.release(); b->d_fb_slot.release(); b->d_fb_cnt.release(); b->d_fb_flag.release();
  b->d_fb_dist.release(); b->d_fb_angle.release(); b->d_fb_skew.release();
}

int cs_batch_set_debug(cs_batch* b, int enable) {
  if (!b) return CS_ERR_INVALID_ARG;
  b->debug = (enable & 1) != 0;
  b->force_host_rank = (enable & 2) != 0;
  b->force_host_setup = (enable & 4) != 0;
  b->force_no_pipeline = (enable & 8) != 0;
  return CS_OK;
}

}  // extern "C"

namespace {

struct Proposal {   // one entry of raw_obj_proposals (box_proposal_detail.cpp:195, :797)
  int res_idx;      // JobResult index
  int cand;         // candidate index inside the job (raw_cube_ind)
  double normalized_error, skew;
};

struct Winner {
  int frame, box, rank;
  int res_idx, cand;
  double normalized_error, skew;
};

// Build one cs_cuboid from the winner's corners (box_proposal_detail.cpp:740-798, object_3d_util.cpp:941-1011).
void finish_cuboid(const FrameIn& F, const cs::RpPose& pose, const double* rows9, const double* corners16, const double raw_euler[3],
                   bool sample_rp, double normalized_error, cs_cuboid& o) {
  std::memset(&o, 0, sizeof(o));
  cs::V2 c[8];
  for (int i = 0; i < 8; i++) c[i] = cs::v2(corners16[i], corners16[8 + i]);
  cs::lift_to_3d(c, pose.R, pose.t, F.invK, pose.plane, o.pos, o.scale);
  o.rotY = rows9[2];
  o.box_config_type[0] = rows9[0]; o.box_config_type[1] = rows9[1];
  static const int left_ids[8] = {6, 5, 8, 7, 2, 3, 4, 1}, right_ids[8] = {5, 6, 7, 8, 3, 2, 1, 4};
  const int* ids = (rows9[1] == 1) ? left_ids : right_ids;
  for (int i = 0; i < 8; i++) {
    o.box_corners_2d[i] = (int)corners16[ids[i] - 1];
    o.box_corners_2d[8 + i] = (int)corners16[8 + ids[i] - 1];
  }
  // compute3D_BoxCorner (object_3d_util.cpp:59-73) with similarityTransformation (:15-44)
  static const double body[3][8] = {{1, 1, -1, -1, 1, 1, -1, -1}, {1, -1, -1, 1, 1, -1, -1, 1}, {-1, -1, -1, -1, 1, 1, 1, 1}};
  double cr = h_cos(o.rotY), sr = h_sin(o.rotY);
  double rot[3][3] = {{cr, -sr, 0}, {sr, cr, 0}, {0, 0, 1}};
  double S[4][4] = {{0}};
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) S[i][j] = rot[i][j] * o.scale[j];
    S[i][3] = o.pos[i];
  }
  S[3][3] = 1;
  for (int k = 0; k < 8; k++) {
    double p[4] = {body[0][k], body[1][k], body[2][k], 1.0}, w[4];
    for (int i = 0; i < 4; i++) w[i] = ((S[i][0] * p[0] + S[i][1] * p[1]) + S[i][2] * p[2]) + S[i][3] * p[3];
    for (int i = 0; i < 3; i++) o.box_corners_3d_world[8 * i + k] = w[i] / w[3];
  }
  o.edge_distance_error = rows9[4];
  o.edge_angle_error = rows9[5];
  o.normalized_error = normalized_error;
  o.skew_ratio = std::max(o.scale[0], o.scale[1]) / std::min(o.scale[0], o.scale[1]);
  o.down_expand_height = rows9[6];
  if (sample_rp) {
    o.camera_roll_delta = rows9[7] - raw_euler[0];
    o.camera_pitch_delta = rows9[8] - raw_euler[1];
  }
}

}  // namespace


// ================================================================== pipelined production path =====
// No roll/pitch sampling, no debug retention: boxes are independent, line setup and ranking run on the device, and
// nothing but the winners comes back.  The batch is cut into chunks that flow through a two-slot software pipeline:
// while the GPU sweeps chunk k, the host packs chunk k+1 and writes the records of chunk k-1.
namespace {

struct PipeCtx {
  cs_detector* d; cs_batch* b; cs_cuboid* out; int* out_counts;
  const std::vector<CamCache>* cam_raw; const std::vector<int>* rp_off;
  cs::SweepParams sp; cs_detect_timing* tm;
  const std::vector<std::vector<CamCache>>* cam_rp_all = nullptr;
};

}  // namespace
struct BatchRunState {
  std::vector<CamCache> cam_raw;
  std::vector<std::vector<CamCache>> cam_rp;
  std::vector<int> rp_off;
  cs_detect_timing tm{};
  double t_begin = 0;
  PipeCtx C{};
  bool deferred = false;    // pipe_launch done, pipe_finish pending (cs_batch_collect)
  bool rp_lean = false;     // ... of the lean roll/pitch path (rp_launch / rp_finish)
};
namespace {

static double g_mark[16];
static int g_runs = 0;
static const bool g_prof = getenv("CS_DETECT_PROF") != nullptr;   // diagnostics: host phase clock of the lean path
static const bool g_split_candidates = getenv("CS_DETECT_SPLIT_CANDIDATES") != nullptr;   // vp_points + candidate + scan + compact as separate kernels (the form before round 6) instead of candidate_compact_kernel
static const bool g_dma_tables = getenv("CS_DETECT_DMA_TABLES") != nullptr;   // a batch's tables / results through hipMemcpyAsync (the form before round 6) instead of multi_copy_kernel
#define MARK(k, t_ref) do { if (g_prof) { double t_now = now_ms(); g_mark[k] += t_now - (t_ref); (t_ref) = t_now; } } while (0)

int pipe_launch(PipeCtx& C, PipeSlot& S, int f0, int f1) {
  cs_detector* d = C.d; cs_batch* b = C.b;
  const cs_detect_params& P = d->prm;
  hipStream_t st = d->stream;
  const int KMAX = P.max_cuboid_num;
  double t0 = now_ms(), tq = t0;
  if (!S.done) { HIP_TRY(hipEventCreate(&S.done)); for (auto& e : S.ev) HIP_TRY(hipEventCreate(&e)); }
  S.f0 = f0; S.f1 = f1;
  const int nf = f1 - f0;
  // ---- per frame, in parallel and straight into the pinned staging pools: job descriptors + sample lists (everything
  // else of the setup happens in line_setup_kernel).  The pools use a capacity layout fixed by the inputs (a frame's
  // jobs, yaw samples and top-edge samples have known upper bounds), so no frame waits for another one's counts; a
  // box the reference skips (:215) leaves null jobs (Y = T = 0) that own no slots.
  const int jb0 = b->job_base[f0], bx0 = b->box_base[f0], tp0 = b->top_base[bx0];
  const long long ln0 = b->line_base[f0];
  const size_t nj = (size_t)(b->job_base[f1] - jb0), n_lines = (size_t)(b->line_base[f1] - ln0);
  const int YCAP = (int)(2.0 * P.yaw_range_deg / std::max(1e-9, P.yaw_step_deg)) + 3;
  const size_t n_yaw = (size_t)nf * YCAP, n_top = (size_t)(b->top_base[b->box_base[f1]] - tp0);
  if ((long long)nj * 1 > 0x7fffffffLL || (long long)n_lines > 0x7fffffffLL) { set_err("chunk too large"); return CS_ERR_CAPACITY; }
  S.nj = nj;
  if (nj == 0) { S.nb = 0; S.slot_total = 0; S.vp_total = 0; S.in_flight = true; C.tm->setup_host_ms += now_ms() - t0; HIP_TRY(hipEventRecord(S.done, st)); return CS_OK; }
  int rc;
#define PENS(buf, n) do { rc = (buf).ensure(n); if (rc) return rc; } while (0)
  PENS(S.h_jobs_in, nj); PENS(S.h_slot_prefix, nj + 1); PENS(S.h_vp_prefix, nj + 1); PENS(S.h_yaw, n_yaw + 1); PENS(S.h_yaw_c, n_yaw + 1); PENS(S.h_yaw_s, n_yaw + 1);
  PENS(S.h_top_x, n_top + 1); PENS(S.h_box_job0, nj); PENS(S.h_box_njobs, nj); PENS(S.h_ls_order, nj);
  std::atomic<int> overflow{0};
  d->pool->run(nf, [&](int q) {
    const int f = f0 + q;
    const FrameIn& F = b->frames[f];
    cs::JobDesc* jout = S.h_jobs_in.p + (b->job_base[f] - jb0);
    double* yw = S.h_yaw.p + (size_t)q * YCAP; double* yc = S.h_yaw_c.p + (size_t)q * YCAP; double* ys = S.h_yaw_s.p + (size_t)q * YCAP;
    int nY = -1, ji = 0;
    for (int bi = 0; bi < F.n_boxes; bi++) {
      const double* bb = &F.boxes[5 * bi];
      int left = bb[0], top = bb[1], w = bb[2], h = bb[3];
      int right = left + bb[2];
      int res = (int)std::round(std::min(20, w / 10));
      const int tb = b->top_base[b->box_base[f] + bi] - tp0, tcap = b->top_base[b->box_base[f] + bi + 1] - tp0 - tb;
      int nT = 0;
      if (res >= 1) {
        if (nY < 0) {  // cam_pose never changes without roll/pitch sampling: one yaw list per frame (:180-184)
          double yaw_init = (*C.cam_raw)[f].cam_yaw - 90.0 / 180.0 * CS_PI;
          std::vector<double> yl;
          linespace<double>(yaw_init - P.yaw_range_deg / 180.0 * CS_PI, yaw_init + P.yaw_range_deg / 180.0 * CS_PI, P.yaw_step_deg / 180.0 * CS_PI, yl);
          nY = (int)yl.size();
          if (nY > YCAP) { overflow = 1; nY = 0; }
          for (int y = 0; y < nY; y++) { yw[y] = yl[y]; yc[y] = h_cos(yl[y]); ys[y] = h_sin(yl[y]); }
        }
        int* tx = S.h_top_x.p + tb;
        for (int x = left + 5; x <= right - 5 && nT < tcap; x += res) tx[nT++] = x;   // linespace<int> (matrix_utils.cpp:368-380)
      }
      for (int k = 0; k < F.n_heights[bi]; k++) {
        const cs_roi& roi = F.rois[3 * bi + k];
        cs::JobDesc jd;
        std::memset(&jd, 0, sizeof(jd));
        int he = h + roi.down_expand;
        jd.g.left = left; jd.g.top = top; jd.g.right = right; jd.g.down = top + he;
        jd.g.el = roi.left; jd.g.et = roi.top; jd.g.er = roi.left + roi.width; jd.g.eb = roi.top + roi.height;
        jd.map_w = roi.width; jd.Y = (res >= 1) ? std::max(nY, 0) : 0; jd.T = nT; jd.RP = 1; jd.down_expand = roi.down_expand;
        jd.frame = f; jd.box = bi; jd.hid = k; jd.map_off = F.map_offs[3 * bi + k]; jd.rp_off = (*C.rp_off)[f];
        jd.diag = std::sqrt(double(w * w + he * he));
        jd.yaw_off = q * YCAP; jd.top_off = tb;
        jd.line_off = (int)(b->line_base[f] - ln0) + ji * F.n_lines;
        jout[ji++] = jd;
      }
    }
  });
  if (overflow) { set_err("yaw sample list exceeds its capacity"); return CS_ERR_CAPACITY; }
  MARK(0, tq);   // per-frame jobs + sample lists
  // ---- slot / vanishing-point prefixes and the box table: one serial pass over the jobs
  size_t nb = 0;
  long long real_slots = 0;      // proposals (the slot space itself is padded per job)
  {
    long long so = 0, vo = 0;
    real_slots = 0;
    for (size_t j = 0; j < nj; j++) {
      cs::JobDesc& jd = S.h_jobs_in.p[j];
      jd.slot_off = so; jd.vp_off = (int)vo;
      S.h_slot_prefix.p[j] = so; S.h_vp_prefix.p[j] = (int)vo;
      // (a job's slots in multiples of 256: its compacted rows start there -- candidate_compact_kernel's capacity layout -- and a scorer
      // workgroup never straddles two jobs)
      real_slots += (long long)jd.Y * jd.T * 2;
      so += ((long long)jd.Y * jd.T * 2 + 255) & ~255LL; vo += (jd.Y + 63) & ~63;   // whole waves per job: a wave of the VP kernels then reads one job (scalar loads)
      if (jd.hid == 0 && jd.Y > 0 && jd.T > 0) { S.h_box_job0.p[nb] = (int)j; S.h_box_njobs.p[nb] = b->frames[jd.frame].n_heights[jd.box]; nb++; }
    }
    S.h_slot_prefix.p[nj] = so; S.h_vp_prefix.p[nj] = (int)vo;
    S.slot_total = so; S.vp_total = (int)vo;
    // launch order of the line setup: jobs with the largest ROI x segment count first (a counting sort on that proxy of
    // their sequential merge work), so that the kernel's tail is not a late-started long job
    {
      enum { NBK = 64 };
      long long wmax = 1;
      for (size_t j = 0; j < nj; j++) { const cs::JobDesc& jd = S.h_jobs_in.p[j]; wmax = std::max(wmax, (long long)(jd.g.er - jd.g.el) * (jd.g.eb - jd.g.et) * b->frames[jd.frame].n_lines); }
      int cnt[NBK + 1] = {0};
      auto bucket = [&](const cs::JobDesc& jd) { return NBK - 1 - (int)(((long long)(jd.g.er - jd.g.el) * (jd.g.eb - jd.g.et) * b->frames[jd.frame].n_lines) * (NBK - 1) / wmax); };
      for (size_t j = 0; j < nj; j++) cnt[bucket(S.h_jobs_in.p[j]) + 1]++;
      for (int q = 0; q < NBK; q++) cnt[q + 1] += cnt[q];
      for (size_t j = 0; j < nj; j++) S.h_ls_order.p[cnt[bucket(S.h_jobs_in.p[j])]++] = (int)j;
    }
    if (vo > 0x7fffffffLL) { set_err("too many yaw samples in one chunk"); return CS_ERR_CAPACITY; }
  }
  S.nb = nb;
  MARK(1, tq);   // prefixes + box table
  MARK(2, tq);   // box table
  C.tm->setup_host_ms += now_ms() - t0;
  C.tm->n_jobs += (long long)nj; C.tm->n_slots += real_slots;
  // ---- device buffers, H2D, kernels, D2H: all asynchronous on the detector's stream
  const long long slot_total = S.slot_total;
  PENS(S.ls_order, nj); PENS(S.jobs, nj); PENS(S.slot_prefix, nj + 1); PENS(S.vp_prefix, nj + 1); PENS(S.job_valid, nj); PENS(S.job_cbase, nj + 1);
  PENS(S.mid_x, n_lines + 1); PENS(S.mid_y, n_lines + 1); PENS(S.ang, n_lines + 1); PENS(S.yaw, n_yaw + 1); PENS(S.yaw_c, n_yaw + 1); PENS(S.yaw_s, n_yaw + 1);
  PENS(S.top_x, n_top + 1); PENS(S.vp, 6 * (size_t)S.vp_total + 6); PENS(S.bound, 6 * (size_t)S.vp_total + 6); PENS(S.bound3, nj * (size_t)cs::vp3_table_doubles_per_job());
  if (g_split_candidates) PENS(S.flag, slot_total + 1); else PENS(S.blk_info, 2 * (size_t)(slot_total >> 8) + 4);
  PENS(S.ls_crowded, 2 * (nj + 1) + 2);
  PENS(S.c_slot, slot_total + 1); PENS(S.c_flag, slot_total + 1); PENS(S.c_dist, slot_total + 1);
  PENS(S.c_angle, slot_total + 1); PENS(S.c_skew, slot_total + 1); PENS(S.box_job0, nb + 1); PENS(S.box_njobs, nb + 1); PENS(S.win_count, nb + 1);
  PENS(S.fallback, nb + 1); PENS(S.winners, nb * KMAX + 1); PENS(S.records, nb * KMAX + 1); PENS(S.h_records, nb * KMAX + 1);
  PENS(S.h_winners, nb * KMAX + 1); PENS(S.h_win_count, nb + 1); PENS(S.h_fallback, nb + 1); PENS(S.h_job_valid, nj); PENS(S.h_job_cbase, nj + 1); PENS(S.h_jobs_out, nj);
#define PH2D(dst, src, n) HIP_TRY(hipMemcpyAsync((dst).p, (src).p, sizeof(*(src).p) * (n), hipMemcpyHostToDevice, st))
  // the tables' device addresses: the slot's own buffers, or -- a call of a frame or two, where ten copies of a few hundred bytes cost ten
  // launch latencies (~150 us of a 0.46 ms call) -- pieces of ONE block that goes up in one copy
  int* p_ls_order = S.ls_order.p; cs::JobDesc* p_jobs = S.jobs.p; long long* p_slot_prefix = S.slot_prefix.p; int* p_vp_prefix = S.vp_prefix.p;
  double *p_yaw = S.yaw.p, *p_yaw_c = S.yaw_c.p, *p_yaw_s = S.yaw_s.p;
  int *p_top_x = S.top_x.p, *p_box_job0 = S.box_job0.p, *p_box_njobs = S.box_njobs.p;
  cs_cuboid* p_records = S.records.p; int *p_win_count = S.win_count.p, *p_fallback = S.fallback.p, *p_job_valid = S.job_valid.p; long long* p_job_cbase = S.job_cbase.p;
  S.merged_io = nj <= 256;
  if (S.merged_io) {
    // layout: [inputs | the job table (in and out) | outputs] -- one copy up ([0, end of the job table)), one copy back (from the job table on)
    size_t need = 0;
    auto room = [&](size_t bytes) { const size_t at = need; need += (bytes + 255) & ~(size_t)255; return at; };
    const size_t o_ls = room(sizeof(int) * nj), o_sp = room(sizeof(long long) * (nj + 1)), o_vp = room(sizeof(int) * (nj + 1));
    const size_t o_y = room(8 * n_yaw), o_yc = room(8 * n_yaw), o_ys = room(8 * n_yaw), o_tx = room(sizeof(int) * n_top), o_b0 = room(sizeof(int) * nb), o_bn = room(sizeof(int) * nb);
    const size_t o_jobs = room(sizeof(cs::JobDesc) * nj), in_end = need;
    S.o_jobs = o_jobs; S.o_rec = room(sizeof(cs_cuboid) * nb * KMAX); S.o_wc = room(sizeof(int) * nb); S.o_fb = room(sizeof(int) * nb); S.o_jv = room(sizeof(int) * nj);
    S.o_cb = room(sizeof(long long) * (nj + 1)); S.o_end = need;
    PENS(S.tab_arena, need + 256); PENS(S.h_tab_arena, need + 256);
    unsigned char *hb = S.h_tab_arena.p, *db = S.tab_arena.p;
    memcpy(hb + o_ls, S.h_ls_order.p, sizeof(int) * nj); memcpy(hb + o_jobs, S.h_jobs_in.p, sizeof(cs::JobDesc) * nj);
    memcpy(hb + o_sp, S.h_slot_prefix.p, sizeof(long long) * (nj + 1)); memcpy(hb + o_vp, S.h_vp_prefix.p, sizeof(int) * (nj + 1));
    if (n_yaw) { memcpy(hb + o_y, S.h_yaw.p, 8 * n_yaw); memcpy(hb + o_yc, S.h_yaw_c.p, 8 * n_yaw); memcpy(hb + o_ys, S.h_yaw_s.p, 8 * n_yaw); }
    if (n_top) memcpy(hb + o_tx, S.h_top_x.p, sizeof(int) * n_top);
    if (nb) { memcpy(hb + o_b0, S.h_box_job0.p, sizeof(int) * nb); memcpy(hb + o_bn, S.h_box_njobs.p, sizeof(int) * nb); }
    HIP_TRY(hipMemcpyAsync(db, hb, in_end, hipMemcpyHostToDevice, st));
    p_ls_order = reinterpret_cast<int*>(db + o_ls); p_jobs = reinterpret_cast<cs::JobDesc*>(db + o_jobs); p_slot_prefix = reinterpret_cast<long long*>(db + o_sp);
    p_vp_prefix = reinterpret_cast<int*>(db + o_vp); p_yaw = reinterpret_cast<double*>(db + o_y); p_yaw_c = reinterpret_cast<double*>(db + o_yc); p_yaw_s = reinterpret_cast<double*>(db + o_ys);
    p_top_x = reinterpret_cast<int*>(db + o_tx); p_box_job0 = reinterpret_cast<int*>(db + o_b0); p_box_njobs = reinterpret_cast<int*>(db + o_bn);
    p_records = reinterpret_cast<cs_cuboid*>(db + S.o_rec); p_win_count = reinterpret_cast<int*>(db + S.o_wc); p_fallback = reinterpret_cast<int*>(db + S.o_fb);
    p_job_valid = reinterpret_cast<int*>(db + S.o_jv); p_job_cbase = reinterpret_cast<long long*>(db + S.o_cb);
  } else {
    // a batch's tables: one kernel reads all ten out of the pinned staging pools (no copy-engine ring in the sweep: see multi_copy_kernel)
    cs::CopySegs cp{};
#define SEG(to_, from_, cnt_) do { if ((cnt_) > 0) { cp.s[cp.n].src = (from_).p; cp.s[cp.n].dst = (to_).p; cp.s[cp.n].bytes = sizeof(*(from_).p) * (unsigned long long)(cnt_); cp.n++; } } while (0)
    if (g_dma_tables) {
      PH2D(S.ls_order, S.h_ls_order, nj); PH2D(S.jobs, S.h_jobs_in, nj); PH2D(S.slot_prefix, S.h_slot_prefix, nj + 1); PH2D(S.vp_prefix, S.h_vp_prefix, nj + 1);
      if (n_yaw) { PH2D(S.yaw, S.h_yaw, n_yaw); PH2D(S.yaw_c, S.h_yaw_c, n_yaw); PH2D(S.yaw_s, S.h_yaw_s, n_yaw); }
      if (n_top) PH2D(S.top_x, S.h_top_x, n_top);
      if (nb) { PH2D(S.box_job0, S.h_box_job0, nb); PH2D(S.box_njobs, S.h_box_njobs, nb); }
    } else {
      SEG(S.ls_order, S.h_ls_order, nj); SEG(S.jobs, S.h_jobs_in, nj); SEG(S.slot_prefix, S.h_slot_prefix, nj + 1); SEG(S.vp_prefix, S.h_vp_prefix, nj + 1);
      SEG(S.yaw, S.h_yaw, n_yaw); SEG(S.yaw_c, S.h_yaw_c, n_yaw); SEG(S.yaw_s, S.h_yaw_s, n_yaw); SEG(S.top_x, S.h_top_x, n_top);
      SEG(S.box_job0, S.h_box_job0, nb); SEG(S.box_njobs, S.h_box_njobs, nb);
      cs::launch_multi_copy(cp, st);
    }
  }
  if (g_split_candidates) HIP_TRY(hipMemsetAsync(p_job_valid, 0, sizeof(int) * nj, st));      // (candidate_kernel counts with atomics; the fused kernel writes every job's count)
  cs::DetectDeviceView& v = S.view;
  v = cs::DetectDeviceView{};
  v.jobs = p_jobs; v.n_jobs = (int)nj; v.slot_prefix = p_slot_prefix; v.vp_prefix = p_vp_prefix; v.maps = b->d_maps.p;
  v.mid_x = S.mid_x.p; v.mid_y = S.mid_y.p; v.line_angle = S.ang.p; v.yaw = p_yaw; v.yaw_cos = p_yaw_c; v.yaw_sin = p_yaw_s; v.top_x = p_top_x;
  v.rp = b->d_rp.p; v.invK = b->d_invK.p; v.vp = S.vp.p; v.bound = S.bound.p; v.bound3 = S.bound3.p; v.flag = S.flag.p; v.job_valid = p_job_valid;
  v.job_cbase = p_job_cbase; v.c_slot = S.c_slot.p; v.c_flag = S.c_flag.p; v.c_dist = S.c_dist.p; v.c_angle = S.c_angle.p; v.c_skew = S.c_skew.p;
  // The corner construction needs the vanishing points but not the segments: it runs on the second stream beside line
  // setup + VP support (a latency-bound and an ALU-bound kernel), and the scorer waits for both.
  hipStream_t stB = d->stream2;
  HIP_TRY(hipEventRecord(S.ev[7], st));                      // inputs resident, job_valid zeroed
  HIP_TRY(hipStreamWaitEvent(stB, S.ev[7], 0));
  HIP_TRY(hipEventRecord(S.ev[8], stB));
  if (g_split_candidates) {
    cs::launch_vp_points(v, S.vp_total, stB);
    cs::launch_candidates(v, C.sp, slot_total, stB);
    HIP_TRY(hipEventRecord(S.ev[9], stB));
    cs::launch_scan_compact(v, stB);
  } else {
    // vanishing points, corner construction and ordered compaction of a job in one workgroup (candidate_compact_kernel); the compacted rows
    // of job j start at slot_prefix[j]: no scan over all jobs between this kernel and the scorer
    v.blk_info = S.blk_info.p;
    HIP_TRY(hipMemsetAsync(S.blk_info.p, 0, sizeof(int), stB));
    cs::launch_candidate_compact(v, C.sp, stB);
    HIP_TRY(hipEventRecord(S.ev[9], stB));
  }
  HIP_TRY(hipEventRecord(S.ev[10], stB));
  HIP_TRY(hipEventRecord(S.ev[0], st));
  static const bool ls_unlisted = getenv("CS_DETECT_LS_UNLISTED") != nullptr;      // (the former form: a workgroup of the crowded instance per job -- A / B)
  if (!ls_unlisted && d->stream3)
    cs::launch_line_setup_listed(p_jobs, (int)nj, b->d_frame_lines.p, b->d_frame_line_ptr.p, S.mid_x.p, S.mid_y.p, S.ang.p, P.pre_merge_dist_thre, P.pre_merge_angle_thre, P.edge_length_threshold, st, p_ls_order,
                                 d->stream3, S.ev[0], S.ev[12], S.ls_crowded.p, d->stream4, S.ev[13]);
  else
    cs::launch_line_setup(p_jobs, (int)nj, b->d_frame_lines.p, b->d_frame_line_ptr.p, S.mid_x.p, S.mid_y.p, S.ang.p, P.pre_merge_dist_thre, P.pre_merge_angle_thre, P.edge_length_threshold, st, p_ls_order,
                          d->stream3, S.ev[0], S.ev[12]);
  HIP_TRY(hipEventRecord(S.ev[1], st));
  cs::launch_vp_support_only(v, C.sp, S.vp_total, st, 1);      // (this path's jobs have one roll/pitch sample: jd.RP = 1 above)
  HIP_TRY(hipEventRecord(S.ev[2], st));
  HIP_TRY(hipStreamWaitEvent(st, S.ev[10], 0));
  HIP_TRY(hipEventRecord(S.ev[4], st));
  cs::launch_score(v, C.sp, slot_total, slot_total, st);
  HIP_TRY(hipEventRecord(S.ev[5], st));
  cs::RankView rv{};
  rv.box_job0 = p_box_job0; rv.box_njobs = p_box_njobs; rv.n_boxes = (int)nb; rv.winners = S.winners.p; rv.win_count = p_win_count; rv.fallback = p_fallback;
  cs::RankParams rkp{P.weight_vp_angle, P.weight_skew_error, P.nominal_skew_ratio, P.max_cut_skew, KMAX, C.sp.short_sq_bound};
  cs::launch_rank(v, rv, rkp, st, 0, false);          // (the host reads no winner of the device: record_kernel rebuilds their corners)
  cs::launch_records(v, rv, KMAX, p_records, st, nullptr, rkp.short_sq_bound);     // the records of the boxes the device ranked: only they come back
  HIP_TRY(hipEventRecord(S.ev[6], st));
  HIP_TRY(hipGetLastError());
  if (S.merged_io) {
    HIP_TRY(hipMemcpyAsync(S.h_tab_arena.p + S.o_jobs, S.tab_arena.p + S.o_jobs, S.o_end - S.o_jobs, hipMemcpyDeviceToHost, st));   // unpacked in pipe_finish
  } else {
    if (g_dma_tables) {
      if (nb) {
        HIP_TRY(hipMemcpyAsync(S.h_records.p, S.records.p, sizeof(cs_cuboid) * nb * KMAX, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(S.h_win_count.p, S.win_count.p, sizeof(int) * nb, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(S.h_fallback.p, S.fallback.p, sizeof(int) * nb, hipMemcpyDeviceToHost, st));
      }
      HIP_TRY(hipMemcpyAsync(S.h_job_valid.p, S.job_valid.p, sizeof(int) * nj, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(S.h_job_cbase.p, S.job_cbase.p, sizeof(long long) * (nj + 1), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(S.h_jobs_out.p, p_jobs, sizeof(cs::JobDesc) * nj, hipMemcpyDeviceToHost, st));
    } else {   // the results: written to the pinned host pools by one kernel
      cs::CopySegs cp{};
      SEG(S.h_records, S.records, nb * KMAX); SEG(S.h_win_count, S.win_count, nb); SEG(S.h_fallback, S.fallback, nb);
      SEG(S.h_job_valid, S.job_valid, nj); SEG(S.h_job_cbase, S.job_cbase, nj + 1);      // (the job table itself is not read back: pipe_finish works from the host's own copy)
      cs::launch_multi_copy(cp, st);
    }
#undef SEG
  }
  HIP_TRY(hipEventRecord(S.done, st));
  S.in_flight = true;
  MARK(3, tq);   // allocations + enqueue of copies and kernels
  return CS_OK;
}

int pipe_finish(PipeCtx& C, PipeSlot& S, const std::vector<std::vector<CamCache>>& cam_rp) {
  cs_detector* d = C.d; cs_batch* b = C.b;
  const cs_detect_params& P = d->prm;
  const int KMAX = P.max_cuboid_num, MB = b->max_boxes;
  cs_detect_timing& tm = *C.tm;
  double tw = now_ms(), tq = tw;
  HIP_TRY(hipEventSynchronize(S.done));
  tm.d2h_ms += now_ms() - tw;   // time the host actually waited for the GPU
  S.in_flight = false;
  const size_t nj = S.nj, nb = S.nb;
  if (nj == 0) return CS_OK;
  if (S.merged_io) {     // a small call's results came back in one block (pipe_submit): to the places the rest of this function reads
    const unsigned char* hb = S.h_tab_arena.p;
    memcpy(S.h_jobs_out.p, hb + S.o_jobs, sizeof(cs::JobDesc) * nj);
    if (nb) { memcpy(S.h_records.p, hb + S.o_rec, sizeof(cs_cuboid) * nb * KMAX); memcpy(S.h_win_count.p, hb + S.o_wc, sizeof(int) * nb); memcpy(S.h_fallback.p, hb + S.o_fb, sizeof(int) * nb); }
    memcpy(S.h_job_valid.p, hb + S.o_jv, sizeof(int) * nj); memcpy(S.h_job_cbase.p, hb + S.o_cb, sizeof(long long) * (nj + 1));
  }
  double t0 = now_ms();
  {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, S.ev[0], S.ev[1])); tm.line_setup_ms += ms;
    HIP_TRY(hipEventElapsedTime(&ms, S.ev[1], S.ev[2])); tm.vp_kernel_ms += ms;
    HIP_TRY(hipEventElapsedTime(&ms, S.ev[8], S.ev[9])); tm.cand_kernel_ms += ms;    // second stream: vanishing points + corners
    HIP_TRY(hipEventElapsedTime(&ms, S.ev[9], S.ev[10])); tm.compact_ms += ms;
    HIP_TRY(hipEventElapsedTime(&ms, S.ev[4], S.ev[5])); tm.score_kernel_ms += ms;
    HIP_TRY(hipEventElapsedTime(&ms, S.ev[5], S.ev[6])); tm.rank_kernel_ms += ms;
    tm.cand_kernel_launches += 1;
    long long n_valid = 0;      // (the capacity layout has no running total: job_cbase[j] = slot_prefix[j])
    for (size_t j = 0; j < nj; j++) n_valid += S.h_job_valid.p[j];
    tm.n_valid += n_valid;
    // algorithmic bytes of the corner construction: the vanishing points it writes (48 B each) + what leaves it per slot -- a 4-byte flag per
    // slot in the four-kernel form, 12 bytes (slot id, flag) per VALID proposal from candidate_compact_kernel
    tm.cand_kernel_bytes += g_split_candidates ? 48LL * S.vp_total + 4LL * S.slot_total : 48LL * S.vp_total + 12LL * n_valid;
    long long sbytes = 96LL * S.vp_total + (28LL + 8LL + 4LL) * n_valid;
    for (size_t j = 0; j < nj; j++) if (S.h_jobs_in.p[j].Y > 0 && S.h_jobs_in.p[j].T > 0) sbytes += 4LL * S.h_jobs_in.p[j].map_w * (S.h_jobs_in.p[j].g.eb - S.h_jobs_in.p[j].g.et);
    tm.score_kernel_bytes += sbytes;
  }
  const cs::JobDesc* jobs = S.h_jobs_in.p;
  hipStream_t st2 = d->stream_hi;
  MARK(4, tq);   // wait for the GPU + timing bookkeeping
  // ---- boxes the kernel flagged (a tie at a cut or at the top): exact std::partial_sort ranking on the host.  Their
  // columns are fetched on the high-priority stream: the host waits for them while other sweeps (the next chunk's, or another
  // batch's of the same process) fill the device, and these few small kernels should not queue behind those.
  std::vector<long long> fb_src, fb_dst;
  std::vector<int> fb_cnt, fb_range_of_job(nj, -1);
  long long tot = 0;
  for (size_t q = 0; q < nb; q++)
    if (S.h_fallback.p[q])
      for (int h = 0; h < S.h_box_njobs.p[q]; h++) {
        int j = S.h_box_job0.p[q] + h;
        fb_range_of_job[j] = (int)fb_src.size();
        fb_src.push_back(S.h_job_cbase.p[j]); fb_cnt.push_back(S.h_job_valid.p[j]); fb_dst.push_back(tot);
        tot += S.h_job_valid.p[j];
      }
  int rc;
  if (!fb_src.empty()) {
    size_t nr = fb_src.size();
    PENS(S.fb_src, nr); PENS(S.fb_dst, nr); PENS(S.fb_cnt, nr); PENS(S.h_fb_src, nr); PENS(S.h_fb_dst, nr); PENS(S.h_fb_cnt, nr);
    PENS(S.fb_dist, tot + 1); PENS(S.fb_angle, tot + 1); PENS(S.fb_skew, tot + 1); PENS(S.fb_flag, tot + 1); PENS(S.fb_slot, tot + 1);
    PENS(S.h_fb_dist, tot + 1); PENS(S.h_fb_angle, tot + 1); PENS(S.h_fb_skew, tot + 1); PENS(S.h_fb_flag, tot + 1); PENS(S.h_fb_slot, tot + 1);
    std::copy(fb_src.begin(), fb_src.end(), S.h_fb_src.p); std::copy(fb_dst.begin(), fb_dst.end(), S.h_fb_dst.p); std::copy(fb_cnt.begin(), fb_cnt.end(), S.h_fb_cnt.p);
    if (g_dma_tables) {
      HIP_TRY(hipMemcpyAsync(S.fb_src.p, S.h_fb_src.p, 8 * nr, hipMemcpyHostToDevice, st2));
      HIP_TRY(hipMemcpyAsync(S.fb_dst.p, S.h_fb_dst.p, 8 * nr, hipMemcpyHostToDevice, st2));
      HIP_TRY(hipMemcpyAsync(S.fb_cnt.p, S.h_fb_cnt.p, 4 * nr, hipMemcpyHostToDevice, st2));
      cs::launch_gather_ranges(S.view, S.fb_src.p, S.fb_cnt.p, S.fb_dst.p, (int)nr, S.fb_dist.p, S.fb_angle.p, S.fb_skew.p, S.fb_flag.p, S.fb_slot.p, st2);
      if (tot) {
        HIP_TRY(hipMemcpyAsync(S.h_fb_dist.p, S.fb_dist.p, 8 * (size_t)tot, hipMemcpyDeviceToHost, st2));
        HIP_TRY(hipMemcpyAsync(S.h_fb_angle.p, S.fb_angle.p, 8 * (size_t)tot, hipMemcpyDeviceToHost, st2));
        HIP_TRY(hipMemcpyAsync(S.h_fb_skew.p, S.fb_skew.p, 8 * (size_t)tot, hipMemcpyDeviceToHost, st2));
        HIP_TRY(hipMemcpyAsync(S.h_fb_flag.p, S.fb_flag.p, 4 * (size_t)tot, hipMemcpyDeviceToHost, st2));
        HIP_TRY(hipMemcpyAsync(S.h_fb_slot.p, S.fb_slot.p, 8 * (size_t)tot, hipMemcpyDeviceToHost, st2));
      }
    } else {
      // the range lists are read from, and the columns written to, the pinned host pools by the gather kernel itself (coalesced rows; no
      // copy engine between the device and a host that is waiting for a few kilobytes)
      cs::launch_gather_ranges(S.view, S.h_fb_src.p, S.h_fb_cnt.p, S.h_fb_dst.p, (int)nr, S.h_fb_dist.p, S.h_fb_angle.p, S.h_fb_skew.p, S.h_fb_flag.p, S.h_fb_slot.p, st2);
      HIP_TRY(hipGetLastError());
    }
  }
  MARK(5, tq);   // tie lists + gather enqueue
  // ---- records of the winners.  The boxes the device ranked are written while the tie boxes' columns travel.
  auto write_box = [&](size_t q, const cs::RankWinner* wl, int nw) {
    const int j0 = S.h_box_job0.p[q], nh = S.h_box_njobs.p[q];
    const int f = jobs[j0].frame, bi = jobs[j0].box;
    const FrameIn& F = b->frames[f];
    const double* bb = &F.boxes[5 * bi];
    for (int r = 0; r < nw; r++) {
      const cs::RankWinner& w = wl[r];
      int h = 0;
      while (h + 1 < nh && w.slot >= jobs[j0 + h + 1].slot_off) h++;
      const cs::JobDesc& jd = jobs[j0 + h];
      long long local = w.slot - jd.slot_off;
      long long rest = local >> 1;
      int t = (int)(rest % jd.T), y = (int)(rest / jd.T);
      const cs::RpPose& pose = cam_rp[f][0].pose;
      double r9[9] = {(double)((local & 1) + 1), (double)(w.flag & cs::CAND_VP_MASK), S.h_yaw.p[jd.yaw_off + y], (double)t, w.dist_err, w.angle_err,
                      (double)jd.down_expand, pose.roll, pose.pitch};
      cs_cuboid& o = C.out[((size_t)f * MB + bi) * KMAX + r];
      finish_cuboid(F, pose, r9, w.corners, (*C.cam_raw)[f].euler, false, w.normalized_error, o);
      o.rect_detect_2d[0] = (int)bb[0]; o.rect_detect_2d[1] = (int)bb[1]; o.rect_detect_2d[2] = (int)bb[2]; o.rect_detect_2d[3] = (int)bb[3];
    }
    C.out_counts[(size_t)f * MB + bi] = nw;
  };
  const double* fb_dist = S.h_fb_dist.p; const double* fb_angle = S.h_fb_angle.p; const double* fb_skew = S.h_fb_skew.p;
  const int* fb_flag = S.h_fb_flag.p; const long long* fb_slot = S.h_fb_slot.p;
  std::vector<std::vector<cs::RankWinner>> fb_winners(nb);
  auto rank_on_host = [&](size_t q) {
    const int j0 = S.h_box_job0.p[q], nh = S.h_box_njobs.p[q];
    struct HP { int h, cand; double score, skew; };
    std::vector<HP> props;
    for (int h = 0; h < nh; h++) {
      int j = j0 + h, V = S.h_job_valid.p[j];
      long long p0 = fb_dst[fb_range_of_job[j]];
      std::vector<int> keep;
      std::vector<double> score;
      fuse_scores(fb_dist + p0, fb_angle + p0, V, P.weight_vp_angle, keep, score);
      for (size_t z = 0; z < keep.size(); z++) {
        if (fb_flag[p0 + keep[z]] & cs::CAND_NEG_SCALE) continue;
        props.push_back(HP{h, keep[z], score[z], fb_skew[p0 + keep[z]]});
      }
    }
    int n = (int)props.size(), kk = std::min(KMAX, n);
    std::vector<double> comb(n);
    for (int i = 0; i < n; i++) {
      double skew_error = P.weight_skew_error * std::max(props[i].skew - P.nominal_skew_ratio, 0.0);
      if (props[i].skew > P.max_cut_skew) skew_error = 100;
      comb[i] = props[i].score + P.weight_skew_error * skew_error;
    }
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::partial_sort(idx.begin(), idx.begin() + kk, idx.end(), [&comb](int a, int c) { return comb[a] < comb[c]; });
    for (int r = 0; r < kk; r++) {
      const HP& hp = props[idx[r]];
      long long p0 = fb_dst[fb_range_of_job[j0 + hp.h]];
      cs::RankWinner w{};
      w.slot = fb_slot[p0 + hp.cand]; w.normalized_error = hp.score; w.dist_err = fb_dist[p0 + hp.cand]; w.angle_err = fb_angle[p0 + hp.cand];
      w.flag = fb_flag[p0 + hp.cand] & cs::CAND_VP_MASK;
      fb_winners[q].push_back(w);
    }
  };
  std::vector<int> fbq;
  for (size_t q = 0; q < nb; q++) if (S.h_fallback.p[q]) fbq.push_back((int)q);
  tm.n_fallback_boxes += (int)fbq.size();
  auto records = [&](int qi) {      // a device-ranked box: its records are complete but for the caller's rectangle
    const size_t q = (size_t)qi;
    if (S.h_fallback.p[q]) return;
    const int j0 = S.h_box_job0.p[q];
    const int f = jobs[j0].frame, bi = jobs[j0].box, nw = S.h_win_count.p[q];
    const double* bb = &b->frames[f].boxes[5 * bi];
    for (int r = 0; r < nw; r++) {
      cs_cuboid& o = C.out[((size_t)f * MB + bi) * KMAX + r];
      o = S.h_records.p[q * KMAX + r];
      o.rect_detect_2d[0] = (int)bb[0]; o.rect_detect_2d[1] = (int)bb[1]; o.rect_detect_2d[2] = (int)bb[2]; o.rect_detect_2d[3] = (int)bb[3];
    }
    C.out_counts[(size_t)f * MB + bi] = nw;
  };
  static const bool tie_separate = getenv("CS_TIE_SEPARATE_PASS") != nullptr;   // diagnostics: always the two-pass order
  if (fbq.size() <= 64 && !tie_separate) {
    // a handful of tie boxes (the usual case): their columns are a few kilobytes and already here; their exact ranking
    // (tens of microseconds each) rides in the same parallel pass as the records of the other boxes, first in the queue
    if (!fb_src.empty()) HIP_TRY(hipStreamSynchronize(st2));
    MARK(7, tq);   // wait for the tie columns
    const int nfb = (int)fbq.size();
    constexpr int RCH = 256;           // records are a 400-byte copy each: hand them out in chunks
    const int nch = ((int)nb + RCH - 1) / RCH;
    d->pool->run(nfb + nch, [&](int z) {
      if (z < nfb) { rank_on_host((size_t)fbq[z]); return; }
      for (int q = (z - nfb) * RCH, q1 = std::min((int)nb, q + RCH); q < q1; q++) records(q);
    });
    MARK(6, tq);   // records of the device-ranked boxes (+ exact ranking of the tie boxes)
  } else {
    d->pool->run((int)nb, records);
    MARK(6, tq);   // records of the device-ranked boxes, written while the tie boxes' columns travel
    if (!fb_src.empty()) HIP_TRY(hipStreamSynchronize(st2));
    MARK(7, tq);   // wait for the tie columns
    d->pool->run((int)fbq.size(), [&](int z) { rank_on_host((size_t)fbq[z]); });
  }
  MARK(8, tq);   // exact ranking of the tie boxes
  {
    std::vector<long long> ws;
    for (int q : fbq) for (auto& w : fb_winners[q]) ws.push_back(w.slot);
    if (!ws.empty()) {
      PENS(S.win_slots, ws.size()); PENS(S.win_corners, 16 * ws.size()); PENS(S.h_win_slots, ws.size()); PENS(S.h_win_corners, 16 * ws.size());
      std::copy(ws.begin(), ws.end(), S.h_win_slots.p);
      if (g_dma_tables) {
        HIP_TRY(hipMemcpyAsync(S.win_slots.p, S.h_win_slots.p, 8 * ws.size(), hipMemcpyHostToDevice, st2));
        cs::launch_gather_corners(S.view, C.sp, S.win_slots.p, (int)ws.size(), S.win_corners.p, st2);
        HIP_TRY(hipMemcpyAsync(S.h_win_corners.p, S.win_corners.p, 8 * 16 * ws.size(), hipMemcpyDeviceToHost, st2));
      } else {
        cs::launch_gather_corners(S.view, C.sp, S.h_win_slots.p, (int)ws.size(), S.h_win_corners.p, st2);     // (pinned host memory on both sides)
      }
      HIP_TRY(hipStreamSynchronize(st2));
      const double* hc = S.h_win_corners.p;
      size_t z = 0;
      for (int q : fbq) for (auto& w : fb_winners[q]) { std::memcpy(w.corners, &hc[16 * z], 128); z++; }
    }
  }
  MARK(9, tq);   // corners of the tie winners
  if (fbq.size() <= 24) { for (size_t z = 0; z < fbq.size(); z++) { const size_t q = (size_t)fbq[z]; write_box(q, fb_winners[q].data(), (int)fb_winners[q].size()); } }
  else d->pool->run((int)fbq.size(), [&](int z) { const size_t q = (size_t)fbq[z]; write_box(q, fb_winners[q].data(), (int)fb_winners[q].size()); });
  MARK(10, tq);  // records of the tie boxes
  tm.finalize_ms += now_ms() - t0;
  return CS_OK;
}
#undef PENS
#undef PH2D

// ================================================================== lean roll/pitch path =========
// whether_sample_cam_roll_pitch = 1 (the reference class's default; main_obj.cpp:623 uses it from the second frame on).  The boxes of a
// frame depend on each other through cam_pose.camera_yaw (box_proposal_detail.cpp:180 reads what :374 / :734 left), so box r of
// every frame forms round r.  The carried value is the camera yaw of one of the frame's RP roll/pitch samples (or the raw pose's before
// the first box): the host lays down all 1 + RP yaw lists of every frame up front -- glibc's cos / sin, as everywhere -- and
// rp_carry_kernel picks the list of the next round on the device.  All rounds are queued back to back; nothing comes back before the
// last one.  Per round the kernels are those of the production path on the round's jobs (slot / vanishing-point numbering relative to
// the round; slots laid out for the longest list a box can get).  A box whose ranking needs the exact host order (a tie that can
// reach the output or the carried proposal) invalidates what the device assumed for the rest of its frame: those frames -- under one
// in a hundred boxes -- are redone through the round-by-round path below, as a small batch that shares this one's distance maps.
int rp_launch(PipeCtx& C, PipeSlot& S, const std::vector<std::vector<CamCache>>& cam_rp) {
  cs_detector* d = C.d; cs_batch* b = C.b;
  const cs_detect_params& P = d->prm;
  hipStream_t st = d->stream;
  const int KMAX = P.max_cuboid_num, NF = b->n_frames, MB = b->max_boxes;
  double t0 = now_ms();
  if (!S.done) { HIP_TRY(hipEventCreate(&S.done)); for (auto& e : S.ev) HIP_TRY(hipEventCreate(&e)); }
  S.rp_mode = true;
  S.f0 = 0; S.f1 = NF;
  int rc;
#define PENS(buf, n) do { rc = (buf).ensure(n); if (rc) return rc; } while (0)
  // ---- yaw lists: NT = 1 + max RP per frame, YCAP samples each
  int max_rp = 1;
  for (int f = 0; f < NF; f++) max_rp = std::max(max_rp, (int)cam_rp[f].size());
  const int NT = 1 + max_rp;
  const int YCAP = (int)(2.0 * P.yaw_range_deg / std::max(1e-9, P.yaw_step_deg)) + 3;
  const size_t n_yaw = (size_t)NF * NT * YCAP;
  if (n_yaw > 0x7fffffffULL) { set_err("too many yaw samples"); return CS_ERR_CAPACITY; }
  PENS(S.h_yaw, n_yaw + 1); PENS(S.h_yaw_c, n_yaw + 1); PENS(S.h_yaw_s, n_yaw + 1); PENS(S.h_rp_tab_count, (size_t)NF * NT + 1); PENS(S.h_rp_raw_euler, 3 * (size_t)NF + 1);
  std::atomic<int> overflow{0};
  std::vector<int> ycap_f(NF, 0);     // longest list of the frame: its boxes' slots are laid out for it
  d->pool->run(NF * NT, [&](int z) {
    const int f = z / NT, i = z % NT;
    const int nrp = (int)cam_rp[f].size();
    int* cnt = S.h_rp_tab_count.p + (size_t)f * NT + i;
    *cnt = 0;
    if (i > nrp) return;
    const double cam_yaw = i == 0 ? (*C.cam_raw)[f].cam_yaw : cam_rp[f][i - 1].cam_yaw;
    const double yaw_init = cam_yaw - 90.0 / 180.0 * CS_PI;
    std::vector<double> yl;
    linespace<double>(yaw_init - P.yaw_range_deg / 180.0 * CS_PI, yaw_init + P.yaw_range_deg / 180.0 * CS_PI, P.yaw_step_deg / 180.0 * CS_PI, yl);
    if ((int)yl.size() > YCAP) { overflow = 1; return; }
    double* yw = S.h_yaw.p + ((size_t)f * NT + i) * YCAP; double* yc = S.h_yaw_c.p + ((size_t)f * NT + i) * YCAP; double* ys = S.h_yaw_s.p + ((size_t)f * NT + i) * YCAP;
    for (size_t y = 0; y < yl.size(); y++) { yw[y] = yl[y]; yc[y] = h_cos(yl[y]); ys[y] = h_sin(yl[y]); }
    *cnt = (int)yl.size();
  });
  if (overflow) { set_err("yaw sample list exceeds its capacity"); return CS_ERR_CAPACITY; }
  for (int f = 0; f < NF; f++) {
    for (int i = 0; i < NT; i++) ycap_f[f] = std::max(ycap_f[f], S.h_rp_tab_count.p[(size_t)f * NT + i]);
    for (int e = 0; e < 3; e++) S.h_rp_raw_euler.p[3 * f + e] = (*C.cam_raw)[f].euler[e];
  }
  // ---- jobs of all rounds: (round, frame, height sample); per round its own slot / vanishing-point numbering
  const int tp0 = 0;
  size_t nj_tot = 0, nb_tot = 0;
  for (int f = 0; f < NF; f++) for (int bi = 0; bi < b->frames[f].n_boxes; bi++) nj_tot += b->frames[f].n_heights[bi];
  const size_t n_top = (size_t)b->top_base[b->box_base[NF]];
  PENS(S.h_jobs_in, nj_tot + 1); PENS(S.h_slot_prefix, nj_tot + MB + 1); PENS(S.h_vp_prefix, nj_tot + MB + 1); PENS(S.h_top_x, n_top + 1);
  PENS(S.h_box_job0, nj_tot + 1); PENS(S.h_box_njobs, nj_tot + 1); PENS(S.h_ls_order, nj_tot + 1); PENS(S.h_rp_maps, 3 * (size_t)MB * NF + 1);
  S.rp_rounds.assign(MB, PipeSlot::RpRound{});
  S.rp_box_frame.clear(); S.rp_box_index.clear();
  size_t ji = 0;
  long long n_lines = 0, slot_cap_max = 0, slots_all = 0, max_job_slots = 0;
  int vp_cap_max = 0;
  size_t nj_round_max = 0;
  for (int r = 0; r < MB; r++) {
    PipeSlot::RpRound& Rr = S.rp_rounds[r];
    Rr.j0 = ji; Rr.b0 = nb_tot;
    long long so = 0, vo = 0;
    int* box_of = S.h_rp_maps.p + (size_t)(3 * r) * NF; int* job0_of = box_of + NF; int* njobs_of = job0_of + NF;
    for (int f = 0; f < NF; f++) {
      box_of[f] = -1; job0_of[f] = -1; njobs_of[f] = 0;
      const FrameIn& F = b->frames[f];
      if (r >= F.n_boxes) continue;
      const int bi = r;
      const double* bb = &F.boxes[5 * bi];
      const int left = bb[0], top = bb[1], w = bb[2], h = bb[3], right = left + bb[2];
      const int res = (int)std::round(std::min(20, w / 10));
      if (res < 1) continue;              // :215: the box is skipped and leaves the carried yaw alone
      const int tb = b->top_base[b->box_base[f] + bi] - tp0, tcap = b->top_base[b->box_base[f] + bi + 1] - tp0 - tb;
      int nT = 0;
      int* tx = S.h_top_x.p + tb;
      for (int x = left + 5; x <= right - 5 && nT < tcap; x += res) tx[nT++] = x;
      const int nrp = (int)cam_rp[f].size();
      box_of[f] = (int)(nb_tot - Rr.b0); job0_of[f] = (int)(ji - Rr.j0); njobs_of[f] = F.n_heights[bi];
      S.h_box_job0.p[nb_tot] = (int)(ji - Rr.j0); S.h_box_njobs.p[nb_tot] = F.n_heights[bi];
      S.rp_box_frame.push_back(f); S.rp_box_index.push_back(bi);
      nb_tot++;
      for (int k = 0; k < F.n_heights[bi]; k++) {
        const cs_roi& roi = F.rois[3 * bi + k];
        cs::JobDesc jd;
        std::memset(&jd, 0, sizeof(jd));
        const int he = h + roi.down_expand;
        jd.g.left = left; jd.g.top = top; jd.g.right = right; jd.g.down = top + he;
        jd.g.el = roi.left; jd.g.et = roi.top; jd.g.er = roi.left + roi.width; jd.g.eb = roi.top + roi.height;
        jd.map_w = roi.width; jd.T = nT; jd.RP = nrp; jd.down_expand = roi.down_expand;
        jd.Y = S.h_rp_tab_count.p[(size_t)f * NT];            // list 0 (the raw pose) until rp_carry_kernel says otherwise
        jd.yaw_off = (int)(((size_t)f * NT) * YCAP);
        jd.frame = f; jd.box = bi; jd.hid = k; jd.map_off = F.map_offs[3 * bi + k]; jd.rp_off = (*C.rp_off)[f];
        jd.diag = std::sqrt(double(w * w + he * he));
        jd.top_off = tb;
        jd.line_off = (int)n_lines; n_lines += F.n_lines;
        jd.slot_off = so; jd.vp_off = (int)vo;
        S.h_slot_prefix.p[ji + r] = so; S.h_vp_prefix.p[ji + r] = (int)vo;
        so += (long long)nrp * ycap_f[f] * nT * 2;
        max_job_slots = std::max(max_job_slots, (long long)nrp * ycap_f[f] * nT * 2);
        vo += ((long long)nrp * ycap_f[f] + 63) & ~63LL;
        S.h_jobs_in.p[ji++] = jd;
      }
    }
    Rr.nj = ji - Rr.j0; Rr.nb = nb_tot - Rr.b0; Rr.slot_cap = so; Rr.vp_cap = (int)vo;
    S.h_slot_prefix.p[ji + r] = so; S.h_vp_prefix.p[ji + r] = (int)vo;
    if (vo > 0x7fffffffLL || so > 0x7fffffff00LL) { set_err("too many samples in one round"); return CS_ERR_CAPACITY; }
    slot_cap_max = std::max(slot_cap_max, so); vp_cap_max = std::max(vp_cap_max, (int)vo); nj_round_max = std::max(nj_round_max, Rr.nj);
    slots_all += so;
  }
  const size_t nj = ji, nb = nb_tot;
  S.nj = nj; S.nb = nb; S.slot_total = slots_all; S.vp_total = vp_cap_max;
  if (n_lines > 0x7fffffffLL) { set_err("chunk too large"); return CS_ERR_CAPACITY; }
  for (size_t j = 0; j < nj; j++) S.h_ls_order.p[j] = (int)j;
  C.tm->setup_host_ms += now_ms() - t0;
  C.tm->n_jobs += (long long)nj;
  if (nj == 0) { S.in_flight = true; HIP_TRY(hipEventRecord(S.done, st)); return CS_OK; }
  // ---- device buffers; the per-round arrays are sized for the largest round and reused from round to round (one stream: in order)
  PENS(S.ls_order, nj); PENS(S.jobs, nj); PENS(S.slot_prefix, nj + MB + 1); PENS(S.vp_prefix, nj + MB + 1); PENS(S.job_valid, nj); PENS(S.job_cbase, nj + MB + 1);
  const int max_trips = (int)std::min<long long>((max_job_slots + 4095) / 4096, 1 << 20);      // compaction trips of the largest job (4096 slots each)
  PENS(S.rp_trip_cnt, (size_t)nj_round_max * (size_t)std::max(1, max_trips) + 1);
  PENS(S.mid_x, (size_t)n_lines + 1); PENS(S.mid_y, (size_t)n_lines + 1); PENS(S.ang, (size_t)n_lines + 1); PENS(S.yaw, n_yaw + 1); PENS(S.yaw_c, n_yaw + 1); PENS(S.yaw_s, n_yaw + 1);
  PENS(S.top_x, n_top + 1); PENS(S.vp, 6 * (size_t)vp_cap_max + 6); PENS(S.bound, 6 * (size_t)vp_cap_max + 6); PENS(S.bound3, nj_round_max * (size_t)cs::vp3_table_doubles_per_job() + 1);
  PENS(S.flag, slot_cap_max + 1); PENS(S.c_slot, slot_cap_max + 1); PENS(S.c_flag, slot_cap_max + 1); PENS(S.c_dist, slot_cap_max + 1); PENS(S.c_angle, slot_cap_max + 1); PENS(S.c_skew, slot_cap_max + 1);
  PENS(S.box_job0, nb + 1); PENS(S.box_njobs, nb + 1); PENS(S.win_count, nb + 1); PENS(S.fallback, nb + 1); PENS(S.rp_last_slot, nb + 1);
  PENS(S.winners, nb * KMAX + 1); PENS(S.records, nb * KMAX + 1); PENS(S.h_records, nb * KMAX + 1); PENS(S.h_win_count, nb + 1); PENS(S.h_fallback, nb + 1);
  PENS(S.h_job_cbase, nj + MB + 1);
  S.rp_NT = NT; S.rp_YCAP = YCAP;
  S.rp_pool_cap = std::max<long long>(1 << 18, slots_all / 64);        // columns of the flagged boxes (about one box in a hundred)
  PENS(S.fb_dist, (size_t)S.rp_pool_cap + 1); PENS(S.fb_angle, (size_t)S.rp_pool_cap + 1); PENS(S.fb_skew, (size_t)S.rp_pool_cap + 1); PENS(S.fb_flag, (size_t)S.rp_pool_cap + 1); PENS(S.fb_slot, (size_t)S.rp_pool_cap + 1);
  PENS(S.ls_crowded, 2 * (nj + 1) + 2);
  PENS(S.rp_box_base, nb + 1); PENS(S.rp_pool_used, 1); PENS(S.h_rp_box_base, nb + 1); PENS(S.h_rp_pool_used, 1); PENS(S.h_rp_last_slot, nb + 1); PENS(S.h_job_valid, nj + 1); PENS(S.h_jobs_out, nj + 1);
  HIP_TRY(hipMemsetAsync(S.rp_pool_used.p, 0, sizeof(unsigned long long), st));
  PENS(S.rp_cur_idx, (size_t)NF + 1); PENS(S.rp_tab_count, (size_t)NF * NT + 1); PENS(S.rp_maps, 3 * (size_t)MB * NF + 1); PENS(S.rp_raw_euler, 3 * (size_t)NF + 1);
#define PH2D(dst, src, n) HIP_TRY(hipMemcpyAsync((dst).p, (src).p, sizeof(*(src).p) * (n), hipMemcpyHostToDevice, st))
#define SEG(to_, from_, cnt_) do { if ((cnt_) > 0) { cp.s[cp.n].src = (from_).p; cp.s[cp.n].dst = (to_).p; cp.s[cp.n].bytes = sizeof(*(from_).p) * (unsigned long long)(cnt_); cp.n++; } } while (0)
  if (g_dma_tables) {
    PH2D(S.ls_order, S.h_ls_order, nj); PH2D(S.jobs, S.h_jobs_in, nj); PH2D(S.slot_prefix, S.h_slot_prefix, nj + MB); PH2D(S.vp_prefix, S.h_vp_prefix, nj + MB);
    PH2D(S.yaw, S.h_yaw, n_yaw); PH2D(S.yaw_c, S.h_yaw_c, n_yaw); PH2D(S.yaw_s, S.h_yaw_s, n_yaw);
    if (n_top) PH2D(S.top_x, S.h_top_x, n_top);
    PH2D(S.box_job0, S.h_box_job0, nb); PH2D(S.box_njobs, S.h_box_njobs, nb);
    PH2D(S.rp_tab_count, S.h_rp_tab_count, (size_t)NF * NT); PH2D(S.rp_maps, S.h_rp_maps, 3 * (size_t)MB * NF); PH2D(S.rp_raw_euler, S.h_rp_raw_euler, 3 * (size_t)NF);
  } else {     // (one kernel reads all thirteen tables out of the pinned pools: multi_copy_kernel)
    cs::CopySegs cp{};
    SEG(S.ls_order, S.h_ls_order, nj); SEG(S.jobs, S.h_jobs_in, nj); SEG(S.slot_prefix, S.h_slot_prefix, nj + MB); SEG(S.vp_prefix, S.h_vp_prefix, nj + MB);
    SEG(S.yaw, S.h_yaw, n_yaw); SEG(S.yaw_c, S.h_yaw_c, n_yaw); SEG(S.yaw_s, S.h_yaw_s, n_yaw); SEG(S.top_x, S.h_top_x, n_top);
    SEG(S.box_job0, S.h_box_job0, nb); SEG(S.box_njobs, S.h_box_njobs, nb);
    SEG(S.rp_tab_count, S.h_rp_tab_count, (size_t)NF * NT); SEG(S.rp_maps, S.h_rp_maps, 3 * (size_t)MB * NF); SEG(S.rp_raw_euler, S.h_rp_raw_euler, 3 * (size_t)NF);
    cs::launch_multi_copy(cp, st);
  }
  HIP_TRY(hipMemsetAsync(S.job_valid.p, 0, sizeof(int) * nj, st));
  HIP_TRY(hipMemsetAsync(S.rp_cur_idx.p, 0, sizeof(int) * NF, st));
  HIP_TRY(hipEventRecord(S.ev[0], st));
  // ---- line setup of every job of the batch at once (it depends on the box and the frame's segments only)
  static const bool ls_unlisted = getenv("CS_DETECT_LS_UNLISTED") != nullptr;
  if (!ls_unlisted && d->stream3)
    cs::launch_line_setup_listed(S.jobs.p, (int)nj, b->d_frame_lines.p, b->d_frame_line_ptr.p, S.mid_x.p, S.mid_y.p, S.ang.p, P.pre_merge_dist_thre, P.pre_merge_angle_thre, P.edge_length_threshold, st, S.ls_order.p,
                                 d->stream3, S.ev[0], S.ev[12], S.ls_crowded.p, d->stream4, S.ev[13]);
  else
    cs::launch_line_setup(S.jobs.p, (int)nj, b->d_frame_lines.p, b->d_frame_line_ptr.p, S.mid_x.p, S.mid_y.p, S.ang.p, P.pre_merge_dist_thre, P.pre_merge_angle_thre, P.edge_length_threshold, st, S.ls_order.p,
                          d->stream3, S.ev[0], S.ev[12]);
  HIP_TRY(hipEventRecord(S.ev[1], st));
  cs::RankParams rkp{P.weight_vp_angle, P.weight_skew_error, P.nominal_skew_ratio, P.max_cut_skew, KMAX, C.sp.short_sq_bound};
  while (S.rp_ev.size() < 5 * (size_t)MB) { hipEvent_t e = nullptr; HIP_TRY(hipEventCreate(&e)); S.rp_ev.push_back(e); }
  for (int r = 0; r < MB; r++) {
    const PipeSlot::RpRound& Rr = S.rp_rounds[r];
    if (Rr.nj == 0) continue;
    hipEvent_t* re = &S.rp_ev[5 * (size_t)r];
    cs::DetectDeviceView v{};
    v.jobs = S.jobs.p + Rr.j0; v.n_jobs = (int)Rr.nj; v.slot_prefix = S.slot_prefix.p + Rr.j0 + r; v.vp_prefix = S.vp_prefix.p + Rr.j0 + r; v.maps = b->d_maps.p;
    v.mid_x = S.mid_x.p; v.mid_y = S.mid_y.p; v.line_angle = S.ang.p; v.yaw = S.yaw.p; v.yaw_cos = S.yaw_c.p; v.yaw_sin = S.yaw_s.p; v.top_x = S.top_x.p;
    v.rp = b->d_rp.p; v.invK = b->d_invK.p; v.vp = S.vp.p; v.bound = S.bound.p; v.bound3 = S.bound3.p; v.flag = S.flag.p; v.job_valid = S.job_valid.p + Rr.j0;
    v.job_cbase = S.job_cbase.p + Rr.j0 + r; v.c_slot = S.c_slot.p; v.c_flag = S.c_flag.p; v.c_dist = S.c_dist.p; v.c_angle = S.c_angle.p; v.c_skew = S.c_skew.p;
    if (r > 0) {      // the carried yaw: this round's jobs get the list the previous round's result selects
      // (the previous round WITH jobs decides for the frames it held a box of; a frame without a box there keeps its list)
      int rq = r - 1;
      while (rq > 0 && S.rp_rounds[rq].nj == 0) rq--;
      const PipeSlot::RpRound& Rq = S.rp_rounds[rq];
      cs::RpCarryView c{};
      c.n_frames = NF; c.NT = NT; c.YCAP = YCAP;
      c.prev_box_of_frame = Rq.nj ? S.rp_maps.p + (size_t)(3 * rq) * NF : nullptr;
      c.prev_last_slot = S.rp_last_slot.p + Rq.b0; c.prev_jobs = S.jobs.p + Rq.j0; c.prev_box_job0 = S.box_job0.p + Rq.b0; c.prev_box_njobs = S.box_njobs.p + Rq.b0;
      c.job0_of_frame = S.rp_maps.p + (size_t)(3 * r + 1) * NF; c.njobs_of_frame = S.rp_maps.p + (size_t)(3 * r + 2) * NF;
      c.tab_count = S.rp_tab_count.p; c.cur_idx = S.rp_cur_idx.p;
      cs::launch_rp_carry(c, S.jobs.p + Rr.j0, st);
    }
    HIP_TRY(hipEventRecord(re[0], st));
    cs::launch_vp_points(v, Rr.vp_cap, st);
    cs::launch_candidates(v, C.sp, Rr.slot_cap, st);
    cs::launch_scan_compact_trips(v, S.rp_trip_cnt.p, max_trips, st);
    HIP_TRY(hipEventRecord(re[1], st));
    cs::launch_vp_support_only(v, C.sp, Rr.vp_cap, st);
    HIP_TRY(hipEventRecord(re[2], st));
    cs::launch_score(v, C.sp, Rr.slot_cap, Rr.slot_cap, st);
    HIP_TRY(hipEventRecord(re[3], st));
    cs::RankView rv{};
    rv.box_job0 = S.box_job0.p + Rr.b0; rv.box_njobs = S.box_njobs.p + Rr.b0; rv.n_boxes = (int)Rr.nb; rv.winners = S.winners.p + Rr.b0 * KMAX;
    rv.win_count = S.win_count.p + Rr.b0; rv.fallback = S.fallback.p + Rr.b0; rv.last_slot = S.rp_last_slot.p + Rr.b0;
    cs::launch_rank(v, rv, rkp, st, Rr.nb ? (long long)(Rr.slot_cap / (long long)Rr.nb) : 0, false);      // (average slots per box of the round: which instance ranks)
    cs::launch_records(v, rv, KMAX, S.records.p + Rr.b0 * KMAX, st, S.rp_raw_euler.p, rkp.short_sq_bound);
    {   // the flagged boxes' columns, before the next round reuses the arrays
      cs::RpSaveView sv{};
      sv.fallback = rv.fallback; sv.box_job0 = rv.box_job0; sv.box_njobs = rv.box_njobs; sv.n_boxes = rv.n_boxes; sv.pool_used = S.rp_pool_used.p; sv.pool_cap = S.rp_pool_cap;
      sv.box_base = S.rp_box_base.p + Rr.b0; sv.p_dist = S.fb_dist.p; sv.p_angle = S.fb_angle.p; sv.p_skew = S.fb_skew.p; sv.p_flag = S.fb_flag.p; sv.p_slot = S.fb_slot.p;
      cs::launch_rp_save_fallback(v, sv, st);
    }
    HIP_TRY(hipEventRecord(re[4], st));
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(S.ev[6], st));
  if (g_dma_tables) {
    HIP_TRY(hipMemcpyAsync(S.h_records.p, S.records.p, sizeof(cs_cuboid) * nb * KMAX, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(S.h_win_count.p, S.win_count.p, sizeof(int) * nb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(S.h_fallback.p, S.fallback.p, sizeof(int) * nb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(S.h_job_cbase.p, S.job_cbase.p, sizeof(long long) * (nj + MB), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(S.h_job_valid.p, S.job_valid.p, sizeof(int) * nj, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(S.h_jobs_out.p, S.jobs.p, sizeof(cs::JobDesc) * nj, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(S.h_rp_last_slot.p, S.rp_last_slot.p, sizeof(long long) * nb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(S.h_rp_box_base.p, S.rp_box_base.p, sizeof(long long) * nb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(S.h_rp_pool_used.p, S.rp_pool_used.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  } else {
    cs::CopySegs cp{};
    SEG(S.h_records, S.records, nb * KMAX); SEG(S.h_win_count, S.win_count, nb); SEG(S.h_fallback, S.fallback, nb); SEG(S.h_job_cbase, S.job_cbase, nj + MB);
    SEG(S.h_job_valid, S.job_valid, nj); SEG(S.h_jobs_out, S.jobs, nj); SEG(S.h_rp_last_slot, S.rp_last_slot, nb); SEG(S.h_rp_box_base, S.rp_box_base, nb);
    SEG(S.h_rp_pool_used, S.rp_pool_used, 1);
    cs::launch_multi_copy(cp, st);
  }
#undef SEG
  HIP_TRY(hipEventRecord(S.done, st));
  S.in_flight = true;
  return CS_OK;
}

int rp_finish(PipeCtx& C, PipeSlot& S) {
  cs_detector* d = C.d; cs_batch* b = C.b;
  const cs_detect_params& P = d->prm;
  const int KMAX = P.max_cuboid_num, MB = b->max_boxes, NF = b->n_frames;
  cs_detect_timing& tm = *C.tm;
  double tw = now_ms();
  HIP_TRY(hipEventSynchronize(S.done));
  tm.d2h_ms += now_ms() - tw;
  S.in_flight = false;
  if (S.nj == 0) return CS_OK;
  double t0 = now_ms();
  {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, S.ev[0], S.ev[1])); tm.line_setup_ms += ms;
    for (int r = 0; r < MB; r++) {
      if (S.rp_rounds[r].nj == 0) continue;
      hipEvent_t* re = &S.rp_ev[5 * (size_t)r];
      HIP_TRY(hipEventElapsedTime(&ms, re[0], re[1])); tm.cand_kernel_ms += ms;      // vanishing points, corners, compaction
      HIP_TRY(hipEventElapsedTime(&ms, re[1], re[2])); tm.vp_kernel_ms += ms;
      HIP_TRY(hipEventElapsedTime(&ms, re[2], re[3])); tm.score_kernel_ms += ms;
      HIP_TRY(hipEventElapsedTime(&ms, re[3], re[4])); tm.rank_kernel_ms += ms;      // ranking, records, the flagged boxes' columns
      tm.cand_kernel_launches += 1;
    }
    tm.n_slots += S.slot_total;
    for (int r = 0; r < MB; r++) if (S.rp_rounds[r].nj) tm.n_valid += S.h_job_cbase.p[S.rp_rounds[r].j0 + r + S.rp_rounds[r].nj];
  }
  // ---- the flagged boxes: exact ranking on the host from their saved columns.  A frame is only redone when the exact ranking
  // carries a different camera yaw to the frame's next box than the device assumed (or the columns did not fit the pool).
  std::vector<char> redo(NF, 0);
  std::vector<char> host_done(S.nb, 0);     // box written from the host ranking
  int n_redo = 0;
  size_t n_fb = 0;
  for (size_t q = 0; q < S.nb; q++) n_fb += S.h_fallback.p[q] ? 1 : 0;
  tm.n_fallback_boxes += (int)n_fb;
  if (n_fb) {
    hipStream_t st2 = d->stream_hi;
    const size_t used = (size_t)std::min<unsigned long long>(*S.h_rp_pool_used.p, (unsigned long long)S.rp_pool_cap);
    int rc;
#define PENS(buf, n) do { rc = (buf).ensure(n); if (rc) return rc; } while (0)
    PENS(S.h_fb_dist, used + 1); PENS(S.h_fb_angle, used + 1); PENS(S.h_fb_skew, used + 1); PENS(S.h_fb_flag, used + 1); PENS(S.h_fb_slot, used + 1);
#undef PENS
    if (used && !g_dma_tables) {
      cs::CopySegs cp{};
      const void* srcs[5] = {S.fb_dist.p, S.fb_angle.p, S.fb_skew.p, S.fb_flag.p, S.fb_slot.p};
      void* dsts[5] = {S.h_fb_dist.p, S.h_fb_angle.p, S.h_fb_skew.p, S.h_fb_flag.p, S.h_fb_slot.p};
      const unsigned long long widths[5] = {8, 8, 8, 4, 8};
      for (int z = 0; z < 5; z++) { cp.s[z].src = srcs[z]; cp.s[z].dst = dsts[z]; cp.s[z].bytes = widths[z] * used; }
      cp.n = 5;
      cs::launch_multi_copy(cp, st2);
      HIP_TRY(hipStreamSynchronize(st2));
    } else if (used) {
      HIP_TRY(hipMemcpyAsync(S.h_fb_dist.p, S.fb_dist.p, 8 * used, hipMemcpyDeviceToHost, st2));
      HIP_TRY(hipMemcpyAsync(S.h_fb_angle.p, S.fb_angle.p, 8 * used, hipMemcpyDeviceToHost, st2));
      HIP_TRY(hipMemcpyAsync(S.h_fb_skew.p, S.fb_skew.p, 8 * used, hipMemcpyDeviceToHost, st2));
      HIP_TRY(hipMemcpyAsync(S.h_fb_flag.p, S.fb_flag.p, 4 * used, hipMemcpyDeviceToHost, st2));
      HIP_TRY(hipMemcpyAsync(S.h_fb_slot.p, S.fb_slot.p, 8 * used, hipMemcpyDeviceToHost, st2));
      HIP_TRY(hipStreamSynchronize(st2));
    }
    const std::vector<std::vector<CamCache>>& cam_rp = *C.cam_rp_all;
    // boxes of a frame in round order
    std::vector<std::vector<size_t>> boxes_of(NF);
    for (size_t q = 0; q < S.nb; q++) boxes_of[S.rp_box_frame[q]].push_back(q);
    auto carry_idx = [&](const cs::JobDesc& jl, long long last) { return last < 0 ? jl.RP : 1 + (int)((((last - jl.slot_off) >> 1) / jl.T) / jl.Y); };
    std::vector<int> fb_frames;
    for (int f = 0; f < NF; f++) { bool any = false; for (size_t q : boxes_of[f]) any = any || S.h_fallback.p[q]; if (any) fb_frames.push_back(f); }
    d->pool->run((int)fb_frames.size(), [&](int z) {
      const int f = fb_frames[z];
      const FrameIn& F = b->frames[f];
      for (size_t bq = 0; bq < boxes_of[f].size(); bq++) {
        const size_t q = boxes_of[f][bq];
        if (!S.h_fallback.p[q]) continue;
        const long long base = S.h_rp_box_base.p[q];
        if (base < 0) { redo[f] = 1; return; }
        // which round the box belongs to: its jobs are relative to that round's first job
        size_t r = 0;
        while (r + 1 < S.rp_rounds.size() && q >= S.rp_rounds[r + 1].b0) r++;
        const cs::JobDesc* jobs = S.h_jobs_out.p + S.rp_rounds[r].j0;
        const int* jvalid = S.h_job_valid.p + S.rp_rounds[r].j0;
        const int j0 = S.h_box_job0.p[q], nh = S.h_box_njobs.p[q], bi = S.rp_box_index[q];
        struct HP { int h, cand; double score, skew; };
        std::vector<HP> props;
        std::vector<long long> p0(nh);
        long long run = base, exact_last = -1;
        for (int h = 0; h < nh; h++) {
          const int V = jvalid[j0 + h];
          p0[h] = run; run += V;
          std::vector<int> keep;
          std::vector<double> score;
          fuse_scores(S.h_fb_dist.p + p0[h], S.h_fb_angle.p + p0[h], V, P.weight_vp_angle, keep, score);
          if (h == nh - 1) exact_last = keep.empty() ? -1 : S.h_fb_slot.p[p0[h] + keep.back()];
          for (size_t k = 0; k < keep.size(); k++) {
            if (S.h_fb_flag.p[p0[h] + keep[k]] & cs::CAND_NEG_SCALE) continue;
            props.push_back(HP{h, keep[k], score[k], S.h_fb_skew.p[p0[h] + keep[k]]});
          }
        }
        const int n = (int)props.size(), kk = std::min(KMAX, n);
        std::vector<double> comb(n);
        for (int i = 0; i < n; i++) {
          double skew_error = P.weight_skew_error * std::max(props[i].skew - P.nominal_skew_ratio, 0.0);
          if (props[i].skew > P.max_cut_skew) skew_error = 100;
          comb[i] = props[i].score + P.weight_skew_error * skew_error;
        }
        std::vector<int> idx(n);
        std::iota(idx.begin(), idx.end(), 0);
        std::partial_sort(idx.begin(), idx.begin() + kk, idx.end(), [&comb](int a, int c) { return comb[a] < comb[c]; });
        const double* bb = &F.boxes[5 * bi];
        for (int w = 0; w < kk; w++) {
          const HP& hp = props[idx[w]];
          const cs::JobDesc& jd = jobs[j0 + hp.h];
          const long long slot = S.h_fb_slot.p[p0[hp.h] + hp.cand];
          const long long local = slot - jd.slot_off, rest = local >> 1;
          const int t = (int)(rest % jd.T), ry = (int)(rest / jd.T), rp = ry / jd.Y, y = ry - rp * jd.Y;
          // the winner's corners: the vanishing points (vp_points_kernel's arithmetic) and build_corners, on the host
          const double* A = cam_rp[f][rp].pose.KinvR;
          const double cy = S.h_yaw_c.p[jd.yaw_off + y], sy = S.h_yaw_s.p[jd.yaw_off + y];
          const double dd[3][3] = {{cy, sy, 0.0}, {-sy, cy, 0.0}, {0.0, 0.0, 1.0}};
          cs::V2 vp[3];
          for (int k = 0; k < 3; k++) {
            const double h0 = (A[0] * dd[k][0] + A[1] * dd[k][1]) + A[2] * dd[k][2];
            const double h1 = (A[3] * dd[k][0] + A[4] * dd[k][1]) + A[5] * dd[k][2];
            const double h2 = (A[6] * dd[k][0] + A[7] * dd[k][1]) + A[8] * dd[k][2];
            vp[k] = cs::v2(h0 / h2, h1 / h2);
          }
          cs::V2 c8[8];
          for (auto& c : c8) c = cs::v2(0.0, 0.0);
          (void)cs::build_corners(jd.g, vp[0], vp[1], vp[2], (double)S.h_top_x.p[jd.top_off + t], (int)(local & 1) + 1, C.sp.short_sq_bound, c8);
          double c16[16];
          for (int k = 0; k < 8; k++) { c16[k] = c8[k].x; c16[8 + k] = c8[k].y; }
          double r9[9] = {(double)((local & 1) + 1), (double)(S.h_fb_flag.p[p0[hp.h] + hp.cand] & cs::CAND_VP_MASK), S.h_yaw.p[jd.yaw_off + y], (double)t,
                          S.h_fb_dist.p[p0[hp.h] + hp.cand], S.h_fb_angle.p[p0[hp.h] + hp.cand], (double)jd.down_expand, cam_rp[f][rp].pose.roll, cam_rp[f][rp].pose.pitch};
          cs_cuboid& o = C.out[((size_t)f * MB + bi) * KMAX + w];
          finish_cuboid(F, cam_rp[f][rp].pose, r9, c16, (*C.cam_raw)[f].euler, true, hp.score, o);
          o.rect_detect_2d[0] = (int)bb[0]; o.rect_detect_2d[1] = (int)bb[1]; o.rect_detect_2d[2] = (int)bb[2]; o.rect_detect_2d[3] = (int)bb[3];
        }
        C.out_counts[(size_t)f * MB + bi] = kk;
        host_done[q] = 1;
        // does the frame's next box start from the yaw the device assumed?
        const cs::JobDesc& jl = jobs[j0 + nh - 1];
        if (bq + 1 < boxes_of[f].size() && carry_idx(jl, exact_last) != carry_idx(jl, S.h_rp_last_slot.p[q])) { redo[f] = 1; return; }
      }
    });
    for (int f = 0; f < NF; f++) n_redo += redo[f] ? 1 : 0;
  }
  constexpr int RCH = 256;
  const int nch = ((int)S.nb + RCH - 1) / RCH;
  d->pool->run(nch, [&](int z) {
    for (size_t q = (size_t)z * RCH, q1 = std::min(S.nb, q + RCH); q < q1; q++) {
      const int f = S.rp_box_frame[q], bi = S.rp_box_index[q];
      if (redo[f] || host_done[q]) continue;
      const int nw = S.h_win_count.p[q];
      const double* bb = &b->frames[f].boxes[5 * bi];
      for (int r = 0; r < nw; r++) {
        cs_cuboid& o = C.out[((size_t)f * MB + bi) * KMAX + r];
        o = S.h_records.p[q * KMAX + r];
        o.rect_detect_2d[0] = (int)bb[0]; o.rect_detect_2d[1] = (int)bb[1]; o.rect_detect_2d[2] = (int)bb[2]; o.rect_detect_2d[3] = (int)bb[3];
      }
      C.out_counts[(size_t)f * MB + bi] = nw;
    }
  });
  if (n_redo) {
    // a small batch of those frames, sharing this batch's map pool (same offsets), through the round-by-round path
    cs_batch sub;
    sub.det = d;
    sub.force_round_path = true;
    std::vector<int> ids;
    for (int f = 0; f < NF; f++) if (redo[f]) ids.push_back(f);
    sub.n_frames = (int)ids.size();
    sub.frames.resize(ids.size());
    sub.max_boxes = 0;
    for (size_t z = 0; z < ids.size(); z++) { sub.frames[z] = b->frames[ids[z]]; sub.max_boxes = std::max(sub.max_boxes, sub.frames[z].n_boxes); }
    sub.d_maps.p = b->d_maps.p; sub.d_maps.cap = b->d_maps.cap;           // borrowed
    struct Unborrow { cs_batch* s; ~Unborrow() { s->d_maps.p = nullptr; s->d_maps.cap = 0; release_batch_buffers(s); } } ub{&sub};
    int rc = batch_layout(d, &sub);
    if (rc) return rc;
    const int MBs = sub.max_boxes;
    std::vector<cs_cuboid> o2((size_t)ids.size() * std::max(1, MBs) * KMAX);
    std::vector<int> c2((size_t)ids.size() * std::max(1, MBs), 0);
    rc = batch_run_impl(d, &sub, o2.data(), c2.data(), false);
    if (rc) return rc;
    for (size_t z = 0; z < ids.size(); z++) {
      const int f = ids[z];
      for (int bi = 0; bi < b->frames[f].n_boxes; bi++) {
        const int nw = c2[z * MBs + bi];
        for (int r = 0; r < nw; r++) C.out[((size_t)f * MB + bi) * KMAX + r] = o2[(z * MBs + bi) * KMAX + r];
        C.out_counts[(size_t)f * MB + bi] = nw;
      }
    }
    tm.n_redo_frames += n_redo;
  }
  tm.finalize_ms += now_ms() - t0;
  return CS_OK;
}
#undef PENS
#undef PH2D

}  // namespace

extern "C" int cs_batch_set_pipeline_chunks(cs_batch* b, int n_chunks) {
  if (!b || n_chunks < 1) return CS_ERR_INVALID_ARG;
  b->pipe_chunks = n_chunks;
  return CS_OK;
}

static int batch_run_impl(cs_detector* d, cs_batch* b, cs_cuboid* out, int* out_counts, bool defer);
static int batch_collect_impl(cs_detector* d, cs_batch* b);
extern "C" int cs_batch_run(cs_detector* d, cs_batch* b, cs_cuboid* out, int* out_counts) {
  CS_GUARD_BEGIN
  return batch_run_impl(d, b, out, out_counts, false);
  CS_GUARD_END("cs_batch_run")
}
// cs_batch_run in two halves: submit packs the batch on the host and queues the whole sweep on the detector's streams, collect waits
// for it and writes the records.  A caller that owns several batches keeps the next one queued while the current one is on the
// device (one submit outstanding per batch; batches of one detector are submitted and collected from one thread, in order).
extern "C" int cs_batch_submit(cs_detector* d, cs_batch* b, cs_cuboid* out, int* out_counts) {
  CS_GUARD_BEGIN
  return batch_run_impl(d, b, out, out_counts, true);
  CS_GUARD_END("cs_batch_submit")
}
extern "C" int cs_batch_collect(cs_detector* d, cs_batch* b) {
  CS_GUARD_BEGIN
  if (!d || !b || b->det != d) return CS_ERR_INVALID_ARG;
  return batch_collect_impl(d, b);
  CS_GUARD_END("cs_batch_collect")
}
static void batch_drop_run_state(cs_batch* b) { delete b->run_state; b->run_state = nullptr; }
static int batch_collect_impl(cs_detector* d, cs_batch* b) {
  if (!b->run_state) return CS_OK;                        // nothing outstanding (or the sweep ran to completion inside submit)
  std::unique_ptr<BatchRunState> rs(b->run_state);
  b->run_state = nullptr;
  if (!rs->deferred) return CS_OK;
  HIP_TRY(hipSetDevice(d->device));
  int rc = rs->rp_lean ? rp_finish(rs->C, b->pipe[0]) : pipe_finish(rs->C, b->pipe[0], rs->cam_rp);
  if (rc) return rc;
  rs->tm.total_ms = now_ms() - rs->t_begin;
  b->timing = rs->tm;
  b->ran = true;
  if (g_prof && (++g_runs % 8) == 0) {
    const char* nm[11] = {"jobs+samples", "prefix+pack", "box table", "alloc+enqueue", "gpu wait", "tie lists", "records", "tie wait", "tie rank", "tie corners", "tie records"};
    fprintf(stderr, "[detect] host ms/run:");
    for (int k = 0; k < 11; k++) { fprintf(stderr, " %s %.3f", nm[k], g_mark[k] / 8); g_mark[k] = 0; }
    fprintf(stderr, " | pre %.3f total %.3f\n", g_mark[11] / 8, rs->tm.total_ms); g_mark[11] = 0;
  }
  return CS_OK;
}
static int batch_run_impl(cs_detector* d, cs_batch* b, cs_cuboid* out, int* out_counts, bool defer) {
  if (!d || !b || b->det != d || !out || !out_counts) return CS_ERR_INVALID_ARG;
  if (b->run_state) { set_err("cs_batch_submit: the previous submit of this batch has not been collected"); return CS_ERR_INVALID_ARG; }
  HIP_TRY(hipSetDevice(d->device));
  { const int rcf = batch_flush_refill(d, b); if (rcf) return rcf; }      // (cs_batch_refill_gray: the new images' maps, in front of this sweep)
  std::unique_ptr<BatchRunState> rs_owner(new BatchRunState());
  BatchRunState* rs = rs_owner.get();
  const cs_detect_params& P = d->prm;
  const bool sample_rp = P.whether_sample_cam_roll_pitch != 0;
  const int NF = b->n_frames, MB = b->max_boxes, KMAX = P.max_cuboid_num;
  auto parallel_for = [d](int n, int /*nt*/, const std::function<void(int)>& fn) { d->pool->run(n, fn); };
  const int NT = d->n_threads;
  hipStream_t st = d->stream;
  cs_detect_timing& tm = rs->tm;
  double t_begin = now_ms();
  rs->t_begin = t_begin;

  std::fill(out_counts, out_counts + (size_t)NF * std::max(MB, 0), 0);
  b->results.clear();
  b->job_index.assign((size_t)NF * std::max(MB, 1) * 3, -1);

  cs::SweepParams sp;
  sp.vp12_thre_rad = P.vp12_edge_angle_thre / 180.0 * CS_PI;
  sp.vp3_thre_rad = P.vp3_edge_angle_thre / 180.0 * CS_PI;
  sp.short_thre = P.shorted_edge_thre;
  sp.short_sq_bound = cs::sqrt_lt_bound(P.shorted_edge_thre);
  sp.consider_config_1 = P.consider_config_1; sp.consider_config_2 = P.consider_config_2;

  // ---- per-frame camera caches: raw pose and the roll/pitch sample poses (:78-79, :344-355, :368-377)
  double t0 = now_ms();
  std::vector<CamCache>& cam_raw = rs->cam_raw; cam_raw.resize(NF);
  std::vector<std::vector<CamCache>>& cam_rp = rs->cam_rp; cam_rp.resize(NF);
  std::vector<double> cur_yaw(NF);  // cam_pose.camera_yaw as the next box will see it (:180)
  parallel_for(NF, NT, [&](int f) {
    const FrameIn& F = b->frames[f];
    make_cam(F.K, F.R, F.t, cam_raw[f]);
    cur_yaw[f] = cam_raw[f].cam_yaw;
    if (sample_rp) {
      std::vector<double> rs, ps;
      linespace<double>(cam_raw[f].euler[0] - 6.0 / 180.0 * CS_PI, cam_raw[f].euler[0] + 6.0 / 180.0 * CS_PI, 3.0 / 180.0 * CS_PI, rs);
      linespace<double>(cam_raw[f].euler[1] - 6.0 / 180.0 * CS_PI, cam_raw[f].euler[1] + 6.0 / 180.0 * CS_PI, 3.0 / 180.0 * CS_PI, ps);
      for (double r : rs)
        for (double p : ps) {
          double Rn[9];
          euler_to_rot(r, p, cam_raw[f].euler[2], Rn);
          CamCache c;
          make_cam(F.K, Rn, F.t, c);
          c.pose.roll = r; c.pose.pitch = p;  // the row stores the *sample* angles (:686)
          cam_rp[f].push_back(c);
        }
    } else {
      cam_rp[f].push_back(cam_raw[f]);
    }
  });
  // rp pool: identical for every round -> upload once
  std::vector<int>& rp_off = rs->rp_off; rp_off.assign(NF + 1, 0);
  for (int f = 0; f < NF; f++) rp_off[f + 1] = rp_off[f] + (int)cam_rp[f].size();
  {
    // (pinned staging owned by the batch: the copy is queued behind whatever the detector's stream still runs -- another batch's sweep --
    // and nobody waits for it here)
    const size_t np_ = (size_t)std::max(1, rp_off[NF]);
    int rc = b->h_rp.ensure(np_);
    if (rc) return rc;
    for (int f = 0; f < NF; f++)
      for (size_t k = 0; k < cam_rp[f].size(); k++) b->h_rp.p[rp_off[f] + k] = cam_rp[f][k].pose;
    rc = b->d_rp.ensure(np_);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(b->d_rp.p, b->h_rp.p, sizeof(cs::RpPose) * np_, hipMemcpyHostToDevice, st));
  }
  tm.setup_host_ms += now_ms() - t0;

  if (g_prof) g_mark[11] += now_ms() - t_begin;   // camera caches + pose pool upload
  // ---- production path: chunked two-slot pipeline (host packs chunk k+1 / finishes chunk k-1 while the GPU sweeps k)
  if (!sample_rp && !b->debug && !b->force_host_rank && !b->force_host_setup && !b->force_no_pipeline && b->device_setup && KMAX <= cs::RANK_KMAX && MB > 0) {
    rs->C = PipeCtx{d, b, out, out_counts, &rs->cam_raw, &rs->rp_off, sp, &rs->tm};
    PipeCtx& C = rs->C;
    const int n_chunks = std::max(1, std::min(NF, b->pipe_chunks));
    if (n_chunks == 1) {    // the usual shape: one launch, one finish -- the finish may be left to cs_batch_collect
      int rc = pipe_launch(C, b->pipe[0], 0, NF);
      if (rc) return rc;
      rs->deferred = true;
      b->run_state = rs_owner.release();
      return defer ? CS_OK : batch_collect_impl(d, b);
    }
    for (int k = 0; k <= n_chunks; k++) {
      if (k < n_chunks) {
        int f0 = (int)((long long)NF * k / n_chunks), f1 = (int)((long long)NF * (k + 1) / n_chunks);
        int rc = pipe_launch(C, b->pipe[k & 1], f0, f1);
        if (rc) return rc;
      }
      if (k >= 1) {
        int rc = pipe_finish(C, b->pipe[(k - 1) & 1], cam_rp);
        if (rc) return rc;
      }
    }
    tm.total_ms = now_ms() - t_begin;
    b->timing = tm;
    b->ran = true;
    if (g_prof && (++g_runs % 8) == 0) {
      const char* nm[11] = {"jobs+samples", "prefix+pack", "box table", "alloc+enqueue", "gpu wait", "tie lists", "records", "tie wait", "tie rank", "tie corners", "tie records"};
      fprintf(stderr, "[detect] host ms/run:");
      for (int k = 0; k < 11; k++) { fprintf(stderr, " %s %.3f", nm[k], g_mark[k] / 8); g_mark[k] = 0; }
      fprintf(stderr, " | pre %.3f total %.3f\n", g_mark[11] / 8, tm.total_ms); g_mark[11] = 0;
    }
    return CS_OK;
  }

  // ---- roll/pitch sampling, lean path: every round queued at once, the carried yaw picked on the device
  {
    int max_rp = 0;
    for (int f = 0; f < NF; f++) max_rp = std::max(max_rp, (int)cam_rp[f].size());
    static const bool rp_rounds_forced = getenv("CS_DETECT_RP_ROUNDS") != nullptr;   // diagnostics: always the round-by-round path
    if (sample_rp && !b->debug && !b->force_host_rank && !b->force_host_setup && !b->force_no_pipeline && !b->force_round_path && !rp_rounds_forced && b->device_setup &&
        KMAX <= cs::RANK_KMAX && MB > 0 && max_rp <= cs::vp3_table_doubles_per_job() / 2) {
      rs->C = PipeCtx{d, b, out, out_counts, &rs->cam_raw, &rs->rp_off, sp, &rs->tm, &rs->cam_rp};
      int rc = rp_launch(rs->C, b->pipe[0], rs->cam_rp);
      if (rc) return rc;
      rs->deferred = true; rs->rp_lean = true;
      b->run_state = rs_owner.release();
      return defer ? CS_OK : batch_collect_impl(d, b);
    }
  }
  // proposals per (frame, box) accumulate over height samples inside a round
  std::vector<Winner> winners;
  const int n_rounds = sample_rp ? MB : (MB > 0 ? 1 : 0);
  for (int round = 0; round < n_rounds; round++) {
    // ------------------------------------------------------------------ setup (host) ---------
    t0 = now_ms();
    const bool dev_setup = b->device_setup && !b->force_host_setup;
    std::vector<FrameRound> fr(NF);
    parallel_for(NF, NT, [&](int f) {
      const FrameIn& F = b->frames[f];
      FrameRound& R = fr[f];
      int b0 = sample_rp ? round : 0, b1 = sample_rp ? std::min(round + 1, F.n_boxes) : F.n_boxes;
      int shared_yoff = -1, shared_Y = 0;  // without roll/pitch sampling cam_pose never changes: every box of the frame sweeps the same yaw list (:180-184)
      for (int bi = b0; bi < b1; bi++) {
        const double* bb = &F.boxes[5 * bi];
        int left = bb[0], top = bb[1], w = bb[2], h = bb[3];
        int right = left + bb[2];
        int res = (int)std::round(std::min(20, w / 10));
        if (res < 1) continue;  // :215 break (same for every height sample)
        // yaw samples (:180-184)
        int yoff, nY;
        if (shared_yoff >= 0) { yoff = shared_yoff; nY = shared_Y; }
        else {
          double yaw_init = cur_yaw[f] - 90.0 / 180.0 * CS_PI;
          std::vector<double> yaws;
          linespace<double>(yaw_init - P.yaw_range_deg / 180.0 * CS_PI, yaw_init + P.yaw_range_deg / 180.0 * CS_PI, P.yaw_step_deg / 180.0 * CS_PI, yaws);
          yoff = (int)R.yaw.size(); nY = (int)yaws.size();
          for (double y : yaws) { R.yaw.push_back(y); R.yaw_c.push_back(h_cos(y)); R.yaw_s.push_back(h_sin(y)); }
          if (!sample_rp) { shared_yoff = yoff; shared_Y = nY; }
        }
        std::vector<int> tops;
        linespace<int>(left + 5, right - 5, res, tops);
        for (int k = 0; k < F.n_heights[bi]; k++) {
          const cs_roi& roi = F.rois[3 * bi + k];
          cs::JobDesc jd;
          std::memset(&jd, 0, sizeof(jd));
          int he = h + roi.down_expand;
          jd.g.left = left; jd.g.top = top; jd.g.right = right; jd.g.down = top + he;
          jd.g.el = roi.left; jd.g.et = roi.top; jd.g.er = roi.left + roi.width; jd.g.eb = roi.top + roi.height;
          jd.map_w = roi.width;
          jd.Y = nY; jd.T = (int)tops.size(); jd.RP = (int)cam_rp[f].size();
          jd.down_expand = roi.down_expand;
          jd.frame = f; jd.box = bi; jd.hid = k;
          jd.map_off = F.map_offs[3 * bi + k];
          jd.rp_off = rp_off[f];
          jd.diag = std::sqrt(double(w * w + he * he));
          JobHost jh;
          jh.frame = f; jh.box = bi; jh.hid = k; jh.yaw_off_local = yoff; jh.tops = tops;
          // ROI line filter (:271-283) + merge (:288-296) + angles/midpoints (:309-315): line_setup_kernel, or here
          if (dev_setup) { jd.m = 0; jh.line_cap = F.n_lines; R.jobs.push_back(jd); R.jh.push_back(std::move(jh)); continue; }
          std::vector<double> in;
          for (int e = 0; e < F.n_lines; e++) {
            const double* l = &F.lines[4 * e];
            if (cs::inside_box(cs::v2(l[0], l[1]), jd.g.el, jd.g.et, jd.g.er, jd.g.eb) &&
                cs::inside_box(cs::v2(l[2], l[3]), jd.g.el, jd.g.et, jd.g.er, jd.g.eb))
              in.insert(in.end(), l, l + 4);
          }
          merge_lines(in, P.pre_merge_dist_thre, P.pre_merge_angle_thre, P.edge_length_threshold);
          jd.m = (int)(in.size() / 4);
          for (int i = 0; i < jd.m; i++) {
            jh.angs.push_back(cs::cs_atan2(in[4 * i + 3] - in[4 * i + 1], in[4 * i + 2] - in[4 * i]));
            jh.mids_x.push_back((in[4 * i] + in[4 * i + 2]) / 2);
            jh.mids_y.push_back((in[4 * i + 1] + in[4 * i + 3]) / 2);
          }
          jh.line_cap = jd.m;
          R.jobs.push_back(jd);
          R.jh.push_back(std::move(jh));
        }
      }
    });
    // pack: per-frame sizes -> offsets (serial prefix over frames) -> parallel copy into the pooled arrays
    std::vector<size_t> f_job(NF + 1, 0), f_line(NF + 1, 0), f_yaw(NF + 1, 0), f_top(NF + 1, 0);
    std::vector<long long> f_slot(NF + 1, 0), f_vp(NF + 1, 0);
    for (int f = 0; f < NF; f++) {
      size_t nl = 0, nt = 0;
      long long ns = 0, nv = 0;
      for (size_t q = 0; q < fr[f].jobs.size(); q++) {
        const cs::JobDesc& jd = fr[f].jobs[q];
        nl += fr[f].jh[q].line_cap; nt += fr[f].jh[q].tops.size();
        nv += (long long)jd.RP * jd.Y; ns += (long long)jd.RP * jd.Y * jd.T * 2;
      }
      f_job[f + 1] = f_job[f] + fr[f].jobs.size(); f_line[f + 1] = f_line[f] + nl; f_yaw[f + 1] = f_yaw[f] + fr[f].yaw.size();
      f_top[f + 1] = f_top[f] + nt; f_slot[f + 1] = f_slot[f] + ns; f_vp[f + 1] = f_vp[f] + nv;
    }
    const size_t nj = f_job[NF], n_lines = f_line[NF], n_yaw = f_yaw[NF], n_top = f_top[NF];
    if (nj == 0) { tm.setup_host_ms += now_ms() - t0; continue; }
    if (f_vp[NF] > 0x7fffffffLL) { set_err("too many (roll,pitch,yaw) samples in one round"); return CS_ERR_CAPACITY; }
    std::vector<cs::JobDesc> jobs(nj);
    std::vector<long long> slot_prefix(nj + 1);
    std::vector<int> vp_prefix(nj + 1);
    const size_t n_lines_host = dev_setup ? 0 : n_lines;  // with the device line setup these tables are only ever written by the kernel
    std::vector<double> mid_x(n_lines_host), mid_y(n_lines_host), ang(n_lines_host), yaw(n_yaw), yaw_c(n_yaw), yaw_s(n_yaw);
    std::vector<int> top_x(n_top);
    parallel_for(NF, NT, [&](int f) {
      FrameRound& R = fr[f];
      size_t ji = f_job[f], lo = f_line[f], yo = f_yaw[f], to = f_top[f];
      long long so = f_slot[f], vo = f_vp[f];
      std::copy(R.yaw.begin(), R.yaw.end(), yaw.begin() + yo);
      std::copy(R.yaw_c.begin(), R.yaw_c.end(), yaw_c.begin() + yo);
      std::copy(R.yaw_s.begin(), R.yaw_s.end(), yaw_s.begin() + yo);
      for (size_t q = 0; q < R.jobs.size(); q++, ji++) {
        cs::JobDesc& jd = R.jobs[q];
        const JobHost& jh = R.jh[q];
        jd.line_off = (int)lo; jd.yaw_off = (int)(yo + jh.yaw_off_local); jd.top_off = (int)to;
        jd.vp_off = (int)vo; jd.slot_off = so;
        if (!dev_setup) {
          std::copy(jh.mids_x.begin(), jh.mids_x.end(), mid_x.begin() + lo);
          std::copy(jh.mids_y.begin(), jh.mids_y.end(), mid_y.begin() + lo);
          std::copy(jh.angs.begin(), jh.angs.end(), ang.begin() + lo);
        }
        std::copy(jh.tops.begin(), jh.tops.end(), top_x.begin() + to);
        lo += jh.line_cap; to += jh.tops.size();
        slot_prefix[ji] = so; vp_prefix[ji] = (int)vo;
        vo += (long long)jd.RP * jd.Y;
        so += (long long)jd.RP * jd.Y * jd.T * 2;
        jobs[ji] = jd;
      }
    });
    slot_prefix[nj] = f_slot[NF]; vp_prefix[nj] = (int)f_vp[NF];
    const long long slot_total = slot_prefix[nj];
    const int vp_total = vp_prefix[nj];
    tm.setup_host_ms += now_ms() - t0;
    tm.n_jobs += (long long)nj; tm.n_slots += slot_total;

    // ------------------------------------------------------------------ H2D -----------------
    t0 = now_ms();
    int rc;
#define ENS(buf, n) do { rc = (buf).ensure(n); if (rc) return rc; } while (0)
    ENS(b->d_jobs, nj); ENS(b->d_slot_prefix, nj + 1); ENS(b->d_vp_prefix, nj + 1); ENS(b->d_job_valid, nj); ENS(b->d_job_cbase, nj + 1);
    ENS(b->d_mid_x, n_lines + 1); ENS(b->d_mid_y, n_lines + 1); ENS(b->d_ang, n_lines + 1);
    ENS(b->d_yaw, n_yaw + 1); ENS(b->d_yaw_c, n_yaw + 1); ENS(b->d_yaw_s, n_yaw + 1); ENS(b->d_top_x, n_top + 1);
    ENS(b->d_vp, 6 * (size_t)vp_total + 6); ENS(b->d_bound, 6 * (size_t)vp_total + 6);
    ENS(b->d_flag, slot_total + 1);
    // The tables are small (a frame: ~10 of them, a few KB each).  From the std::vectors every hipMemcpyAsync is a staged, effectively
    // synchronous copy (10-25 us apiece, ~150 us per call); copied first into ONE pinned block they go out back to back and the host
    // does not wait (the round's results are awaited further down, before the block is written again).
    {
      size_t need = 0;
      auto room = [&](size_t bytes) { const size_t at = need; need += (bytes + 63) & ~(size_t)63; return at; };
      const size_t o_jobs = room(sizeof(cs::JobDesc) * jobs.size()), o_sp = room(sizeof(long long) * slot_prefix.size()), o_vp = room(sizeof(int) * vp_prefix.size());
      const size_t o_mx = room(8 * mid_x.size()), o_my = room(8 * mid_y.size()), o_an = room(8 * ang.size());
      const size_t o_y = room(8 * yaw.size()), o_yc = room(8 * yaw_c.size()), o_ys = room(8 * yaw_s.size()), o_tx = room(sizeof(int) * top_x.size());
      ENS(b->h_tables, need + 64);
      unsigned char* hb = b->h_tables.p;
#define H2DP(dst, vec, off) do { if (!(vec).empty()) { memcpy(hb + (off), (vec).data(), sizeof((vec)[0]) * (vec).size()); \
                                   HIP_TRY(hipMemcpyAsync((dst).p, hb + (off), sizeof((vec)[0]) * (vec).size(), hipMemcpyHostToDevice, st)); } } while (0)
      H2DP(b->d_jobs, jobs, o_jobs); H2DP(b->d_slot_prefix, slot_prefix, o_sp); H2DP(b->d_vp_prefix, vp_prefix, o_vp);
      if (n_lines && !dev_setup) { H2DP(b->d_mid_x, mid_x, o_mx); H2DP(b->d_mid_y, mid_y, o_my); H2DP(b->d_ang, ang, o_an); }
      if (n_yaw) { H2DP(b->d_yaw, yaw, o_y); H2DP(b->d_yaw_c, yaw_c, o_yc); H2DP(b->d_yaw_s, yaw_s, o_ys); }
      if (n_top) H2DP(b->d_top_x, top_x, o_tx);
    }
#define H2D(dst, vec) HIP_TRY(hipMemcpyAsync((dst).p, (vec).data(), sizeof((vec)[0]) * (vec).size(), hipMemcpyHostToDevice, st))
    HIP_TRY(hipMemsetAsync(b->d_job_valid.p, 0, sizeof(int) * nj, st));
    tm.h2d_ms += now_ms() - t0;

    // ------------------------------------------------------------------ sweep (HIP) ----------
    cs::DetectDeviceView v{};
    v.jobs = b->d_jobs.p; v.n_jobs = (int)nj; v.slot_prefix = b->d_slot_prefix.p; v.vp_prefix = b->d_vp_prefix.p;
    v.maps = b->d_maps.p; v.mid_x = b->d_mid_x.p; v.mid_y = b->d_mid_y.p; v.line_angle = b->d_ang.p;
    v.yaw = b->d_yaw.p; v.yaw_cos = b->d_yaw_c.p; v.yaw_sin = b->d_yaw_s.p; v.top_x = b->d_top_x.p; v.rp = b->d_rp.p; v.invK = b->d_invK.p;
    v.vp = b->d_vp.p; v.bound = b->d_bound.p; v.flag = b->d_flag.p;
    v.job_valid = b->d_job_valid.p; v.job_cbase = b->d_job_cbase.p;
    HIP_TRY(hipEventRecord(d->ev[6], st));
    if (dev_setup) {
      cs::launch_line_setup(b->d_jobs.p, (int)nj, b->d_frame_lines.p, b->d_frame_line_ptr.p, b->d_mid_x.p, b->d_mid_y.p, b->d_ang.p,
                            P.pre_merge_dist_thre, P.pre_merge_angle_thre, P.edge_length_threshold, st);
    }
    HIP_TRY(hipEventRecord(d->ev[7], st));
    if (dev_setup) {
      // the merged segment counts come back with the results (byte accounting, debug getters)
      ENS(b->h_jobs, nj);
      HIP_TRY(hipMemcpyAsync(b->h_jobs.p, b->d_jobs.p, sizeof(cs::JobDesc) * nj, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipEventRecord(d->ev[0], st));
    cs::launch_vp_support(v, sp, vp_total, st);
    HIP_TRY(hipEventRecord(d->ev[1], st));
    cs::launch_candidates(v, sp, slot_total, st);
    HIP_TRY(hipEventRecord(d->ev[2], st));
    HIP_TRY(hipGetLastError());

    // ------------------------------------------------------------------ rank on the device --------
    // (no roll/pitch sampling: boxes are independent, so nothing has to come back to the host before the ranking)
    const bool device_rank = !b->debug && !b->force_host_rank && KMAX <= cs::RANK_KMAX;
    if (device_rank) {
      ENS(b->d_c_slot, slot_total + 1); ENS(b->d_c_flag, slot_total + 1); ENS(b->d_c_dist, slot_total + 1); ENS(b->d_c_angle, slot_total + 1); ENS(b->d_c_skew, slot_total + 1);
      v.c_slot = b->d_c_slot.p; v.c_flag = b->d_c_flag.p; v.c_dist = b->d_c_dist.p; v.c_angle = b->d_c_angle.p; v.c_skew = b->d_c_skew.p;
      HIP_TRY(hipEventRecord(d->ev[3], st));
      cs::launch_scan_compact(v, st);
      HIP_TRY(hipEventRecord(d->ev[8], st));
      cs::launch_score(v, sp, slot_total, slot_total, st);   // the exact number of valid proposals stays on the device
      HIP_TRY(hipEventRecord(d->ev[4], st));
      std::vector<int> box_job0, box_njobs;
      for (size_t j = 0; j < nj; j++)
        if (jobs[j].hid == 0) { box_job0.push_back((int)j); box_njobs.push_back(b->frames[jobs[j].frame].n_heights[jobs[j].box]); }
      const size_t nb = box_job0.size();
      ENS(b->d_box_job0, nb); ENS(b->d_box_njobs, nb); ENS(b->d_win_count, nb); ENS(b->d_fallback, nb); ENS(b->d_winners, nb * KMAX);
      ENS(b->h_winners, nb * KMAX); ENS(b->h_win_count, nb); ENS(b->h_fallback, nb); ENS(b->h_job_valid, nj); ENS(b->h_job_cbase, nj + 1);
      {
        ENS(b->h_tables2, 2 * sizeof(int) * nb + 128);
        unsigned char* hb = b->h_tables2.p;
        const size_t o1 = (sizeof(int) * nb + 63) & ~(size_t)63;
        H2DP(b->d_box_job0, box_job0, 0); H2DP(b->d_box_njobs, box_njobs, o1);
      }
      cs::RankView rv{};
      rv.box_job0 = b->d_box_job0.p; rv.box_njobs = b->d_box_njobs.p; rv.n_boxes = (int)nb;
      rv.winners = b->d_winners.p; rv.win_count = b->d_win_count.p; rv.fallback = b->d_fallback.p;
      if (sample_rp) { ENS(b->d_last_slot, nb); ENS(b->h_last_slot, nb); rv.last_slot = b->d_last_slot.p; }
      cs::RankParams rkp{P.weight_vp_angle, P.weight_skew_error, P.nominal_skew_ratio, P.max_cut_skew, KMAX, sp.short_sq_bound};
      cs::launch_rank(v, rv, rkp, st);
      HIP_TRY(hipEventRecord(d->ev[5], st));
      HIP_TRY(hipGetLastError());
      double t_d2h = now_ms();
      HIP_TRY(hipMemcpyAsync(b->h_winners.p, b->d_winners.p, sizeof(cs::RankWinner) * nb * KMAX, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(b->h_win_count.p, b->d_win_count.p, sizeof(int) * nb, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(b->h_fallback.p, b->d_fallback.p, sizeof(int) * nb, hipMemcpyDeviceToHost, st));
      if (sample_rp) HIP_TRY(hipMemcpyAsync(b->h_last_slot.p, b->d_last_slot.p, sizeof(long long) * nb, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(b->h_job_valid.p, b->d_job_valid.p, sizeof(int) * nj, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(b->h_job_cbase.p, b->d_job_cbase.p, sizeof(long long) * (nj + 1), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      tm.d2h_ms += now_ms() - t_d2h;
      if (dev_setup) for (size_t j = 0; j < nj; j++) jobs[j].m = b->h_jobs.p[j].m;
      const long long n_valid = b->h_job_cbase.p[nj];
      tm.n_valid += n_valid;
      {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, d->ev[0], d->ev[1])); tm.vp_kernel_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, d->ev[1], d->ev[2])); tm.cand_kernel_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, d->ev[3], d->ev[8])); tm.compact_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, d->ev[8], d->ev[4])); tm.score_kernel_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, d->ev[4], d->ev[5])); tm.rank_kernel_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, d->ev[6], d->ev[7])); tm.line_setup_ms += ms;
        tm.cand_kernel_launches += 1;
        // algorithmic bytes: geometry kernel = vanishing points read once per (job, rp, yaw) + one flag per slot
        // (the corners stay in registers); scoring kernel = each distance map once + the vanishing points and the VP support table
        // once per (job, rp, yaw) + per valid proposal 12 B read (slot id, flag) and 28 B of scores written
        tm.cand_kernel_bytes += 48LL * vp_total + 4LL * slot_total;
        long long sbytes = 96LL * vp_total + (28LL + 8LL + 4LL) * n_valid;
        for (size_t j = 0; j < nj; j++) sbytes += 4LL * jobs[j].map_w * (jobs[j].g.eb - jobs[j].g.et);
        tm.score_kernel_bytes += sbytes;
      }
      // ---- finish on the host: records of the winners; boxes flagged by the kernel are re-ranked exactly
      t0 = now_ms();
      auto rows_of = [&](const cs::JobDesc& jd, long long slot, int flag, double de, double ae, double* r9, int* rp_out) {
        long long local = slot - jd.slot_off;
        int cfg = (int)(local & 1) + 1;
        long long rest = local >> 1;
        int t = (int)(rest % jd.T);
        int ry = (int)(rest / jd.T);
        int rp = ry / jd.Y, y = ry - rp * jd.Y;
        *rp_out = rp;
        r9[0] = cfg; r9[1] = flag & cs::CAND_VP_MASK; r9[2] = yaw[jd.yaw_off + y]; r9[3] = t; r9[4] = de; r9[5] = ae; r9[6] = jd.down_expand;
        r9[7] = cam_rp[jd.frame][rp].pose.roll; r9[8] = cam_rp[jd.frame][rp].pose.pitch;
      };
      // columns of every flagged box, fetched with one gather + one copy per column
      std::vector<long long> fb_src, fb_dst;
      std::vector<int> fb_cnt, fb_range_of_job(nj, -1);
      std::vector<double> fb_dist, fb_angle, fb_skew;
      std::vector<int> fb_flag;
      std::vector<long long> fb_slot;
      {
        long long tot = 0;
        for (size_t q = 0; q < nb; q++)
          if (b->h_fallback.p[q])
            for (int h = 0; h < box_njobs[q]; h++) {
              int j = box_job0[q] + h;
              fb_range_of_job[j] = (int)fb_src.size();
              fb_src.push_back(b->h_job_cbase.p[j]); fb_cnt.push_back(b->h_job_valid.p[j]); fb_dst.push_back(tot);
              tot += b->h_job_valid.p[j];
            }
        if (!fb_src.empty()) {
          size_t nr = fb_src.size();
          ENS(b->d_fb_src, nr); ENS(b->d_fb_dst, nr); ENS(b->d_fb_cnt, nr);
          ENS(b->d_fb_dist, tot + 1); ENS(b->d_fb_angle, tot + 1); ENS(b->d_fb_skew, tot + 1); ENS(b->d_fb_flag, tot + 1); ENS(b->d_fb_slot, tot + 1);
          H2D(b->d_fb_src, fb_src); H2D(b->d_fb_dst, fb_dst); H2D(b->d_fb_cnt, fb_cnt);
          cs::launch_gather_ranges(v, b->d_fb_src.p, b->d_fb_cnt.p, b->d_fb_dst.p, (int)nr, b->d_fb_dist.p, b->d_fb_angle.p, b->d_fb_skew.p, b->d_fb_flag.p, b->d_fb_slot.p, st);
          fb_dist.resize(tot); fb_angle.resize(tot); fb_skew.resize(tot); fb_flag.resize(tot); fb_slot.resize(tot);
          if (tot) {
            HIP_TRY(hipMemcpyAsync(fb_dist.data(), b->d_fb_dist.p, 8 * (size_t)tot, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(fb_angle.data(), b->d_fb_angle.p, 8 * (size_t)tot, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(fb_skew.data(), b->d_fb_skew.p, 8 * (size_t)tot, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(fb_flag.data(), b->d_fb_flag.p, 4 * (size_t)tot, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(fb_slot.data(), b->d_fb_slot.p, 8 * (size_t)tot, hipMemcpyDeviceToHost, st));
          }
          HIP_TRY(hipStreamSynchronize(st));
        }
      }
      std::vector<std::vector<cs::RankWinner>> fb_winners(nb);
      int hip_err = 0;
      auto finish_box = [&](size_t q, int phase) -> int {
        const int j0 = box_job0[q], nh = box_njobs[q];
        const int f = jobs[j0].frame, bi = jobs[j0].box;
        const FrameIn& F = b->frames[f];
        const double* bb = &F.boxes[5 * bi];
        std::vector<cs::RankWinner> wl;
        // cam_pose.camera_yaw as the next box of this frame reads it (:180): the reference's last set_cam_pose() of this box is
        // the one for the last kept proposal of the last height sample (:727-737), or, when nothing was kept, the sweep's last
        // (roll, pitch) sample (:368-377).  (One box per frame and round: no two workers write the same entry.)
        auto carry_yaw = [&](long long last_slot) {
          const cs::JobDesc& jl = jobs[j0 + nh - 1];
          if (last_slot < 0) { cur_yaw[f] = cam_rp[f].back().cam_yaw; return; }
          const long long ry = ((last_slot - jl.slot_off) >> 1) / jl.T;
          cur_yaw[f] = cam_rp[f][(size_t)(ry / jl.Y)].cam_yaw;
        };
        if (!b->h_fallback.p[q]) {
          if (sample_rp && phase == 0) carry_yaw(b->h_last_slot.p[q]);
          for (int r = 0; r < b->h_win_count.p[q]; r++) wl.push_back(b->h_winners.p[q * KMAX + r]);
        } else if (phase == 1) {
          wl = fb_winners[q];
        } else {
          struct HP { int h, cand; double score, skew; };
          std::vector<HP> props;
          std::vector<std::vector<double>> hd(nh), ha(nh), hs(nh);
          std::vector<std::vector<int>> hf(nh);
          std::vector<std::vector<long long>> hsl(nh);
          for (int h = 0; h < nh; h++) {
            int j = j0 + h;
            int V = b->h_job_valid.p[j];
            long long p0 = fb_dst[fb_range_of_job[j]];
            hd[h].assign(fb_dist.begin() + p0, fb_dist.begin() + p0 + V); ha[h].assign(fb_angle.begin() + p0, fb_angle.begin() + p0 + V);
            hs[h].assign(fb_skew.begin() + p0, fb_skew.begin() + p0 + V); hf[h].assign(fb_flag.begin() + p0, fb_flag.begin() + p0 + V);
            hsl[h].assign(fb_slot.begin() + p0, fb_slot.begin() + p0 + V);
            std::vector<int> keep;
            std::vector<double> score;
            fuse_scores(hd[h].data(), ha[h].data(), V, P.weight_vp_angle, keep, score);
            if (sample_rp && h == nh - 1) carry_yaw(keep.empty() ? -1 : hsl[h][keep.back()]);
            for (size_t z = 0; z < keep.size(); z++) {
              if (hf[h][keep[z]] & cs::CAND_NEG_SCALE) continue;
              props.push_back(HP{h, keep[z], score[z], hs[h][keep[z]]});
            }
          }
          int n = (int)props.size(), kk = std::min(KMAX, n);
          std::vector<double> comb(n);
          for (int i = 0; i < n; i++) {
            double skew_error = P.weight_skew_error * std::max(props[i].skew - P.nominal_skew_ratio, 0.0);
            if (props[i].skew > P.max_cut_skew) skew_error = 100;
            comb[i] = props[i].score + P.weight_skew_error * skew_error;
          }
          std::vector<int> idx(n);
          std::iota(idx.begin(), idx.end(), 0);
          std::partial_sort(idx.begin(), idx.begin() + kk, idx.end(), [&comb](int a, int c) { return comb[a] < comb[c]; });
          for (int r = 0; r < kk; r++) {
            const HP& hp = props[idx[r]];
            cs::RankWinner w{};
            w.slot = hsl[hp.h][hp.cand]; w.normalized_error = hp.score; w.dist_err = hd[hp.h][hp.cand]; w.angle_err = ha[hp.h][hp.cand];
            w.flag = hf[hp.h][hp.cand] & cs::CAND_VP_MASK;
            wl.push_back(w);  // corners fetched below, in one gather for all fallback winners
          }
          fb_winners[q] = wl;
          return CS_OK;
        }
        for (size_t r = 0; r < wl.size(); r++) {
          const cs::RankWinner& w = wl[r];
          int h = 0;
          while (h + 1 < nh && w.slot >= jobs[j0 + h + 1].slot_off) h++;
          double r9[9];
          int rpi = 0;
          rows_of(jobs[j0 + h], w.slot, w.flag, w.dist_err, w.angle_err, r9, &rpi);
          cs_cuboid& o = out[((size_t)f * MB + bi) * KMAX + r];
          finish_cuboid(F, cam_rp[f][rpi].pose, r9, w.corners, cam_raw[f].euler, sample_rp, w.normalized_error, o);
          o.rect_detect_2d[0] = (int)bb[0]; o.rect_detect_2d[1] = (int)bb[1]; o.rect_detect_2d[2] = (int)bb[2]; o.rect_detect_2d[3] = (int)bb[3];
        }
        out_counts[(size_t)f * MB + bi] = (int)wl.size();
        return CS_OK;
      };
      parallel_for((int)nb, NT, [&](int q) { (void)finish_box((size_t)q, 0); });  // flagged boxes: exact host ranking only
      {
        std::vector<long long> ws;
        for (size_t q = 0; q < nb; q++) { if (b->h_fallback.p[q]) { tm.n_fallback_boxes++; for (auto& w : fb_winners[q]) ws.push_back(w.slot); } }
        if (!ws.empty()) {
          ENS(b->d_win_slots, ws.size()); ENS(b->d_win_corners, 16 * ws.size()); ENS(b->h_win_corners, 16 * ws.size());
          H2D(b->d_win_slots, ws);
          cs::launch_gather_corners(v, sp, b->d_win_slots.p, (int)ws.size(), b->d_win_corners.p, st);
          HIP_TRY(hipMemcpyAsync(b->h_win_corners.p, b->d_win_corners.p, sizeof(double) * 16 * ws.size(), hipMemcpyDeviceToHost, st));
          HIP_TRY(hipStreamSynchronize(st));
          size_t z = 0;
          for (size_t q = 0; q < nb; q++)
            if (b->h_fallback.p[q])
              for (auto& w : fb_winners[q]) { std::memcpy(w.corners, b->h_win_corners.p + 16 * z, 128); z++; }
        }
        parallel_for((int)nb, NT, [&](int q) { if (b->h_fallback.p[q]) (void)finish_box((size_t)q, 1); });
      }
      if (hip_err) return hip_err;
      tm.finalize_ms += now_ms() - t0;
      continue;
    }
    // valid counts -> host -> size the compact arrays
    ENS(b->h_job_valid, nj); ENS(b->h_job_cbase, nj + 1);
    HIP_TRY(hipMemcpyAsync(b->h_job_valid.p, b->d_job_valid.p, sizeof(int) * nj, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (dev_setup) for (size_t j = 0; j < nj; j++) jobs[j].m = b->h_jobs.p[j].m;
    long long n_valid = 0;
    for (size_t j = 0; j < nj; j++) { b->h_job_cbase.p[j] = n_valid; n_valid += b->h_job_valid.p[j]; }
    b->h_job_cbase.p[nj] = n_valid;
    tm.n_valid += n_valid;
    ENS(b->d_c_slot, n_valid + 1); ENS(b->d_c_flag, n_valid + 1); ENS(b->d_c_dist, n_valid + 1); ENS(b->d_c_angle, n_valid + 1); ENS(b->d_c_skew, n_valid + 1);
    v.c_slot = b->d_c_slot.p; v.c_flag = b->d_c_flag.p; v.c_dist = b->d_c_dist.p; v.c_angle = b->d_c_angle.p; v.c_skew = b->d_c_skew.p;
    HIP_TRY(hipEventRecord(d->ev[3], st));
    cs::launch_scan_compact(v, st);
    HIP_TRY(hipEventRecord(d->ev[8], st));
    cs::launch_score(v, sp, n_valid, slot_total, st);
    HIP_TRY(hipEventRecord(d->ev[4], st));
    HIP_TRY(hipGetLastError());
    ENS(b->h_c_slot, n_valid + 1); ENS(b->h_c_flag, n_valid + 1); ENS(b->h_c_dist, n_valid + 1); ENS(b->h_c_angle, n_valid + 1); ENS(b->h_c_skew, n_valid + 1);
    double t_d2h = now_ms();
    if (n_valid) {
      HIP_TRY(hipMemcpyAsync(b->h_c_slot.p, b->d_c_slot.p, sizeof(long long) * n_valid, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(b->h_c_flag.p, b->d_c_flag.p, sizeof(int) * n_valid, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(b->h_c_dist.p, b->d_c_dist.p, sizeof(double) * n_valid, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(b->h_c_angle.p, b->d_c_angle.p, sizeof(double) * n_valid, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(b->h_c_skew.p, b->d_c_skew.p, sizeof(double) * n_valid, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    tm.d2h_ms += now_ms() - t_d2h;
    {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, d->ev[0], d->ev[1])); tm.vp_kernel_ms += ms;
      HIP_TRY(hipEventElapsedTime(&ms, d->ev[1], d->ev[2])); tm.cand_kernel_ms += ms;
      HIP_TRY(hipEventElapsedTime(&ms, d->ev[3], d->ev[8])); tm.compact_ms += ms;
      HIP_TRY(hipEventElapsedTime(&ms, d->ev[8], d->ev[4])); tm.score_kernel_ms += ms;
      HIP_TRY(hipEventElapsedTime(&ms, d->ev[6], d->ev[7])); tm.line_setup_ms += ms;
      tm.cand_kernel_launches += 1;
      // algorithmic bytes of the candidate kernel: maps + line arrays + vp/bound read once,
      // 200 B written per valid proposal, 4 B flag per slot
      // algorithmic bytes: as above
      tm.cand_kernel_bytes += 48LL * vp_total + 4LL * slot_total;
      long long sbytes = 96LL * vp_total + (28LL + 8LL + 4LL) * n_valid;
      for (size_t j = 0; j < nj; j++) sbytes += 4LL * jobs[j].map_w * (jobs[j].g.eb - jobs[j].g.et);
      tm.score_kernel_bytes += sbytes;
    }

    // ------------------------------------------------------------------ rank (host) ----------
    t0 = now_ms();
    size_t res_base = b->results.size();
    b->results.resize(res_base + nj);
    parallel_for((int)nj, NT, [&](int j) {
      const cs::JobDesc& jd = jobs[j];
      JobResult& R = b->results[res_base + j];
      R.frame = jd.frame; R.box = jd.box; R.hid = jd.hid;
      long long c0 = b->h_job_cbase.p[j];
      int V = b->h_job_valid.p[j];
      R.n_valid = V;
      R.rows9.resize(9 * (size_t)V);
      R.slots.assign(b->h_c_slot.p + c0, b->h_c_slot.p + c0 + V);
      R.rp_idx.resize(V);
      const cs::RpPose* rpp = nullptr;
      for (int i = 0; i < V; i++) {
        long long local = R.slots[i] - jd.slot_off;
        int cfg = (int)(local & 1) + 1;
        long long rest = local >> 1;
        int t = (int)(rest % jd.T);
        int ry = (int)(rest / jd.T);
        int rp = ry / jd.Y, y = ry - rp * jd.Y;
        rpp = &cam_rp[jd.frame][rp].pose;
        R.rp_idx[i] = rp;
        double* r9 = &R.rows9[9 * (size_t)i];
        r9[0] = cfg; r9[1] = b->h_c_flag.p[c0 + i] & cs::CAND_VP_MASK; r9[2] = yaw[jd.yaw_off + y]; r9[3] = t;
        r9[4] = b->h_c_dist.p[c0 + i]; r9[5] = b->h_c_angle.p[c0 + i]; r9[6] = jd.down_expand;
        r9[7] = rpp->roll; r9[8] = rpp->pitch;
      }
      fuse_scores(b->h_c_dist.p + c0, b->h_c_angle.p + c0, V, P.weight_vp_angle, R.keep, R.score);
    });
    for (size_t j = 0; j < nj; j++) b->job_index[((size_t)jobs[j].frame * MB + jobs[j].box) * 3 + jobs[j].hid] = (int)(res_base + j);

    // per box: proposals over its height samples (in order), final ranking (:804-838)
    std::vector<std::pair<int, int>> round_boxes;  // (frame, box)
    for (size_t j = 0; j < nj; j++)
      if (jobs[j].hid == 0) round_boxes.emplace_back(jobs[j].frame, jobs[j].box);
    std::vector<std::vector<Winner>> win_per_box(round_boxes.size());
    parallel_for((int)round_boxes.size(), NT, [&](int q) {
      int f = round_boxes[q].first, bi = round_boxes[q].second;
      const FrameIn& F = b->frames[f];
      std::vector<Proposal> props;
      int last_ri = -1;
      for (int k = 0; k < F.n_heights[bi]; k++) {
        int ri = b->job_index[((size_t)f * MB + bi) * 3 + k];
        if (ri < 0) continue;
        last_ri = ri;
        const JobResult& R = b->results[ri];
        long long c0 = b->h_job_cbase.p[ri - res_base];
        for (size_t q2 = 0; q2 < R.keep.size(); q2++) {
          int cand = R.keep[q2];
          if (b->h_c_flag.p[c0 + cand] & cs::CAND_NEG_SCALE) continue;  // :766
          props.push_back(Proposal{ri, cand, R.score[q2], b->h_c_skew.p[c0 + cand]});
        }
      }
      if (sample_rp && last_ri >= 0) {
        // cam_pose.camera_yaw as the next box of this frame reads it (:180): the reference's last
        // set_cam_pose() of this box is the one for the last kept proposal of the last height sample
        // (:727-737), or, when nothing was kept, the sweep's last (roll, pitch) sample (:368-377).
        const JobResult& R = b->results[last_ri];
        if (!R.keep.empty()) cur_yaw[f] = cam_rp[f][R.rp_idx[R.keep.back()]].cam_yaw;
        else cur_yaw[f] = cam_rp[f].back().cam_yaw;
      }
      int n = (int)props.size();
      int kk = std::min(KMAX, n);
      std::vector<double> comb(n);
      for (int i = 0; i < n; i++) {
        double skew_error = P.weight_skew_error * std::max(props[i].skew - P.nominal_skew_ratio, 0.0);
        if (props[i].skew > P.max_cut_skew) skew_error = 100;
        comb[i] = props[i].normalized_error + P.weight_skew_error * skew_error;  // weight applied twice (:813,:820)
      }
      std::vector<int> idx(n);
      std::iota(idx.begin(), idx.end(), 0);
      std::partial_sort(idx.begin(), idx.begin() + kk, idx.end(), [&comb](int a, int c) { return comb[a] < comb[c]; });
      for (int r = 0; r < kk; r++) {
        const Proposal& p = props[idx[r]];
        win_per_box[q].push_back(Winner{f, bi, r, p.res_idx, p.cand, p.normalized_error, p.skew});
      }
    });
    size_t w0 = winners.size();
    for (auto& wv : win_per_box) winners.insert(winners.end(), wv.begin(), wv.end());
    tm.rank_host_ms += now_ms() - t0;

    // ------------------------------------------------------------------ finish ---------------
    t0 = now_ms();
    size_t nw = winners.size() - w0;
    std::vector<long long> wslots(nw);
    for (size_t i = 0; i < nw; i++) wslots[i] = b->results[winners[w0 + i].res_idx].slots[winners[w0 + i].cand];
    // debug: corners of every valid candidate of this round
    size_t n_gather = nw + (b->debug ? (size_t)n_valid : 0);
    if (b->debug) wslots.insert(wslots.end(), b->h_c_slot.p, b->h_c_slot.p + n_valid);
    if (n_gather) {
      ENS(b->d_win_slots, n_gather); ENS(b->d_win_corners, 16 * n_gather); ENS(b->h_win_corners, 16 * n_gather);
      HIP_TRY(hipMemcpyAsync(b->d_win_slots.p, wslots.data(), sizeof(long long) * n_gather, hipMemcpyHostToDevice, st));
      cs::launch_gather_corners(v, sp, b->d_win_slots.p, (int)n_gather, b->d_win_corners.p, st);
      HIP_TRY(hipMemcpyAsync(b->h_win_corners.p, b->d_win_corners.p, sizeof(double) * 16 * n_gather, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
    for (size_t i = 0; i < nw; i++) {
      const Winner& w = winners[w0 + i];
      const JobResult& R = b->results[w.res_idx];
      const FrameIn& F = b->frames[w.frame];
      const double* r9 = &R.rows9[9 * (size_t)w.cand];
      const cs::RpPose* pose = &cam_rp[w.frame][R.rp_idx[w.cand]].pose;
      cs_cuboid& o = out[((size_t)w.frame * MB + w.box) * KMAX + w.rank];
      finish_cuboid(F, *pose, r9, b->h_win_corners.p + 16 * i, cam_raw[w.frame].euler, sample_rp, w.normalized_error, o);
      const double* bb = &F.boxes[5 * w.box];
      o.rect_detect_2d[0] = (int)bb[0]; o.rect_detect_2d[1] = (int)bb[1]; o.rect_detect_2d[2] = (int)bb[2]; o.rect_detect_2d[3] = (int)bb[3];
      out_counts[(size_t)w.frame * MB + w.box] = std::max(out_counts[(size_t)w.frame * MB + w.box], w.rank + 1);
    }
    if (b->debug) {
      for (size_t j = 0; j < nj; j++) {
        JobResult& R = b->results[res_base + j];
        long long c0 = b->h_job_cbase.p[j];
        R.corners.assign(b->h_win_corners.p + 16 * (nw + c0), b->h_win_corners.p + 16 * (nw + c0 + R.n_valid));
      }
    }
    tm.finalize_ms += now_ms() - t0;
  }
  tm.total_ms = now_ms() - t_begin;
  b->timing = tm;
  b->ran = true;
  return CS_OK;
}

extern "C" {

int cs_batch_last_timing(const cs_batch* b, cs_detect_timing* t) {
  if (!b || !t) return CS_ERR_INVALID_ARG;
  if (!b->ran) return CS_ERR_NOT_RUN;
  *t = b->timing;
  return CS_OK;
}

int cs_batch_debug_candidates(cs_batch* b, int frame, int box, int k, int cap, double* rows9, double* corners16) {
  if (!b || frame < 0 || frame >= b->n_frames || box < 0 || box >= b->max_boxes || k < 0 || k > 2) return CS_ERR_INVALID_ARG;
  if (!b->ran) return CS_ERR_NOT_RUN;
  int ri = b->job_index[((size_t)frame * b->max_boxes + box) * 3 + k];
  if (ri < 0) return 0;
  const JobResult& R = b->results[ri];
  int n = std::min(cap, R.n_valid);
  if (rows9) std::memcpy(rows9, R.rows9.data(), sizeof(double) * 9 * (size_t)n);
  if (corners16) {
    if (R.corners.size() < 16 * (size_t)R.n_valid) { set_err("corners not retained: call cs_batch_set_debug(b,1) before cs_batch_run"); return CS_ERR_NOT_RUN; }
    std::memcpy(corners16, R.corners.data(), sizeof(double) * 16 * (size_t)n);
  }
  return R.n_valid;
}

int cs_batch_debug_kept(cs_batch* b, int frame, int box, int k, int cap, int* keep_ids, double* scores) {
  if (!b || frame < 0 || frame >= b->n_frames || box < 0 || box >= b->max_boxes || k < 0 || k > 2) return CS_ERR_INVALID_ARG;
  if (!b->ran) return CS_ERR_NOT_RUN;
  int ri = b->job_index[((size_t)frame * b->max_boxes + box) * 3 + k];
  if (ri < 0) return 0;
  const JobResult& R = b->results[ri];
  int n = std::min(cap, (int)R.keep.size());
  if (keep_ids) std::memcpy(keep_ids, R.keep.data(), sizeof(int) * (size_t)n);
  if (scores) std::memcpy(scores, R.score.data(), sizeof(double) * (size_t)n);
  return (int)R.keep.size();
}

// ------------------------------------------------------------------ distance-map front end -----
int cs_bgr_to_gray(const unsigned char* bgr, int n_pixels, unsigned char* gray) {   // cvtColor(BGR2GRAY), box_proposal_detail.cpp:84
  if (n_pixels < 0 || (n_pixels && (!bgr || !gray))) return CS_ERR_INVALID_ARG;
  for (int i = 0; i < n_pixels; i++) gray[i] = (unsigned char)((bgr[3 * i] * 1868 + bgr[3 * i + 1] * 9617 + bgr[3 * i + 2] * 4899 + (1 << 13)) >> 14);
  return CS_OK;
}

int cs_edge_distance_maps_multi(cs_detector* d, const unsigned char* const* grays, int n_images, int img_w, int img_h, const cs_roi* rois, const int* roi_image,
                                int n_rois, float* const* out_maps, double* kernel_ms) {
  if (!d || n_images <= 0 || !grays || img_w <= 0 || img_h <= 0 || n_rois < 0 || (n_rois && (!rois || !roi_image))) return CS_ERR_INVALID_ARG;
  CS_GUARD_BEGIN
  HIP_TRY(hipSetDevice(d->device));
  if (kernel_ms) *kernel_ms = 0;
  if (n_rois == 0) return CS_OK;
  const size_t img_px = (size_t)img_w * img_h;
  std::vector<cs::EdgeRoi> er(n_rois);
  long long tot = 0, max_px = 1;
  int max_w = 1;
  for (int k = 0; k < n_rois; k++) {
    const cs_roi& r = rois[k];
    if (r.width <= 0 || r.height <= 0 || r.left < 0 || r.top < 0 || r.left + r.width > img_w || r.top + r.height > img_h || roi_image[k] < 0 || roi_image[k] >= n_images ||
        (out_maps && !out_maps[k])) {
      set_err("cs_edge_distance_maps: ROI outside the image");
      return CS_ERR_INVALID_ARG;
    }
    er[k] = cs::EdgeRoi{r.left, r.top, r.width, r.height, (long long)(roi_image[k] * img_px), tot, tot};
    tot += (long long)r.width * r.height;
    max_w = std::max(max_w, r.width); max_px = std::max(max_px, 4LL * ((r.width + 5) / 4) * (r.height + 2));
  }
  DevBuf<unsigned char> d_gray, d_cls;
  DevBuf<cs::EdgeRoi> d_rois;
  DevBuf<float> d_map;
  int rc;
  if ((rc = d_gray.ensure(img_px * n_images)) || (rc = d_cls.ensure((size_t)tot + 8)) || (rc = d_rois.ensure((size_t)n_rois)) || (rc = d_map.ensure((size_t)tot))) return rc;
  hipStream_t st = d->stream;
  for (int i = 0; i < n_images; i++) {
    if (!grays[i]) { set_err("cs_edge_distance_maps: null image"); return CS_ERR_INVALID_ARG; }
    HIP_TRY(hipMemcpyAsync(d_gray.p + img_px * i, grays[i], img_px, hipMemcpyHostToDevice, st));
  }
  std::vector<cs::EdgeRoi> er_launch(er.begin(), er.end());      // (er keeps the caller's order for the copies back)
  edge_rois_largest_first(er_launch);
  HIP_TRY(hipMemcpyAsync(d_rois.p, er_launch.data(), sizeof(cs::EdgeRoi) * n_rois, hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(d->ev[0], st));
  // cv::Canny(gray_img(object_bbox), im_canny, 80, 200): the thresholds are literals of the reference (:324)
  cs::launch_edge_maps(d_gray.p, img_w, img_h, d_rois.p, n_rois, d_cls.p, d_map.p, max_w, max_px, 80, 200, st);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(d->ev[1], st));
  if (out_maps)
    for (int k = 0; k < n_rois; k++)
      HIP_TRY(hipMemcpyAsync(out_maps[k], d_map.p + er[k].map_off, sizeof(float) * (size_t)er[k].w * er[k].h, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (kernel_ms) { float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, d->ev[0], d->ev[1])); *kernel_ms = ms; }
  d_gray.release(); d_cls.release(); d_rois.release(); d_map.release();
  return CS_OK;
  CS_GUARD_END("cs_edge_distance_maps")
}

int cs_edge_distance_maps(cs_detector* d, const unsigned char* gray, int img_w, int img_h, const cs_roi* rois, int n_rois, float* const* out_maps) {
  if (!gray || (n_rois > 0 && !out_maps)) return CS_ERR_INVALID_ARG;
  std::vector<int> zero(std::max(n_rois, 1), 0);
  return cs_edge_distance_maps_multi(d, &gray, 1, img_w, img_h, rois, zero.data(), n_rois, out_maps, nullptr);
}

// One frame through the detector's resident single-frame batch (gray == nullptr: the frame's dist_maps are uploaded; else they are
// produced on the device from the gray image, in place in the map pool).
static int detect_single(cs_detector* d, const cs_frame_desc* frame, const unsigned char* gray, cs_cuboid* out, int* out_counts) {
  if (!d || !frame || !out || !out_counts) return CS_ERR_INVALID_ARG;
  CS_GUARD_BEGIN
  std::lock_guard<std::mutex> lk(d->single_mu);
  HIP_TRY(hipSetDevice(d->device));
  if (!d->single) { d->single = new cs_batch(); d->single->det = d; }
  int rc = batch_fill(d, d->single, frame, gray ? &gray : nullptr, 1);
  if (rc) return rc;
  return cs_batch_run(d, d->single, out, out_counts);
  CS_GUARD_END("cs_detect_cuboids")
}
// image in, cuboids out: the frame's dist_maps are ignored and computed from the gray image on the device
int cs_detect_cuboids_gray(cs_detector* d, const cs_frame_desc* frame, const unsigned char* gray, cs_cuboid* out, int* out_counts) {
  if (!gray) return CS_ERR_INVALID_ARG;
  return detect_single(d, frame, gray, out, out_counts);
}

int cs_detect_cuboids(cs_detector* d, const cs_frame_desc* frame, cs_cuboid* out, int* out_counts) {
  return detect_single(d, frame, nullptr, out, out_counts);
}

}  // extern "C"
