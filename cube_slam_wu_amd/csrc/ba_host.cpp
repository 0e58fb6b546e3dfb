// ba_host.cpp -- host side of path B (g2o bundle adjustment) behind the C ABI of include/cubeslam_hip.h.
//
// Mirrors, step for step, what g2o's BlockSolver + OptimizationAlgorithmLevenberg do
// (object_slam/Thirdparty/g2o/g2o/core/block_solver.hpp, optimization_algorithm_levenberg.cpp): the
// structure phase (index mapping, edge orderings, Schur pattern) runs once on the host when the edges are
// set; every numeric step is a HIP kernel on resident data; the LM control flow (lambda schedule, accept /
// reject, termination) is scalar host code fed by one chi2 read-back per trial.  The reduced camera/cuboid
// system is dense and factorised with rocSOLVER potrf/potrs (the reference's LinearSolverDense uses a dense
// Eigen::LDLT, solvers/linear_solver_dense.h:104-111).  There is no CPU fallback.
#include <hip/hip_runtime.h>
#include <rocsolver/rocsolver.h>

#include "ba_sparse.h"
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <string>
#include <mutex>
#include <thread>
#include <map>
#include <vector>

#include "../../include/cubeslam_hip.h"
#include "ba_types.h"

namespace cs {
int ba_chi2_blocks(int n_proj);
void ba_launch_chi2(const BaView& v, int nb_proj, hipStream_t st);
void ba_launch_fill_rows4(double* dst, const double* rec4, long long rows, hipStream_t st);
void ba_launch_linearize(const BaView& v, hipStream_t st, hipStream_t st2, hipEvent_t ev_fork, hipEvent_t ev_join, hipStream_t st3, hipEvent_t ev_join3, hipEvent_t ev_pre = nullptr);
void ba_launch_reduce(const BaView& v, const double* lambda_dev, hipStream_t st, hipStream_t st2, hipEvent_t ev_fork, hipEvent_t ev_join, const BaSidePrologue* sp = nullptr);
void ba_launch_gather_rows(const double* src, const int* idx, int n, int width, double* dst, hipStream_t st);
void ba_launch_backsub(const BaView& v, hipStream_t st);
int ba_scale_blocks();
void ba_launch_scale(const BaView& v, const double* lambda_dev, double* partial, hipStream_t st);
void ba_launch_update(const BaView& v, hipStream_t st, double* bak_cams = nullptr, double* bak_points = nullptr, double* bak_cubes = nullptr);
void ba_launch_scale_update(const BaView& v, const double* lambda_dev, double* partial, hipStream_t st, double* bak_cams, double* bak_points, double* bak_cubes);
void ba_launch_band_cholesky(double* Sb, double* work, int n, int LD, double* rhs, int* info, bool solve, hipStream_t st, bool one_sided = false);
void ba_launch_sep_reduce(const double* S, int LD, const double* Linv, int ci, int ni, int zl, int wl, int zr, int wr, double* Y, const double* rhs, double* msg, int wm, hipStream_t st);
void ba_launch_sep_assemble(const double* msgs, size_t msg_doubles, int wm, int R, const int* sep_off, int n, int LDs, double* Ssep, double* rsep, hipStream_t st);
void ba_launch_sep_scatter(const double* xsep, int n, int R, const int* sep_off, const int* sep_col, double* x, hipStream_t st);
void ba_launch_sep_backsolve(double* S, int LD, double* work, int ci, int ni, int zl, int wl, int zr, int wr, double* Y, double* rhs, int* info, hipStream_t st);
void ba_launch_fail_flag(const int* a, const int* b, const int* c, double* out, hipStream_t st);
size_t ba_band_workspace_doubles(int n, int LD);
int ba_band_team(int LD, int* rw_out);
bool ba_band_fits_device(int n, int LD, bool one_sided = false);
bool ba_bcr_ok(int n, int LD);
double ba_bcr_estimate_ms(int n);
size_t ba_bcr_workspace_doubles(int n, int Bv);
void ba_launch_bcr(const double* Sb, double* work, int n, int LD, int Bv, double* rhs, int* info, hipStream_t st);
bool ba_bcr_sep_ok(int wm, int R);
size_t ba_bcr_sep_workspace_doubles(int R);
void ba_launch_bcr_sep(const double* msgs, size_t msg_doubles, int wm, int R, const int* sep_off, const int* sep_col, int ns, double* work, double* x, int* info, hipStream_t st);
void ba_launch_sum2(const double* a, int na, const double* b, int nb, double* out, hipStream_t st);
void ba_launch_multi_zero(const std::pair<void*, size_t>* list, int n, hipStream_t st);
void ba_launch_multi_copy(const BaCopyItem* list, int n, hipStream_t st);
void ba_launch_sum2_flag(const double* a, int na, const double* b, int nb, const int* f0, const int* f1, double* out, hipStream_t st, double* host_out = nullptr, double seq = 0.0);
void ba_launch_max_diag(const BaView& v, double* out, hipStream_t st);
void ba_launch_trial_prologue(double* d_lam, double lam0, double lam1, int* info24, int* elim_fail, double* S, size_t n_clear, hipStream_t st);
void ba_launch_ext_add(const BaView& v, const double* cam36, const double* cam6, const double* cub81, const double* cub9, const double* pt9, const double* pt3, hipStream_t st);
void ba_launch_ext_offdiag(const BaView& v, int n_groups, const int* gptr, const int* order, const int* e4, const double* Hij, hipStream_t st);
void ba_launch_scan_finite(const double* p, long long n, int* out, hipStream_t st);
void ba_launch_edge_chi(const BaView& v, double* out, hipStream_t st);
}  // namespace cs

extern "C" const char* cs_last_error(void);
void cs_set_error_ba(const std::string& s);

namespace {

#define BA_TRY(expr)                                                           \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      cs_set_error_ba(std::string(#expr) + ": " + hipGetErrorString(_e));      \
      return CS_ERR_HIP;                                                       \
    }                                                                          \
  } while (0)
// a phase mark on the handle's stream -- only while the stage split is asked for (cs_ba_set_stage_timing)
#define BA_MARK(B, e) do { if ((B)->stage_timing) BA_TRY(hipEventRecord((e), (B)->st)); } while (0)
#define BA_NCCL(expr)                                                          \
  do {                                                                         \
    ncclResult_t _r = (expr);                                                  \
    if (_r != ncclSuccess) {                                                   \
      cs_set_error_ba(std::string(#expr) + ": " + ncclGetErrorString(_r));     \
      return CS_ERR_HIP;                                                       \
    }                                                                          \
  } while (0)
#define BA_ROC(expr)                                                           \
  do {                                                                         \
    rocblas_status _s = (expr);                                                \
    if (_s != rocblas_status_success) {                                        \
      cs_set_error_ba(std::string(#expr) + ": rocblas status " + std::to_string((int)_s)); \
      return CS_ERR_HIP;                                                       \
    }                                                                          \
  } while (0)

// no C++ exception crosses the C boundary
#define BA_GUARD_BEGIN try {
#define BA_GUARD_END(fn_name)                                                                                                   \
  } catch (const std::bad_alloc&) { cs_set_error_ba(std::string(fn_name) + ": out of host memory"); return CS_ERR_CAPACITY; }   \
    catch (const std::exception& ex) { cs_set_error_ba(std::string(fn_name) + ": " + ex.what()); return CS_ERR_CAPACITY; }

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Pinned staging for the structure phase's many small uploads: a blocking hipMemcpy from pageable memory costs ~40 us whatever its
// size, and the phase makes a few dozen; through this arena they are queued on the handle's stream and waited for once.
struct StageArena {
  char* p = nullptr;
  size_t cap = 0, off = 0;
  void* take(size_t bytes) {          // nullptr: no room (the caller copies directly)
    const size_t at = (off + 63) & ~(size_t)63;
    if (!p || at + bytes > cap) return nullptr;
    off = at + bytes;
    return p + at;
  }
  int begin(size_t want) {            // at the start of a structure phase, the stream idle
    off = 0;
    if (p && cap >= want) return CS_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    BA_TRY(hipHostMalloc((void**)&p, want));
    cap = want;
    return CS_OK;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = off = 0; }
};

static int g_dbuf_reallocs = 0;     // (diagnostics: device (re)allocations, printed with the structure phase's clock under CS_BA_PROF)
template <class T>
struct DBuf {
  T* p = nullptr;
  size_t n = 0, cap = 0;       // entries in use / allocated (grow-only: a graph that is extended frame by frame re-runs the structure phase,
                               // and ~100 hipFree + hipMalloc pairs were a quarter of it at 200 cameras)
  int reserve(size_t count) {
    count = std::max<size_t>(1, count);
    if (p && count <= cap) return CS_OK;
    // (a buffer that has to grow belongs to a graph that is being extended: a quarter of head room, so that the next frames fit --
    // without it every appended frame re-allocated every buffer whose size follows the edges or the points, ~60 of them)
    const size_t want = p ? count + count / 4 + 64 : count;
    g_dbuf_reallocs++;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    BA_TRY(hipMalloc((void**)&p, want * sizeof(T)));
    cap = want;
    return CS_OK;
  }
  int upload_ptr(const T* h, size_t count) {   // from the caller's memory
    int rc = reserve(count); if (rc) return rc;
    n = count;
    if (n) BA_TRY(hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice));
    return CS_OK;
  }
  int append_ptr(const T* h, size_t count) {   // the old contents stay where they are on the device, `count` new entries follow
    if (!count) return CS_OK;
    if (!p || n + count > cap) {               // (with head room: the next frames append in place)
      const size_t want = std::max(n + count, cap + cap / 2 + 16);
      T* q = nullptr;
      BA_TRY(hipMalloc((void**)&q, want * sizeof(T)));
      if (n) BA_TRY(hipMemcpy(q, p, n * sizeof(T), hipMemcpyDeviceToDevice));
      if (p) (void)hipFree(p);
      p = q; cap = want;
    }
    BA_TRY(hipMemcpy(p + n, h, count * sizeof(T), hipMemcpyHostToDevice));
    n += count;
    return CS_OK;
  }
  // the same through a pinned arena, queued on st (a frame's appends: a dozen blocking copies from pageable memory were 0.3-0.5 ms); the
  // arena's contents must stay until st has run (the handle rewinds it after the next structure phase's wait)
  int append_ptr_staged(const T* h, size_t count, StageArena& stage, hipStream_t st) {
    if (!count) return CS_OK;
    const size_t bytes = count * sizeof(T);
    void* pin = (p && n + count <= cap && bytes <= (256u << 10)) ? stage.take(bytes) : nullptr;
    if (!pin) {                                   // growing (the old contents are copied: nothing may still be writing them) or no room
      BA_TRY(hipStreamSynchronize(st));
      return append_ptr(h, count);
    }
    std::memcpy(pin, h, bytes);
    BA_TRY(hipMemcpyAsync(p + n, pin, bytes, hipMemcpyHostToDevice, st));
    n += count;
    return CS_OK;
  }
  int upload(const std::vector<T>& h) { return upload_ptr(h.data(), h.size()); }
  int upload_staged(const std::vector<T>& h, StageArena& stage, hipStream_t st) { return upload_ptr_staged(h.data(), h.size(), stage, st); }
  int upload_ptr_staged(const T* h, size_t count, StageArena& stage, hipStream_t st) {   // up to 1 MB: through the pinned arena, queued on st (a blocking copy from pageable memory costs 50-100 us)
    const size_t bytes = count * sizeof(T);
    void* pin = (bytes && bytes <= (1u << 20)) ? stage.take(bytes) : nullptr;
    if (!pin) return upload_ptr(h, count);
    int rc = reserve(count); if (rc) return rc;
    n = count;
    std::memcpy(pin, h, bytes);
    BA_TRY(hipMemcpyAsync(p, pin, bytes, hipMemcpyHostToDevice, st));
    return CS_OK;
  }
  // like upload_ptr_staged, but the copy itself is left to the caller's batched launch (cs::ba_launch_multi_copy): destination, staged
  // source and size are appended to the list.  No room in the arena: a blocking copy now.
  // Tables of more than 64 KB keep their own hipMemcpyAsync (the copy engine moves a large table faster than a kernel reads it over the link:
  // at a million edges the all-kernel form cost the phase 1.5 ms).
  int upload_ptr_deferred(const T* h, size_t count, StageArena& stage, std::vector<cs::BaCopyItem>& list, hipStream_t st) {
    const size_t bytes = count * sizeof(T);
    if (bytes > (64u << 10)) return upload_ptr_staged(h, count, stage, st);
    void* pin = bytes ? stage.take(bytes) : nullptr;
    if (!pin) return upload_ptr(h, count);
    int rc = reserve(count); if (rc) return rc;
    n = count;
    std::memcpy(pin, h, bytes);
    list.push_back(cs::BaCopyItem{p, pin, bytes});
    return CS_OK;
  }
  int alloc(size_t count, hipStream_t st = nullptr) {   // zeroed; st: queued on that stream instead of a blocking call per buffer
    int rc = reserve(count); if (rc) return rc;
    n = count;
    if (st) BA_TRY(hipMemsetAsync(p, 0, std::max<size_t>(1, n) * sizeof(T), st));
    else BA_TRY(hipMemset(p, 0, std::max<size_t>(1, n) * sizeof(T)));
    return CS_OK;
  }
  // zeroed like alloc(), but the fill is left to the caller's batched launch (cs::ba_launch_multi_zero): ptr / bytes are appended to the list
  int alloc_deferred(size_t count, std::vector<std::pair<void*, size_t>>& zero_list) {
    int rc = reserve(count); if (rc) return rc;
    n = count;
    zero_list.emplace_back((void*)p, std::max<size_t>(1, n) * sizeof(T));
    return CS_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; cap = 0; }
};

// Host scratch of the structure phase: UNINITIALISED storage (std::vector<int>(n) zero-fills on the constructing thread, which takes the
// page faults of a fresh multi-megabyte array one by one; the worker threads that fill these arrays then fault their own ranges side by side)
template <class T>
struct UBuf {
  std::unique_ptr<T[]> p;
  size_t n = 0, cap = 0;
  UBuf() {}
  explicit UBuf(size_t m) : p(new T[std::max<size_t>(1, m)]), n(m), cap(std::max<size_t>(1, m)) {}
  // a buffer kept on the handle between structure phases: the same (already faulted-in) memory again, grown with head room -- a fresh
  // 400 KB array is ~100 page faults, 0.1-0.2 ms on the thread that fills it; contents are NOT kept when it grows
  void ensure(size_t m) { if (m > cap || !p) { cap = m + m / 4 + 1024; p.reset(new T[cap]); } n = m; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  T* begin() { return p.get(); }
  const T* begin() const { return p.get(); }
  T* data() { return p.get(); }
  const T* data() const { return p.get(); }
  size_t size() const { return n; }
};

}  // namespace

struct cs_ba {
  int device = 0;
  hipStream_t st = nullptr;
  StageArena stage;                          // pinned staging of the structure phase's small uploads
  // the landmarks' camera lists of the last structure phase (host): a graph that is only appended to extends them instead of re-sorting all edges
  int st_lists_edges = 0, st_cur = 0; std::vector<int> st_cam_cnt; UBuf<int> st_cams_of[2], st_edge_of[2];   // (two sets: a phase reads the last one's and fills the other)
  UBuf<int> st_pm_pt, st_pm_cam, st_cm_pm, st_cm_pt;   // host scratch of the edge orders, kept for its memory only
  StageArena append_stage;                   // ... and of cs_ba_append_*'s rows (queued on st, rewound by the structure phase)
  hipStream_t st2 = nullptr;                 // side stream of the reduce phase (cuboid elimination beside the landmark segments)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipStream_t st3 = nullptr;                 // third stream of the linearisation: the landmark kernel beside the camera kernel (which fills a fraction of the CUs)
  hipEvent_t ev_join3 = nullptr;
  rocblas_handle blas = nullptr;
  hipEvent_t ev[8] = {};   // phase marks on the stream (linearise: 0-1; solve: 2 reduce 3 factor 4 back-substitution 5)
  hipEvent_t sev[4] = {};  // separator mode, inside [3, 4]: interior factorised, separator message formed, messages gathered, separator system solved
  double sep_ms[5] = {0, 0, 0, 0, 0};   // accumulated: interior factorisation, message (Y, T, t), gather, separator solve, interior back-substitution
  bool lin_pending = false;  // ev[0..1] recorded but not yet read
  int* h_status = nullptr;   // pinned: [status of the (interior's) factorisation, a cuboid block failed, status of the separator system's factorisation]
  // sharded BA over RCCL (cs_ba_comm_init): the collectives are issued from here, on this handle's stream
  ncclComm_t comm = nullptr;
  DBuf<double> d_scalars;    // [chi2, LM scale term] of a trial; lambda_0's diagonal on iteration 0
  // lambda of the current trial lives in device memory (d_lam: [lambda, lambda of the pose diagonals in LM's scale term]), copied from
  // the pinned h_lam at the head of a trial's launch sequence: the sequence itself never changes
  double* h_lam = nullptr;
  DBuf<double> d_lam;
  double* h_scalars = nullptr;   // pinned mirror
  // the trial's [chi2, scale term, failure flag, sequence number], written by the trial's last kernel itself (ba_sum2_flag_kernel) and polled by
  // cs_ba_optimize: pinned, device-visible
  double* h_trial = nullptr;
  double trial_seq = 0.0;
  // G2OBatchStatistics-like stage split (core/batch_stats.h:48-62): like g2o's setComputeBatchStatistics it is OFF unless asked for
  // (cs_ba_set_stage_timing) -- every phase mark is an event on the stream, ~6 us of dispatch gap each, eight per LM trial
  bool stage_timing = false;
  // OptimizationAlgorithmLevenberg's two properties (optimization_algorithm_levenberg.cpp:50-51, setters :191-199): cs_ba_set_lm_params
  double user_lambda_init = 0.0;      // "initialLambda": > 0 replaces tau * max |H_jj| (computeLambdaInit :168-169)
  int max_trials_after_failure = 10;  // "maxTrialsAfterFailure": trials of one iteration (:149-151)
  hipEvent_t ev_upd = nullptr;     // recorded behind a trial's update kernel when the next linearisation is speculated: its side streams fork there
  size_t scalars_cap = 0;
  // host copy of the problem description
  int nc = 0, no = 0, np = 0, cuboids_first = 0;
  std::vector<int> cam_fixed, cub_fixed, pt_fixed, cam_col, cub_col, pt_lm;  // pt_lm: landmark index among free points or -1
  std::vector<int> cam_col_ref, cub_col_ref;  // columns in g2o's sort-by-id order (inspection only); cam_col / cub_col are in solver (RCM) order
  int band_ld = 0;                            // 0 = dense reduced system
  bool use_bcr = false;
  int force_dense = 0;
  // sharded BA: landmarks (with all their projection edges) are dealt to ranks by the camera subsequence of their
  // first observation; cuboid / odometry edges follow their camera.  Every rank keeps all vertices.
  int shard_rank = 0, shard_n = 1;
  // Separator mode of the sharded solve (shard_n > 1, banded system, every rank's interior wide enough): rank r owns the columns
  // [cut[r], cut[r + 1]) of the band -- its first sepw[r] >= bandwidth columns are the separator Z_r (sepw[0] = 0), the rest its interior
  // -- and everything whose lowest column falls into that range (landmarks with their projection edges, cuboids, odometry edges).
  // What a rank builds then lies in its own columns and in the diagonal block of the next separator; it factorises its interior only,
  // the ranks exchange the separators' Schur complements (3 w^2 + 2 w doubles each) and the interiors' solutions.  Otherwise
  // (sep_mode == false) the whole [S | b] is summed over the ranks and every rank factorises it.
  bool sep_mode = false;
  std::vector<int> cut, sepw, sep_off, lm_owner;   // sep_off[k]: row of separator k in the separator system (k = 1 .. R - 1; sep_off[R] = n_sep)
  int n_sep = 0, w_max = 0;
  size_t msg_doubles = 0;                          // one rank's message: [LL | RL | RR | tL | tR]
  int int_c = 0, int_n = 0, zl = 0, wl = 0, zr = 0, wr = 0;   // this rank's interior and its two separators
  DBuf<double> sepY, sep_msgs, sepS, int_work, sep_work;
  DBuf<int> d_sep_off, d_sep_col, d_int_info, d_sep_info;
  long long bytes_per_trial = 0, bytes_per_trial_allreduce = 0;   // payload this rank contributes to the collectives of one LM trial; what the all-reduce of [S | b] would be
  size_t s_doubles = 0;                       // size of S; rhs follows it in the same allocation (one all-reduce)
  int n_pose = 0, n_lm = 0;
  int n_red = 0;            // dimension of the system the solver factorises: n_pose, or the cameras' part when the cuboids are eliminated too
  bool elim = false;        // free cuboids eliminated like landmarks (single rank, fused Schur schedule)
  int elim_max_slots = 1;   // observing cameras of the widest free cuboid
  // general sparse Cholesky of the reduced system (ba_sparse.h): graphs the ordering cannot band
  bool sparse = false;
  bool sp_S_clean = false;     // S holds nothing outside the plan's pattern (set by the first trial's full clear)
  cs::SparsePlan sp_plan;
  cs::SparseGrids sp_grids{0, 0};   // launch grids of the sparse factorisation / substitution, decided with the plan
  DBuf<int> sp_ndim, sp_ncol, sp_sptr, sp_srow, sp_sroff, sp_prow, sp_rbase, sp_rent, sp_rptr, sp_rcol, sp_rpos, sp_order, sp_info;
  DBuf<long long> sp_poff;
  DBuf<double> sp_L, sp_xs, sp_T;
  DBuf<int> sp_tcol;
  DBuf<unsigned> sp_done, sp_xdone;
  DBuf<int> d_cub_mine;     // sharded + eliminated cuboids: 1 = this rank owns the cuboid (holds all its edges)
  DBuf<int> d_cubS_ptr, d_cubS_cam, d_ce_slot, d_cub_tile, d_cub_coef, d_elim_fail, d_slotE_ptr, d_slotE_idx;
  DBuf<double> cub_M, cub_Dinv;
  int n_proj = 0, n_cub = 0, n_odom = 0;   // n_cub = EdgeSE3Cuboid + EdgeSE3CuboidProj edges (the combined list ce_cam / ce_cub)
  int n_cub3 = 0;                          // of which EdgeSE3Cuboid (they come first)
  std::vector<int> u3_cam, u3_cub, up_cam, up_cub;      // the caller's two lists
  std::vector<double> h_pe_meas, h_pe_info, h_pe_K;
  DBuf<double> pe_meas, pe_info, pe_K;
  std::vector<int> e_pt, e_cam;      // projection edges, caller order
  DBuf<int> d_src;                   // device copy of slot_src (the gathers of the structure phase)
  std::unique_ptr<int[]> slot_src;   // point-major slot -> caller edge (uninitialised storage with head room: kept across structure phases)
  size_t slot_src_cap = 0; int slot_src_n = 0;
  std::vector<int> ce_cam, ce_cub, oe_i, oe_j;
  bool structure_dirty = true;
  // device buffers
  DBuf<double> cams, points, cubes, cams_bak, points_bak, cubes_bak;
  DBuf<int> d_cam_col, d_cub_col, d_pt_free;
  DBuf<int> pm_pt, pm_cam, pt_ptr, cm_pm, cm_pt, cam_ptr;
  DBuf<double> pm_uv, pm_info, pm_intr, pm_huber, cm_uv, cm_info, cm_intr, cm_huber;
  DBuf<int> d_ce_cam, d_ce_cub, d_oe_i, d_oe_j, d_ce_active, d_oe_active;
  DBuf<double> ce_meas, ce_info, ce_Hcc, ce_Hoo, ce_Hco, ce_bc, ce_bo, oe_meas, oe_info, oe_Hii, oe_Hjj, oe_Hij, oe_bi, oe_bj;
  DBuf<int> cam_ce_ptr, cam_ce_idx, cam_oei_ptr, cam_oei_idx, cam_oej_ptr, cam_oej_idx, cub_ce_ptr, cub_ce_idx;
  DBuf<double> Hcam, bcam, Hcub, bcub, Hll, bl, W, WD, Dinv, dbl, S, rhs, xl, chi_partial, band_linv, scale_partial;
  DBuf<int> pair_ptr, pair_i1, pair_i2, ent_a, ent_b;
  // fused Schur schedule (BaView::fused)
  bool fused = false;
  int n_seg = 0, n_gpairs = 0, seg_class[5] = {0, 0, 0, 0, 0};
  DBuf<int> d_run_lm, d_seg_ptr, d_seg_k, d_seg_tile, d_seg_slot, d_gp_ptr, d_gp_i1, d_gp_i2, d_gtile, d_gcam_ptr, d_gslot;
  DBuf<int> d_run_e0, d_seg_cam;     // round 6: a run entry's first point-major edge (= pt_ptr[run_lm]) and a segment slot's camera (= pm_cam of the segment's first landmark): two dependent loads less at the head of ba_lin_schur_kernel
  DBuf<double> part_tiles, part_coef;
  DBuf<rocblas_int> d_info;
  DBuf<int> d_band_info;
  int n_pairs = 0, nb_chi = 1, n_chi_partials = 1;
  long long schur_entries = 0;
  // raw edge payloads kept until finalisation
  std::vector<double> h_ce_meas, h_ce_info, h_oe_meas, h_oe_info;
  // the projection edges' payload (88 bytes per edge) goes straight to the device when the edges are set, in the caller's order;
  // the structure phase permutes it there (ba_gather_rows_kernel)
  DBuf<double> raw_uv, raw_info, raw_intr, raw_huber;
  // every projection edge with the same information matrix / intrinsics (decided by comparing the caller's records when they are set or
  // appended): the kernels then read these 4 + 4 doubles instead of the per-edge records (BaView::info_u / intr_u).  CS_BA_UNIFORM=0: never.
  bool info_uniform = false, intr_uniform = false;
  // the per-edge records are not stored while every edge carries the handle's reference record (uni8): 64 bytes per edge that no kernel would
  // read -- a third of a C4 problem's upload; they are written out from the reference record the moment an appended edge brings another one
  bool raw_info_virtual = false, raw_intr_virtual = false;
  double uni8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  DBuf<double> d_uni;
  bool have_huber = false;
  // robust kernels (cs_ba_set_robust_kernels; cs_robust.h): kinds per edge in the caller's order, empty = none of the class has one.
  // The projection edges' deltas live in raw_huber (Huber is the fast path: kinds all 0 / 1 -> no kind array on the device).
  std::vector<int> rk_proj, rk_cub3, rk_cproj, rk_odom;
  std::vector<double> rd_cub3, rd_cproj, rd_odom;
  DBuf<int> d_pm_rk, d_cm_rk, d_ce_rk, d_oe_rk;
  DBuf<double> d_ce_rdelta, d_oe_rdelta;
  // external (host-evaluated) edges: the coupling pattern of the binary ones (structure), the terms of the current linearisation
  int ext_n = 0;
  std::vector<int> ext_e4;                       // (class_i, idx_i, class_j, idx_j) per edge; idx_j < 0: unary
  DBuf<int> d_ext_e4, d_ext_order, d_ext_gptr;   // (+ the binary ones grouped by destination pair: ba_ext_offdiag_kernel)
  int ext_groups = 0;
  DBuf<double> ext_cam36, ext_cam6, ext_cub81, ext_cub9, ext_pt9, ext_pt3, ext_Hij;
  std::vector<double> h_ext_Hij;                 // host copy (cs_ba_get_system's dense H_pp)
  bool ext_has_cam = false, ext_has_cub = false, ext_has_pt = false, ext_terms_set = false;
  double ext_chi2 = 0;
  cs_external_fn ext_fn = nullptr; void* ext_ctx = nullptr;
  // last solution / rhs on the host (for LM's scale term and for inspection)
  std::vector<double> h_b, h_x;
  bool have_system = false;
  cs_ba_timing tm{};
  cs::BaView view{};
};


namespace {

// The structure phase's threaded loops (seven or eight short bursts of <= 8 items): a process-wide set of seven parked threads instead of
// 8 x std::thread per burst (20-30 us each to create and join: over a millisecond of the phase at a million edges).  One user at a time;
// a second handle building its structure at the same moment (the sharded tests' rank threads) falls back to plain threads.
class StructPool {
 public:
  static StructPool& get() { static StructPool p; return p; }
  bool try_run(int n, const std::function<void(int)>& fn) {
    std::unique_lock<std::mutex> use(use_, std::try_to_lock);
    if (!use.owns_lock()) return false;
    if (th_.empty()) for (int i = 0; i < 7; i++) th_.emplace_back([this] { loop(); });
    fn_ = &fn; n_ = n; next_.store(0); pending_.store((int)th_.size());
    gen_.fetch_add(1);                                   // (publishes fn_ / n_ / next_ / pending_ to the workers)
    if (sleepers_.load() > 0) { std::lock_guard<std::mutex> lk(m_); cv_.notify_all(); }
    work();
    // every worker acknowledges the burst (fn must outlive their last look at it): the hot ones within a microsecond, a parked one after its wake-up
    for (int spins = 0; pending_.load(std::memory_order_acquire) != 0; spins++) { if (spins < 2000) cpu_relax(); else std::this_thread::yield(); }
    fn_ = nullptr;
    return true;
  }
  ~StructPool() {
    stop_.store(true); gen_.fetch_add(1);
    { std::lock_guard<std::mutex> lk(m_); cv_.notify_all(); }
    for (auto& t : th_) t.join();
  }
 private:
  static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  void work() { for (;;) { const int i = next_.fetch_add(1); if (i >= n_) break; (*fn_)(i); } }
  // A worker can stay hot for SPIN_US after a burst (CS_BA_POOL_SPIN_US; the structure phase is seven or eight bursts inside 1-2 ms, and
  // waking a parked thread is 20-40 us of the calling thread's wait each time) before it parks on the condition variable.  Default 0: on the
  // measurement boxes (a 16-CPU cgroup quota) seven spinning threads bought nothing -- 1.2 ms per appended frame at 200 cameras either way --
  // and their burnt quota came back as 5 ms throttling stalls in one frame out of ten.
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      const auto t0 = std::chrono::steady_clock::now();
      for (int spins = 0; gen_.load(std::memory_order_acquire) == seen; spins++) {
        cpu_relax();
        if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(SPIN_US)) {
          std::unique_lock<std::mutex> lk(m_);
          sleepers_.fetch_add(1);                        // (seq_cst against try_run's gen_ increment + sleepers_ read: one of the two sees the other)
          cv_.wait(lk, [&] { return gen_.load() != seen; });
          sleepers_.fetch_sub(1);
          break;
        }
      }
      seen = gen_.load(std::memory_order_acquire);
      if (stop_.load()) return;
      work();
      pending_.fetch_sub(1, std::memory_order_release);
    }
  }
  const int SPIN_US = [] { const char* e = getenv("CS_BA_POOL_SPIN_US"); return e ? atoi(e) : 0; }();
  std::vector<std::thread> th_;
  std::mutex use_, m_;
  std::condition_variable cv_;
  const std::function<void(int)>* fn_ = nullptr;
  std::atomic<int> next_{0}, pending_{0}, sleepers_{0};
  int n_ = 0;
  std::atomic<unsigned long long> gen_{0};
  std::atomic<bool> stop_{false};
};
// threads of a structure-phase burst: min(8, hardware threads), or fewer with CS_BA_STRUCT_THREADS (= 1: every loop on the calling thread --
// tests hold the threaded tables to the sequential ones, cs_ba_structure_digest)
inline int struct_threads() {
  static const int cap = [] { const char* e = getenv("CS_BA_STRUCT_THREADS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 8; }();
  return (int)std::max(1u, std::min((unsigned)cap, std::thread::hardware_concurrency()));
}
// items 0 .. n - 1 side by side (n <= 8 in the structure phase)
inline void run_items(int n, const std::function<void(int)>& fn) {
  if (n <= 1) { if (n == 1) fn(0); return; }
  if (StructPool::get().try_run(n, fn)) return;
  std::vector<std::thread> th;
  for (int t = 0; t < n; t++) th.emplace_back(fn, t);
  for (auto& t : th) t.join();
}

// std::sort on chunks in a few host threads, then pairwise merges (the structure phase's large sorts)
template <class It, class Cmp>
void parallel_sort(It b, It e, Cmp cmp, int nt) {
  const size_t n = (size_t)(e - b);
  if (nt <= 1 || n < 50000) { std::sort(b, e, cmp); return; }
  std::vector<size_t> cut(nt + 1);
  for (int t = 0; t <= nt; t++) cut[t] = n * t / nt;
  {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([&, t] { std::sort(b + cut[t], b + cut[t + 1], cmp); });
    for (auto& t : th) t.join();
  }
  for (int step = 1; step < nt; step *= 2) {
    std::vector<std::thread> th;
    for (int t = 0; t + step < nt; t += 2 * step)
      th.emplace_back([&, t] { std::inplace_merge(b + cut[t], b + cut[t + step], b + cut[std::min(nt, t + 2 * step)], cmp); });
    for (auto& t : th) t.join();
  }
}

// camera -> rank (contiguous subsequences), landmark -> rank of the subsequence of its lowest-index observing camera
inline int cam_rank(int cam, int n_cams, int n_ranks) { return (int)(((long long)cam * n_ranks) / std::max(1, n_cams)); }
// The per-landmark camera lists of the last structure phase (st_cams_of / st_edge_of, st_cam_cnt) stay valid only while the projection edges
// [0, st_lists_edges) and the landmark count they were built for are untouched -- cs_ba_append_* only adds behind them.  EVERY entry point that
// rewrites e_pt / e_cam or the vertex arrays calls this, so that the next phase checks every edge's indices again and rebuilds the lists.
inline void proj_edge_lists_invalidate(cs_ba* B) { B->st_lists_edges = 0; B->st_cam_cnt.clear(); }
void landmark_owners(int n_ranks, int n_cams, int n_points, int n_proj, const int* e_pt, const int* e_cam, std::vector<int>& owner) {
  std::vector<int> first(n_points, 0x7fffffff);
  for (int k = 0; k < n_proj; k++) if (e_pt[k] >= 0 && e_pt[k] < n_points) first[e_pt[k]] = std::min(first[e_pt[k]], e_cam[k]);
  owner.assign(n_points, 0);
  for (int p = 0; p < n_points; p++) owner[p] = (first[p] == 0x7fffffff) ? 0 : cam_rank(first[p], n_cams, n_ranks);
}

int finalize_structure(cs_ba* B) {
  if (!B->structure_dirty) return CS_OK;
  BA_TRY(hipSetDevice(B->device));
  BA_TRY(hipStreamSynchronize(B->st));             // nothing of an earlier call may still read the buffers (or the staging arena) reused below
  { const int rc0 = B->stage.begin((size_t)8 << 20); if (rc0) return rc0; }
  static const bool prof = getenv("CS_BA_PROF") != nullptr;   // diagnostics: host phase clock of the structure phase
  double t_ph = now_ms();
  auto mark = [&](const char* what) { if (prof) { double t = now_ms(); fprintf(stderr, "[ba structure] %-28s %8.2f ms   (%d device allocations so far)\n", what, t - t_ph, g_dbuf_reallocs); t_ph = t; } };
  const int nc = B->nc, no = B->no, np = B->np;
  // camera-cuboid edges of both kinds as one list: EdgeSE3Cuboid first, then EdgeSE3CuboidProj
  B->ce_cam = B->u3_cam; B->ce_cam.insert(B->ce_cam.end(), B->up_cam.begin(), B->up_cam.end());
  B->ce_cub = B->u3_cub; B->ce_cub.insert(B->ce_cub.end(), B->up_cub.begin(), B->up_cub.end());
  B->n_cub3 = (int)B->u3_cam.size();
  B->n_cub = (int)B->ce_cam.size();
  // ---- index mapping (sparse_optimizer.cpp:166-190): non-marginalised vertices by id, then the points
  B->cam_col_ref.assign(nc, -1); B->cub_col_ref.assign(no, -1); B->pt_lm.assign(np, -1);
  {
    int col = 0;
    auto do_cams = [&]() { for (int i = 0; i < nc; i++) if (!B->cam_fixed[i]) { B->cam_col_ref[i] = col; col += 6; } };
    auto do_cubs = [&]() { for (int i = 0; i < no; i++) if (!B->cub_fixed[i]) { B->cub_col_ref[i] = col; col += 9; } };
    if (B->cuboids_first) { do_cubs(); do_cams(); } else { do_cams(); do_cubs(); }
    B->n_pose = col;
  }
  // ---- the free landmarks grouped by the set of cameras that see them.  cams_of: every landmark's cameras sorted by id;
  // gorder: free landmarks with >= 1 edge sorted by (number of cameras, camera list); run_first: where each distinct set starts
  // (edges that an earlier phase of this handle has seen were checked then, and append only raises nc / np)
  for (int k = std::min(B->st_lists_edges, B->n_proj); k < B->n_proj; k++)
    if (B->e_pt[k] < 0 || B->e_pt[k] >= np || B->e_cam[k] < 0 || B->e_cam[k] >= nc) { cs_set_error_ba("projection edge index out of range"); return CS_ERR_INVALID_ARG; }
  mark("  checks, index mapping");
  std::vector<int> cam_cnt(np + 1, 0), gorder, run_first;
  // edges grouped by landmark: camera (sorted by id) and caller edge index
  const int st_nb = 1 - B->st_cur;
  UBuf<int>& cams_of = B->st_cams_of[st_nb]; UBuf<int>& edge_of = B->st_edge_of[st_nb];
  cams_of.ensure((size_t)B->n_proj); edge_of.ensure((size_t)B->n_proj);
  {
    {   // a counting sort of the edges by landmark (the edge order inside a landmark stays the caller's), then every landmark's camera list
        // sorted, its edges with it.  Threaded in two levels: the edges are first dealt into landmark ranges (thread t walks its share of the
        // edges and appends to its own list per range), then range r's thread counts, fills and sorts from the lists of its range only --
        // every edge is read three times in all, every write (and the page faults of the two fresh arrays) is partitioned.  (First form: one
        // thread counted all edges, then every thread scanned ALL edges for those of its landmarks -- 5.9 ms at a million edges.)
      // A graph that was only APPENDED to since the last structure phase (cs_ba_append_*): the lists of that phase are kept on the handle, every
      // landmark's old list is copied, its new edges follow in the caller's order, and the same stable insertion sort runs over the landmarks
      // that got edges -- the same lists as the counting sort below gives (old edges precede new ones in the caller's order), in one linear
      // pass instead of three (200 cameras / 100 k edges, a frame of 500 edges appended: 0.28 -> 0.08 ms).
      const int E0 = B->st_lists_edges, np0 = (int)B->st_cam_cnt.size() - 1;
      const bool grown = E0 > 0 && np0 >= 0 && np0 <= np && E0 <= B->n_proj && (B->n_proj - E0) <= E0 / 4 && B->st_cams_of[B->st_cur].size() == (size_t)E0 && B->st_edge_of[B->st_cur].size() == (size_t)E0;
      const int NT = grown ? 0 : (B->n_proj > 20000 ? struct_threads() : 1);   // (200 cameras / 100 k edges: 0.67 -> 0.35 ms)
      auto sort_lists = [&](int p0, int p1) {
        for (int p = p0; p < p1; p++) {
          const int a0 = cam_cnt[p], a1 = cam_cnt[p + 1];
          for (int a = a0 + 1; a < a1; a++) {      // insertion sort (a handful of edges; stable: caller order among equal cameras)
            const int c = cams_of[a], e = edge_of[a];
            int q = a;
            while (q > a0 && cams_of[q - 1] > c) { cams_of[q] = cams_of[q - 1]; edge_of[q] = edge_of[q - 1]; q--; }
            cams_of[q] = c; edge_of[q] = e;
          }
        }
      };
      if (grown) {
        std::vector<int> add(np + 1, 0), used(np, 0);
        for (int k = E0; k < B->n_proj; k++) add[B->e_pt[k] + 1]++;
        for (int p = 0; p < np; p++) add[p + 1] += add[p];                          // new edges of the landmarks before p
        for (int p = 0; p <= np; p++) cam_cnt[p] = (p <= np0 ? B->st_cam_cnt[p] : E0) + add[p];
        const int* oc = B->st_cams_of[B->st_cur].data(); const int* oe = B->st_edge_of[B->st_cur].data();
        const int NTc = B->n_proj > 20000 ? struct_threads() : 1;                   // (the old lists are out of cache by now: the copy is a burst of misses)
        run_items(NTc, [&](int t) {
          const int p0 = (int)((long long)np0 * t / NTc), p1 = (int)((long long)np0 * (t + 1) / NTc);
          if (p1 <= p0) return;
          // (the landmarks of a range that got no edge in between move as one block: runs are copied whole)
          int p = p0;
          while (p < p1) {
            int q = p + 1;
            while (q < p1 && add[q] == add[p]) q++;                                 // landmarks p .. q - 1: the same shift (only the last may have new edges)
            const int a0 = B->st_cam_cnt[p], a1 = B->st_cam_cnt[q], d0 = cam_cnt[p];
            std::memcpy(&cams_of[d0], oc + a0, sizeof(int) * (size_t)(a1 - a0));
            std::memcpy(&edge_of[d0], oe + a0, sizeof(int) * (size_t)(a1 - a0));
            p = q;
          }
        });
        for (int k = E0; k < B->n_proj; k++) {
          const int p = B->e_pt[k], m_old = p < np0 ? B->st_cam_cnt[p + 1] - B->st_cam_cnt[p] : 0, q = cam_cnt[p] + m_old + used[p]++;
          cams_of[q] = B->e_cam[k]; edge_of[q] = k;
        }
        for (int k = E0; k < B->n_proj; k++) sort_lists(B->e_pt[k], B->e_pt[k] + 1);
      } else if (NT == 1) {
        for (int k = 0; k < B->n_proj; k++) cam_cnt[B->e_pt[k] + 1]++;
        for (int i = 0; i < np; i++) cam_cnt[i + 1] += cam_cnt[i];
        std::vector<int> fill(cam_cnt.begin(), cam_cnt.end() - 1);
        for (int k = 0; k < B->n_proj; k++) { const int q = fill[B->e_pt[k]]++; cams_of[q] = B->e_cam[k]; edge_of[q] = k; }
        sort_lists(0, np);
      } else {
        auto range_of = [&](int p) { return (int)(((long long)p * NT) / np); };                    // landmark -> range
        auto range_lo = [&](int r) { return (int)(((long long)r * np + NT - 1) / NT); };            // first landmark of range r (inverse of range_of)
        std::vector<std::vector<int>> dealt((size_t)NT * NT);                                      // [dealing thread][range]: edge indices, ascending
        run_items(NT, [&](int t) {
          const int k0 = (int)((long long)B->n_proj * t / NT), k1 = (int)((long long)B->n_proj * (t + 1) / NT);
          for (int r = 0; r < NT; r++) dealt[(size_t)t * NT + r].reserve((size_t)(k1 - k0) / NT + 64);
          for (int k = k0; k < k1; k++) dealt[(size_t)t * NT + range_of(B->e_pt[k])].push_back(k);
        });
        std::vector<int> range_edges(NT + 1, 0);       // where a range's edges start in the grouped arrays
        for (int r = 0; r < NT; r++) { size_t m = 0; for (int t = 0; t < NT; t++) m += dealt[(size_t)t * NT + r].size(); range_edges[r + 1] = range_edges[r] + (int)m; }
        run_items(NT, [&](int r) {       // (cam_cnt[p + 1] of the range's landmarks is this thread's alone: count, then end offset)
          const int p0 = range_lo(r), p1 = r + 1 < NT ? range_lo(r + 1) : np;
          for (int t = 0; t < NT; t++) for (int k : dealt[(size_t)t * NT + r]) cam_cnt[B->e_pt[k] + 1]++;
          std::vector<int> fill((size_t)(p1 - p0));
          int at = range_edges[r];
          for (int p = p0; p < p1; p++) { fill[p - p0] = at; at += cam_cnt[p + 1]; cam_cnt[p + 1] = at; }
          for (int t = 0; t < NT; t++) for (int k : dealt[(size_t)t * NT + r]) { const int q = fill[B->e_pt[k] - p0]++; cams_of[q] = B->e_cam[k]; edge_of[q] = k; }
          // (the lists' bounds: cam_cnt[p0] belongs to the range before -- its value is range_edges[r])
          for (int p = p0; p < p1; p++) {
            const int a0 = p == p0 ? range_edges[r] : cam_cnt[p], a1 = cam_cnt[p + 1];
            for (int a = a0 + 1; a < a1; a++) {
              const int c = cams_of[a], e = edge_of[a];
              int q = a;
              while (q > a0 && cams_of[q - 1] > c) { cams_of[q] = cams_of[q - 1]; edge_of[q] = edge_of[q - 1]; q--; }
              cams_of[q] = c; edge_of[q] = e;
            }
          }
        });
      }
    }
    mark("  camera lists (count, fill, sort)");
    for (int p = 0; p < np; p++) {
      for (int a = cam_cnt[p] + 1; a < cam_cnt[p + 1]; a++)
        if (cams_of[a] == cams_of[a - 1]) { cs_set_error_ba("two projection edges between the same point and camera"); return CS_ERR_INVALID_ARG; }
      if (!B->pt_fixed[p] && cam_cnt[p + 1] > cam_cnt[p]) gorder.push_back(p);
    }
    mark("  duplicate check");
    // Group the landmarks by camera set.  A KITTI-shaped problem has ~70x fewer distinct sets than landmarks (2 862 for 200 k at C4), so
    // instead of sorting the landmarks (9.6 ms of comparisons at C4) every thread hashes its range of landmarks into its own table of
    // sets, the tables are merged, the few DISTINCT sets are sorted by (number of cameras, camera list), and the landmarks are laid
    // out group by group -- in landmark order inside a group, because every thread walks its range in order and the ranges are laid
    // down in order.  Same gorder / run_first as the sort produced.
    {
      const int NTg = (gorder.size() > 5000) ? struct_threads() : 1;
      struct LGroup { int rep; std::vector<int> members; };
      std::vector<std::vector<LGroup>> local(NTg);
      auto set_hash = [&](int p) {
        unsigned long long h = 1469598103934665603ull;
        for (int a = cam_cnt[p]; a < cam_cnt[p + 1]; a++) { h ^= (unsigned)cams_of[a] + 0x9e3779b9u; h *= 1099511628211ull; }
        return h ^ (h >> 29);
      };
      auto same_set = [&](int p, int q) {
        const int kp = cam_cnt[p + 1] - cam_cnt[p];
        return kp == cam_cnt[q + 1] - cam_cnt[q] && std::equal(cams_of.begin() + cam_cnt[p], cams_of.begin() + cam_cnt[p + 1], cams_of.begin() + cam_cnt[q]);
      };
      auto group_range = [&](int t) {
        const size_t i0 = gorder.size() * t / NTg, i1 = gorder.size() * (t + 1) / NTg;
        std::vector<LGroup>& G = local[t];
        size_t cap = 1024;
        std::vector<int> table(cap, -1);
        for (size_t i = i0; i < i1; i++) {
          const int p = gorder[i];
          if ((G.size() + 1) * 2 > cap) {            // grow: re-insert the groups' representatives
            cap *= 4; table.assign(cap, -1);
            for (size_t g = 0; g < G.size(); g++) { size_t s = set_hash(G[g].rep) & (cap - 1); while (table[s] >= 0) s = (s + 1) & (cap - 1); table[s] = (int)g; }
          }
          size_t s = set_hash(p) & (cap - 1);
          while (table[s] >= 0 && !same_set(G[table[s]].rep, p)) s = (s + 1) & (cap - 1);
          if (table[s] < 0) { table[s] = (int)G.size(); G.push_back(LGroup{p, {}}); }
          G[table[s]].members.push_back(p);
        }
      };
      run_items(NTg, group_range);
      // merge: global groups = distinct sets over all threads
      struct GGroup { int rep; std::vector<std::pair<int, int>> parts; };     // (thread, local group), thread order = landmark order
      std::vector<GGroup> GG;
      {
        size_t cap = 4096;
        std::vector<int> table(cap, -1);
        for (int t = 0; t < NTg; t++)
          for (size_t g = 0; g < local[t].size(); g++) {
            const int p = local[t][g].rep;
            if ((GG.size() + 1) * 2 > cap) {
              cap *= 4; table.assign(cap, -1);
              for (size_t q = 0; q < GG.size(); q++) { size_t s = set_hash(GG[q].rep) & (cap - 1); while (table[s] >= 0) s = (s + 1) & (cap - 1); table[s] = (int)q; }
            }
            size_t s = set_hash(p) & (cap - 1);
            while (table[s] >= 0 && !same_set(GG[table[s]].rep, p)) s = (s + 1) & (cap - 1);
            if (table[s] < 0) { table[s] = (int)GG.size(); GG.push_back(GGroup{p, {}}); }
            GG[table[s]].parts.push_back({t, (int)g});
          }
      }
      std::vector<int> gidx(GG.size());
      for (size_t q = 0; q < GG.size(); q++) gidx[q] = (int)q;
      std::sort(gidx.begin(), gidx.end(), [&](int x, int y) {
        const int p = GG[x].rep, q = GG[y].rep;
        const int kp = cam_cnt[p + 1] - cam_cnt[p], kq = cam_cnt[q + 1] - cam_cnt[q];
        if (kp != kq) return kp < kq;
        return std::lexicographical_compare(cams_of.begin() + cam_cnt[p], cams_of.begin() + cam_cnt[p + 1], cams_of.begin() + cam_cnt[q], cams_of.begin() + cam_cnt[q + 1]);
      });
      std::vector<int> sorted;
      sorted.reserve(gorder.size());
      for (int q : gidx) {
        run_first.push_back((int)sorted.size());
        for (auto& pt : GG[q].parts) { const std::vector<int>& m = local[pt.first][pt.second].members; sorted.insert(sorted.end(), m.begin(), m.end()); }
      }
      run_first.push_back((int)sorted.size());
      gorder.swap(sorted);
    }
    mark("  group by camera set");
  }
  mark("camera sets");
  // ---- cuboid / odometry edge indices are used below: check them first
  for (int k = 0; k < B->n_cub; k++)
    if (B->ce_cam[k] < 0 || B->ce_cam[k] >= nc || B->ce_cub[k] < 0 || B->ce_cub[k] >= no) { cs_set_error_ba("cuboid edge index out of range"); return CS_ERR_INVALID_ARG; }
  for (int k = 0; k < B->n_odom; k++)
    if (B->oe_i[k] < 0 || B->oe_i[k] >= nc || B->oe_j[k] < 0 || B->oe_j[k] >= nc) { cs_set_error_ba("odometry edge index out of range"); return CS_ERR_INVALID_ARG; }
  // ---- solver ordering of the pose vertices: reverse Cuthill-McKee on the block graph of the reduced system
  // (camera-camera through shared landmarks and odometry edges, camera-cuboid through cuboid edges), so that S is
  // banded for trajectory-shaped graphs.  The ordering only permutes the linear system; g2o's order is kept for x/b
  // inspection (cam_col_ref).
  // Two candidate systems.  (a) g2o's: cameras and cuboids (only the points are marginalised).  (b) The cuboids eliminated as
  // well: a cuboid is coupled to the cameras that observe it and to nothing else, exactly like a landmark with a 9 x 9 block, so
  // S_cc -= H_co D_oo^-1 H_co^T is the same exact block elimination -- a different elimination order of one Cholesky
  // factorisation, not a different system -- and the reduced system shrinks to the cameras (C4: 10 494 -> 5 994 unknowns,
  // bandwidth 182 -> 119: a cuboid couples the ~20 consecutive cameras that see it, which the landmarks' band nearly contains).
  // The one with the cheaper banded factorisation (n bw^2) is taken; (b) needs the fused Schur schedule and at most
  // BA_ELIM_MAX_SLOTS cameras per cuboid; sharded, a cuboid's edges then all live on ONE rank (that of its lowest-index observing
  // camera) so that its block is complete where it is eliminated, and the cuboids' increments are summed over the ranks after the
  // back-substitution (zero on every rank but the owner).  CS_BA_KEEP_CUBOIDS=1 forces (a).
  B->cam_col.assign(nc, -1); B->cub_col.assign(no, -1);
  // external (host-evaluated) edges: indices, and what their pattern means for the choices below
  bool ext_binary_on_cuboid = false;
  for (int k = 0; k < B->ext_n; k++) {
    const int ci = B->ext_e4[4 * k], ii = B->ext_e4[4 * k + 1], cj = B->ext_e4[4 * k + 2], ij = B->ext_e4[4 * k + 3];
    auto bad = [&](int c, int i) { return c < 0 || c > 2 || i < 0 || i >= (c == 0 ? nc : c == 1 ? no : np); };
    if (bad(ci, ii) || (ij >= 0 && bad(cj, ij))) { cs_set_error_ba("external edge: vertex class / index out of range"); return CS_ERR_INVALID_ARG; }
    if (ij >= 0 && (ci == 2 || cj == 2)) { cs_set_error_ba("external edge: a binary edge may join cameras and cuboids only (a marginalised point takes unary terms only: its coupling to a pose would change the Schur structure)"); return CS_ERR_INVALID_ARG; }
    if (ij >= 0 && ci == cj && ii == ij) { cs_set_error_ba("external edge: both ends are the same vertex"); return CS_ERR_INVALID_ARG; }
    if (ij >= 0 && (ci == 1 || cj == 1)) ext_binary_on_cuboid = true;
  }
  if (B->shard_n > 1 && (B->ext_n > 0 || B->ext_fn || B->ext_terms_set)) { cs_set_error_ba("external (host-evaluated) edges are not supported on a sharded handle"); return CS_ERR_INVALID_ARG; }
  std::vector<std::vector<int>> cub_cams(no);     // free cameras observing a free cuboid, distinct, by camera id
  int max_slots = 0, n_free_cub = 0;
  for (int k = 0; k < B->n_cub; k++) if (!B->cub_fixed[B->ce_cub[k]] && !B->cam_fixed[B->ce_cam[k]]) cub_cams[B->ce_cub[k]].push_back(B->ce_cam[k]);
  for (int o = 0; o < no; o++) {
    std::sort(cub_cams[o].begin(), cub_cams[o].end());
    cub_cams[o].erase(std::unique(cub_cams[o].begin(), cub_cams[o].end()), cub_cams[o].end());
    max_slots = std::max(max_slots, (int)cub_cams[o].size());
    if (!B->cub_fixed[o]) n_free_cub++;
  }
  bool fused_ok = getenv("CS_BA_SCHUR_PAIRS") == nullptr;
  {
    // (tracks of more than BA_FUSED_KMAX views go through the segments' plain multiply-add kernel: meant for the tail of a map -- when a
    // quarter of the landmarks are such tracks the pair-major path is the faster build again, measured at C4's size with 20 views each)
    size_t n_long = 0;
    for (int p : gorder) {
      const int kk = cam_cnt[p + 1] - cam_cnt[p];
      if (kk > cs::BA_LONG_KMAX) { fused_ok = false; break; }
      if (kk > cs::BA_FUSED_KMAX) n_long++;
    }
    if (4 * n_long > gorder.size()) fused_ok = false;
  }
  mark("  cuboid cameras, track lengths");
  struct Ordering { std::vector<int> cam_col, cub_col; int n_red = 0, bw = 0; std::vector<std::vector<int>> adj; std::vector<int> free_ids; };
  auto make_ordering = [&](bool elim) -> Ordering {
    Ordering O;
    O.cam_col.assign(nc, -1); O.cub_col.assign(no, -1);
    const int NV = nc + no;
    std::vector<std::vector<int>> adj(NV);
    auto is_free = [&](int v) { return v < nc ? !B->cam_fixed[v] : (!elim && !B->cub_fixed[v - nc]); };
    // (up to 8 192 vertices the links are collected as bits -- a camera pair is linked by many camera sets and, cuboids eliminated, by
    // every cuboid both see: 0.2 M pushes + a sort and unique per vertex were 4 ms of this phase at 1 000 cameras -- and read out in order)
    const bool as_bits = NV <= 8192;
    const size_t Wd = as_bits ? (size_t)(NV + 63) / 64 : 0;
    std::vector<unsigned long long> bits(as_bits ? (size_t)NV * Wd : 0, 0ull);
    auto link = [&](int a, int b) {
      if (a == b || !is_free(a) || !is_free(b)) return;
      if (as_bits) { bits[(size_t)a * Wd + (b >> 6)] |= 1ull << (b & 63); bits[(size_t)b * Wd + (a >> 6)] |= 1ull << (a & 63); }
      else { adj[a].push_back(b); adj[b].push_back(a); }
    };
    // landmarks couple the cameras that see them: one clique per DISTINCT camera set (the landmarks were grouped by camera set
    // above; KITTI-shaped problems have ~100x fewer sets than landmarks)
    for (size_t r = 0; r + 1 < run_first.size(); r++) {
      const int p = gorder[run_first[r]];
      for (int a = cam_cnt[p]; a < cam_cnt[p + 1]; a++) for (int b = a + 1; b < cam_cnt[p + 1]; b++) link(cams_of[a], cams_of[b]);
    }
    for (int k = 0; k < B->n_odom; k++) link(B->oe_i[k], B->oe_j[k]);
    for (int k = 0; k < B->ext_n; k++) if (B->ext_e4[4 * k + 3] >= 0) link((B->ext_e4[4 * k] ? nc : 0) + B->ext_e4[4 * k + 1], (B->ext_e4[4 * k + 2] ? nc : 0) + B->ext_e4[4 * k + 3]);
    if (!elim) { for (int k = 0; k < B->n_cub; k++) link(B->ce_cam[k], nc + B->ce_cub[k]); }
    else for (int o = 0; o < no; o++) if (!B->cub_fixed[o]) for (size_t a = 0; a < cub_cams[o].size(); a++) for (size_t b = a + 1; b < cub_cams[o].size(); b++) link(cub_cams[o][a], cub_cams[o][b]);
    if (as_bits) {
      for (int v = 0; v < NV; v++) {
        const unsigned long long* r = bits.data() + (size_t)v * Wd;
        for (size_t w = 0; w < Wd; w++) for (unsigned long long m = r[w]; m; m &= m - 1) adj[v].push_back((int)(64 * w) + __builtin_ctzll(m));
      }
    } else {
      for (auto& a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
    }
    std::vector<int> order;  // Cuthill-McKee, component by component, starting from a minimum-degree vertex
    std::vector<char> seen(NV, 0);
    std::vector<int> by_deg;
    for (int v = 0; v < NV; v++) if (is_free(v)) by_deg.push_back(v);
    std::stable_sort(by_deg.begin(), by_deg.end(), [&](int a, int b) { return adj[a].size() < adj[b].size(); });
    for (int s0 : by_deg) {
      if (seen[s0]) continue;
      // pseudo-peripheral start: two BFS sweeps from the minimum-degree vertex of the component
      int start = s0;
      for (int sweep = 0; sweep < 2; sweep++) {
        std::vector<int> q{start}, dist(NV, -1);
        dist[start] = 0;
        for (size_t h = 0; h < q.size(); h++) for (int w : adj[q[h]]) if (dist[w] < 0) { dist[w] = dist[q[h]] + 1; q.push_back(w); }
        int far = q.back();
        for (int v : q) if (dist[v] == dist[far] && adj[v].size() < adj[far].size()) far = v;
        start = far;
      }
      size_t head = order.size();
      order.push_back(start); seen[start] = 1;
      for (; head < order.size(); head++) {
        std::vector<int> nb;
        for (int w : adj[order[head]]) if (!seen[w]) { seen[w] = 1; nb.push_back(w); }
        std::stable_sort(nb.begin(), nb.end(), [&](int a, int b) { return adj[a].size() < adj[b].size(); });
        order.insert(order.end(), nb.begin(), nb.end());
      }
    }
    std::reverse(order.begin(), order.end());
    int col = 0;
    for (int v : order) { if (v < nc) { O.cam_col[v] = col; col += 6; } else { O.cub_col[v - nc] = col; col += 9; } }
    O.n_red = col;
    if (elim) for (int o = 0; o < no; o++) if (!B->cub_fixed[o]) { O.cub_col[o] = col; col += 9; }   // increments of the eliminated cuboids: behind the reduced system's
    auto vcol = [&](int v) { return v < nc ? O.cam_col[v] : O.cub_col[v - nc]; };
    auto vdim = [&](int v) { return v < nc ? 6 : 9; };
    int bw = 0;
    for (int v : order) {
      bw = std::max(bw, vdim(v) - 1);
      for (int w : adj[v]) { int lo = std::min(vcol(v), vcol(w)); int hi = (vcol(v) > vcol(w)) ? vcol(v) + vdim(v) - 1 : vcol(w) + vdim(w) - 1; bw = std::max(bw, hi - lo); }
    }
    O.bw = bw;
    O.free_ids = order;          // (for the sparse plan: the block graph of the free vertices)
    O.adj = std::move(adj);
    return O;
  };
  // banded path only where the persistent kernel's team is guaranteed to be resident on THIS device (occupancy query x CUs:
  // 256 CUs admit bandwidths up to ~1900; a partitioned or smaller device proportionally less); dense rocSOLVER otherwise
  auto band_ok = [&](const Ordering& O) { return !B->force_dense && O.n_red > 128 && O.bw + 1 <= O.n_red / 2 && cs::ba_band_fits_device(O.n_red, O.bw + 1); };
  auto cost = [&](const Ordering& O) { return band_ok(O) ? (double)O.n_red * (O.bw + 1.0) * (O.bw + 1.0) : (double)O.n_red * O.n_red * O.n_red / 3.0; };
  {
    // (an external binary edge on a cuboid couples it to something besides its observing cameras: the cuboids then stay in the system)
    B->elim_max_slots = std::max(1, std::min(max_slots, (int)cs::BA_ELIM_MAX_SLOTS));
    const bool try_elim = fused_ok && n_free_cub > 0 && max_slots <= cs::BA_ELIM_MAX_SLOTS && getenv("CS_BA_KEEP_CUBOIDS") == nullptr && !ext_binary_on_cuboid;
    // (the two candidate orderings only read shared data: g2o's system on a second thread while this one orders the cameras-only system)
    Ordering keep_o, elim_o;
    if (try_elim && nc + no > 64) {
      run_items(2, [&](int t) { if (t == 0) elim_o = make_ordering(true); else keep_o = make_ordering(false); });
    } else {
      keep_o = make_ordering(false);
      if (try_elim) elim_o = make_ordering(true);
    }
    // (an ordering without unknowns -- every camera fixed, the driver's frame-0 graph -- is not a candidate: nothing would be
    // factorised and the cuboids' elimination / back-substitution hang off the reduced solve)
    mark("  two orderings");
    B->elim = try_elim && elim_o.n_red > 0 && cost(elim_o) < cost(keep_o);
    const Ordering& O = B->elim ? elim_o : keep_o;
    B->cam_col = O.cam_col; B->cub_col = O.cub_col; B->n_red = O.n_red;
    B->band_ld = band_ok(O) ? O.bw + 1 : 0;
    // A general sparse factorisation (fill-reducing order of the block graph, ba_sparse.h) where it is the fastest of the three.  The
    // estimates are fits to measurements on MI355X (tools/ba_mesh_quick.py): banded -- 1.15 n / 64 dependent steps of
    // (10 + 0.043 bw) us; dense rocSOLVER potrf -- n^3 / 3 at 11 Tflop/s + 1 ms; sparse -- 25 us per level of the elimination tree +
    // 1.5 ms per Gflop + the dense tail's n^3 / 3 at 3 Tflop/s (rocSOLVER at 1-2 k unknowns) + the dense assembly's extra cost (1 ms + the
    // n x n fill).  The plan (an O(N^2) minimum-degree sweep) is only built when the alternative costs more than 5 ms.  CS_BA_SPARSE=0 never, =1 whenever the plan fits.
    mark("  band fits device");
    B->sparse = false; B->sp_S_clean = false;
    {
      const char* e = getenv("CS_BA_SPARSE");
      const int mode = e ? atoi(e) : -1;
      const double nn = (double)O.n_red;
      double est_band = B->band_ld ? 1.15 * nn / 64.0 * (10.0 + 0.043 * (B->band_ld - 1)) * 1e-3 : 1e30;
      // (wide bands: the line above was fitted on bandwidths of 100-200; the 1 920-camera survey mesh -- 11 514 unknowns, bandwidth 1 121 -- takes the banded
      // kernels 14.2 ms where it says 12.0 (bench.py, ba.solver_paths: 17.2 ms per damped solve against 14.8 through the sparse path, ~3 ms of either the reduce))
      if (B->band_ld > 128) est_band *= 1.0 + 0.00018 * (B->band_ld - 128);
      const double est_dense = nn * nn * nn / 3.0 / 11e12 * 1e3 + 1.0;
      // block cyclic reduction where the band is narrow enough for 128-unknown blocks and its few levels beat the banded kernels' chain of steps
      B->use_bcr = false;
      if (B->band_ld && cs::ba_bcr_ok(O.n_red, B->band_ld)) {
        const double est_bcr = cs::ba_bcr_estimate_ms(O.n_red);
        const char* eb = getenv("CS_BAND_BCR");
        if (est_bcr < est_band || (eb && atoi(eb) == 2)) { B->use_bcr = true; est_band = est_bcr; }
      }
      const double est_other = std::min(est_band, est_dense);
      const bool consider = mode != 0 && !B->force_dense && B->shard_n == 1 && O.n_red >= 256 && (mode == 1 || est_other > 5.0);
      if (consider) {
        std::vector<int> dim(nc + no, 0), col(nc + no, 0);
        for (int v : O.free_ids) { dim[v] = v < nc ? 6 : 9; col[v] = v < nc ? O.cam_col[v] : O.cub_col[v - nc]; }
        cs::SparsePlan plan;
        cs::SparseGrids sp_grids{0, 0};
        const bool fits = cs::sparse_plan_build(O.adj, O.free_ids, dim, col, cs::sparse_max_panel_doubles(), 0.35, plan, getenv("CS_BA_SPARSE_NO_TAIL") ? 0 : (getenv("CS_BA_SPARSE_TAIL_MAX") ? atoi(getenv("CS_BA_SPARSE_TAIL_MAX")) : 9000)) && cs::sparse_grids(cs::sparse_max_panel_doubles(), plan.N, &sp_grids);
        const double nt = (double)plan.n_tail;
        const double est_sparse = 0.025 * plan.levels + 1.5 * plan.flops * 2e-9 + (nt > 0 ? nt * nt * nt / 3.0 / 3e12 * 1e3 + 1.0 : 0.0) + 1.0 + nn * nn * 8.0 / 2e12 * 1e3
                                  + 0.3;   // (the sparse solve is not deferrable: its trials take cs_ba_optimize's synchronising flow -- four more host round trips than the stream flow of the band)
        if (prof && fits) fprintf(stderr, "[ba structure] sparse plan: %d vertices, %d levels, %lld values (%.1f%% of the dense triangle), largest panel %d, %.2f Gflop + a dense tail of %d unknowns; estimates ms: sparse %.1f, band %.1f, dense %.1f\n",
                                  plan.N, plan.levels, plan.nvals, 100.0 * plan.nvals / (0.5 * nn * nn), plan.max_panel, plan.flops * 2e-9, plan.n_tail, est_sparse, est_band < 1e29 ? est_band : -1.0, est_dense);
        if (fits && (mode == 1 || est_sparse < est_other)) {
          B->sparse = true; B->band_ld = 0; B->use_bcr = false;
          B->sp_plan = std::move(plan); B->sp_grids = sp_grids;
        }
      }
    }
  }
  // ---- sharded: the column cut (separator mode) and who owns what
  mark("ordering (RCM)");
  B->sep_mode = false; B->cut.clear(); B->sepw.clear(); B->sep_off.clear(); B->n_sep = 0; B->w_max = 0; B->msg_doubles = 0;
  B->int_c = B->int_n = B->zl = B->wl = B->zr = B->wr = 0;
  const int R = B->shard_n;
  if (R > 1 && B->band_ld > 0 && getenv("CS_BA_SHARD_ALLREDUCE") == nullptr) {
    struct Blk { int col, dim; };
    std::vector<Blk> blks;
    for (int i = 0; i < nc; i++) if (B->cam_col[i] >= 0) blks.push_back(Blk{B->cam_col[i], 6});
    for (int i = 0; i < no; i++) if (B->cub_col[i] >= 0 && B->cub_col[i] < B->n_red) blks.push_back(Blk{B->cub_col[i], 9});
    std::sort(blks.begin(), blks.end(), [](const Blk& a, const Blk& b) { return a.col < b.col; });
    const int bw = B->band_ld - 1;
    std::vector<int> cut(R + 1, 0), sepw(R, 0);
    cut[R] = B->n_red;
    bool ok = true;
    for (int r = 1; r < R && ok; r++) {
      const long long target = (long long)B->n_red * r / R;
      size_t q = 0;
      while (q < blks.size() && blks[q].col < target) q++;
      if (q >= blks.size()) { ok = false; break; }
      cut[r] = blks[q].col;
      int w = 0;
      while (q < blks.size() && w < bw) { w += blks[q].dim; q++; }
      if (w < bw) { ok = false; break; }
      sepw[r] = w;
    }
    for (int r = 0; r < R && ok; r++) {
      const int ni = cut[r + 1] - cut[r] - sepw[r];
      if (ni < 129 || ni <= B->band_ld) ok = false;     // (also: interiors further apart than the bandwidth, the regime the band kernels are tested in)
    }
    if (ok) {   // the separator system (block tridiagonal, blocks <= w_max: a band of 2 w_max) goes through the same persistent kernels
      int wm = 0, nsep = 0;
      for (int k = 1; k < R; k++) { wm = std::max(wm, sepw[k]); nsep += sepw[k]; }
      // the interiors go through the ONE-SIDED kernel (grid = team + 1), whose residency depends on LD only: every rank decides alike
      int ni_max = 0;
      for (int r = 0; r < R; r++) ni_max = std::max(ni_max, cut[r + 1] - cut[r] - sepw[r]);
      if (!cs::ba_band_fits_device(nsep, 2 * wm) || !cs::ba_band_fits_device(ni_max, B->band_ld, true)) ok = false;
    }
    if (ok) {
      B->sep_mode = true; B->cut = cut; B->sepw = sepw;
      B->sep_off.assign(R + 1, 0);
      for (int k = 1; k < R; k++) { B->sep_off[k + 1] = B->sep_off[k] + sepw[k]; B->w_max = std::max(B->w_max, sepw[k]); }
      B->n_sep = B->sep_off[R];
      B->msg_doubles = 3 * (size_t)B->w_max * B->w_max + 2 * (size_t)B->w_max;
      const int r = B->shard_rank;
      B->zl = cut[r]; B->wl = sepw[r]; B->int_c = cut[r] + sepw[r]; B->int_n = cut[r + 1] - B->int_c;
      B->zr = cut[r + 1]; B->wr = (r + 1 < R) ? sepw[r + 1] : 0;
    }
  }
  auto rank_of_col = [&](int col) { int r = 0; while (r + 1 < R && col >= B->cut[r + 1]) r++; return r; };
  std::vector<int> owner;
  if (B->sep_mode) {   // a landmark goes to the rank that owns the lowest column among its free cameras (none: rank 0)
    owner.assign(np, 0);
    for (int p = 0; p < np; p++) {
      int lo = 0x7fffffff;
      for (int a = cam_cnt[p]; a < cam_cnt[p + 1]; a++) { const int c = B->cam_col[cams_of[a]]; if (c >= 0) lo = std::min(lo, c); }
      if (lo != 0x7fffffff) owner[p] = rank_of_col(lo);
    }
  } else if (B->shard_n > 1) {
    landmark_owners(B->shard_n, nc, np, B->n_proj, B->e_pt.data(), B->e_cam.data(), owner);
  } else {
    owner.assign(np, 0);
  }
  B->lm_owner = owner;
  int nl = 0;
  std::vector<int> pt_free(np);
  for (int i = 0; i < np; i++) { pt_free[i] = B->pt_fixed[i] ? 0 : 1; if (pt_free[i]) B->pt_lm[i] = nl++; }
  B->n_lm = nl;
  int rc;
  // (the ~80 small uploads of a structure phase likewise: staged in the pinned arena, copied by one kernel per group of up to 48 tables at
  // the end of the phase -- nothing launched inside the phase reads them.  CS_BA_UPLOAD_KERNEL=0: one hipMemcpyAsync per table, as before.)
  std::vector<cs::BaCopyItem> copy_list;
  const bool upload_kernel = [] { const char* e = getenv("CS_BA_UPLOAD_KERNEL"); return !e || atoi(e) != 0; }();
  // (a launch per 24 tables or 2 MB staged, so that a large graph's transfers run beside the host work that follows them)
  auto cflush = [&](bool all) -> int {
    size_t bytes = 0;
    for (const cs::BaCopyItem& c : copy_list) bytes += c.bytes;
    if (!all && copy_list.size() < 24 && bytes < ((size_t)2 << 20)) return CS_OK;
    for (size_t i = 0; i < copy_list.size(); i += 48) cs::ba_launch_multi_copy(copy_list.data() + i, (int)std::min<size_t>(48, copy_list.size() - i), B->st);
    copy_list.clear();
    BA_TRY(hipGetLastError());
    return CS_OK;
  };
#define UP(buf, vec) do { rc = upload_kernel ? (buf).upload_ptr_deferred((vec).data(), (vec).size(), B->stage, copy_list, B->st) : (buf).upload_staged(vec, B->stage, B->st); if (!rc) rc = cflush(false); if (rc) return rc; } while (0)
  // (the ~50 zero fills of a structure phase -- one hipMemsetAsync each, 4-6 us of host time apiece, a third of an appended frame's phase --
  // are collected and go out as one kernel per group of up to 48 buffers: ZFLUSH() where the order against other device work matters)
  std::vector<std::pair<void*, size_t>> zero_list;
  auto zflush = [&]() -> int {
    for (size_t i = 0; i < zero_list.size(); i += 48) cs::ba_launch_multi_zero(zero_list.data() + i, (int)std::min<size_t>(48, zero_list.size() - i), B->st);
    zero_list.clear();
    BA_TRY(hipGetLastError());
    return CS_OK;
  };
#define AL(buf, n) do { rc = (buf).alloc_deferred(n, zero_list); if (rc) return rc; } while (0)
#define ZFLUSH() do { rc = zflush(); if (rc) return rc; } while (0)
#define UPB(buf, ub) do { rc = upload_kernel ? (buf).upload_ptr_deferred((ub).data(), (ub).size(), B->stage, copy_list, B->st) : (buf).upload_ptr_staged((ub).data(), (ub).size(), B->stage, B->st); if (!rc) rc = cflush(false); if (rc) return rc; } while (0)
  UP(B->d_cam_col, B->cam_col); UP(B->d_cub_col, B->cub_col); UP(B->d_pt_free, pt_free);
  { std::vector<int> e4(B->ext_e4); if (e4.empty()) e4.assign(4, 0); UP(B->d_ext_e4, e4); }
  {  // the host-evaluated binary edges grouped by the unordered pair of vertices they join (stable: the caller's order inside a group)
    std::vector<int> order, gptr(1, 0);
    auto key = [&](int k) {
      const long long a = ((long long)B->ext_e4[4 * k] << 32) | (unsigned)B->ext_e4[4 * k + 1], b = ((long long)B->ext_e4[4 * k + 2] << 32) | (unsigned)B->ext_e4[4 * k + 3];
      return std::make_pair(std::min(a, b), std::max(a, b));
    };
    for (int k = 0; k < B->ext_n; k++) if (B->ext_e4[4 * k + 3] >= 0) order.push_back(k);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return key(x) < key(y); });
    for (size_t q = 0; q < order.size(); q++) if (q + 1 == order.size() || key(order[q]) != key(order[q + 1])) gptr.push_back((int)q + 1);
    B->ext_groups = (int)gptr.size() - 1;
    if (order.empty()) order.push_back(0);
    UP(B->d_ext_order, order); UP(B->d_ext_gptr, gptr);
  }
  // ---- projection edges: point-major order (sorted by pose column inside a point), camera-major copy.  The edges are already
  // grouped by landmark with their cameras sorted by id (cams_of / edge_of above); a rank's point-major table is its own landmarks'
  // groups, each re-ordered by (column, camera id) -- k <= a handful of entries -- in a few host threads on disjoint ranges.
  std::vector<int> pt_ptr(np + 1, 0);
  for (int p = 0; p < np; p++) pt_ptr[p + 1] = pt_ptr[p] + (owner[p] == B->shard_rank ? cam_cnt[p + 1] - cam_cnt[p] : 0);
  const int E = pt_ptr[np];   // local edges
  if (B->slot_src_cap < (size_t)E) { B->slot_src_cap = (size_t)E + (B->slot_src ? (size_t)E / 4 : 0) + 1; B->slot_src.reset(new int[B->slot_src_cap]); }
  B->slot_src_n = E;
  int* const src_of_slot = B->slot_src.get();
  UBuf<int>& pm_pt = B->st_pm_pt; UBuf<int>& pm_cam = B->st_pm_cam;
  pm_pt.ensure((size_t)E); pm_cam.ensure((size_t)E);
  const int NTH = (E > 20000) ? struct_threads() : 1;
  {
    auto build_points = [&](int p0, int p1) {
      for (int p = p0; p < p1; p++) {
        if (owner[p] != B->shard_rank) continue;
        const int a0 = cam_cnt[p], k = cam_cnt[p + 1] - a0, s0 = pt_ptr[p];
        for (int a = 0; a < k; a++) { pm_cam[s0 + a] = cams_of[a0 + a]; src_of_slot[s0 + a] = edge_of[a0 + a]; pm_pt[s0 + a] = p; }
        for (int a = 1; a < k; a++) {     // by (column, camera id): fixed cameras (column -1) first, by id; stable
          const int c = pm_cam[s0 + a], e = src_of_slot[s0 + a], col = B->cam_col[c];
          int q = a;
          while (q > 0 && (B->cam_col[pm_cam[s0 + q - 1]] > col || (B->cam_col[pm_cam[s0 + q - 1]] == col && pm_cam[s0 + q - 1] > c))) {
            pm_cam[s0 + q] = pm_cam[s0 + q - 1]; src_of_slot[s0 + q] = src_of_slot[s0 + q - 1]; q--;
          }
          pm_cam[s0 + q] = c; src_of_slot[s0 + q] = e;
        }
      }
    };
    run_items(NTH, [&](int t) { build_points((int)((long long)np * t / NTH), (int)((long long)np * (t + 1) / NTH)); });
  }
  mark("  point-major order");
  // camera-major: a stable counting sort of the point-major slots by camera, the slots cut into NTH ranges (per-range histograms)
  UBuf<int>& cm_pm = B->st_cm_pm; UBuf<int>& cm_pt = B->st_cm_pt;
  cm_pm.ensure((size_t)E); cm_pt.ensure((size_t)E);
  std::vector<int> cam_ptr(nc + 1, 0);
  {
    std::vector<std::vector<int>> hist(NTH, std::vector<int>(nc, 0));
    auto count = [&](int t) { for (int sl = (int)((long long)E * t / NTH), s1 = (int)((long long)E * (t + 1) / NTH); sl < s1; sl++) hist[t][pm_cam[sl]]++; };
    auto fillr = [&](int t) { std::vector<int>& h = hist[t]; for (int sl = (int)((long long)E * t / NTH), s1 = (int)((long long)E * (t + 1) / NTH); sl < s1; sl++) { const int q = h[pm_cam[sl]]++; cm_pm[q] = sl; cm_pt[q] = pm_pt[sl]; } };
    auto run_threads = [&](const std::function<void(int)>& fn) {
      run_items(NTH, fn);
    };
    run_threads(count);
    int run = 0;
    for (int c = 0; c < nc; c++) {
      cam_ptr[c] = run;
      for (int t = 0; t < NTH; t++) { const int v = hist[t][c]; hist[t][c] = run; run += v; }    // where range t starts writing camera c
    }
    cam_ptr[nc] = run;
    run_threads(fillr);
  }
  mark("  index arrays (point-major, camera-major)");
  // The edge tables go to the device from a second host thread while this one builds the Schur schedule (host work only): five
  // arrays of E ints through blocking copies (0.3-0.4 ms each at a million edges, from pageable memory), the zero-fills and the eight
  // gathers that put the caller's measurement rows into both edge orders.  Everything is queued on B->st; the phase's one wait is
  // at its end.  (The staging arena is this thread's: the helper copies directly.)
  int edge_rc = CS_OK;
  std::thread edge_th([&]() {
    edge_rc = [&]() -> int {
      BA_TRY(hipSetDevice(B->device));
      int r;
      const size_t Ez = (size_t)E;
#define ER(call) do { r = (call); if (r) return r; } while (0)
      ER(B->cm_pm.upload_ptr(cm_pm.data(), Ez)); ER(B->d_src.upload_ptr(src_of_slot, Ez));
      // (the information / intrinsics records in the two edge orders only when they differ between edges: BaView::info_u / intr_u otherwise)
      const bool per_info = !B->info_uniform, per_intr = !B->intr_uniform;
      ER(B->pm_uv.alloc(2 * Ez, B->st)); ER(B->pm_huber.alloc(Ez, B->st)); ER(B->cm_uv.alloc(2 * Ez, B->st)); ER(B->cm_huber.alloc(Ez, B->st));
      if (per_info) { ER(B->pm_info.alloc(4 * Ez, B->st)); ER(B->cm_info.alloc(4 * Ez, B->st)); }
      if (per_intr) { ER(B->pm_intr.alloc(4 * Ez, B->st)); ER(B->cm_intr.alloc(4 * Ez, B->st)); }
      cs::ba_launch_gather_rows(B->raw_uv.p, B->d_src.p, E, 2, B->pm_uv.p, B->st);
      if (per_info) cs::ba_launch_gather_rows(B->raw_info.p, B->d_src.p, E, 4, B->pm_info.p, B->st);
      if (per_intr) cs::ba_launch_gather_rows(B->raw_intr.p, B->d_src.p, E, 4, B->pm_intr.p, B->st);
      if (B->have_huber) cs::ba_launch_gather_rows(B->raw_huber.p, B->d_src.p, E, 1, B->pm_huber.p, B->st);   // (else: zeros from the allocation)
      cs::ba_launch_gather_rows(B->pm_uv.p, B->cm_pm.p, E, 2, B->cm_uv.p, B->st);
      if (per_info) cs::ba_launch_gather_rows(B->pm_info.p, B->cm_pm.p, E, 4, B->cm_info.p, B->st);
      if (per_intr) cs::ba_launch_gather_rows(B->pm_intr.p, B->cm_pm.p, E, 4, B->cm_intr.p, B->st);
      cs::ba_launch_gather_rows(B->pm_huber.p, B->cm_pm.p, E, 1, B->cm_huber.p, B->st);
      BA_TRY(hipGetLastError());
      ER(B->pm_pt.upload_ptr(pm_pt.data(), Ez)); ER(B->pm_cam.upload_ptr(pm_cam.data(), Ez)); ER(B->cm_pt.upload_ptr(cm_pt.data(), Ez));
#undef ER
      return CS_OK;
    }();
  });
  struct JoinEdge { std::thread& t; ~JoinEdge() { if (t.joinable()) t.join(); } } join_edge{edge_th};
  UP(B->pt_ptr, pt_ptr);
  {   // kernel kinds of the projection edges in both orders -- only if some edge has a kernel other than Huber
    bool generic = false;
    for (int kd : B->rk_proj) if (kd != cs::RK_NONE && kd != cs::RK_HUBER) { generic = true; break; }
    if (generic) {
      std::vector<int> pk(std::max(1, E), 0), ck(std::max(1, E), 0);
      for (int sl = 0; sl < E; sl++) pk[sl] = B->rk_proj[src_of_slot[sl]];
      for (int q = 0; q < E; q++) ck[q] = pk[cm_pm[q]];
      UP(B->d_pm_rk, pk); UP(B->d_cm_rk, ck);
    } else { B->d_pm_rk.release(); B->d_cm_rk.release(); }
  }
  UP(B->cam_ptr, cam_ptr);
  mark("edge orderings + upload");
  // ---- Schur pattern (block_solver.hpp:262-292).  Fused path: segments of landmarks with one camera set + the destination
  // schedule of their partial blocks (BaView::fused).  It needs every landmark to be seen by <= BA_LONG_KMAX cameras; otherwise
  // (or with CS_BA_SCHUR_PAIRS=1, diagnostics) the pair-major path: (landmark, i1 <= i2) entries grouped by camera pair.
  B->fused = fused_ok;        // (CS_BA_SCHUR_PAIRS unset, no track beyond BA_LONG_KMAX views, long tracks a minority: decided with the orderings above)
  for (int p : gorder) if (owner[p] == B->shard_rank && cam_cnt[p + 1] - cam_cnt[p] > cs::BA_LONG_KMAX) { B->fused = false; break; }
  B->n_seg = 0; B->n_gpairs = 0; B->seg_class[0] = B->seg_class[1] = B->seg_class[2] = B->seg_class[3] = B->seg_class[4] = 0;
  B->schur_entries = 0;
  for (int p : gorder) if (owner[p] == B->shard_rank) { const long long k = cam_cnt[p + 1] - cam_cnt[p]; B->schur_entries += k * (k + 1) / 2; }
  if (B->fused) {
    std::vector<int> run_lm, seg_ptr{0}, seg_k, seg_tile, seg_slot, run_e0, seg_cam;
    struct Dst { int a, b, id; };
    std::vector<Dst> dst;                      // (destination block = its two columns, partial block id), in creation order: ids grow
    std::vector<std::pair<int, int>> cdst;     // (camera, partial vector id)
    const long long NP = std::max(1, B->n_pose);
    int n_tiles = 0, n_slots = 0;
    // Two passes over the runs (camera sets), both threaded: what a run contributes to every table is a function of its size and its cameras
    // alone -- count, prefix over the runs, then every run writes its own ranges.  (Round 6; one thread appending to ten vectors was 2.5-3.3 ms
    // at C4.  Same tables, entry for entry: the no-GPU digest test holds the threaded build to the sequential one.)
    {
      const size_t R = run_first.size() > 0 ? run_first.size() - 1 : 0;
      struct RunCnt { int m, k, f, dps; };       // owned landmarks, cameras, free cameras, destination blocks per segment
      std::vector<RunCnt> rc_(R);
      std::vector<int> slot_cam_all;             // every run's cameras in slot order (by column, then id), run after run
      std::vector<size_t> sc_off(R + 1, 0);
      for (size_t r = 0; r < R; r++) { const int p0 = gorder[run_first[r]]; sc_off[r + 1] = sc_off[r] + (size_t)(cam_cnt[p0 + 1] - cam_cnt[p0]); }
      slot_cam_all.resize(sc_off[R]);
      const int NTs = R > 256 ? struct_threads() : 1;
      run_items(NTs, [&](int t) {
        for (size_t r = R * t / NTs, r1 = R * (t + 1) / NTs; r < r1; r++) {
          const int p0 = gorder[run_first[r]], k = cam_cnt[p0 + 1] - cam_cnt[p0];
          int* sc = slot_cam_all.data() + sc_off[r];
          std::copy(cams_of.begin() + cam_cnt[p0], cams_of.begin() + cam_cnt[p0 + 1], sc);
          std::sort(sc, sc + k, [&](int x, int y) { return B->cam_col[x] != B->cam_col[y] ? B->cam_col[x] < B->cam_col[y] : x < y; });
          int m = 0, f = 0, dps = 0;
          for (int q = run_first[r]; q < run_first[r + 1]; q++) m += owner[gorder[q]] == B->shard_rank ? 1 : 0;
          for (int a2 = 0; a2 < k; a2++) if (B->cam_col[sc[a2]] >= 0) { f++; dps += k - a2; }
          rc_[r] = RunCnt{m, k, f, dps};
        }
      });
      // where every run starts in every table
      std::vector<size_t> o_lm(R + 1, 0), o_seg(R + 1, 0), o_cam(R + 1, 0), o_dst(R + 1, 0), o_cd(R + 1, 0);
      std::vector<int> o_tile(R + 1, 0), o_slot(R + 1, 0);
      for (size_t r = 0; r < R; r++) {
        const int ns = (rc_[r].m + cs::BA_SEG_LM - 1) / cs::BA_SEG_LM, k = rc_[r].k;
        o_lm[r + 1] = o_lm[r] + rc_[r].m; o_seg[r + 1] = o_seg[r] + ns; o_cam[r + 1] = o_cam[r] + (size_t)ns * k;
        o_dst[r + 1] = o_dst[r] + (size_t)ns * rc_[r].dps; o_cd[r + 1] = o_cd[r] + (size_t)ns * rc_[r].f;
        o_tile[r + 1] = o_tile[r] + ns * (k * (k + 1) / 2); o_slot[r + 1] = o_slot[r] + ns * k;
      }
      run_lm.resize(o_lm[R]); run_e0.resize(o_lm[R]); seg_ptr.assign(o_seg[R] + 1, 0); seg_k.resize(o_seg[R]); seg_tile.resize(o_seg[R]); seg_slot.resize(o_seg[R]);
      seg_cam.resize(o_cam[R]);
      {   // (room for what the eliminated cuboids append below: without it their first push_back copies the whole table)
        size_t cub_dst = 0, cub_cd = 0;
        if (B->elim) for (int o = 0; o < no; o++) if (!B->cub_fixed[o]) { const size_t kc = cub_cams[o].size(); cub_dst += kc * (kc + 1) / 2; cub_cd += kc; }
        dst.reserve(o_dst[R] + cub_dst); cdst.reserve(o_cd[R] + cub_cd);
      }
      dst.resize(o_dst[R]); cdst.resize(o_cd[R]);
      run_items(NTs, [&](int t) {
        for (size_t r = R * t / NTs, r1 = R * (t + 1) / NTs; r < r1; r++) {
          const int k = rc_[r].k;
          const int* sc = slot_cam_all.data() + sc_off[r];
          size_t lm = o_lm[r], sg = o_seg[r], cm = o_cam[r], ds = o_dst[r], cd = o_cd[r];
          int tiles = o_tile[r], slots = o_slot[r], in_seg = 0;
          for (int q = run_first[r]; q < run_first[r + 1]; q++) {
            const int p = gorder[q];
            if (owner[p] != B->shard_rank) continue;
            if (in_seg == 0) {   // open a segment
              seg_k[sg] = k; seg_tile[sg] = tiles; seg_slot[sg] = slots;
              for (int a2 = 0; a2 < k; a2++) seg_cam[cm++] = sc[a2];      // (= pm_cam[pt_ptr[p] + a]: the point-major order of a landmark's edges is this slot order)
              for (int a2 = 0; a2 < k; a2++) {
                const int ca = B->cam_col[sc[a2]];
                if (ca < 0) continue;
                cdst[cd++] = {sc[a2], slots + a2};
                for (int b2 = a2; b2 < k; b2++) dst[ds++] = Dst{ca, B->cam_col[sc[b2]], tiles + a2 * k - a2 * (a2 - 1) / 2 + (b2 - a2)};
              }
              tiles += k * (k + 1) / 2; slots += k;
            }
            run_lm[lm] = p; run_e0[lm] = pt_ptr[p]; lm++;
            if (++in_seg == cs::BA_SEG_LM) { seg_ptr[++sg] = (int)lm; in_seg = 0; }
          }
          if (in_seg) seg_ptr[++sg] = (int)lm;
        }
      });
      n_tiles = R ? o_tile[R] : 0; n_slots = R ? o_slot[R] : 0;
    }
    mark("  segments of the runs");
    // the eliminated cuboids join the destination schedule: per free cuboid one slot per observing camera (by column), the upper
    // triangle of slot pairs as partial blocks, one partial vector per slot
    std::vector<int> cubS_ptr(no + 1, 0), cubS_cam, ce_slot(B->n_cub, -1), cub_tile(no, 0), cub_coef(no, 0);
    if (B->elim) {
      for (int o = 0; o < no; o++) {
        cubS_ptr[o] = (int)cubS_cam.size();
        if (B->cub_fixed[o]) continue;
        std::vector<int> sc = cub_cams[o];
        std::sort(sc.begin(), sc.end(), [&](int a, int b) { return B->cam_col[a] != B->cam_col[b] ? B->cam_col[a] < B->cam_col[b] : a < b; });
        const int k = (int)sc.size();
        cub_tile[o] = n_tiles; cub_coef[o] = n_slots;
        for (int a = 0; a < k; a++) {
          cubS_cam.push_back(sc[a]);
          cdst.push_back({sc[a], n_slots + a});
          for (int b = a; b < k; b++) dst.push_back(Dst{B->cam_col[sc[a]], B->cam_col[sc[b]], n_tiles + a * k - a * (a - 1) / 2 + (b - a)});
        }
        n_tiles += k * (k + 1) / 2; n_slots += k;
      }
      cubS_ptr[no] = (int)cubS_cam.size();
      for (int k = 0; k < B->n_cub; k++) {
        const int o = B->ce_cub[k], c = B->ce_cam[k];
        if (B->cub_fixed[o] || B->cam_fixed[c]) continue;
        for (int q = cubS_ptr[o]; q < cubS_ptr[o + 1]; q++) if (cubS_cam[q] == c) { ce_slot[k] = q; break; }
      }
    } else {
      for (int o = 0; o <= no; o++) cubS_ptr[o] = 0;
    }
    // edges of a slot, in edge order (a camera may hold an EdgeSE3Cuboid and an EdgeSE3CuboidProj to one cuboid)
    std::vector<int> slotE_ptr(cubS_cam.size() + 1, 0), slotE_idx;
    {
      for (int k = 0; k < B->n_cub; k++) if (ce_slot[k] >= 0) slotE_ptr[ce_slot[k] + 1]++;
      for (size_t q = 0; q < cubS_cam.size(); q++) slotE_ptr[q + 1] += slotE_ptr[q];
      slotE_idx.assign(std::max(1, slotE_ptr.back()), 0);
      std::vector<int> fill(slotE_ptr.begin(), slotE_ptr.end() - 1);
      for (int k = 0; k < B->n_cub; k++) if (ce_slot[k] >= 0) slotE_idx[fill[ce_slot[k]]++] = k;
    }
    UP(B->d_slotE_ptr, slotE_ptr); UP(B->d_slotE_idx, slotE_idx);
    if (cubS_cam.empty()) cubS_cam.push_back(0);
    if (ce_slot.empty()) ce_slot.push_back(-1);
    if (cub_tile.empty()) { cub_tile.push_back(0); cub_coef.push_back(0); }
    UP(B->d_cubS_ptr, cubS_ptr); UP(B->d_cubS_cam, cubS_cam); UP(B->d_ce_slot, ce_slot); UP(B->d_cub_tile, cub_tile); UP(B->d_cub_coef, cub_coef);
    AL(B->cub_M, 54 * cubS_cam.size()); AL(B->cub_Dinv, 81 * (size_t)std::max(1, no)); AL(B->d_elim_fail, 1); ZFLUSH();
    B->n_seg = (int)seg_k.size();
    for (int sgi = 0; sgi < B->n_seg; sgi++) {   // segments are sorted by k
      if (seg_k[sgi] <= 2) B->seg_class[0] = sgi + 1;
      if (seg_k[sgi] <= 5) B->seg_class[1] = sgi + 1;
      if (seg_k[sgi] <= 7) B->seg_class[2] = sgi + 1;
      if (seg_k[sgi] <= 10) B->seg_class[3] = sgi + 1;
      if (seg_k[sgi] <= cs::BA_FUSED_KMAX) B->seg_class[4] = sgi + 1;
    }
    mark("  cuboid slots");
    // by destination block, the partial blocks of one destination in creation (= segment) order: ids grow with creation, so (key, id) is the stable order
    // (two stable counting passes over the columns -- second column, then first -- instead of a comparison sort: 0.66 -> 0.06 ms at 200 cameras,
    // a third of this phase at 1 000)
    {
      std::vector<Dst> tmp(dst.size());
      std::vector<int> cnt((size_t)NP + 2);
      auto pass = [&](const std::vector<Dst>& in, std::vector<Dst>& out, bool by_b) {
        std::fill(cnt.begin(), cnt.end(), 0);
        for (const Dst& e : in) cnt[(by_b ? e.b : e.a) + 1]++;
        for (size_t c = 0; c + 1 < cnt.size(); c++) cnt[c + 1] += cnt[c];
        for (const Dst& e : in) out[cnt[by_b ? e.b : e.a]++] = e;
      };
      pass(dst, tmp, true);
      pass(tmp, dst, false);
    }
    mark("  destination sort");
    std::vector<int> gp_ptr, gp_i1, gp_i2, gtile(dst.size());
    for (size_t i = 0; i < dst.size(); i++) {
      if (i == 0 || dst[i].a != dst[i - 1].a || dst[i].b != dst[i - 1].b) { gp_ptr.push_back((int)i); gp_i1.push_back(dst[i].a); gp_i2.push_back(dst[i].b); }
      gtile[i] = dst[i].id;
    }
    gp_ptr.push_back((int)dst.size());
    B->n_gpairs = (int)gp_i1.size();
    std::vector<int> gcam_ptr(nc + 1, 0), gslot(cdst.size());
    for (auto& e : cdst) gcam_ptr[e.first + 1]++;
    for (int i = 0; i < nc; i++) gcam_ptr[i + 1] += gcam_ptr[i];
    { std::vector<int> fill(gcam_ptr.begin(), gcam_ptr.end() - 1); for (auto& e : cdst) gslot[fill[e.first]++] = e.second; }
    mark("  destination lists");
    UP(B->d_run_lm, run_lm); UP(B->d_seg_ptr, seg_ptr); UP(B->d_seg_k, seg_k); UP(B->d_seg_tile, seg_tile); UP(B->d_seg_slot, seg_slot);
    UP(B->d_run_e0, run_e0); UP(B->d_seg_cam, seg_cam);
    UP(B->d_gp_ptr, gp_ptr); UP(B->d_gp_i1, gp_i1); UP(B->d_gp_i2, gp_i2); UP(B->d_gtile, gtile); UP(B->d_gcam_ptr, gcam_ptr); UP(B->d_gslot, gslot);
    AL(B->part_tiles, 36 * (size_t)n_tiles); AL(B->part_coef, 6 * (size_t)n_slots); ZFLUSH();
    B->n_pairs = 0;
    std::vector<int> none(1, 0);
    UP(B->pair_ptr, none); UP(B->pair_i1, none); UP(B->pair_i2, none); UP(B->ent_a, none); UP(B->ent_b, none);
  } else {
    struct Ent { long long key; int a, b; };
    std::vector<Ent> ents;
    const long long NP = std::max(1, B->n_pose);
    for (int p = 0; p < np; p++) {
      if (!pt_free[p]) continue;
      for (int a = pt_ptr[p]; a < pt_ptr[p + 1]; a++) {
        int ca = B->cam_col[pm_cam[a]];
        if (ca < 0) continue;
        for (int b = a; b < pt_ptr[p + 1]; b++) {
          int cb = B->cam_col[pm_cam[b]];
          if (cb < 0) continue;
          ents.push_back(Ent{(long long)ca * NP + cb, a, b});
        }
      }
    }
    std::stable_sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) { return x.key < y.key; });
    std::vector<int> pair_ptr, pair_i1, pair_i2, ent_a(ents.size()), ent_b(ents.size());
    for (size_t i = 0; i < ents.size(); i++) {
      if (i == 0 || ents[i].key != ents[i - 1].key) {
        pair_ptr.push_back((int)i);
        pair_i1.push_back((int)(ents[i].key / NP));
        pair_i2.push_back((int)(ents[i].key % NP));
      }
      ent_a[i] = ents[i].a; ent_b[i] = ents[i].b;
    }
    pair_ptr.push_back((int)ents.size());
    B->n_pairs = (int)pair_i1.size();
    UP(B->pair_ptr, pair_ptr); UP(B->pair_i1, pair_i1); UP(B->pair_i2, pair_i2); UP(B->ent_a, ent_a); UP(B->ent_b, ent_b);
    std::vector<int> none(1, 0);
    UP(B->d_run_lm, none); UP(B->d_seg_ptr, none); UP(B->d_seg_k, none); UP(B->d_seg_tile, none); UP(B->d_seg_slot, none); UP(B->d_run_e0, none); UP(B->d_seg_cam, none);
    UP(B->d_gp_ptr, none); UP(B->d_gp_i1, none); UP(B->d_gp_i2, none); UP(B->d_gtile, none); UP(B->d_gcam_ptr, none); UP(B->d_gslot, none);
    AL(B->part_tiles, 1); AL(B->part_coef, 1); ZFLUSH();
    std::vector<int> zp(no + 1, 0);
    UP(B->d_slotE_ptr, none); UP(B->d_slotE_idx, none);
    UP(B->d_cubS_ptr, zp); UP(B->d_cubS_cam, none); UP(B->d_ce_slot, none); UP(B->d_cub_tile, none); UP(B->d_cub_coef, none);
    AL(B->cub_M, 1); AL(B->cub_Dinv, 1); AL(B->d_elim_fail, 1); ZFLUSH();
  }
  mark("Schur schedule");
  // ---- cuboid / odometry edges and their vertex adjacency
  auto csr = [&](int nv, const std::vector<int>& owner, std::vector<int>& ptr, std::vector<int>& idx) {
    ptr.assign(nv + 1, 0);
    for (int o : owner) ptr[o + 1]++;
    for (int i = 0; i < nv; i++) ptr[i + 1] += ptr[i];
    idx.resize(owner.size());
    std::vector<int> fill(ptr.begin(), ptr.end() - 1);
    for (size_t k = 0; k < owner.size(); k++) idx[fill[owner[k]]++] = (int)k;
  };
  std::vector<int> p1, i1, p2, i2, p3, i3, p4, i4;
  csr(nc, B->ce_cam, p1, i1); csr(nc, B->oe_i, p2, i2); csr(nc, B->oe_j, p3, i3); csr(no, B->ce_cub, p4, i4);
  UP(B->cam_ce_ptr, p1); UP(B->cam_ce_idx, i1); UP(B->cam_oei_ptr, p2); UP(B->cam_oei_idx, i2);
  UP(B->cam_oej_ptr, p3); UP(B->cam_oej_idx, i3); UP(B->cub_ce_ptr, p4); UP(B->cub_ce_idx, i4);
  UP(B->d_ce_cam, B->ce_cam); UP(B->d_ce_cub, B->ce_cub); UP(B->d_oe_i, B->oe_i); UP(B->d_oe_j, B->oe_j);
  {
    std::vector<int> ca(B->n_cub), oa(B->n_odom);
    // cuboid edges: with their camera's rank, or -- cuboids eliminated -- all edges of a cuboid with the rank of its lowest-index camera.
    // Separator mode: by the lowest column among the free vertices involved (an eliminated cuboid: among its observing cameras).
    std::vector<int> cub_owner(no, 0);
    if (B->sep_mode) {
      for (int o = 0; o < no; o++) {
        int lo = 0x7fffffff;
        for (int c : cub_cams[o]) lo = std::min(lo, B->cam_col[c]);
        cub_owner[o] = lo == 0x7fffffff ? 0 : rank_of_col(lo);
      }
      for (int k = 0; k < B->n_cub; k++) {
        const int o = B->ce_cub[k], c = B->ce_cam[k];
        int own = 0;
        if (B->elim && !B->cub_fixed[o]) own = cub_owner[o];
        else {
          int lo = 0x7fffffff;
          if (B->cam_col[c] >= 0) lo = std::min(lo, B->cam_col[c]);
          if (B->cub_col[o] >= 0 && B->cub_col[o] < B->n_red) lo = std::min(lo, B->cub_col[o]);
          own = lo == 0x7fffffff ? 0 : rank_of_col(lo);
        }
        ca[k] = own == B->shard_rank;
      }
      for (int k = 0; k < B->n_odom; k++) {
        int lo = 0x7fffffff;
        if (B->cam_col[B->oe_i[k]] >= 0) lo = std::min(lo, B->cam_col[B->oe_i[k]]);
        if (B->cam_col[B->oe_j[k]] >= 0) lo = std::min(lo, B->cam_col[B->oe_j[k]]);
        oa[k] = (lo == 0x7fffffff ? 0 : rank_of_col(lo)) == B->shard_rank;
      }
    } else {
      std::vector<int> first(no, 0x7fffffff);
      for (int k = 0; k < B->n_cub; k++) first[B->ce_cub[k]] = std::min(first[B->ce_cub[k]], B->ce_cam[k]);
      for (int o = 0; o < no; o++) cub_owner[o] = first[o] == 0x7fffffff ? 0 : cam_rank(first[o], nc, B->shard_n);
      for (int k = 0; k < B->n_cub; k++) ca[k] = (B->elim ? cub_owner[B->ce_cub[k]] : cam_rank(B->ce_cam[k], nc, B->shard_n)) == B->shard_rank;
      for (int k = 0; k < B->n_odom; k++) oa[k] = cam_rank(B->oe_j[k], nc, B->shard_n) == B->shard_rank;
    }
    {
      std::vector<int> mine(std::max(1, no), 0);
      for (int o = 0; o < no; o++) mine[o] = cub_owner[o] == B->shard_rank;
      UP(B->d_cub_mine, mine);
    }
    UP(B->d_ce_active, ca); UP(B->d_oe_active, oa);
  }
  {   // kernels of the camera-cuboid edges (EdgeSE3Cuboid list, then EdgeSE3CuboidProj list) and of the odometry edges
    bool any = false;
    for (int kd : B->rk_cub3) any |= kd != 0;
    for (int kd : B->rk_cproj) any |= kd != 0;
    if (any) {
      std::vector<int> kk(std::max(1, B->n_cub), 0); std::vector<double> dd(std::max(1, B->n_cub), 0.0);
      for (size_t k = 0; k < B->rk_cub3.size() && (int)k < B->n_cub3; k++) { kk[k] = B->rk_cub3[k]; dd[k] = B->rd_cub3[k]; }
      for (size_t k = 0; k < B->rk_cproj.size() && B->n_cub3 + (int)k < B->n_cub; k++) { kk[B->n_cub3 + k] = B->rk_cproj[k]; dd[B->n_cub3 + k] = B->rd_cproj[k]; }
      UP(B->d_ce_rk, kk); UP(B->d_ce_rdelta, dd);
    } else { B->d_ce_rk.release(); B->d_ce_rdelta.release(); }
    any = false;
    for (int kd : B->rk_odom) any |= kd != 0;
    if (any) {
      std::vector<int> kk(B->rk_odom); std::vector<double> dd(B->rd_odom);
      kk.resize(std::max(1, B->n_odom), 0); dd.resize(std::max(1, B->n_odom), 0.0);
      UP(B->d_oe_rk, kk); UP(B->d_oe_rdelta, dd);
    } else { B->d_oe_rk.release(); B->d_oe_rdelta.release(); }
  }
  mark("  pose edge lists");
  UP(B->ce_meas, B->h_ce_meas); UP(B->ce_info, B->h_ce_info); UP(B->oe_meas, B->h_oe_meas); UP(B->oe_info, B->h_oe_info);
  UP(B->pe_meas, B->h_pe_meas); UP(B->pe_info, B->h_pe_info); UP(B->pe_K, B->h_pe_K);
  AL(B->ce_Hcc, 36 * (size_t)B->n_cub); AL(B->ce_Hoo, 81 * (size_t)B->n_cub); AL(B->ce_Hco, 54 * (size_t)B->n_cub); AL(B->ce_bc, 6 * (size_t)B->n_cub); AL(B->ce_bo, 9 * (size_t)B->n_cub);
  AL(B->oe_Hii, 36 * (size_t)B->n_odom); AL(B->oe_Hjj, 36 * (size_t)B->n_odom); AL(B->oe_Hij, 36 * (size_t)B->n_odom); AL(B->oe_bi, 6 * (size_t)B->n_odom); AL(B->oe_bj, 6 * (size_t)B->n_odom);
  mark("  pose edge measurements");
  // ---- linear system storage
  AL(B->Hcam, 36 * (size_t)nc); AL(B->bcam, 6 * (size_t)nc); AL(B->Hcub, 81 * (size_t)no); AL(B->bcub, 9 * (size_t)no);
  AL(B->Hll, 9 * (size_t)np); AL(B->bl, 3 * (size_t)np); AL(B->W, 18 * (size_t)E); AL(B->WD, 18 * (size_t)E);
  AL(B->Dinv, 9 * (size_t)np); AL(B->dbl, 3 * (size_t)np); B->s_doubles = (size_t)B->n_red * (B->band_ld ? B->band_ld : B->n_red);
  AL(B->S, B->s_doubles + B->n_pose);   // [S | rhs]: one buffer, one all-reduce in the sharded solve
  AL(B->xl, 3 * (size_t)np);
  AL(B->d_band_info, 24);  // [first bad pivot + 1, grid-barrier counters, a zero double]
  AL(B->band_linv, B->band_ld ? std::max(cs::ba_band_workspace_doubles(B->n_red, B->band_ld), B->use_bcr ? cs::ba_bcr_workspace_doubles(B->n_red, 128) : (size_t)1) : 1);   // inverted diagonal blocks (+ the separator's rows in the nested order)
  B->nb_chi = cs::ba_chi2_blocks(E);
  B->n_chi_partials = B->nb_chi + (B->n_cub + B->n_odom + 63) / 64;
  AL(B->chi_partial, B->n_chi_partials);
  AL(B->scale_partial, (size_t)cs::ba_scale_blocks());
  AL(B->d_info, 1);
  if (B->sparse) {
    const cs::SparsePlan& SP = B->sp_plan;
    UP(B->sp_ndim, SP.ndim); UP(B->sp_ncol, SP.ncol); UP(B->sp_sptr, SP.sptr); UP(B->sp_srow, SP.srow); UP(B->sp_sroff, SP.sroff); UP(B->sp_prow, SP.prow);
    UP(B->sp_rbase, SP.rbase); UP(B->sp_rent, SP.rent); UP(B->sp_rptr, SP.rptr); UP(B->sp_order, SP.order); UP(B->sp_poff, SP.poff);
    { std::vector<int> v_rcol(SP.rcol), v_rpos(SP.rpos); if (v_rcol.empty()) { v_rcol.push_back(0); v_rpos.push_back(0); } UP(B->sp_rcol, v_rcol); UP(B->sp_rpos, v_rpos); }
    UP(B->sp_tcol, SP.tcol); AL(B->sp_T, (size_t)SP.n_tail * SP.n_tail + SP.n_tail + 1);
    AL(B->sp_L, (size_t)SP.nvals); AL(B->sp_xs, 9 * (size_t)(SP.N + 1)); AL(B->sp_done, (size_t)SP.N + 1); AL(B->sp_xdone, (size_t)SP.N + 2); AL(B->sp_info, 2);
  }
  if (B->sep_mode) {
    AL(B->sepY, (size_t)(B->wl + B->wr) * B->int_n);
    AL(B->sep_msgs, B->msg_doubles * (size_t)R);
    AL(B->sepS, (size_t)B->n_sep * 2 * B->w_max + B->n_sep);   // [S_sep (band of 2 w_max) | rhs_sep]
    AL(B->sep_work, std::max(cs::ba_band_workspace_doubles(B->n_sep, 2 * B->w_max), cs::ba_bcr_sep_ok(B->w_max, B->shard_n) ? cs::ba_bcr_sep_workspace_doubles(B->shard_n) : (size_t)1));
    AL(B->d_sep_info, 24);
    AL(B->int_work, cs::ba_band_workspace_doubles(B->int_n, B->band_ld));
    AL(B->d_int_info, 24);
    std::vector<int> sep_col(R + 1, 0);
    for (int k = 1; k < R; k++) sep_col[k] = B->cut[k];
    UP(B->d_sep_off, B->sep_off); UP(B->d_sep_col, sep_col);
    // what this rank contributes to the collectives of one LM trial: its separator message, the solution vector, three scalars
    B->bytes_per_trial = 8 * ((long long)B->msg_doubles + B->n_pose + 3);
  } else {
    AL(B->sepY, 1); AL(B->sep_msgs, 1); AL(B->sepS, 1); AL(B->int_work, 1); AL(B->d_int_info, 24); AL(B->sep_work, 1); AL(B->d_sep_info, 24);
    std::vector<int> none1(1, 0);
    UP(B->d_sep_off, none1); UP(B->d_sep_col, none1);
    B->bytes_per_trial = R > 1 ? 8 * ((long long)B->s_doubles + B->n_pose + 3 + (B->elim ? B->n_pose - B->n_red : 0)) : 0;
  }
  B->bytes_per_trial_allreduce = 8 * ((long long)B->s_doubles + B->n_pose + 3 + (B->elim ? B->n_pose - B->n_red : 0));
  {
    const size_t need = std::max<size_t>(16, (size_t)B->n_pose + 2);
    AL(B->d_scalars, need);
    if (B->scalars_cap < need) {     // (with head room: a graph that grows by a camera per frame would re-pin this buffer every frame, ~0.1 ms)
      if (B->h_scalars) (void)hipHostFree(B->h_scalars);
      B->h_scalars = nullptr;
      const size_t want = need + need / 4 + 256;
      BA_TRY(hipHostMalloc((void**)&B->h_scalars, want * sizeof(double)));
      B->scalars_cap = want;
    }
  }
  AL(B->cams_bak, 7 * (size_t)nc); AL(B->points_bak, 3 * (size_t)np); AL(B->cubes_bak, 10 * (size_t)no);
  { std::vector<double> u8(B->uni8, B->uni8 + 8); UP(B->d_uni, u8); }
#undef UP
#undef AL
#undef UPB
  // (the helper thread allocates the edge tables' device buffers: their addresses are final only once it is done)
  mark("  pose edge tables staged");
  edge_th.join();
  mark("  edge-table thread joined");
  if (edge_rc) return edge_rc;
  cs::BaView& v = B->view;
  v.cams = B->cams.p; v.points = B->points.p; v.cubes = B->cubes.p; v.cam_col = B->d_cam_col.p; v.cub_col = B->d_cub_col.p; v.pt_free = B->d_pt_free.p;
  v.nc = nc; v.np = np; v.no = no; v.n_pose = B->n_pose; v.n_red = B->n_red; v.elim = B->elim ? 1 : 0; v.elim_max_slots = B->elim_max_slots;
  v.cubS_ptr = B->d_cubS_ptr.p; v.cubS_cam = B->d_cubS_cam.p; v.ce_slot = B->d_ce_slot.p; v.cub_tile = B->d_cub_tile.p; v.cub_coef = B->d_cub_coef.p;
  v.cub_mine = B->d_cub_mine.p;
  v.cub_M = B->cub_M.p; v.cub_Dinv = B->cub_Dinv.p; v.elim_fail = B->d_elim_fail.p; v.slotE_ptr = B->d_slotE_ptr.p; v.slotE_idx = B->d_slotE_idx.p;
  v.n_proj = E; v.pm_pt = B->pm_pt.p; v.pm_cam = B->pm_cam.p; v.pm_uv = B->pm_uv.p; v.pm_info = B->pm_info.p; v.pm_intr = B->pm_intr.p; v.pm_huber = B->pm_huber.p;
  v.pm_rk = B->d_pm_rk.p; v.cm_rk = B->d_cm_rk.p; v.ce_rk = B->d_ce_rk.p; v.ce_rdelta = B->d_ce_rdelta.p; v.oe_rk = B->d_oe_rk.p; v.oe_rdelta = B->d_oe_rdelta.p;
  v.pt_ptr = B->pt_ptr.p; v.cm_pm = B->cm_pm.p; v.cm_pt = B->cm_pt.p; v.cm_uv = B->cm_uv.p; v.cm_info = B->cm_info.p; v.cm_intr = B->cm_intr.p; v.cm_huber = B->cm_huber.p; v.cam_ptr = B->cam_ptr.p;
  v.n_cub3 = B->n_cub3; v.pe_meas = B->pe_meas.p; v.pe_info = B->pe_info.p; v.pe_K = B->pe_K.p;
  v.n_cub = B->n_cub; v.ce_cam = B->d_ce_cam.p; v.ce_cub = B->d_ce_cub.p; v.ce_meas = B->ce_meas.p; v.ce_info = B->ce_info.p; v.ce_active = B->d_ce_active.p;
  v.ce_Hcc = B->ce_Hcc.p; v.ce_Hoo = B->ce_Hoo.p; v.ce_Hco = B->ce_Hco.p; v.ce_bc = B->ce_bc.p; v.ce_bo = B->ce_bo.p;
  v.n_odom = B->n_odom; v.oe_i = B->d_oe_i.p; v.oe_j = B->d_oe_j.p; v.oe_meas = B->oe_meas.p; v.oe_info = B->oe_info.p; v.oe_active = B->d_oe_active.p;
  v.oe_Hii = B->oe_Hii.p; v.oe_Hjj = B->oe_Hjj.p; v.oe_Hij = B->oe_Hij.p; v.oe_bi = B->oe_bi.p; v.oe_bj = B->oe_bj.p;
  v.cam_ce_ptr = B->cam_ce_ptr.p; v.cam_ce_idx = B->cam_ce_idx.p; v.cam_oei_ptr = B->cam_oei_ptr.p; v.cam_oei_idx = B->cam_oei_idx.p;
  v.cam_oej_ptr = B->cam_oej_ptr.p; v.cam_oej_idx = B->cam_oej_idx.p; v.cub_ce_ptr = B->cub_ce_ptr.p; v.cub_ce_idx = B->cub_ce_idx.p;
  v.Hcam = B->Hcam.p; v.bcam = B->bcam.p; v.Hcub = B->Hcub.p; v.bcub = B->bcub.p; v.Hll = B->Hll.p; v.bl = B->bl.p; v.W = B->W.p; v.WD = B->WD.p;
  v.fuse_lin = 0;
  {
    v.info_u = B->info_uniform && E > 0 ? B->d_uni.p : nullptr;
    v.intr_u = B->intr_uniform && E > 0 ? B->d_uni.p + 4 : nullptr;
  }
  v.Dinv = B->Dinv.p; v.dbl = B->dbl.p; v.S = B->S.p; v.band_ld = B->band_ld; v.lam_lo = B->sep_mode ? B->cut[B->shard_rank] : 0; v.lam_hi = B->sep_mode ? B->cut[B->shard_rank + 1] : (B->shard_rank == 0 ? 0x7fffffff : 0); v.rhs = B->S.p + B->s_doubles; v.xl = B->xl.p;
  v.n_pairs = B->n_pairs; v.pair_ptr = B->pair_ptr.p; v.pair_i1 = B->pair_i1.p; v.pair_i2 = B->pair_i2.p; v.ent_a = B->ent_a.p; v.ent_b = B->ent_b.p;
  v.fused = B->fused ? 1 : 0; v.n_seg = B->n_seg; for (int q = 0; q < 5; q++) v.seg_class[q] = B->seg_class[q];
  v.seg_ptr = B->d_seg_ptr.p; v.seg_k = B->d_seg_k.p; v.seg_tile = B->d_seg_tile.p; v.seg_slot = B->d_seg_slot.p; v.run_lm = B->d_run_lm.p; v.run_e0 = B->d_run_e0.p; v.seg_cam = B->d_seg_cam.p;
  v.part_tiles = B->part_tiles.p; v.part_coef = B->part_coef.p;
  v.n_gpairs = B->n_gpairs; v.gpair_ptr = B->d_gp_ptr.p; v.gpair_i1 = B->d_gp_i1.p; v.gpair_i2 = B->d_gp_i2.p; v.gtile = B->d_gtile.p;
  v.gcam_ptr = B->d_gcam_ptr.p; v.gslot = B->d_gslot.p;
  v.chi_partial = B->chi_partial.p;
  // the allocations above are zeroed on B->st (one batched fill, one wait here); uploads went through blocking copies
  ZFLUSH();
  rc = cflush(true); if (rc) return rc;
  BA_TRY(hipStreamSynchronize(B->st));
  B->append_stage.off = 0;           // (the appended rows have arrived)
  B->st_lists_edges = B->n_proj; B->st_cam_cnt = std::move(cam_cnt); B->st_cur = st_nb;   // (for a grown graph's next phase)
  mark("pose edges + allocations");
  B->structure_dirty = false;
  B->have_system = false;
  return CS_OK;
}

}  // namespace
static void debug_nan_scan(cs_ba* B, const char* where);
static bool debug_nan_enabled();
namespace {

int chi2_device(cs_ba* B, double* chi) {
  cs::ba_launch_chi2(B->view, B->nb_chi, B->st);
  BA_TRY(hipGetLastError());
  std::vector<double> part(B->n_chi_partials);
  BA_TRY(hipMemcpyAsync(part.data(), B->chi_partial.p, sizeof(double) * part.size(), hipMemcpyDeviceToHost, B->st));
  BA_TRY(hipStreamSynchronize(B->st));
  double s = 0;
  for (double p : part) s += p;  // fixed order
  *chi = s;
  return CS_OK;
}

// b (poses then landmarks) to the host: LM's scale term needs it (optimization_algorithm_levenberg.cpp:182-189)
int fetch_b(cs_ba* B) {
  B->h_b.assign(B->n_pose + 3 * (size_t)B->n_lm, 0.0);
  std::vector<double> bc(6 * (size_t)B->nc), bo(9 * (size_t)B->no), bl(3 * (size_t)B->np);
  if (B->nc) BA_TRY(hipMemcpyAsync(bc.data(), B->bcam.p, 8 * bc.size(), hipMemcpyDeviceToHost, B->st));
  if (B->no) BA_TRY(hipMemcpyAsync(bo.data(), B->bcub.p, 8 * bo.size(), hipMemcpyDeviceToHost, B->st));
  if (B->np) BA_TRY(hipMemcpyAsync(bl.data(), B->bl.p, 8 * bl.size(), hipMemcpyDeviceToHost, B->st));
  BA_TRY(hipStreamSynchronize(B->st));
  for (int i = 0; i < B->nc; i++) if (B->cam_col[i] >= 0) std::memcpy(&B->h_b[B->cam_col[i]], &bc[6 * (size_t)i], 48);
  for (int i = 0; i < B->no; i++) if (B->cub_col[i] >= 0) std::memcpy(&B->h_b[B->cub_col[i]], &bo[9 * (size_t)i], 72);
  for (int i = 0; i < B->np; i++) if (B->pt_lm[i] >= 0) std::memcpy(&B->h_b[B->n_pose + 3 * (size_t)B->pt_lm[i]], &bl[3 * (size_t)i], 24);
  return CS_OK;
}

int fetch_x(cs_ba* B) {
  B->h_x.assign(B->n_pose + 3 * (size_t)B->n_lm, 0.0);
  std::vector<double> xl(3 * (size_t)B->np);
  if (B->n_pose) BA_TRY(hipMemcpyAsync(B->h_x.data(), B->view.rhs, 8 * (size_t)B->n_pose, hipMemcpyDeviceToHost, B->st));
  if (B->np) BA_TRY(hipMemcpyAsync(xl.data(), B->xl.p, 8 * xl.size(), hipMemcpyDeviceToHost, B->st));
  BA_TRY(hipStreamSynchronize(B->st));
  for (int i = 0; i < B->np; i++) if (B->pt_lm[i] >= 0) std::memcpy(&B->h_x[B->n_pose + 3 * (size_t)B->pt_lm[i]], &xl[3 * (size_t)i], 24);
  return CS_OK;
}

// phase times come from events read after the next stream synchronisation, so that no phase boundary stalls the host
int collect_lin_time(cs_ba* B) {
  if (!B->stage_timing) { B->lin_pending = false; return CS_OK; }
  if (!B->lin_pending) return CS_OK;
  float ms = 0;
  BA_TRY(hipEventElapsedTime(&ms, B->ev[0], B->ev[1]));
  B->tm.linearize_ms += ms;
  B->lin_pending = false;
  return CS_OK;
}

int build_system_device(cs_ba* B, hipEvent_t ev_pre = nullptr) {
  BA_MARK(B, B->ev[0]);
  static const int lin_mode = [] { const char* e = getenv("CS_BA_LIN_STREAMS"); return e ? atoi(e) : 3; }();   // diagnostics: 3 = camera / landmark / pose-edge kernels side by side, 2 = landmark kernel behind the camera kernel, 1 = one stream
  cs::ba_launch_linearize(B->view, B->st, lin_mode >= 2 ? B->st2 : nullptr, B->ev_fork, B->ev_join, lin_mode >= 3 ? B->st3 : nullptr, B->ev_join3, lin_mode >= 2 ? ev_pre : nullptr);
  if (B->ext_terms_set) {
    if (B->ext_cam36.n != 36 * (size_t)B->nc || B->ext_cub81.n != 81 * (size_t)B->no || B->ext_pt9.n != 9 * (size_t)B->np) { cs_set_error_ba("external terms were set for a graph of another size: call cs_ba_set_external_terms again"); return CS_ERR_INVALID_ARG; }
    cs::ba_launch_ext_add(B->view, B->ext_has_cam ? B->ext_cam36.p : nullptr, B->ext_cam6.p, B->ext_has_cub ? B->ext_cub81.p : nullptr, B->ext_cub9.p,
                          B->ext_has_pt ? B->ext_pt9.p : nullptr, B->ext_pt3.p, B->st);
  }
  BA_TRY(hipGetLastError());
  BA_MARK(B, B->ev[1]);
  B->lin_pending = true;
  B->tm.n_linearizations++;
  B->have_system = true;
  return CS_OK;
}

// setLambda + solve + restoreDiagonal (block_solver.hpp:353-486, :563-604): lambda is applied while the
// reduced system is assembled, so the stored blocks are never modified and nothing needs restoring.
// The banded factorisation is a persistent kernel whose workgroups wait for each other: all of them must be resident.
// One such kernel fits many times over, but an unbounded number of concurrent solves (handles on different streams of
// one process) would not; they take turns.
static std::mutex g_coop_mutex;
// The turn is held across the collectives of a trial when they are queued on the stream (RCCL): two communicator handles in one
// process would wait for each other (one holds the turn inside a collective, the other needs the turn to enqueue its side).  One
// process per GPU is the contract; cs_ba_comm_init enforces it.
static std::atomic<int> g_comm_handles{0};

// phase times of the last solve, read once the stream has been synchronised
int collect_solve_times(cs_ba* B) {
  if (!B->stage_timing) { B->lin_pending = false; return CS_OK; }
  float ms = 0;
  BA_TRY(hipEventElapsedTime(&ms, B->ev[2], B->ev[3])); B->tm.reduce_ms += ms;
  BA_TRY(hipEventElapsedTime(&ms, B->ev[3], B->ev[4])); B->tm.factor_ms += ms;
  BA_TRY(hipEventElapsedTime(&ms, B->ev[4], B->ev[5])); B->tm.backsub_ms += ms;
  if (B->sep_mode && B->shard_n > 1) {
    hipEvent_t seq[6] = {B->ev[3], B->sev[0], B->sev[1], B->sev[2], B->sev[3], B->ev[4]};
    for (int i = 0; i < 5; i++) { BA_TRY(hipEventElapsedTime(&ms, seq[i], seq[i + 1])); B->sep_ms[i] += ms; }
  }
  return collect_lin_time(B);
}

// Sharded with the cuboids eliminated: a cuboid's increment is computed by the rank that owns it (zero elsewhere); every rank updates
// all vertices, so the increments (9 doubles per free cuboid, behind the reduced system's in `rhs`) are summed over the ranks.
int share_cuboid_increments(cs_ba* B, cs_allreduce_fn fn, void* ctx) {
  if (!B->elim || B->shard_n <= 1 || B->n_pose <= B->n_red) return CS_OK;
  double* xo = B->view.rhs + B->n_red;
  const size_t n = (size_t)(B->n_pose - B->n_red);
  if (fn) {
    BA_TRY(hipStreamSynchronize(B->st));
    if (fn(ctx, xo, n, 1, 0) != 0) { cs_set_error_ba("all-reduce callback failed"); return CS_ERR_HIP; }
  } else if (B->comm) {
    BA_NCCL(ncclAllReduce(xo, xo, n, ncclDouble, ncclSum, B->comm, B->st));
  }
  return CS_OK;
}

// defer != nullptr (banded path only): everything is queued and the function returns WITHOUT synchronising; *defer then holds the
// persistent-kernel turn, and the caller synchronises, reads *h_status, calls collect_solve_times() and releases the turn.
// lambda of the damped solve that is about to be queued: into the pinned word and, queued on the stream, into device memory.  (Inside a
// captured trial the copy is a node of the graph: every replay reads whatever the host left in h_lam.)
int put_lambda(cs_ba* B, double lambda) {
  B->h_lam[0] = lambda;
  B->h_lam[1] = B->shard_rank == 0 ? lambda : 0.0;      // x^T (lambda x + b): the poses' lambda x^2 is counted once, on rank 0
  BA_TRY(hipMemcpyAsync(B->d_lam.p, B->h_lam, 2 * sizeof(double), hipMemcpyHostToDevice, B->st));
  return CS_OK;
}
int solve_device_sep(cs_ba* B, double lambda, bool* ok, cs_allreduce_fn fn, void* ctx, std::unique_lock<std::mutex>* defer);
static cs::SparseView sparse_view(cs_ba* B) {
  cs::SparseView SV;
  SV.N = B->sp_plan.N; SV.n = B->n_red;
  SV.ndim = B->sp_ndim.p; SV.ncol = B->sp_ncol.p; SV.sptr = B->sp_sptr.p; SV.srow = B->sp_srow.p; SV.sroff = B->sp_sroff.p; SV.prow = B->sp_prow.p;
  SV.rbase = B->sp_rbase.p; SV.rent = B->sp_rent.p; SV.rptr = B->sp_rptr.p; SV.rcol = B->sp_rcol.p; SV.rpos = B->sp_rpos.p; SV.order = B->sp_order.p; SV.poff = B->sp_poff.p;
  SV.tcol = B->sp_tcol.p; SV.tail_start = B->sp_plan.tail_start; SV.n_tail = B->sp_plan.n_tail; SV.T = B->sp_T.p; SV.rhs_t = B->sp_T.p + (size_t)SV.n_tail * SV.n_tail;
  SV.S = B->S.p; SV.rhs = B->view.rhs; SV.L = B->sp_L.p; SV.xs = B->sp_xs.p; SV.done = B->sp_done.p; SV.xdone = B->sp_xdone.p; SV.info = B->sp_info.p;
  return SV;
}

int solve_device(cs_ba* B, double lambda, bool* ok, cs_allreduce_fn fn = nullptr, void* ctx = nullptr, std::unique_lock<std::mutex>* defer = nullptr) {
  const int n = B->n_red;
  *ok = true;
  if (B->sep_mode && B->shard_n > 1) return solve_device_sep(B, lambda, ok, fn, ctx, defer);
  if (n > 0) {
    const bool lean_head = B->band_ld > 0 && !B->sparse;      // (banded path: one prologue kernel instead of a copy and three fills)
    // (the prologue's launch is ba_launch_reduce's when nothing reads the phase marks and no host-evaluated edge writes S in between: it then
    // runs on the side stream, beside the landmark segments -- BaSidePrologue)
    cs::BaSidePrologue side_pro{};
    const bool pro_in_reduce = lean_head && defer != nullptr && !B->stage_timing && B->shard_n == 1 && !(B->ext_n > 0 && B->ext_terms_set);
    if (lean_head) {
      B->h_lam[0] = lambda; B->h_lam[1] = B->shard_rank == 0 ? lambda : 0.0;
      side_pro = cs::BaSidePrologue{B->d_lam.p, B->h_lam[0], B->h_lam[1], B->d_band_info.p, B->d_elim_fail.p, B->S.p, B->s_doubles + (size_t)B->n_pose};
      if (!pro_in_reduce) cs::ba_launch_trial_prologue(B->d_lam.p, B->h_lam[0], B->h_lam[1], B->d_band_info.p, B->d_elim_fail.p, B->S.p, B->s_doubles + (size_t)B->n_pose, B->st);   // (+ [S | rhs] cleared: no fill of its own)
    } else { int rcl = put_lambda(B, lambda); if (rcl) return rcl; }
    BA_MARK(B, B->ev[2]);
    if (lean_head) {
    } else if (B->sparse && B->sp_S_clean) {
      // (sparse path: S was cleared by the structure phase and only the plan's pattern is ever written -- the pattern and the right-hand side)
      cs::launch_sparse_zero_pattern(sparse_view(B), B->S.p, B->st);
      BA_TRY(hipMemsetAsync(B->S.p + B->s_doubles, 0, sizeof(double) * B->n_pose, B->st));
    } else {
      BA_TRY(hipMemsetAsync(B->S.p, 0, sizeof(double) * (B->s_doubles + B->n_pose), B->st));
      B->sp_S_clean = B->sparse;
    }
    if (!lean_head) BA_TRY(hipMemsetAsync(B->d_elim_fail.p, 0, sizeof(int), B->st));
    cs::ba_launch_reduce(B->view, B->d_lam.p, B->st, B->st2, B->ev_fork, B->ev_join, pro_in_reduce ? &side_pro : nullptr);
    if (B->ext_n > 0 && B->ext_terms_set) cs::ba_launch_ext_offdiag(B->view, B->ext_groups, B->d_ext_gptr.p, B->d_ext_order.p, B->d_ext_e4.p, B->ext_Hij.p, B->st);
    // (block cyclic reduction in a deferred trial: no kernel of it waits for another -- no time-out word to read --, and the two failure words
    // reach the host folded into the trial's scalars by the caller's sum kernel: the two 4-byte copies, a blit launch each, stay away)
    const bool status_in_scalars = defer != nullptr && B->band_ld > 0 && B->use_bcr;
    if (status_in_scalars) { B->h_status[0] = 0; B->h_status[1] = 0; }
    else BA_TRY(hipMemcpyAsync(B->h_status + 1, B->d_elim_fail.p, sizeof(int), hipMemcpyDeviceToHost, B->st));
    BA_TRY(hipGetLastError());
    if (fn && B->shard_n > 1) {  // sum the ranks' partial reduced systems: [S | rhs] in one message
      BA_TRY(hipStreamSynchronize(B->st));
      if (fn(ctx, B->S.p, B->s_doubles + B->n_pose, 1, 0) != 0) { cs_set_error_ba("all-reduce callback failed"); return CS_ERR_HIP; }
    } else if (!fn && B->comm) {  // RCCL, queued on this stream behind the kernels that produced the partial system: no host round trip
      BA_NCCL(ncclAllReduce(B->S.p, B->S.p, B->s_doubles + B->n_pose, ncclDouble, ncclSum, B->comm, B->st));
    }
    BA_MARK(B, B->ev[3]);
    if (B->band_ld) {
      // banded: factorisation, both substitutions and the landmark back-substitution are queued back to back; the
      // pivot flag comes home with the single synchronisation (a failed factorisation just leaves garbage increments
      // that the caller discards)
      std::unique_lock<std::mutex> coop_turn(g_coop_mutex);
      if (B->use_bcr) cs::ba_launch_bcr(B->S.p, B->band_linv.p, n, B->band_ld, 128, B->view.rhs, B->d_band_info.p, B->st);
      else cs::ba_launch_band_cholesky(B->S.p, B->band_linv.p, n, B->band_ld, B->view.rhs, B->d_band_info.p, true, B->st);
      BA_TRY(hipGetLastError());
      BA_MARK(B, B->ev[4]);
      // (Round 5 tried clearing the band for the NEXT trial on the side stream here -- block cyclic reduction is done with it after its first
      // level -- to take the 10 us fill off the head of a trial: the event record / wait / fill / record sequence stalled the host's enqueue of the
      // kernels behind it by 35-60 us each, profiles/r5_ba_timeline_prezero.txt; the fill stays at the head.)
      cs::ba_launch_backsub(B->view, B->st);
      BA_TRY(hipGetLastError());
      if (fn && B->elim && B->shard_n > 1) {   // the callback waits for the other ranks: the persistent kernel's turn must be free by then
        BA_TRY(hipStreamSynchronize(B->st));
        coop_turn.unlock();
      }
      { int rc2 = share_cuboid_increments(B, fn, ctx); if (rc2) return rc2; }
      BA_MARK(B, B->ev[5]);
      if (!status_in_scalars) BA_TRY(hipMemcpyAsync(B->h_status, B->d_band_info.p, sizeof(int), hipMemcpyDeviceToHost, B->st));
      if (defer) { *defer = std::move(coop_turn); B->tm.n_solves++; return CS_OK; }   // (the caller's sum kernel folds the two status words into the trial's scalars)
      cs::ba_launch_fail_flag(B->d_band_info.p, B->d_elim_fail.p, nullptr, B->d_scalars.p + 2, B->st);
      BA_TRY(hipStreamSynchronize(B->st));
      if (*B->h_status == 0x7fffffff) {   // a workgroup waited ~1 s for its team: the device is shared with another persistent kernel
        cs_set_error_ba("banded solver: team not co-resident (wait timed out); set CS_BA_FORCE_DENSE=1 on a shared device");
        return CS_ERR_HIP;
      }
      if (B->h_status[0] != 0 || B->h_status[1] != 0) *ok = false;
    } else if (B->sparse) {
      // general sparse: the pattern's blocks are gathered from the dense S by the factorisation itself (sparse_kernels.hip)
      std::unique_lock<std::mutex> coop_turn(g_coop_mutex);
      BA_TRY(hipMemsetAsync(B->sp_info.p, 0, 2 * sizeof(int), B->st));
      const cs::SparseView SV = sparse_view(B);
      if (!cs::launch_sparse_cholesky(SV, cs::sparse_max_panel_doubles(), B->sp_grids, B->st)) { cs_set_error_ba("sparse solver: the factorisation could not be launched (grid " + std::to_string(B->sp_grids.chol) + " for " + std::to_string(SV.N) + " vertices)"); return CS_ERR_HIP; }
      BA_TRY(hipMemsetAsync(B->d_info.p, 0, sizeof(int), B->st));
      if (SV.n_tail > 0) {   // the top of the elimination tree as one dense block (same storage convention as the dense path's S)
        BA_ROC(rocsolver_dpotrf(B->blas, rocblas_fill_upper, SV.n_tail, SV.T, SV.n_tail, B->d_info.p));
        BA_ROC(rocsolver_dpotrs(B->blas, rocblas_fill_upper, SV.n_tail, 1, SV.T, SV.n_tail, SV.rhs_t, SV.n_tail));
      }
      if (!cs::launch_sparse_backsolve(SV, B->sp_grids, B->st)) { cs_set_error_ba("sparse solver: the substitution could not be launched (grid " + std::to_string(B->sp_grids.back) + " for " + std::to_string(SV.N) + " vertices)"); return CS_ERR_HIP; }
      BA_MARK(B, B->ev[4]);
      cs::ba_launch_backsub(B->view, B->st);
      BA_TRY(hipGetLastError());
      { int rc2 = share_cuboid_increments(B, fn, ctx); if (rc2) return rc2; }
      BA_MARK(B, B->ev[5]);
      BA_TRY(hipMemcpyAsync(B->h_status, B->sp_info.p, sizeof(int), hipMemcpyDeviceToHost, B->st));
      BA_TRY(hipMemcpyAsync(B->h_status + 2, B->d_info.p, sizeof(int), hipMemcpyDeviceToHost, B->st));
      BA_TRY(hipStreamSynchronize(B->st));
      if (B->h_status[2] != 0 && B->h_status[0] == 0) B->h_status[0] = B->h_status[2];     // (the dense tail's pivot)
      if (*B->h_status == 0x7fffffff) {
        cs_set_error_ba("sparse solver: grid not co-resident (wait timed out); set CS_BA_SPARSE=0 on a shared device");
        return CS_ERR_HIP;
      }
      if (B->h_status[0] != 0 || B->h_status[1] != 0) *ok = false;
    } else {
      // dense: the lower triangle of the row-major S is the upper triangle of the column-major matrix rocSOLVER sees
      BA_ROC(rocsolver_dpotrf(B->blas, rocblas_fill_upper, n, B->S.p, n, B->d_info.p));
      BA_TRY(hipMemcpyAsync(B->h_status, B->d_info.p, sizeof(int), hipMemcpyDeviceToHost, B->st));
      BA_TRY(hipStreamSynchronize(B->st));
      if (B->h_status[0] != 0 || B->h_status[1] != 0) *ok = false;
      else BA_ROC(rocsolver_dpotrs(B->blas, rocblas_fill_upper, n, 1, B->S.p, n, B->view.rhs, n));
      BA_MARK(B, B->ev[4]);
      // (sharded: the collective is issued whether or not THIS rank's factorisation went through -- the ranks decide together, below)
      if (*ok || B->shard_n > 1) { cs::ba_launch_backsub(B->view, B->st); BA_TRY(hipGetLastError()); int rc2 = share_cuboid_increments(B, fn, ctx); if (rc2) return rc2; }
      BA_MARK(B, B->ev[5]);
      BA_TRY(hipStreamSynchronize(B->st));
    }
    int rc = collect_solve_times(B); if (rc) return rc;
  }
  B->tm.n_solves++;
  return CS_OK;
}


// collectives of the sharded solve: RCCL on the handle's stream (queued, no host round trip), or the caller's callback (which is
// host-synchronous: the stream is drained first)
int coll_allreduce(cs_ba* B, cs_allreduce_fn fn, void* ctx, double* d, size_t n) {
  if (fn) {
    BA_TRY(hipStreamSynchronize(B->st));
    if (fn(ctx, d, n, 1, 0) != 0) { cs_set_error_ba("all-reduce callback failed"); return CS_ERR_HIP; }
  } else if (B->comm) {
    BA_NCCL(ncclAllReduce(d, d, n, ncclDouble, ncclSum, B->comm, B->st));
  }
  return CS_OK;
}
// in place: rank r's part sits at buf + r * per_rank.  Through the callback an all-gather is the sum of the zero-padded buffers
// (the caller zeroes the other ranks' slots first).
int coll_allgather(cs_ba* B, cs_allreduce_fn fn, void* ctx, double* buf, size_t per_rank) {
  if (fn) {
    BA_TRY(hipStreamSynchronize(B->st));
    if (fn(ctx, buf, per_rank * (size_t)B->shard_n, 1, 0) != 0) { cs_set_error_ba("all-reduce callback failed"); return CS_ERR_HIP; }
  } else if (B->comm) {
    BA_NCCL(ncclAllGather(buf + (size_t)B->shard_rank * per_rank, buf, per_rank, ncclDouble, B->comm, B->st));
  }
  return CS_OK;
}

// The damped solve in separator mode (cs_ba::sep_mode; kernels: "sharded reduced solve" in ba_kernels.hip).  What is built here lies
// in this rank's columns and in the next separator's diagonal block; only the interior is factorised here.  Collectives per solve:
// one all-gather of the separator messages (3 w^2 + 2 w doubles per rank) and one all-reduce of the solution vector (n_pose doubles:
// every rank contributes its interior's increments and those of the cuboids it eliminated).  block_solver.hpp:385-485 for what a
// shard builds and solves.
int solve_device_sep(cs_ba* B, double lambda, bool* ok, cs_allreduce_fn fn, void* ctx, std::unique_lock<std::mutex>* defer) {
  *ok = true;
  // Without a collective the separator system would be assembled from the other ranks' stale / zero messages and "solve" to a
  // meaningless x that reports positive_definite = 1 (the low-level path cs_ba_solve -> solve_device has neither callback nor, possibly,
  // a communicator).  A sharded handle solves only through cs_ba_optimize_sharded or after cs_ba_comm_init.
  if (!fn && !B->comm) { cs_set_error_ba("sharded handle in separator mode: the damped solve needs the ranks' separator messages -- use cs_ba_optimize_sharded (callback) or cs_ba_comm_init (RCCL); cs_ba_solve alone cannot"); return CS_ERR_INVALID_ARG; }
  const int R = B->shard_n, LD = B->band_ld, ns = B->n_sep;
  double* rhs = B->view.rhs;
  const int LDs = 2 * B->w_max;
  double* rsep = B->sepS.p + (size_t)ns * LDs;
  { int rcl = put_lambda(B, lambda); if (rcl) return rcl; }
  BA_MARK(B, B->ev[2]);
  BA_TRY(hipMemsetAsync(B->S.p, 0, sizeof(double) * (B->s_doubles + B->n_pose), B->st));
  BA_TRY(hipMemsetAsync(B->d_elim_fail.p, 0, sizeof(int), B->st));
  if (fn) BA_TRY(hipMemsetAsync(B->sep_msgs.p, 0, sizeof(double) * B->msg_doubles * (size_t)R, B->st));
  cs::ba_launch_reduce(B->view, B->d_lam.p, B->st, B->st2, B->ev_fork, B->ev_join);
  BA_TRY(hipMemcpyAsync(B->h_status + 1, B->d_elim_fail.p, sizeof(int), hipMemcpyDeviceToHost, B->st));
  BA_TRY(hipGetLastError());
  BA_MARK(B, B->ev[3]);
  std::unique_lock<std::mutex> coop_turn(g_coop_mutex);
  BA_TRY(hipMemsetAsync(B->d_int_info.p, 0, 24 * sizeof(int), B->st));
  // the interior: L L^T = S(I, I), y = L^-1 b_I in place (one-sided order, right-hand side riding along)
  cs::ba_launch_band_cholesky(B->S.p + (size_t)B->int_c * LD, B->int_work.p, B->int_n, LD, rhs + B->int_c, B->d_int_info.p, false, B->st, true);
  BA_TRY(hipGetLastError());
  BA_MARK(B, B->sev[0]);
  if (fn) {   // the callback waits for the other ranks (threads of this process in the tests): the persistent kernel's turn must be free by then
    BA_TRY(hipStreamSynchronize(B->st));
    coop_turn.unlock();
  }
  cs::ba_launch_sep_reduce(B->S.p, LD, B->int_work.p, B->int_c, B->int_n, B->zl, B->wl, B->zr, B->wr, B->sepY.p, rhs, B->sep_msgs.p + (size_t)B->shard_rank * B->msg_doubles, B->w_max, B->st);
  BA_TRY(hipGetLastError());
  BA_MARK(B, B->sev[1]);
  { int rc = coll_allgather(B, fn, ctx, B->sep_msgs.p, B->msg_doubles); if (rc) return rc; }
  BA_MARK(B, B->sev[2]);
  // every rank assembles and solves the (small) separator system: nothing to broadcast afterwards.  Block tridiagonal = a band of
  // 2 w_max: the persistent banded Cholesky again (two fronts at 7 separators: 18 dependent steps)
  if (cs::ba_bcr_sep_ok(B->w_max, R)) {
    // ... by block cyclic reduction, straight from the messages' blocks (bcr_kernels.hip): 7 separators = 3 levels instead of 18 dependent steps
    BA_TRY(hipMemsetAsync(B->d_sep_info.p, 0, 24 * sizeof(int), B->st));
    cs::ba_launch_bcr_sep(B->sep_msgs.p, B->msg_doubles, B->w_max, R, B->d_sep_off.p, B->d_sep_col.p, ns, B->sep_work.p, rhs, B->d_sep_info.p, B->st);
    BA_TRY(hipGetLastError());
  } else {
    cs::ba_launch_sep_assemble(B->sep_msgs.p, B->msg_doubles, B->w_max, R, B->d_sep_off.p, ns, LDs, B->sepS.p, rsep, B->st);
    BA_TRY(hipGetLastError());
    if (!coop_turn.owns_lock()) coop_turn.lock();
    BA_TRY(hipMemsetAsync(B->d_sep_info.p, 0, 24 * sizeof(int), B->st));
    cs::ba_launch_band_cholesky(B->sepS.p, B->sep_work.p, ns, LDs, rsep, B->d_sep_info.p, true, B->st);
    BA_TRY(hipGetLastError());
    if (fn) {
      BA_TRY(hipStreamSynchronize(B->st));
      coop_turn.unlock();
    }
    cs::ba_launch_sep_scatter(rsep, ns, R, B->d_sep_off.p, B->d_sep_col.p, rhs, B->st);
  }
  BA_MARK(B, B->sev[3]);
  cs::ba_launch_sep_backsolve(B->S.p, LD, B->int_work.p, B->int_c, B->int_n, B->zl, B->wl, B->zr, B->wr, B->sepY.p, rhs, B->d_int_info.p, B->st);
  BA_TRY(hipGetLastError());
  BA_MARK(B, B->ev[4]);
  cs::ba_launch_backsub(B->view, B->st);   // this rank's landmarks and cuboids see its own columns and the next separator only
  BA_TRY(hipGetLastError());
  // the solution vector: every rank contributes ITS columns (its separator as it solved it, its interior) and its cuboids' increments
  // behind the reduced system, zeros elsewhere -- every entry then has one source, so all ranks end with the same bits
  if (B->zl > 0) BA_TRY(hipMemsetAsync(rhs, 0, sizeof(double) * (size_t)B->zl, B->st));
  if (B->int_c + B->int_n < B->n_red) BA_TRY(hipMemsetAsync(rhs + B->int_c + B->int_n, 0, sizeof(double) * (size_t)(B->n_red - B->int_c - B->int_n), B->st));
  { int rc = coll_allreduce(B, fn, ctx, rhs, (size_t)B->n_pose); if (rc) return rc; }
  BA_MARK(B, B->ev[5]);
  cs::ba_launch_fail_flag(B->d_int_info.p, B->d_elim_fail.p, B->d_sep_info.p, B->d_scalars.p + 2, B->st);
  // both persistent factorisations of the trial report their own time-out (a team that was not co-resident): the interior's and the
  // separator system's -- the latter must not be mistaken for "not positive definite" (LM would raise lambda and retry, ~1 s a trial)
  BA_TRY(hipMemcpyAsync(B->h_status, B->d_int_info.p, sizeof(int), hipMemcpyDeviceToHost, B->st));
  BA_TRY(hipMemcpyAsync(B->h_status + 2, B->d_sep_info.p, sizeof(int), hipMemcpyDeviceToHost, B->st));
  BA_TRY(hipGetLastError());
  B->tm.n_solves++;
  if (defer) { if (coop_turn.owns_lock()) *defer = std::move(coop_turn); return CS_OK; }
  BA_TRY(hipMemcpyAsync(B->h_scalars + 2, B->d_scalars.p + 2, sizeof(double), hipMemcpyDeviceToHost, B->st));
  BA_TRY(hipStreamSynchronize(B->st));
  if (coop_turn.owns_lock()) coop_turn.unlock();
  if (B->h_status[0] == 0x7fffffff || B->h_status[2] == 0x7fffffff) { cs_set_error_ba("banded solver: team not co-resident (wait timed out); set CS_BA_FORCE_DENSE=1 on a shared device"); return CS_ERR_HIP; }
  if (B->h_scalars[2] != 0.0) *ok = false;
  return collect_solve_times(B);
}

}  // namespace

extern "C" {

int cs_ba_create(int device, cs_ba** out) {
  if (!out) return CS_ERR_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { cs_set_error_ba("no HIP device visible; libcubeslam_hip has no CPU fallback"); return CS_ERR_NO_DEVICE; }
  if (device < 0 || device >= n) { cs_set_error_ba("device index out of range"); return CS_ERR_INVALID_ARG; }
  BA_GUARD_BEGIN
  struct Guard { cs_ba* b; ~Guard() { if (b) cs_ba_destroy(b); } } g{new cs_ba()};   // freed on every early return
  cs_ba* B = g.b;
  B->device = device;
  { const char* e = getenv("CS_BA_FORCE_DENSE"); B->force_dense = (e && atoi(e)) ? 1 : 0; }  // diagnostics: rocSOLVER dense path
  BA_TRY(hipSetDevice(device));
  BA_TRY(hipStreamCreateWithFlags(&B->st, hipStreamNonBlocking));
  BA_TRY(hipStreamCreateWithFlags(&B->st2, hipStreamNonBlocking));   // (a high-priority side stream was measured in round 5: no effect on the kernels that run beside a device-filling one)
  BA_TRY(hipEventCreateWithFlags(&B->ev_fork, hipEventDisableTiming));
  BA_TRY(hipEventCreateWithFlags(&B->ev_join, hipEventDisableTiming));
  BA_TRY(hipStreamCreateWithFlags(&B->st3, hipStreamNonBlocking));
  BA_TRY(hipEventCreateWithFlags(&B->ev_join3, hipEventDisableTiming));
  BA_TRY(hipEventCreateWithFlags(&B->ev_upd, hipEventDisableTiming));
  for (auto& e : B->ev) BA_TRY(hipEventCreate(&e));
  for (auto& e : B->sev) BA_TRY(hipEventCreate(&e));
  BA_TRY(hipHostMalloc((void**)&B->h_lam, 2 * sizeof(double)));
  B->h_lam[0] = B->h_lam[1] = 0.0;
  { int rc0 = B->d_lam.alloc(2); if (rc0) return rc0; }
  BA_TRY(hipHostMalloc((void**)&B->h_trial, 8 * sizeof(double)));
  for (int i = 0; i < 8; i++) B->h_trial[i] = 0.0;
  BA_TRY(hipHostMalloc((void**)&B->h_status, 3 * sizeof(int)));   // [factorisation status, a cuboid block failed, separator system's status]
  B->h_status[0] = B->h_status[1] = B->h_status[2] = 0;
  BA_ROC(rocblas_create_handle(&B->blas));
  BA_ROC(rocblas_set_stream(B->blas, B->st));
  *out = B;
  g.b = nullptr;
  return CS_OK;
  BA_GUARD_END("cs_ba_create")
}

void cs_ba_destroy(cs_ba* B) {
  if (!B) return;
  (void)hipSetDevice(B->device);
  DBuf<double>* dd[] = {&B->cams, &B->points, &B->cubes, &B->cams_bak, &B->points_bak, &B->cubes_bak, &B->pm_uv, &B->pm_info, &B->pm_intr, &B->pm_huber,
                        &B->cm_uv, &B->cm_info, &B->cm_intr, &B->cm_huber, &B->ce_meas, &B->ce_info, &B->ce_Hcc, &B->ce_Hoo, &B->ce_Hco, &B->ce_bc, &B->ce_bo,
                        &B->oe_meas, &B->oe_info, &B->oe_Hii, &B->oe_Hjj, &B->oe_Hij, &B->oe_bi, &B->oe_bj, &B->Hcam, &B->bcam, &B->Hcub, &B->bcub, &B->Hll, &B->bl,
                        &B->W, &B->WD, &B->Dinv, &B->dbl, &B->S, &B->rhs, &B->xl, &B->chi_partial, &B->band_linv, &B->scale_partial, &B->pe_meas, &B->pe_info, &B->pe_K, &B->part_tiles, &B->part_coef, &B->cub_M, &B->cub_Dinv, &B->raw_uv, &B->raw_info, &B->raw_intr, &B->raw_huber, &B->sepY, &B->sep_msgs, &B->sepS, &B->int_work, &B->sep_work, &B->d_ce_rdelta, &B->d_oe_rdelta, &B->ext_cam36, &B->ext_cam6, &B->ext_cub81, &B->ext_cub9, &B->ext_pt9, &B->ext_pt3, &B->ext_Hij, &B->sp_L, &B->sp_xs, &B->sp_T};
  for (auto* d : dd) d->release();
  B->stage.release(); B->append_stage.release();
  DBuf<int>* di[] = {&B->d_ce_active, &B->d_oe_active, &B->d_cam_col, &B->d_cub_col, &B->d_pt_free, &B->pm_pt, &B->pm_cam, &B->pt_ptr, &B->cm_pm, &B->cm_pt, &B->cam_ptr, &B->d_ce_cam, &B->d_ce_cub,
                     &B->d_oe_i, &B->d_oe_j, &B->cam_ce_ptr, &B->cam_ce_idx, &B->cam_oei_ptr, &B->cam_oei_idx, &B->cam_oej_ptr, &B->cam_oej_idx, &B->cub_ce_ptr,
                     &B->cub_ce_idx, &B->pair_ptr, &B->pair_i1, &B->pair_i2, &B->ent_a, &B->ent_b, &B->d_run_lm, &B->d_seg_ptr, &B->d_seg_k, &B->d_seg_tile, &B->d_seg_slot, &B->d_run_e0, &B->d_seg_cam,
                     &B->d_gp_ptr, &B->d_gp_i1, &B->d_gp_i2, &B->d_gtile, &B->d_gcam_ptr, &B->d_gslot, &B->d_cubS_ptr, &B->d_cubS_cam, &B->d_ce_slot, &B->d_cub_tile, &B->d_cub_coef,
                     &B->d_elim_fail, &B->d_slotE_ptr, &B->d_slotE_idx, &B->d_cub_mine, &B->d_sep_off, &B->d_sep_col, &B->d_int_info, &B->d_sep_info, &B->d_pm_rk, &B->d_cm_rk, &B->d_ce_rk, &B->d_oe_rk, &B->d_ext_e4, &B->d_ext_order, &B->d_ext_gptr, &B->d_src, &B->sp_ndim, &B->sp_ncol, &B->sp_sptr, &B->sp_srow, &B->sp_sroff, &B->sp_prow, &B->sp_rbase, &B->sp_rent, &B->sp_rptr, &B->sp_rcol, &B->sp_rpos,
                     &B->sp_order, &B->sp_info, &B->sp_tcol};
  for (auto* d : di) d->release();
  B->sp_poff.release(); B->sp_done.release(); B->sp_xdone.release();
  B->d_info.release(); B->d_band_info.release();
  for (auto& e : B->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : B->sev) if (e) (void)hipEventDestroy(e);
  if (B->h_lam) (void)hipHostFree(B->h_lam);
  B->d_lam.release();
  if (B->h_status) (void)hipHostFree(B->h_status);
  if (B->h_trial) (void)hipHostFree(B->h_trial);
  if (B->h_scalars) (void)hipHostFree(B->h_scalars);
  if (B->comm) { (void)ncclCommDestroy(B->comm); g_comm_handles--; }
  B->d_scalars.release();
  if (B->blas) rocblas_destroy_handle(B->blas);
  if (B->ev_fork) (void)hipEventDestroy(B->ev_fork);
  if (B->ev_join) (void)hipEventDestroy(B->ev_join);
  if (B->st2) (void)hipStreamDestroy(B->st2);
  if (B->st3) (void)hipStreamDestroy(B->st3);
  if (B->ev_join3) (void)hipEventDestroy(B->ev_join3);
  if (B->ev_upd) (void)hipEventDestroy(B->ev_upd);
  if (B->st) (void)hipStreamDestroy(B->st);
  delete B;
}

static int cs_ba_set_vertices_impl(cs_ba* B, const double* cams7, const int* cam_fixed, int nc, const double* cuboids10, const int* cub_fixed, int no,
                       const double* points3, const int* pt_fixed, int np, int cuboids_first) {
  if (!B || nc < 0 || no < 0 || np < 0 || (nc && (!cams7 || !cam_fixed)) || (no && (!cuboids10 || !cub_fixed)) || (np && (!points3 || !pt_fixed))) return CS_ERR_INVALID_ARG;
  BA_TRY(hipSetDevice(B->device));
  BA_TRY(hipStreamSynchronize(B->st));       // (appended rows may still be on their way: cs_ba_append_* queues them on st)
  B->nc = nc; B->no = no; B->np = np; B->cuboids_first = cuboids_first;
  proj_edge_lists_invalidate(B);
  B->cam_fixed.assign(cam_fixed, cam_fixed + nc); B->cub_fixed.assign(cub_fixed, cub_fixed + no); B->pt_fixed.assign(pt_fixed, pt_fixed + np);
  // SE3Quat(Vector7d) normalises the rotation and makes w >= 0 (se3quat.h:68-71); cuboid::fromVector does not.
  std::vector<double> c(cams7, cams7 + 7 * (size_t)nc);
  for (int i = 0; i < nc; i++) { cs::Pose p = cs::pose_load(&c[7 * (size_t)i]); cs::pose_normalize(p); cs::pose_store(p, &c[7 * (size_t)i]); }
  int rc = B->cams.upload(c); if (rc) return rc;
  rc = B->cubes.upload(std::vector<double>(cuboids10, cuboids10 + 10 * (size_t)no)); if (rc) return rc;
  rc = B->points.upload(std::vector<double>(points3, points3 + 3 * (size_t)np)); if (rc) return rc;
  B->structure_dirty = true;
  return CS_OK;
}
int cs_ba_set_vertices(cs_ba* B, const double* cams7, const int* cam_fixed, int nc, const double* cuboids10, const int* cub_fixed, int no,
                       const double* points3, const int* pt_fixed, int np, int cuboids_first) {
  BA_GUARD_BEGIN
  return cs_ba_set_vertices_impl(B, cams7, cam_fixed, nc, cuboids10, cub_fixed, no, points3, pt_fixed, np, cuboids_first);
  BA_GUARD_END("cs_ba_set_vertices")
}

// ---- growing graphs (the reference's pattern: main_obj.cpp:802-803 adds a frame and calls optimize(5); g2o's seam is
// Solver::updateStructure, core/solver.h:62, core/block_solver.hpp:297-350).  New vertices and edges are appended behind the existing
// ones; the estimates that live on the device (possibly optimised there) stay untouched; the structure phase runs again on the
// next solve (49 ms at C4, a few ms at C3 -- DESIGN.md section 3).
// the appends' pinned staging arena (1 MB, made by the first append; rewound once the structure phase has waited for the stream)
static int append_arena(cs_ba* B) {
  if (B->append_stage.p) return CS_OK;
  return B->append_stage.begin((size_t)1 << 20);
}
static int cs_ba_append_vertices_impl(cs_ba* B, const double* cams7, const int* cam_fixed, int n_cams, const double* cuboids10, const int* cub_fixed, int n_cub,
                                      const double* points3, const int* pt_fixed, int n_pts) {
  if (!B || n_cams < 0 || n_cub < 0 || n_pts < 0 || (n_cams && (!cams7 || !cam_fixed)) || (n_cub && (!cuboids10 || !cub_fixed)) || (n_pts && (!points3 || !pt_fixed))) return CS_ERR_INVALID_ARG;
  BA_TRY(hipSetDevice(B->device));
  BA_TRY(hipStreamSynchronize(B->st));
  std::vector<double> c(cams7, cams7 + 7 * (size_t)n_cams);
  for (int i = 0; i < n_cams; i++) { cs::Pose p = cs::pose_load(&c[7 * (size_t)i]); cs::pose_normalize(p); cs::pose_store(p, &c[7 * (size_t)i]); }
  int rc;
  if ((rc = append_arena(B))) return rc;
  if ((rc = B->cams.append_ptr_staged(c.data(), 7 * (size_t)n_cams, B->append_stage, B->st)) || (rc = B->cubes.append_ptr_staged(cuboids10, 10 * (size_t)n_cub, B->append_stage, B->st)) ||
      (rc = B->points.append_ptr_staged(points3, 3 * (size_t)n_pts, B->append_stage, B->st))) return rc;
  B->cam_fixed.insert(B->cam_fixed.end(), cam_fixed, cam_fixed + n_cams);
  B->cub_fixed.insert(B->cub_fixed.end(), cub_fixed, cub_fixed + n_cub);
  B->pt_fixed.insert(B->pt_fixed.end(), pt_fixed, pt_fixed + n_pts);
  B->nc += n_cams; B->no += n_cub; B->np += n_pts;
  B->structure_dirty = true;
  return CS_OK;
}
int cs_ba_append_vertices(cs_ba* B, const double* cams7, const int* cam_fixed, int n_cams, const double* cuboids10, const int* cub_fixed, int n_cub,
                          const double* points3, const int* pt_fixed, int n_pts) {
  BA_GUARD_BEGIN
  return cs_ba_append_vertices_impl(B, cams7, cam_fixed, n_cams, cuboids10, cub_fixed, n_cub, points3, pt_fixed, n_pts);
  BA_GUARD_END("cs_ba_append_vertices")
}
// are the n new records all equal to the handle's reference record (the first record ever set)?  A few threads over the caller's arrays.
static void scan_uniform_records(cs_ba* B, bool first, const double* info4, const double* intr4, int n) {
  if (first) {
    const char* env = getenv("CS_BA_UNIFORM");   // (read per call: the parity test sets the same graph up both ways in one process)
    const bool on = !env || atoi(env) != 0;
    B->info_uniform = B->intr_uniform = on && n > 0;
    if (n > 0) { std::memcpy(B->uni8, info4, 32); std::memcpy(B->uni8 + 4, intr4, 32); }
  }
  if (!(B->info_uniform || B->intr_uniform) || n <= 0) return;
  const double t_scan = now_ms();
  const int nt = n > (1 << 16) ? 8 : 1;
  std::vector<char> ok_i(nt, 1), ok_k(nt, 1);
  auto work = [&](int t) {
    const size_t lo = (size_t)n * t / nt, hi = (size_t)n * (t + 1) / nt;
    bool a = B->info_uniform, b = B->intr_uniform;
    for (size_t k = lo; k < hi && (a || b); k++) {
      if (a && std::memcmp(info4 + 4 * k, B->uni8, 32) != 0) a = false;
      if (b && std::memcmp(intr4 + 4 * k, B->uni8 + 4, 32) != 0) b = false;
    }
    ok_i[t] = a; ok_k[t] = b;
  };
  if (nt == 1) work(0);
  else { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(work, t); for (auto& x : th) x.join(); }
  for (int t = 0; t < nt; t++) { B->info_uniform = B->info_uniform && ok_i[t]; B->intr_uniform = B->intr_uniform && ok_k[t]; }
  if (getenv("CS_BA_PROF")) fprintf(stderr, "[ba set edges] uniform-record scan of %d edges: %.2f ms\n", n, now_ms() - t_scan);
}
int cs_ba_append_edges_proj(cs_ba* B, int n, const int* pt, const int* cam, const double* uv, const double* info4, const double* intr4, const double* huber) {
  if (!B || n < 0 || (n && (!pt || !cam || !uv || !info4 || !intr4))) return CS_ERR_INVALID_ARG;
  BA_GUARD_BEGIN
  if (n == 0) return CS_OK;
  if (B->n_proj > 0 && (huber != nullptr) != B->have_huber) { cs_set_error_ba("cs_ba_append_edges_proj: Huber deltas must be given for all projection edges or for none"); return CS_ERR_INVALID_ARG; }
  BA_TRY(hipSetDevice(B->device));
  BA_TRY(hipStreamSynchronize(B->st));
  int rc;
  const bool first_edges = B->n_proj == 0;
  scan_uniform_records(B, first_edges, info4, intr4, n);
  if (first_edges) { B->raw_info_virtual = B->info_uniform; B->raw_intr_virtual = B->intr_uniform; B->raw_info.n = 0; B->raw_intr.n = 0; }
  // an appended edge with another record: the records of the edges so far are written out from the reference record, then kept per edge
  auto write_out = [&](DBuf<double>& raw, const double* rec4, bool& is_virtual) -> int {
    int r = raw.reserve(4 * (size_t)(B->n_proj + n));
    if (r) return r;
    cs::ba_launch_fill_rows4(raw.p, rec4, B->n_proj, B->st);
    BA_TRY(hipGetLastError());
    raw.n = 4 * (size_t)B->n_proj;
    is_virtual = false;
    return CS_OK;
  };
  if (B->raw_info_virtual && !B->info_uniform && (rc = write_out(B->raw_info, B->uni8, B->raw_info_virtual))) return rc;
  if (B->raw_intr_virtual && !B->intr_uniform && (rc = write_out(B->raw_intr, B->uni8 + 4, B->raw_intr_virtual))) return rc;
  if ((rc = append_arena(B))) return rc;
  if ((rc = B->raw_uv.append_ptr_staged(uv, 2 * (size_t)n, B->append_stage, B->st))) return rc;
  if (!B->raw_info_virtual && (rc = B->raw_info.append_ptr_staged(info4, 4 * (size_t)n, B->append_stage, B->st))) return rc;
  if (!B->raw_intr_virtual && (rc = B->raw_intr.append_ptr_staged(intr4, 4 * (size_t)n, B->append_stage, B->st))) return rc;
  if (huber) { rc = B->raw_huber.append_ptr_staged(huber, (size_t)n, B->append_stage, B->st); if (rc) return rc; }
  B->have_huber = huber != nullptr;
  B->e_pt.insert(B->e_pt.end(), pt, pt + n); B->e_cam.insert(B->e_cam.end(), cam, cam + n);
  if (!B->rk_proj.empty()) for (int k = 0; k < n; k++) B->rk_proj.push_back((huber && huber[k] > 0) ? cs::RK_HUBER : cs::RK_NONE);
  B->n_proj += n;
  B->structure_dirty = true;
  return CS_OK;
  BA_GUARD_END("cs_ba_append_edges_proj")
}
int cs_ba_append_edges_cuboid(cs_ba* B, int n, const int* cam, const int* cub, const double* meas10, const double* info81) {
  if (!B || n < 0 || (n && (!cam || !cub || !meas10 || !info81))) return CS_ERR_INVALID_ARG;
  BA_GUARD_BEGIN
  B->u3_cam.insert(B->u3_cam.end(), cam, cam + n); B->u3_cub.insert(B->u3_cub.end(), cub, cub + n);
  if (!B->rk_cub3.empty()) { B->rk_cub3.resize(B->u3_cam.size(), 0); B->rd_cub3.resize(B->u3_cam.size(), 0.0); }
  B->h_ce_meas.insert(B->h_ce_meas.end(), meas10, meas10 + 10 * (size_t)n); B->h_ce_info.insert(B->h_ce_info.end(), info81, info81 + 81 * (size_t)n);
  B->structure_dirty = true;
  return CS_OK;
  BA_GUARD_END("cs_ba_append_edges_cuboid")
}
int cs_ba_append_edges_cuboid_proj(cs_ba* B, int n, const int* cam, const int* cub, const double* meas4, const double* info16, const double* K9) {
  if (!B || n < 0 || (n && (!cam || !cub || !meas4 || !info16 || !K9))) return CS_ERR_INVALID_ARG;
  BA_GUARD_BEGIN
  B->up_cam.insert(B->up_cam.end(), cam, cam + n); B->up_cub.insert(B->up_cub.end(), cub, cub + n);
  if (!B->rk_cproj.empty()) { B->rk_cproj.resize(B->up_cam.size(), 0); B->rd_cproj.resize(B->up_cam.size(), 0.0); }
  B->h_pe_meas.insert(B->h_pe_meas.end(), meas4, meas4 + 4 * (size_t)n); B->h_pe_info.insert(B->h_pe_info.end(), info16, info16 + 16 * (size_t)n);
  B->h_pe_K.insert(B->h_pe_K.end(), K9, K9 + 9 * (size_t)n);
  B->structure_dirty = true;
  return CS_OK;
  BA_GUARD_END("cs_ba_append_edges_cuboid_proj")
}
int cs_ba_append_edges_odom(cs_ba* B, int n, const int* ci, const int* cj, const double* meas7, const double* info36) {
  if (!B || n < 0 || (n && (!ci || !cj || !meas7 || !info36))) return CS_ERR_INVALID_ARG;
  BA_GUARD_BEGIN
  B->oe_i.insert(B->oe_i.end(), ci, ci + n); B->oe_j.insert(B->oe_j.end(), cj, cj + n);
  const size_t m0 = B->h_oe_meas.size();
  B->h_oe_meas.insert(B->h_oe_meas.end(), meas7, meas7 + 7 * (size_t)n);
  for (int k = 0; k < n; k++) { cs::Pose p = cs::pose_load(&B->h_oe_meas[m0 + 7 * (size_t)k]); cs::pose_normalize(p); cs::pose_store(p, &B->h_oe_meas[m0 + 7 * (size_t)k]); }
  B->h_oe_info.insert(B->h_oe_info.end(), info36, info36 + 36 * (size_t)n);
  B->n_odom += n;
  if (!B->rk_odom.empty()) { B->rk_odom.resize(B->n_odom, 0); B->rd_odom.resize(B->n_odom, 0.0); }
  B->structure_dirty = true;
  return CS_OK;
  BA_GUARD_END("cs_ba_append_edges_odom")
}

static int cs_ba_set_estimates_impl(cs_ba* B, const double* cams7, const double* cuboids10, const double* points3) {
  if (!B) return CS_ERR_INVALID_ARG;
  BA_TRY(hipSetDevice(B->device));
  BA_TRY(hipStreamSynchronize(B->st));       // (appended rows may still be on their way)
  if (cams7 && B->nc) {
    std::vector<double> c(cams7, cams7 + 7 * (size_t)B->nc);
    for (int i = 0; i < B->nc; i++) { cs::Pose p = cs::pose_load(&c[7 * (size_t)i]); cs::pose_normalize(p); cs::pose_store(p, &c[7 * (size_t)i]); }
    BA_TRY(hipMemcpy(B->cams.p, c.data(), 56 * (size_t)B->nc, hipMemcpyHostToDevice));
  }
  if (cuboids10 && B->no) BA_TRY(hipMemcpy(B->cubes.p, cuboids10, 80 * (size_t)B->no, hipMemcpyHostToDevice));
  if (points3 && B->np) BA_TRY(hipMemcpy(B->points.p, points3, 24 * (size_t)B->np, hipMemcpyHostToDevice));
  B->have_system = false;
  return CS_OK;
}
int cs_ba_set_estimates(cs_ba* B, const double* cams7, const double* cuboids10, const double* points3) {
  BA_GUARD_BEGIN
  return cs_ba_set_estimates_impl(B, cams7, cuboids10, points3);
  BA_GUARD_END("cs_ba_set_estimates")
}

static int cs_ba_set_edges_proj_impl(cs_ba* B, int n, const int* pt, const int* cam, const double* uv, const double* info4, const double* intr4, const double* huber) {
  if (!B || n < 0 || (n && (!pt || !cam || !uv || !info4 || !intr4))) return CS_ERR_INVALID_ARG;
  B->n_proj = n;
  proj_edge_lists_invalidate(B);
  B->e_pt.assign(pt, pt + n); B->e_cam.assign(cam, cam + n);
  BA_TRY(hipSetDevice(B->device));
  BA_TRY(hipStreamSynchronize(B->st));       // (appended rows may still be on their way)
  int rc;
  scan_uniform_records(B, true, info4, intr4, n);
  if ((rc = B->raw_uv.upload_ptr(uv, 2 * (size_t)n))) return rc;
  B->raw_info_virtual = B->info_uniform; B->raw_intr_virtual = B->intr_uniform;
  if (B->raw_info_virtual) B->raw_info.n = 0; else if ((rc = B->raw_info.upload_ptr(info4, 4 * (size_t)n))) return rc;
  if (B->raw_intr_virtual) B->raw_intr.n = 0; else if ((rc = B->raw_intr.upload_ptr(intr4, 4 * (size_t)n))) return rc;
  B->have_huber = huber != nullptr;
  B->rk_proj.clear();
  if (huber) { rc = B->raw_huber.upload_ptr(huber, (size_t)n); if (rc) return rc; } else B->raw_huber.release();
  B->structure_dirty = true;
  return CS_OK;
}
int cs_ba_set_edges_proj(cs_ba* B, int n, const int* pt, const int* cam, const double* uv, const double* info4, const double* intr4, const double* huber) {
  BA_GUARD_BEGIN
  return cs_ba_set_edges_proj_impl(B, n, pt, cam, uv, info4, intr4, huber);
  BA_GUARD_END("cs_ba_set_edges_proj")
}

static int cs_ba_set_edges_cuboid_impl(cs_ba* B, int n, const int* cam, const int* cub, const double* meas10, const double* info81) {
  if (!B || n < 0 || (n && (!cam || !cub || !meas10 || !info81))) return CS_ERR_INVALID_ARG;
  B->u3_cam.assign(cam, cam + n); B->u3_cub.assign(cub, cub + n);
  B->rk_cub3.clear(); B->rd_cub3.clear();
  B->h_ce_meas.assign(meas10, meas10 + 10 * (size_t)n); B->h_ce_info.assign(info81, info81 + 81 * (size_t)n);
  B->structure_dirty = true;
  return CS_OK;
}
int cs_ba_set_edges_cuboid(cs_ba* B, int n, const int* cam, const int* cub, const double* meas10, const double* info81) {
  BA_GUARD_BEGIN
  return cs_ba_set_edges_cuboid_impl(B, n, cam, cub, meas10, info81);
  BA_GUARD_END("cs_ba_set_edges_cuboid")
}

static int cs_ba_set_edges_cuboid_proj_impl(cs_ba* B, int n, const int* cam, const int* cub, const double* meas4, const double* info16, const double* K9) {
  if (!B || n < 0 || (n && (!cam || !cub || !meas4 || !info16 || !K9))) return CS_ERR_INVALID_ARG;
  B->up_cam.assign(cam, cam + n); B->up_cub.assign(cub, cub + n);
  B->rk_cproj.clear(); B->rd_cproj.clear();
  B->h_pe_meas.assign(meas4, meas4 + 4 * (size_t)n); B->h_pe_info.assign(info16, info16 + 16 * (size_t)n); B->h_pe_K.assign(K9, K9 + 9 * (size_t)n);
  B->structure_dirty = true;
  return CS_OK;
}
int cs_ba_set_edges_cuboid_proj(cs_ba* B, int n, const int* cam, const int* cub, const double* meas4, const double* info16, const double* K9) {
  BA_GUARD_BEGIN
  return cs_ba_set_edges_cuboid_proj_impl(B, n, cam, cub, meas4, info16, K9);
  BA_GUARD_END("cs_ba_set_edges_cuboid_proj")
}

static int cs_ba_set_edges_odom_impl(cs_ba* B, int n, const int* ci, const int* cj, const double* meas7, const double* info36) {
  if (!B || n < 0 || (n && (!ci || !cj || !meas7 || !info36))) return CS_ERR_INVALID_ARG;
  B->n_odom = n;
  B->oe_i.assign(ci, ci + n); B->oe_j.assign(cj, cj + n);
  B->rk_odom.clear(); B->rd_odom.clear();
  B->h_oe_meas.assign(meas7, meas7 + 7 * (size_t)n);
  for (int k = 0; k < n; k++) { cs::Pose p = cs::pose_load(&B->h_oe_meas[7 * (size_t)k]); cs::pose_normalize(p); cs::pose_store(p, &B->h_oe_meas[7 * (size_t)k]); }
  B->h_oe_info.assign(info36, info36 + 36 * (size_t)n);
  B->structure_dirty = true;
  return CS_OK;
}
int cs_ba_set_edges_odom(cs_ba* B, int n, const int* ci, const int* cj, const double* meas7, const double* info36) {
  BA_GUARD_BEGIN
  return cs_ba_set_edges_odom_impl(B, n, ci, cj, meas7, info36);
  BA_GUARD_END("cs_ba_set_edges_odom")
}

// ---- external (host-evaluated) edges: the CPU path for edge types the library does not evaluate.  The caller runs such an edge through
// its own virtuals -- computeError, linearizeOplus, constructQuadraticForm (core/optimizable_graph.h:394-454, base_binary_edge.hpp:54-205,
// base_unary_edge.hpp:42-123) -- and hands over what they accumulate.
int cs_ba_set_external_edges(cs_ba* B, int n, const int* class_i, const int* idx_i, const int* class_j, const int* idx_j) {
  if (!B || n < 0 || (n && (!class_i || !idx_i || !class_j || !idx_j))) return CS_ERR_INVALID_ARG;
  BA_GUARD_BEGIN
  // an edge whose two ends are the same vertex has no off-diagonal block (its whole quadratic form belongs to the vertex's diagonal terms,
  // cs_ba_set_external_terms' cam36 / cub81 / pt9): the off-diagonal kernel would address the upper triangle of a diagonal block
  for (int k = 0; k < n; k++)
    if (class_i[k] == class_j[k] && idx_i[k] == idx_j[k]) { cs_set_error_ba("cs_ba_set_external_edges: edge " + std::to_string(k) + " joins a vertex to itself; add its terms to the vertex's diagonal block instead"); return CS_ERR_INVALID_ARG; }
  B->ext_n = n;
  B->ext_e4.resize(4 * (size_t)n);
  for (int k = 0; k < n; k++) { B->ext_e4[4 * k] = class_i[k]; B->ext_e4[4 * k + 1] = idx_i[k]; B->ext_e4[4 * k + 2] = class_j[k]; B->ext_e4[4 * k + 3] = idx_j[k]; }
  B->ext_terms_set = false;
  B->structure_dirty = true;
  return CS_OK;
  BA_GUARD_END("cs_ba_set_external_edges")
}
static int cs_ba_set_external_terms_impl(cs_ba* B, const double* cam36, const double* cam6, const double* cub81, const double* cub9, const double* pt9, const double* pt3,
                                         const double* Hij81, double chi2) {
  if (!B || (cam36 && !cam6) || (cub81 && !cub9) || (pt9 && !pt3) || (B->ext_n > 0 && !Hij81)) return CS_ERR_INVALID_ARG;
  BA_TRY(hipSetDevice(B->device));
  BA_TRY(hipStreamSynchronize(B->st));     // the previous linearisation may still read the buffers
  int rc;
  auto put = [&](DBuf<double>& d, const double* h, size_t n) -> int { return h ? d.upload_ptr(h, n) : d.alloc(n, B->st); };
  if ((rc = put(B->ext_cam36, cam36, 36 * (size_t)B->nc)) || (rc = put(B->ext_cam6, cam6, 6 * (size_t)B->nc)) || (rc = put(B->ext_cub81, cub81, 81 * (size_t)B->no)) ||
      (rc = put(B->ext_cub9, cub9, 9 * (size_t)B->no)) || (rc = put(B->ext_pt9, pt9, 9 * (size_t)B->np)) || (rc = put(B->ext_pt3, pt3, 3 * (size_t)B->np)) ||
      (rc = put(B->ext_Hij, Hij81, 81 * (size_t)B->ext_n))) return rc;
  BA_TRY(hipStreamSynchronize(B->st));
  B->ext_has_cam = cam36 != nullptr; B->ext_has_cub = cub81 != nullptr; B->ext_has_pt = pt9 != nullptr;
  if (Hij81) B->h_ext_Hij.assign(Hij81, Hij81 + 81 * (size_t)B->ext_n); else B->h_ext_Hij.clear();
  B->ext_chi2 = chi2;
  B->ext_terms_set = true;
  B->have_system = false;
  return CS_OK;
}
int cs_ba_set_external_terms(cs_ba* B, const double* cam36, const double* cam6, const double* cub81, const double* cub9, const double* pt9, const double* pt3,
                             const double* Hij81, double chi2) {
  BA_GUARD_BEGIN
  return cs_ba_set_external_terms_impl(B, cam36, cam6, cub81, cub9, pt9, pt3, Hij81, chi2);
  BA_GUARD_END("cs_ba_set_external_terms")
}
int cs_ba_set_external_chi2(cs_ba* B, double chi2) {
  if (!B) return CS_ERR_INVALID_ARG;
  if (!B->ext_terms_set) { cs_set_error_ba("cs_ba_set_external_chi2: call cs_ba_set_external_terms first"); return CS_ERR_NOT_RUN; }
  B->ext_chi2 = chi2;
  return CS_OK;
}
int cs_ba_set_external_callback(cs_ba* B, cs_external_fn fn, void* ctx) {
  if (!B) return CS_ERR_INVALID_ARG;
  B->ext_fn = fn; B->ext_ctx = ctx;
  return CS_OK;
}

// Robust kernels of one edge class (OptimizableGraph::Edge::setRobustKernel, core/optimizable_graph.h:419-423; the kernels:
// core/robust_kernel_impl.cpp:78-165).  Replaces the class's kernels; n = the class's edge count.
static int cs_ba_set_robust_kernels_impl(cs_ba* B, int edge_class, int n, const int* kind, const double* delta) {
  if (!B || n < 0 || (n && kind && !delta)) return CS_ERR_INVALID_ARG;
  const int have = edge_class == CS_EDGE_PROJ ? B->n_proj : edge_class == CS_EDGE_CUBOID ? (int)B->u3_cam.size() : edge_class == CS_EDGE_CUBOID_PROJ ? (int)B->up_cam.size()
                 : edge_class == CS_EDGE_ODOM ? B->n_odom : -1;
  if (have < 0) { cs_set_error_ba("cs_ba_set_robust_kernels: unknown edge class"); return CS_ERR_INVALID_ARG; }
  if (!kind) n = have;         // removing the class's kernels: the count is the library's own (n is ignored)
  if (n != have) { cs_set_error_ba("cs_ba_set_robust_kernels: n must equal the number of edges of the class (set the edges first)"); return CS_ERR_INVALID_ARG; }
  std::vector<int> kk(n, 0); std::vector<double> dd(n, 0.0);
  for (int k = 0; k < n && kind; k++) {
    if (kind[k] < 0 || kind[k] >= cs::RK_KINDS) { cs_set_error_ba("cs_ba_set_robust_kernels: unknown kernel kind"); return CS_ERR_INVALID_ARG; }
    if (kind[k] != cs::RK_NONE && !(delta[k] > 0)) { cs_set_error_ba("cs_ba_set_robust_kernels: a kernel needs delta > 0"); return CS_ERR_INVALID_ARG; }
    kk[k] = kind[k]; dd[k] = kind[k] != cs::RK_NONE ? delta[k] : 0.0;
  }
  if (edge_class == CS_EDGE_PROJ) {
    BA_TRY(hipSetDevice(B->device));
    BA_TRY(hipStreamSynchronize(B->st));
    int rc = B->raw_huber.upload_ptr(dd.data(), (size_t)n); if (rc) return rc;    // delta per edge, 0 = none
    B->have_huber = true;
    B->rk_proj = kk;
  } else if (edge_class == CS_EDGE_CUBOID) { B->rk_cub3 = kk; B->rd_cub3 = dd; }
  else if (edge_class == CS_EDGE_CUBOID_PROJ) { B->rk_cproj = kk; B->rd_cproj = dd; }
  else { B->rk_odom = kk; B->rd_odom = dd; }
  B->structure_dirty = true;
  return CS_OK;
}
int cs_ba_set_robust_kernels(cs_ba* B, int edge_class, int n, const int* kind, const double* delta) {
  BA_GUARD_BEGIN
  return cs_ba_set_robust_kernels_impl(B, edge_class, n, kind, delta);
  BA_GUARD_END("cs_ba_set_robust_kernels")
}

static int cs_ba_compute_errors_impl(cs_ba* B, double* chi2) {
  if (!B || !chi2) return CS_ERR_INVALID_ARG;
  BA_TRY(hipSetDevice(B->device));
  int rc = finalize_structure(B); if (rc) return rc;
  double t0 = now_ms();
  rc = chi2_device(B, chi2);
  if (B->ext_terms_set) *chi2 += B->ext_chi2;      // the host-evaluated edges' share (cs_ba_set_external_terms / _chi2)
  B->tm.errors_ms += now_ms() - t0;
  return rc;
}
int cs_ba_compute_errors(cs_ba* B, double* chi2) {
  BA_GUARD_BEGIN
  return cs_ba_compute_errors_impl(B, chi2);
  BA_GUARD_END("cs_ba_compute_errors")
}

static int cs_ba_build_system_impl(cs_ba* B) {
  if (!B) return CS_ERR_INVALID_ARG;
  BA_TRY(hipSetDevice(B->device));
  int rc = finalize_structure(B); if (rc) return rc;
  rc = build_system_device(B); if (rc) return rc;
  rc = fetch_b(B); if (rc) return rc;
  debug_nan_scan(B, "after cs_ba_build_system");
  return CS_OK;
}
int cs_ba_build_system(cs_ba* B) {
  BA_GUARD_BEGIN
  return cs_ba_build_system_impl(B);
  BA_GUARD_END("cs_ba_build_system")
}

static int cs_ba_solve_impl(cs_ba* B, double lambda, int* pd) {
  if (!B) return CS_ERR_INVALID_ARG;
  BA_TRY(hipSetDevice(B->device));
  if (B->structure_dirty || !B->have_system) { cs_set_error_ba("cs_ba_solve: call cs_ba_build_system first"); return CS_ERR_NOT_RUN; }
  bool ok = false;
  int rc = solve_device(B, lambda, &ok); if (rc) return rc;
  if (ok) debug_nan_scan(B, "after cs_ba_solve");
  if (pd) *pd = ok ? 1 : 0;
  return ok ? fetch_x(B) : CS_OK;
}
int cs_ba_solve(cs_ba* B, double lambda, int* pd) {
  BA_GUARD_BEGIN
  return cs_ba_solve_impl(B, lambda, pd);
  BA_GUARD_END("cs_ba_solve")
}

int cs_ba_update(cs_ba* B) {
  if (!B) return CS_ERR_INVALID_ARG;
  BA_GUARD_BEGIN
  if (B->structure_dirty || !B->have_system) { cs_set_error_ba("cs_ba_update: no solution to apply (call cs_ba_build_system and cs_ba_solve first)"); return CS_ERR_NOT_RUN; }
  BA_TRY(hipSetDevice(B->device));
  double t0 = now_ms();
  cs::ba_launch_update(B->view, B->st);
  BA_TRY(hipGetLastError());
  BA_TRY(hipStreamSynchronize(B->st));
  B->tm.update_ms += now_ms() - t0;
  debug_nan_scan(B, "after cs_ba_update");
  return CS_OK;
  BA_GUARD_END("cs_ba_update")
}

int cs_ba_push(cs_ba* B) {
  if (!B) return CS_ERR_INVALID_ARG;
  BA_GUARD_BEGIN
  BA_TRY(hipSetDevice(B->device));
  int rc = finalize_structure(B); if (rc) return rc;
  if (B->nc) BA_TRY(hipMemcpyAsync(B->cams_bak.p, B->cams.p, 56 * (size_t)B->nc, hipMemcpyDeviceToDevice, B->st));
  if (B->np) BA_TRY(hipMemcpyAsync(B->points_bak.p, B->points.p, 24 * (size_t)B->np, hipMemcpyDeviceToDevice, B->st));
  if (B->no) BA_TRY(hipMemcpyAsync(B->cubes_bak.p, B->cubes.p, 80 * (size_t)B->no, hipMemcpyDeviceToDevice, B->st));
  return CS_OK;
  BA_GUARD_END("cs_ba_push")
}

int cs_ba_pop(cs_ba* B) {
  if (!B) return CS_ERR_INVALID_ARG;
  BA_GUARD_BEGIN
  if (B->structure_dirty) { cs_set_error_ba("cs_ba_pop: nothing was pushed since the graph changed"); return CS_ERR_NOT_RUN; }
  BA_TRY(hipSetDevice(B->device));
  if (B->nc) BA_TRY(hipMemcpyAsync(B->cams.p, B->cams_bak.p, 56 * (size_t)B->nc, hipMemcpyDeviceToDevice, B->st));
  if (B->np) BA_TRY(hipMemcpyAsync(B->points.p, B->points_bak.p, 24 * (size_t)B->np, hipMemcpyDeviceToDevice, B->st));
  if (B->no) BA_TRY(hipMemcpyAsync(B->cubes.p, B->cubes_bak.p, 80 * (size_t)B->no, hipMemcpyDeviceToDevice, B->st));
  return CS_OK;
  BA_GUARD_END("cs_ba_pop")
}

// optimization_algorithm_levenberg.cpp:61-163 + sparse_optimizer.cpp:354-419
int cs_ba_set_shard(cs_ba* B, int rank, int n_ranks) {
  if (!B || n_ranks < 1 || rank < 0 || rank >= n_ranks) return CS_ERR_INVALID_ARG;
  B->shard_rank = rank; B->shard_n = n_ranks;
  B->structure_dirty = true;
  return CS_OK;
}

int cs_ba_shard_landmark_owners(int n_ranks, int n_cams, int n_points, int n_proj, const int* e_pt, const int* e_cam, int* owner_out) {
  if (n_ranks < 1 || n_cams < 0 || n_points < 0 || n_proj < 0 || (n_proj && (!e_pt || !e_cam)) || (n_points && !owner_out)) return CS_ERR_INVALID_ARG;
  std::vector<int> owner;
  landmark_owners(n_ranks, n_cams, n_points, n_proj, e_pt, e_cam, owner);
  std::copy(owner.begin(), owner.end(), owner_out);
  return CS_OK;
}

int cs_ba_optimize(cs_ba* B, int iterations, int* iterations_done, double* chi_hist, double* lambda_hist, int* trials_hist, int cap) {
  return cs_ba_optimize_sharded(B, iterations, nullptr, nullptr, iterations_done, chi_hist, lambda_hist, trials_hist, cap);
}

static int cs_ba_optimize_sharded_impl(cs_ba* B, int iterations, cs_allreduce_fn fn, void* ctx, int* iterations_done, double* chi_hist, double* lambda_hist, int* trials_hist, int cap) {
  if (!B || iterations < 0) return CS_ERR_INVALID_ARG;
  if (B->shard_n > 1 && !fn && !B->comm) { cs_set_error_ba("sharded problem needs cs_ba_comm_init() (RCCL) or an all-reduce callback"); return CS_ERR_INVALID_ARG; }
  const bool cb = fn && B->shard_n > 1;          // collectives through the caller's callback (CPU / gloo tests, ranks as threads)
  const bool rccl = !fn && B->comm != nullptr;   // collectives issued here, on this handle's stream
  BA_TRY(hipSetDevice(B->device));
  int rc = finalize_structure(B); if (rc) return rc;
  // host scalars: sum / max over the ranks
  auto reduce_host = [&](double* v, size_t n, int op) -> int {
    if (cb) return fn(ctx, v, n, 0, op);
    if (rccl) {
      if (n > B->scalars_cap) return 1;
      std::memcpy(B->h_scalars, v, n * sizeof(double));
      if (hipMemcpyAsync(B->d_scalars.p, B->h_scalars, n * sizeof(double), hipMemcpyHostToDevice, B->st) != hipSuccess) return 1;
      if (ncclAllReduce(B->d_scalars.p, B->d_scalars.p, n, ncclDouble, op == 0 ? ncclSum : ncclMax, B->comm, B->st) != ncclSuccess) return 1;
      if (hipMemcpyAsync(B->h_scalars, B->d_scalars.p, n * sizeof(double), hipMemcpyDeviceToHost, B->st) != hipSuccess) return 1;
      if (hipStreamSynchronize(B->st) != hipSuccess) return 1;
      std::memcpy(v, B->h_scalars, n * sizeof(double));
    }
    return 0;
  };
  // One trial entirely on the stream (banded solver, no callback): damped solve, LM scale term, update, chi2 of the new state and --
  // sharded -- ONE RCCL all-reduce of the pair [chi2, scale] are queued back to back; the host synchronises once per trial.
  const bool stream_flow = !cb && B->band_ld > 0 && B->n_red > 0;
  double t_begin = now_ms();
  if (B->shard_n > 1) {
    // every rank decides the solver layout from its own structure phase (and device query): the collectives below only match if
    // they all decided alike
    const double q[5] = {(double)B->n_red, (double)B->band_ld, B->elim ? 1.0 : 0.0, B->sep_mode ? 1.0 : 0.0, (double)B->n_sep};
    double v[10];
    for (int i = 0; i < 5; i++) { v[i] = q[i]; v[5 + i] = -q[i]; }
    if (reduce_host(v, 10, 1)) { cs_set_error_ba("all-reduce failed"); return CS_ERR_HIP; }
    for (int i = 0; i < 5; i++)
      if (v[i] != q[i] || v[5 + i] != -q[i]) { cs_set_error_ba("sharded BA: the ranks disagree on the solver layout (reduced size / bandwidth / cuboid elimination / separator mode); check CS_BA_* environment variables and devices"); return CS_ERR_INVALID_ARG; }
  }
  // external (host-evaluated) edges: their terms depend on the state, so the caller's callback re-evaluates them before every
  // linearisation (want_system = 1: cs_ba_set_external_terms) and after every trial's update (want_system = 0: cs_ba_set_external_chi2)
  const bool ext_active = B->ext_n > 0 || B->ext_terms_set || B->ext_fn;
  if (ext_active && !B->ext_fn) { cs_set_error_ba("cs_ba_optimize: the graph has external (host-evaluated) edges but no cs_ba_set_external_callback; drive the stepwise calls instead"); return CS_ERR_NOT_RUN; }
  auto ext_refresh = [&](int want_system) -> int {
    if (!ext_active) return CS_OK;
    BA_TRY(hipStreamSynchronize(B->st));
    if (B->ext_fn(B->ext_ctx, B, want_system) != 0) { cs_set_error_ba("external-edge callback failed"); return CS_ERR_INVALID_ARG; }
    if (!B->ext_terms_set) { cs_set_error_ba("external-edge callback did not call cs_ba_set_external_terms"); return CS_ERR_NOT_RUN; }
    return CS_OK;
  };
  const bool fuse_lin_env = [] { const char* e = getenv("CS_BA_FUSE_LIN"); return !e || atoi(e) != 0; }();     // (read per call: the tests hold the two paths to each other in one process)
  const bool fuse_lin_ok = fuse_lin_env && stream_flow && B->fused && B->shard_n == 1 && !ext_active && B->n_proj > 0 &&
                           B->n_seg == std::max(std::max(std::max(B->seg_class[0], B->seg_class[1]), std::max(B->seg_class[2], B->seg_class[3])), B->seg_class[4]) &&
                           getenv("CS_BA_DEBUG_NAN") == nullptr;
  struct FuseLinGuard { cs_ba* b; ~FuseLinGuard() { b->view.fuse_lin = 0; } } fuse_lin_guard{B};   // (every other entry point linearises with the classic kernels)
  // The NEXT iteration's linearisation is queued right behind a trial, before the host has the trial's verdict: nearly every trial is accepted, and
  // the ~70 us the host needs to read the scalars, decide and queue again then pass while the device linearises (the timeline showed the device
  // idle for exactly that long after every trial).  A rejected trial pops the estimates and linearises the restored state once more -- the same
  // kernels on the same state: the same bits as the system the speculation overwrote.  Off whenever something else must see the stream between
  // trials: external (host-evaluated) edges, sharded runs, the NaN scans, the stage split's phase marks.
  const bool can_spec = stream_flow && B->shard_n == 1 && !ext_active && !B->stage_timing && !debug_nan_enabled();
  bool spec_lin = false;         // this iteration's system is already queued (speculated behind the last iteration's accepted trial)
  // a trial's scalars straight from its last kernel into pinned memory, polled here (single rank: no collective behind the kernel)
  const bool direct_scalars = stream_flow && !rccl && B->h_trial != nullptr;
  auto wait_trial = [&](double seq) -> int {
    volatile double* h = B->h_trial;
    const double t_w = now_ms();
    unsigned spins = 0;
    while (h[3] != seq) {
      if ((++spins & 0xfffu) == 0 && now_ms() - t_w > 5000.0) {      // five seconds without the word: ask the runtime what happened
        BA_TRY(hipStreamSynchronize(B->st));
        if (h[3] != seq) { cs_set_error_ba("cs_ba_optimize: the trial's scalars never arrived in pinned memory"); return CS_ERR_HIP; }
        break;
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return CS_OK;
  };
  double lambda = -1, ni = 2;
  int nBad = 0, done = 0;
  double carriedChi = 0;
  bool have_carried = false;     // chi2 of the current state is known from the previous iteration's last trial
  for (int it = 0; it < iterations; it++) {
    double currentChi = 0;
    double t0 = now_ms();
    rc = ext_refresh(1); if (rc) return rc;        // (the terms of this iteration's linearisation, and the chi2 share at the current state)
    if (have_carried) {
      // computeActiveErrors + activeRobustChi2 at the top of an iteration (:67-69) would re-evaluate the state the last trial left:
      // the accepted trial's chi2 (same kernels, same state, fixed-shape sums: the same bits), or -- after ten rejections --
      // the popped state's, which is the previous currentChi
      currentChi = carriedChi;
    } else {
      rc = chi2_device(B, &currentChi); if (rc) return rc;
      if (ext_active) currentChi += B->ext_chi2;
      if (reduce_host(&currentChi, 1, 0)) { cs_set_error_ba("all-reduce failed"); return CS_ERR_HIP; }
      B->tm.errors_ms += now_ms() - t0;
    }
    double tempChi = currentChi, iniChi = currentChi;
    // From the second iteration on the landmark side of the projection edges is linearised inside the trial's Schur kernels
    // (ba_lin_schur_kernel: one pass over the edges, H_pl never read back); the first iteration needs H_ll for lambda's initial value
    // before any trial.  Same bits either way (CS_BA_FUSE_LIN=0: the classic pair of kernels throughout).
    B->view.fuse_lin = (it > 0 && fuse_lin_ok) ? 1 : 0;
    if (!spec_lin) { rc = build_system_device(B); if (rc) return rc; }
    spec_lin = false;
    debug_nan_scan(B, "cs_ba_optimize: after the linearisation");
    if (it == 0 && B->shard_n == 1) {  // computeLambdaInit (:166-180): tau * max |H_jj| over all non-fixed vertices, landmarks included
      // one rank: the maximum is taken where the blocks are (ba_max_diag_kernel) and eight bytes come back, instead of every H_cc / H_oo / H_ll
      // block (14 MB at C4, 0.6 ms of every cs_ba_optimize call)
      BA_TRY(hipMemsetAsync(B->d_scalars.p, 0, sizeof(double), B->st));
      cs::ba_launch_max_diag(B->view, B->d_scalars.p, B->st);
      BA_TRY(hipGetLastError());
      BA_TRY(hipMemcpyAsync(B->h_scalars, B->d_scalars.p, sizeof(double), hipMemcpyDeviceToHost, B->st));
      BA_TRY(hipStreamSynchronize(B->st));
      lambda = B->user_lambda_init > 0 ? B->user_lambda_init : 1e-5 * B->h_scalars[0];      // (:168-169: a user value wins)
      ni = 2; nBad = 0;
    } else if (it == 0) {
      BA_TRY(hipStreamSynchronize(B->st));   // the copies below run on the NULL stream, which B->st does not order with
      std::vector<double> hc(36 * (size_t)B->nc), ho(81 * (size_t)B->no), hl(9 * (size_t)B->np);
      if (B->nc) BA_TRY(hipMemcpy(hc.data(), B->Hcam.p, 8 * hc.size(), hipMemcpyDeviceToHost));
      if (B->no) BA_TRY(hipMemcpy(ho.data(), B->Hcub.p, 8 * ho.size(), hipMemcpyDeviceToHost));
      if (B->np) BA_TRY(hipMemcpy(hl.data(), B->Hll.p, 8 * hl.size(), hipMemcpyDeviceToHost));
      // pose diagonals are partial sums on every rank: sum them before taking the maximum
      std::vector<double> pd(std::max(1, B->n_pose), 0.0);
      for (int i = 0; i < B->nc; i++) if (B->cam_col[i] >= 0) for (int d = 0; d < 6; d++) pd[B->cam_col[i] + d] = hc[36 * (size_t)i + 7 * d];
      for (int i = 0; i < B->no; i++) if (B->cub_col[i] >= 0) for (int d = 0; d < 9; d++) pd[B->cub_col[i] + d] = ho[81 * (size_t)i + 10 * d];
      if (reduce_host(pd.data(), pd.size(), 0)) { cs_set_error_ba("all-reduce failed"); return CS_ERR_HIP; }
      double md = 0;
      for (int i = 0; i < B->n_pose; i++) md = std::max(std::fabs(pd[i]), md);
      for (int i = 0; i < B->np; i++) if (B->pt_lm[i] >= 0) for (int d = 0; d < 3; d++) md = std::max(std::fabs(hl[9 * (size_t)i + 4 * d]), md);
      if (reduce_host(&md, 1, 1)) { cs_set_error_ba("all-reduce failed"); return CS_ERR_HIP; }
      lambda = B->user_lambda_init > 0 ? B->user_lambda_init : 1e-5 * md;
      ni = 2; nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      if (!stream_flow) { rc = cs_ba_push(B); if (rc) return rc; }      // (stream flow: the update kernel below saves the estimates it replaces)
      bool ok2 = true;
      bool spec_now = false;
      double scale = 0;
      if (stream_flow) {
        std::unique_lock<std::mutex> turn;
        // x^T (lambda x + b): b is a per-rank partial sum, x is replicated for the poses -- their lambda x^2 term is counted once
        // (rank 0); a landmark's terms live on exactly one rank.  A failed factorisation leaves garbage in x; the update below then
        // writes garbage estimates, which the pop restores (the decision is taken after the synchronisation).
        // [chi2, scale term, "a factorisation failed somewhere"]: one message; every rank takes the same accept / reject branch.
        auto enqueue_trial = [&](std::unique_lock<std::mutex>* t) -> int {
          bool okq = true;
          int rq = solve_device(B, lambda, &okq, nullptr, nullptr, t); if (rq) return rq;
          cs::ba_launch_scale_update(B->view, B->d_lam.p, B->scale_partial.p, B->st, B->cams_bak.p, B->points_bak.p, B->cubes_bak.p);
          if (spec_now) BA_TRY(hipEventRecord(B->ev_upd, B->st));      // the state the next linearisation reads is final from here
          BA_MARK(B, B->ev[6]);
          cs::ba_launch_chi2(B->view, B->nb_chi, B->st);
          if (B->sep_mode && B->shard_n > 1) cs::ba_launch_sum2(B->chi_partial.p, B->n_chi_partials, B->scale_partial.p, cs::ba_scale_blocks(), B->d_scalars.p, B->st);   // (separator mode set its flag itself: three status words)
          else cs::ba_launch_sum2_flag(B->chi_partial.p, B->n_chi_partials, B->scale_partial.p, cs::ba_scale_blocks(), B->d_band_info.p, B->d_elim_fail.p, B->d_scalars.p, B->st,
                                       direct_scalars ? B->h_trial : nullptr, direct_scalars ? (B->trial_seq += 1.0) : 0.0);
          BA_TRY(hipGetLastError());
          if (rccl) BA_NCCL(ncclAllReduce(B->d_scalars.p, B->d_scalars.p, 3, ncclDouble, ncclSum, B->comm, B->st));
          if (!direct_scalars) BA_TRY(hipMemcpyAsync(B->h_scalars, B->d_scalars.p, 3 * sizeof(double), hipMemcpyDeviceToHost, B->st));
          BA_MARK(B, B->ev[7]);
          return CS_OK;
        };
        // (The sequence is the same ~30 launches, fills and copies for every trial of a structure -- lambda is read from device memory -- and was
        // replayed as one hipGraph in round 4: measured no faster on MI355X / ROCm 7.0 and +0.8 ms of instantiation per structure, DESIGN.md section 3;
        // removed in round 5.)
        spec_now = can_spec && direct_scalars && it + 1 < iterations;
        rc = enqueue_trial(&turn); if (rc) return rc;
        if (spec_now) {      // the next iteration's linearisation (it + 1 > 0: the fused form where the graph allows it), queued before the verdict
          const int fl = B->view.fuse_lin;
          B->view.fuse_lin = fuse_lin_ok ? 1 : 0;
          rc = build_system_device(B, B->ev_upd);      // (its numeric-Jacobian edges start behind the update, beside this trial's chi2 kernels)
          B->view.fuse_lin = fl;       // (a retry of THIS iteration reduces the way this iteration linearised)
          if (rc) return rc;
        }
        const double* hs = B->h_scalars;
        if (direct_scalars) { rc = wait_trial(B->trial_seq); if (rc) return rc; hs = B->h_trial; }
        else BA_TRY(hipStreamSynchronize(B->st));
        turn.unlock();
        if (B->h_status[0] == 0x7fffffff || (B->sep_mode && B->shard_n > 1 && B->h_status[2] == 0x7fffffff)) { cs_set_error_ba("banded solver: team not co-resident (wait timed out); set CS_BA_FORCE_DENSE=1 on a shared device"); return CS_ERR_HIP; }
        ok2 = hs[2] == 0.0;
        tempChi = hs[0];
        scale = ok2 ? hs[1] : 0.0;
        if (B->stage_timing) {
          if (direct_scalars) BA_TRY(hipStreamSynchronize(B->st));     // (the phase marks are read below: the last one must have passed)
          rc = collect_solve_times(B); if (rc) return rc;
          float ms = 0;
          BA_TRY(hipEventElapsedTime(&ms, B->ev[6], B->ev[7])); B->tm.errors_ms += ms;
        }
        if (ext_active && ok2) { rc = ext_refresh(0); if (rc) return rc; tempChi += B->ext_chi2; }
      } else {
        rc = solve_device(B, lambda, &ok2, fn, ctx); if (rc) return rc;
        if (B->shard_n > 1) {   // a failed factorisation on one rank is everybody's rejected trial
          double ff = ok2 ? 0.0 : 1.0;
          if (reduce_host(&ff, 1, 1)) { cs_set_error_ba("all-reduce failed"); return CS_ERR_HIP; }
          ok2 = ff == 0.0;
        }
        if (ok2) {
          const int nsb = cs::ba_scale_blocks();
          std::vector<double> sp(nsb);
          cs::ba_launch_scale(B->view, B->d_lam.p, B->scale_partial.p, B->st);     // (d_lam still holds this trial's lambda: put_lambda in solve_device)
          BA_TRY(hipGetLastError());
          BA_TRY(hipMemcpyAsync(sp.data(), B->scale_partial.p, sizeof(double) * nsb, hipMemcpyDeviceToHost, B->st));
          rc = cs_ba_update(B); if (rc) return rc;   // synchronises the stream
          for (double p : sp) scale += p;
        }
        t0 = now_ms();
        rc = chi2_device(B, &tempChi); if (rc) return rc;
        if (ext_active && ok2) { rc = ext_refresh(0); if (rc) return rc; tempChi += B->ext_chi2; }
        if (reduce_host(&tempChi, 1, 0)) { cs_set_error_ba("all-reduce failed"); return CS_ERR_HIP; }
        B->tm.errors_ms += now_ms() - t0;
        if (reduce_host(&scale, 1, 0)) { cs_set_error_ba("all-reduce failed"); return CS_ERR_HIP; }
      }
      if (ok2) debug_nan_scan(B, "cs_ba_optimize: after a trial's solve + update");
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
        spec_lin = spec_now;       // the system of the state this trial left is already on the stream
      } else {
        lambda *= ni;
        ni *= 2;
        rc = cs_ba_pop(B); if (rc) return rc;
        if (spec_now) {            // the speculation linearised the rejected state: the restored one again, as this iteration linearises
          rc = build_system_device(B); if (rc) return rc;
        }
      }
      qmax++;
    } while (rho < 0 && qmax < B->max_trials_after_failure);
    if (!spec_lin) BA_TRY(hipStreamSynchronize(B->st));
    if (done < cap) {
      if (chi_hist) chi_hist[done] = currentChi;
      if (lambda_hist) lambda_hist[done] = lambda;
      if (trials_hist) trials_hist[done] = qmax;
    }
    done++;
    carriedChi = currentChi; have_carried = true;
    if (qmax == B->max_trials_after_failure || rho == 0) break;
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) break;
  }
  if (spec_lin) BA_TRY(hipStreamSynchronize(B->st));      // (a stopping rule ended the loop behind an accepted trial: its speculated system is the current state's)
  if (iterations_done) *iterations_done = done;
  B->tm.total_ms += now_ms() - t_begin;
  return CS_OK;
}
int cs_ba_optimize_sharded(cs_ba* B, int iterations, cs_allreduce_fn fn, void* ctx, int* iterations_done, double* chi_hist, double* lambda_hist, int* trials_hist, int cap) {
  BA_GUARD_BEGIN
  return cs_ba_optimize_sharded_impl(B, iterations, fn, ctx, iterations_done, chi_hist, lambda_hist, trials_hist, cap);
  BA_GUARD_END("cs_ba_optimize_sharded")
}

// ---- RCCL: one process per GPU, the library issues the collectives itself (ncclAllReduce on the handle's stream)
int cs_ba_comm_unique_id(unsigned char id128[128]) {
  if (!id128) return CS_ERR_INVALID_ARG;
  ncclUniqueId id;
  BA_NCCL(ncclGetUniqueId(&id));
  static_assert(sizeof(id.internal) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(id128, id.internal, 128);
  return CS_OK;
}

int cs_ba_comm_init(cs_ba* B, int rank, int n_ranks, const unsigned char id128[128]) {
  if (!B || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return CS_ERR_INVALID_ARG;
  BA_TRY(hipSetDevice(B->device));
  if (B->comm) { (void)ncclCommDestroy(B->comm); B->comm = nullptr; g_comm_handles--; }
  if (g_comm_handles.load() > 0) { cs_set_error_ba("cs_ba_comm_init: this process already holds a communicator handle (one process per GPU)"); return CS_ERR_INVALID_ARG; }
  ncclUniqueId id;
  std::memcpy(id.internal, id128, 128);
  BA_NCCL(ncclCommInitRank(&B->comm, n_ranks, id, rank));
  g_comm_handles++;
  return cs_ba_set_shard(B, rank, n_ranks);
}

// How the sharded solve is organised (after the structure phase): separator mode or the all-reduce of the whole [S | b]; the size of
// the separator system; the bytes this rank contributes to the collectives of one LM trial, and what the all-reduce would be.
int cs_ba_shard_info(cs_ba* B, int* sep_mode, int* n_sep, int* w_max, long long* bytes_per_trial, long long* bytes_per_trial_allreduce, int* interior_n) {
  if (!B) return CS_ERR_INVALID_ARG;
  BA_GUARD_BEGIN
  int rc = finalize_structure(B); if (rc) return rc;
  if (sep_mode) *sep_mode = B->sep_mode ? 1 : 0;
  if (n_sep) *n_sep = B->n_sep;
  if (w_max) *w_max = B->w_max;
  if (bytes_per_trial) *bytes_per_trial = B->bytes_per_trial;
  if (bytes_per_trial_allreduce) *bytes_per_trial_allreduce = B->bytes_per_trial_allreduce;
  if (interior_n) *interior_n = B->sep_mode ? B->int_n : B->n_red;
  return CS_OK;
  BA_GUARD_END("cs_ba_shard_info")
}
// Accumulated stage times of the separator-mode solves (ms; divide by cs_ba_timing::n_solves): interior factorisation, separator
// message (Y = B L^-T, T, t), gather of the messages (through a callback this includes the host round trip), separator system
// (assembly + dense Cholesky + solve), interior back-substitution.  They partition cs_ba_timing::factor_ms.
int cs_ba_shard_timing(cs_ba* B, double out5[5]) {
  if (!B || !out5) return CS_ERR_INVALID_ARG;
  for (int i = 0; i < 5; i++) out5[i] = B->sep_ms[i];
  return CS_OK;
}
// The rank that owns each landmark under the rule in force (separator mode: by the lowest column of its free cameras).
int cs_ba_get_landmark_owners(cs_ba* B, int* owner_out) {
  if (!B || (B->np && !owner_out)) return CS_ERR_INVALID_ARG;
  BA_GUARD_BEGIN
  int rc = finalize_structure(B); if (rc) return rc;
  std::copy(B->lm_owner.begin(), B->lm_owner.end(), owner_out);
  return CS_OK;
  BA_GUARD_END("cs_ba_get_landmark_owners")
}

int cs_ba_get_state(cs_ba* B, double* cams7, double* cuboids10, double* points3) {
  if (!B) return CS_ERR_INVALID_ARG;
  BA_GUARD_BEGIN
  BA_TRY(hipSetDevice(B->device));
  BA_TRY(hipStreamSynchronize(B->st));
  if (cams7 && B->nc) BA_TRY(hipMemcpy(cams7, B->cams.p, 56 * (size_t)B->nc, hipMemcpyDeviceToHost));
  if (cuboids10 && B->no) BA_TRY(hipMemcpy(cuboids10, B->cubes.p, 80 * (size_t)B->no, hipMemcpyDeviceToHost));
  if (points3 && B->np) BA_TRY(hipMemcpy(points3, B->points.p, 24 * (size_t)B->np, hipMemcpyDeviceToHost));
  return CS_OK;
  BA_GUARD_END("cs_ba_get_state")
}

static int cs_ba_sizes_impl(cs_ba* B, int* size_pose, int* size_lm) {
  if (!B) return CS_ERR_INVALID_ARG;
  int rc = finalize_structure(B); if (rc) return rc;
  if (size_pose) *size_pose = B->n_pose;
  if (size_lm) *size_lm = 3 * B->n_lm;
  return CS_OK;
}
int cs_ba_sizes(cs_ba* B, int* size_pose, int* size_lm) {
  BA_GUARD_BEGIN
  return cs_ba_sizes_impl(B, size_pose, size_lm);
  BA_GUARD_END("cs_ba_sizes")
}

static int cs_ba_solver_layout_impl(cs_ba* B, int* band_ld, int* team) {
  if (!B) return CS_ERR_INVALID_ARG;
  int rc = finalize_structure(B); if (rc) return rc;
  int rw = 0;
  if (band_ld) *band_ld = B->band_ld;
  if (team) *team = B->band_ld ? cs::ba_band_team(B->band_ld, &rw) : 0;
  return CS_OK;
}
int cs_ba_solver_layout(cs_ba* B, int* band_ld, int* team) {
  BA_GUARD_BEGIN
  return cs_ba_solver_layout_impl(B, band_ld, team);
  BA_GUARD_END("cs_ba_solver_layout")
}

static int cs_ba_get_system_impl(cs_ba* B, double* Hpp, double* Hll9, double* Hpl18, double* b, double* x) {
  if (!B) return CS_ERR_INVALID_ARG;
  if (B->structure_dirty || !B->have_system) return CS_ERR_NOT_RUN;
  BA_TRY(hipSetDevice(B->device));
  const int n = B->n_pose;
  if (Hpp) {
    std::memset(Hpp, 0, sizeof(double) * (size_t)n * n);
    std::vector<double> hc(36 * (size_t)B->nc), ho(81 * (size_t)B->no), hco(54 * (size_t)B->n_cub), hij(36 * (size_t)B->n_odom);
    if (B->nc) BA_TRY(hipMemcpy(hc.data(), B->Hcam.p, 8 * hc.size(), hipMemcpyDeviceToHost));
    if (B->no) BA_TRY(hipMemcpy(ho.data(), B->Hcub.p, 8 * ho.size(), hipMemcpyDeviceToHost));
    if (B->n_cub) BA_TRY(hipMemcpy(hco.data(), B->ce_Hco.p, 8 * hco.size(), hipMemcpyDeviceToHost));
    if (B->n_odom) BA_TRY(hipMemcpy(hij.data(), B->oe_Hij.p, 8 * hij.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < B->nc; i++) { int c = B->cam_col_ref[i]; if (c < 0) continue; for (int r = 0; r < 6; r++) for (int q = 0; q < 6; q++) Hpp[(size_t)(c + r) * n + c + q] = hc[36 * (size_t)i + 6 * r + q]; }
    for (int i = 0; i < B->no; i++) { int c = B->cub_col_ref[i]; if (c < 0) continue; for (int r = 0; r < 9; r++) for (int q = 0; q < 9; q++) Hpp[(size_t)(c + r) * n + c + q] = ho[81 * (size_t)i + 9 * r + q]; }
    for (int k = 0; k < B->n_cub; k++) {
      int ca = B->cam_col_ref[B->ce_cam[k]], cb = B->cub_col_ref[B->ce_cub[k]];
      if (ca < 0 || cb < 0) continue;
      for (int r = 0; r < 6; r++) for (int q = 0; q < 9; q++) { double val = hco[54 * (size_t)k + 9 * r + q]; Hpp[(size_t)(ca + r) * n + cb + q] += val; Hpp[(size_t)(cb + q) * n + ca + r] += val; }
    }
    for (int k = 0; k < B->n_odom; k++) {
      int ca = B->cam_col_ref[B->oe_i[k]], cb = B->cam_col_ref[B->oe_j[k]];
      if (ca < 0 || cb < 0) continue;
      for (int r = 0; r < 6; r++) for (int q = 0; q < 6; q++) { double val = hij[36 * (size_t)k + 6 * r + q]; Hpp[(size_t)(ca + r) * n + cb + q] += val; Hpp[(size_t)(cb + q) * n + ca + r] += val; }
    }
    for (int k = 0; k < B->ext_n && B->ext_terms_set; k++) {     // the host-evaluated binary edges' blocks
      const int ci = B->ext_e4[4 * k], ii = B->ext_e4[4 * k + 1], cj = B->ext_e4[4 * k + 2], ij = B->ext_e4[4 * k + 3];
      if (ij < 0) continue;
      const int ca = ci ? B->cub_col_ref[ii] : B->cam_col_ref[ii], cb = cj ? B->cub_col_ref[ij] : B->cam_col_ref[ij], di = ci ? 9 : 6, dj = cj ? 9 : 6;
      if (ca < 0 || cb < 0) continue;
      for (int r = 0; r < di; r++) for (int q = 0; q < dj; q++) { const double val = B->h_ext_Hij[81 * (size_t)k + dj * r + q]; Hpp[(size_t)(ca + r) * n + cb + q] += val; Hpp[(size_t)(cb + q) * n + ca + r] += val; }
    }
  }
  if (Hll9) {
    std::vector<double> hl(9 * (size_t)B->np);
    if (B->np) BA_TRY(hipMemcpy(hl.data(), B->Hll.p, 8 * hl.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < B->np; i++) if (B->pt_lm[i] >= 0) std::memcpy(Hll9 + 9 * (size_t)B->pt_lm[i], &hl[9 * (size_t)i], 72);
  }
  if (Hpl18) {
    std::vector<double> w(18 * (size_t)B->slot_src_n);
    if (!w.empty()) BA_TRY(hipMemcpy(w.data(), B->W.p, 8 * w.size(), hipMemcpyDeviceToHost));
    std::memset(Hpl18, 0, 144 * (size_t)B->n_proj);      // (an edge owned by another rank stays zero)
    for (int sl = 0; sl < B->slot_src_n; sl++) std::memcpy(Hpl18 + 18 * (size_t)B->slot_src[sl], &w[18 * (size_t)sl], 144);
  }
  auto to_ref = [&](const std::vector<double>& src, double* dst) {  // solver order -> g2o's order
    for (int i = 0; i < B->nc; i++) if (B->cam_col[i] >= 0) std::memcpy(dst + B->cam_col_ref[i], &src[B->cam_col[i]], 48);
    for (int i = 0; i < B->no; i++) if (B->cub_col[i] >= 0) std::memcpy(dst + B->cub_col_ref[i], &src[B->cub_col[i]], 72);
    std::memcpy(dst + B->n_pose, &src[B->n_pose], 8 * (src.size() - B->n_pose));
  };
  if (b) to_ref(B->h_b, b);
  if (x && !B->h_x.empty()) to_ref(B->h_x, x);
  return CS_OK;
}
int cs_ba_get_system(cs_ba* B, double* Hpp, double* Hll9, double* Hpl18, double* b, double* x) {
  BA_GUARD_BEGIN
  return cs_ba_get_system_impl(B, Hpp, Hll9, Hpl18, b, x);
  BA_GUARD_END("cs_ba_get_system")
}

// Solver::computeMarginals (core/block_solver.hpp:488-499): LinearSolver::solvePattern(spinv, blockIndices, *_Hpp) -- blocks of the INVERSE of
// the pose-pose Hessian H_pp as buildSystem left it (no lambda: restoreDiagonal has run; no Schur complement: the reference's call factorises
// _Hpp itself, core/marginal_covariance_cholesky.cpp:154-222).  Block k = rows of vertex (class_i[k], idx_i[k]) x columns of vertex
// (class_j[k], idx_j[k]), row-major, d_i x d_j doubles (6 per camera, 9 per cuboid), one after the other in `out`.  H_pp is assembled dense in
// g2o's index order (as cs_ba_get_system), factorised by rocSOLVER, and the columns of the requested j-vertices are solved for: a call for
// inspection and for covariance queries of a handful of vertices, not a per-iteration path.  Returns CS_ERR_NOT_RUN before cs_ba_build_system,
// CS_ERR_INVALID_ARG for a fixed vertex or a point, *positive_definite = 0 (and CS_OK) when the factorisation fails (g2o: false).
static int cs_ba_pose_marginals_impl(cs_ba* B, int n_pairs, const int* class_i, const int* idx_i, const int* class_j, const int* idx_j, double* out, int* positive_definite) {
  if (!B || n_pairs < 0 || (n_pairs && (!class_i || !idx_i || !class_j || !idx_j || !out))) return CS_ERR_INVALID_ARG;
  if (positive_definite) *positive_definite = 1;
  if (B->structure_dirty || !B->have_system) { cs_set_error_ba("cs_ba_pose_marginals: call cs_ba_build_system first"); return CS_ERR_NOT_RUN; }
  if (B->shard_n > 1) { cs_set_error_ba("cs_ba_pose_marginals: not on a sharded handle (a rank holds a partial system)"); return CS_ERR_INVALID_ARG; }
  if (n_pairs == 0) return CS_OK;
  const int n = B->n_pose;
  if (n <= 0) { cs_set_error_ba("cs_ba_pose_marginals: the graph has no free camera or cuboid"); return CS_ERR_INVALID_ARG; }
  if ((long long)n * n > (1ll << 28)) { cs_set_error_ba("cs_ba_pose_marginals: the dense pose Hessian would exceed 2 GB"); return CS_ERR_CAPACITY; }
  auto col_of = [&](int cls, int idx, int& dim) -> int {
    if (cls == CS_VERTEX_CAM) { dim = 6; return (idx >= 0 && idx < B->nc) ? B->cam_col_ref[idx] : -1; }
    if (cls == CS_VERTEX_CUBOID) { dim = 9; return (idx >= 0 && idx < B->no) ? B->cub_col_ref[idx] : -1; }
    dim = 0; return -1;
  };
  // the distinct column vertices: one block column of right-hand sides each
  std::vector<int> col_base, col_dim, rhs_of_pair(n_pairs);
  std::map<int, int> rhs_at;            // pose column -> first right-hand-side column
  int m = 0;
  for (int k = 0; k < n_pairs; k++) {
    int di, dj;
    const int ci = col_of(class_i[k], idx_i[k], di), cj = col_of(class_j[k], idx_j[k], dj);
    if (ci < 0 || cj < 0) { cs_set_error_ba("cs_ba_pose_marginals: pair " + std::to_string(k) + " names a fixed vertex, a point or an index out of range"); return CS_ERR_INVALID_ARG; }
    auto it = rhs_at.find(cj);
    if (it == rhs_at.end()) { it = rhs_at.emplace(cj, m).first; col_base.push_back(cj); col_dim.push_back(dj); m += dj; }
    rhs_of_pair[k] = it->second;
  }
  BA_TRY(hipSetDevice(B->device));
  BA_TRY(hipStreamSynchronize(B->st));
  std::vector<double> H((size_t)n * n);
  { const int rc = cs_ba_get_system_impl(B, H.data(), nullptr, nullptr, nullptr, nullptr); if (rc) return rc; }
  std::vector<double> E((size_t)n * m, 0.0);      // column-major n x m: unit vectors
  for (size_t q = 0, c0 = 0; q < col_base.size(); c0 += col_dim[q], q++)
    for (int d = 0; d < col_dim[q]; d++) E[(c0 + d) * (size_t)n + col_base[q] + d] = 1.0;
  DBuf<double> dH, dE;
  DBuf<int> dinfo;
  struct Rel { DBuf<double>&a, &b; DBuf<int>& c; ~Rel() { a.release(); b.release(); c.release(); } } rel{dH, dE, dinfo};
  int rc;
  if ((rc = dH.reserve((size_t)n * n)) || (rc = dE.reserve((size_t)n * m)) || (rc = dinfo.reserve(1))) return rc;
  BA_TRY(hipMemcpyAsync(dH.p, H.data(), 8 * H.size(), hipMemcpyHostToDevice, B->st));
  BA_TRY(hipMemcpyAsync(dE.p, E.data(), 8 * E.size(), hipMemcpyHostToDevice, B->st));
  BA_ROC(rocblas_set_stream(B->blas, B->st));
  BA_ROC(rocsolver_dpotrf(B->blas, rocblas_fill_upper, n, dH.p, n, dinfo.p));      // (symmetric: row- and column-major are the same matrix)
  int info = 0;
  BA_TRY(hipMemcpyAsync(&info, dinfo.p, sizeof(int), hipMemcpyDeviceToHost, B->st));
  BA_TRY(hipStreamSynchronize(B->st));
  if (info != 0) { if (positive_definite) *positive_definite = 0; return CS_OK; }
  BA_ROC(rocsolver_dpotrs(B->blas, rocblas_fill_upper, n, m, dH.p, n, dE.p, n));
  BA_TRY(hipMemcpyAsync(E.data(), dE.p, 8 * E.size(), hipMemcpyDeviceToHost, B->st));
  BA_TRY(hipStreamSynchronize(B->st));
  size_t o = 0;
  for (int k = 0; k < n_pairs; k++) {
    int di, dj;
    const int ci = col_of(class_i[k], idx_i[k], di);
    (void)col_of(class_j[k], idx_j[k], dj);
    for (int r = 0; r < di; r++) for (int c = 0; c < dj; c++) out[o++] = E[(size_t)(rhs_of_pair[k] + c) * n + ci + r];
  }
  return CS_OK;
}
int cs_ba_pose_marginals(cs_ba* B, int n_pairs, const int* class_i, const int* idx_i, const int* class_j, const int* idx_j, double* out, int* positive_definite) {
  BA_GUARD_BEGIN
  return cs_ba_pose_marginals_impl(B, n_pairs, class_i, idx_i, class_j, idx_j, out, positive_definite);
  BA_GUARD_END("cs_ba_pose_marginals")
}

// Inspection for parity tests: the damped reduced system exactly as the solver is about to factorise it -- dense symmetric n_red x n_red
// (n_red = cs_ba_reduced_size), its right-hand side, and the column of every camera / cuboid in it (solver order = reverse
// Cuthill-McKee; -1: fixed; a cuboid column >= n_red: the cuboid was eliminated and is not part of S).  Runs the Schur-complement build
// (block_solver.hpp:373-439) at `lambda` on the current linearisation; no factorisation.
static int cs_ba_get_reduced_system_impl(cs_ba* B, double lambda, double* S_dense, double* rhs, int* cam_col, int* cub_col) {
  if (!B) return CS_ERR_INVALID_ARG;
  if (B->structure_dirty || !B->have_system) { cs_set_error_ba("cs_ba_get_reduced_system: call cs_ba_build_system first"); return CS_ERR_NOT_RUN; }
  if (B->shard_n > 1) { cs_set_error_ba("cs_ba_get_reduced_system: not on a sharded handle (a rank holds a partial system)"); return CS_ERR_INVALID_ARG; }
  BA_TRY(hipSetDevice(B->device));
  const int n = B->n_red;
  if (cam_col) std::copy(B->cam_col.begin(), B->cam_col.end(), cam_col);
  if (cub_col) std::copy(B->cub_col.begin(), B->cub_col.end(), cub_col);
  if (n <= 0) return CS_OK;
  { int rcl = put_lambda(B, lambda); if (rcl) return rcl; }
  BA_TRY(hipMemsetAsync(B->S.p, 0, sizeof(double) * (B->s_doubles + B->n_pose), B->st));
  BA_TRY(hipMemsetAsync(B->d_elim_fail.p, 0, sizeof(int), B->st));
  cs::ba_launch_reduce(B->view, B->d_lam.p, B->st, B->st2, B->ev_fork, B->ev_join);
  if (B->ext_n > 0 && B->ext_terms_set) cs::ba_launch_ext_offdiag(B->view, B->ext_groups, B->d_ext_gptr.p, B->d_ext_order.p, B->d_ext_e4.p, B->ext_Hij.p, B->st);
  BA_TRY(hipGetLastError());
  std::vector<double> h(B->s_doubles + B->n_pose);
  BA_TRY(hipMemcpyAsync(h.data(), B->S.p, 8 * h.size(), hipMemcpyDeviceToHost, B->st));
  BA_TRY(hipStreamSynchronize(B->st));
  if (rhs) std::copy(h.begin() + B->s_doubles, h.begin() + B->s_doubles + n, rhs);
  if (S_dense) {
    std::memset(S_dense, 0, sizeof(double) * (size_t)n * n);
    if (B->band_ld) {
      for (int c = 0; c < n; c++)
        for (int d = 0; d < B->band_ld && c + d < n; d++) { const double v = h[(size_t)c * B->band_ld + d]; S_dense[(size_t)(c + d) * n + c] = v; S_dense[(size_t)c * n + c + d] = v; }
    } else {
      for (int r = 0; r < n; r++) for (int c = 0; c <= r; c++) { const double v = h[(size_t)r * n + c]; S_dense[(size_t)r * n + c] = v; S_dense[(size_t)c * n + r] = v; }
    }
  }
  return CS_OK;
}
int cs_ba_get_reduced_system(cs_ba* B, double lambda, double* S_dense, double* rhs, int* cam_col, int* cub_col) {
  BA_GUARD_BEGIN
  return cs_ba_get_reduced_system_impl(B, lambda, S_dense, rhs, cam_col, cub_col);
  BA_GUARD_END("cs_ba_get_reduced_system")
}

int cs_ba_reduced_size(cs_ba* B, int* n_reduced, int* cuboids_eliminated) {
  if (!B) return CS_ERR_INVALID_ARG;
  BA_GUARD_BEGIN
  int rc = finalize_structure(B); if (rc) return rc;
  if (n_reduced) *n_reduced = B->n_red;
  if (cuboids_eliminated) *cuboids_eliminated = B->elim ? 1 : 0;
  return CS_OK;
  BA_GUARD_END("cs_ba_reduced_size")
}

int cs_ba_band_order(cs_ba* B, int* block_cyclic_reduction, int* levels) {
  if (!B) return CS_ERR_INVALID_ARG;
  BA_GUARD_BEGIN
  int rc = finalize_structure(B); if (rc) return rc;
  const bool bcr = B->band_ld > 0 && B->use_bcr;
  if (block_cyclic_reduction) *block_cyclic_reduction = bcr ? 1 : 0;
  if (levels) { int L = 0; for (int N = (B->n_red + 127) / 128; bcr && N >= 1; N /= 2) L++; *levels = L; }
  return CS_OK;
  BA_GUARD_END("cs_ba_band_order")
}
int cs_ba_solver_path(cs_ba* B, int* path, int* bandwidth, double* sparse_fill) {
  if (!B) return CS_ERR_INVALID_ARG;
  BA_GUARD_BEGIN
  int rc = finalize_structure(B); if (rc) return rc;
  if (path) *path = B->band_ld ? CS_BA_PATH_BAND : (B->sparse ? CS_BA_PATH_SPARSE : CS_BA_PATH_DENSE);
  if (bandwidth) *bandwidth = B->band_ld ? B->band_ld - 1 : 0;
  if (sparse_fill) *sparse_fill = B->sparse ? (double)B->sp_plan.nvals / (0.5 * (double)B->n_red * (double)B->n_red) : 0.0;
  return CS_OK;
  BA_GUARD_END("cs_ba_solver_path")
}

int cs_ba_schur_layout(cs_ba* B, int* fused, int* n_segments, int* n_partial_blocks, int* n_blocks) {
  if (!B) return CS_ERR_INVALID_ARG;
  BA_GUARD_BEGIN
  int rc = finalize_structure(B); if (rc) return rc;
  if (fused) *fused = B->fused ? 1 : 0;
  if (n_segments) *n_segments = B->n_seg;
  if (n_partial_blocks) *n_partial_blocks = (int)(B->part_tiles.n / 36);
  if (n_blocks) *n_blocks = B->fused ? B->n_gpairs : B->n_pairs;
  return CS_OK;
  BA_GUARD_END("cs_ba_schur_layout")
}

// Inspection for tests: a fingerprint of every index table the structure phase leaves on the device (edge orders, per-vertex lists,
// the Schur schedule, the solver's column maps), one 64-bit FNV-1a hash per table in a fixed order.  Two builds of one graph -- the
// threaded loops and CS_BA_STRUCT_THREADS=1, an appended graph and the same graph set up at once -- must agree table by table.
int cs_ba_structure_digest(cs_ba* B, unsigned long long* out, int cap, int* n_tables) {
  if (!B || !n_tables || cap < 0 || (cap && !out)) return CS_ERR_INVALID_ARG;
  BA_GUARD_BEGIN
  int rc = finalize_structure(B); if (rc) return rc;
  BA_TRY(hipSetDevice(B->device));
  BA_TRY(hipStreamSynchronize(B->st));
  const DBuf<int>* tabs[] = {&B->d_cam_col, &B->d_cub_col, &B->d_pt_free, &B->pm_pt, &B->pm_cam, &B->pt_ptr, &B->cm_pm, &B->cm_pt, &B->cam_ptr, &B->d_src,
                             &B->d_run_lm, &B->d_seg_ptr, &B->d_seg_k, &B->d_seg_tile, &B->d_seg_slot, &B->d_gp_ptr, &B->d_gp_i1, &B->d_gp_i2, &B->d_gtile, &B->d_gcam_ptr, &B->d_gslot,
                             &B->pair_ptr, &B->pair_i1, &B->pair_i2, &B->ent_a, &B->ent_b, &B->d_cubS_ptr, &B->d_cubS_cam, &B->d_ce_slot, &B->d_cub_tile, &B->d_cub_coef,
                             &B->d_slotE_ptr, &B->d_slotE_idx, &B->cam_ce_ptr, &B->cam_ce_idx, &B->cam_oei_ptr, &B->cam_oei_idx, &B->cam_oej_ptr, &B->cam_oej_idx, &B->cub_ce_ptr, &B->cub_ce_idx,
                             &B->d_ce_active, &B->d_oe_active};
  const int nt = (int)(sizeof(tabs) / sizeof(tabs[0]));
  *n_tables = nt;
  std::vector<int> h;
  for (int t = 0; t < nt && t < cap; t++) {
    h.resize(tabs[t]->n);
    if (tabs[t]->n) BA_TRY(hipMemcpy(h.data(), tabs[t]->p, sizeof(int) * tabs[t]->n, hipMemcpyDeviceToHost));
    unsigned long long f = 1469598103934665603ull ^ (unsigned long long)tabs[t]->n;
    for (int v : h) { f ^= (unsigned)v; f *= 1099511628211ull; }
    out[t] = f;
  }
  return CS_OK;
  BA_GUARD_END("cs_ba_structure_digest")
}

// The diagonal Hessian blocks g2o keeps mapped into its vertices (BaseVertex::_hessian, core/base_vertex.hpp:30,52-54; mapped by
// BlockSolver::buildStructure, block_solver.hpp:185,191) and that OptimizationAlgorithmLevenberg::computeLambdaInit reads
// (optimization_algorithm_levenberg.cpp:166-180): A_ii of every vertex, caller's vertex order, zeros for fixed vertices.
int cs_ba_get_vertex_hessians(cs_ba* B, double* cam36, double* cub81, double* pt9) {
  if (!B) return CS_ERR_INVALID_ARG;
  if (B->structure_dirty || !B->have_system) { cs_set_error_ba("cs_ba_get_vertex_hessians: call cs_ba_build_system first"); return CS_ERR_NOT_RUN; }
  BA_TRY(hipSetDevice(B->device));
  BA_TRY(hipStreamSynchronize(B->st));
  if (cam36 && B->nc) {
    BA_TRY(hipMemcpy(cam36, B->Hcam.p, 8 * 36 * (size_t)B->nc, hipMemcpyDeviceToHost));
    for (int i = 0; i < B->nc; i++) if (B->cam_fixed[i]) std::memset(cam36 + 36 * (size_t)i, 0, 288);
  }
  if (cub81 && B->no) {
    BA_TRY(hipMemcpy(cub81, B->Hcub.p, 8 * 81 * (size_t)B->no, hipMemcpyDeviceToHost));
    for (int i = 0; i < B->no; i++) if (B->cub_fixed[i]) std::memset(cub81 + 81 * (size_t)i, 0, 648);
  }
  if (pt9 && B->np) {
    BA_TRY(hipMemcpy(pt9, B->Hll.p, 8 * 9 * (size_t)B->np, hipMemcpyDeviceToHost));
    for (int i = 0; i < B->np; i++) if (B->pt_fixed[i]) std::memset(pt9 + 9 * (size_t)i, 0, 72);
  }
  return CS_OK;
}


int cs_ba_set_lm_params(cs_ba* B, double user_lambda_init, int max_trials_after_failure) {
  if (!B || !(user_lambda_init == user_lambda_init) || max_trials_after_failure < 1) return CS_ERR_INVALID_ARG;
  B->user_lambda_init = user_lambda_init > 0 ? user_lambda_init : 0.0;
  B->max_trials_after_failure = max_trials_after_failure;
  return CS_OK;
}
int cs_ba_set_stage_timing(cs_ba* B, int on) {
  if (!B) return CS_ERR_INVALID_ARG;
  B->stage_timing = on != 0;
  B->lin_pending = false;
  return CS_OK;
}
int cs_ba_last_timing(cs_ba* B, cs_ba_timing* t) {
  if (!B || !t) return CS_ERR_INVALID_ARG;
  *t = B->tm;
  t->schur_entries = B->schur_entries;
  // algorithmic bytes per linearisation + Schur build: per projection edge 136 B read
  // + 144 B Hpl written, re-read once by the Schur stage; per camera 336 B; per point 96 B written, 96 B read,
  // 72 B Dinv written; per cuboid edge 864 B read + 432 B written.
  t->linearize_bytes = (long long)B->slot_src_n * (136 + 144 + 144) + (long long)B->nc * 336 + (long long)B->np * 264 + (long long)B->n_cub3 * (864 + 432) + (long long)(B->n_cub - B->n_cub3) * (368 + 1400);
  return CS_OK;
}

}  // extern "C"

// ---- debug / repro aids --------------------------------------------------------------------------------------------------------------
// cs_ba_check_finite: the stand-in for the NaN checks of g2o's debug builds (errors: sparse_optimizer.cpp:78-86; Jacobians:
// block_solver.hpp:533-544) -- see ba_scan_finite_kernel.  CS_BA_DEBUG_NAN=1 runs it after every linearisation, solve and update and
// prints what it finds to stderr.
static int check_finite_impl(cs_ba* B, char* report, int report_cap, int* n_bad_out) {
  if (!B) return CS_ERR_INVALID_ARG;
  BA_TRY(hipSetDevice(B->device));
  int rc = finalize_structure(B); if (rc) return rc;
  const cs::BaView& v = B->view;
  const int E = v.n_proj, n_edges = E + B->n_cub + B->n_odom;
  DBuf<double> chi; DBuf<int> out;
  struct Free { DBuf<double>* a; DBuf<int>* b; ~Free() { a->release(); b->release(); } } guard{&chi, &out};
  if ((rc = chi.alloc(std::max(1, n_edges), B->st))) return rc;
  cs::ba_launch_edge_chi(v, chi.p, B->st);
  struct Arr { const char* name; const double* p; long long n; int per; const char* owner; };
  std::vector<Arr> arrs = {
    {"squared error of a projection edge", chi.p, E, 1, "projection edge (caller's index)"},
    {"squared error of a camera-cuboid edge", chi.p + E, B->n_cub, 1, "camera-cuboid edge (EdgeSE3Cuboid list, then EdgeSE3CuboidProj list)"},
    {"squared error of an odometry edge", chi.p + E + B->n_cub, B->n_odom, 1, "odometry edge"},
    {"camera estimates", v.cams, 7LL * B->nc, 7, "camera"}, {"cuboid estimates", v.cubes, 10LL * B->no, 10, "cuboid"}, {"point estimates", v.points, 3LL * B->np, 3, "point"},
  };
  if (B->have_system) {
    const Arr sys[] = {
      {"A_ii of a camera", v.Hcam, 36LL * B->nc, 36, "camera"}, {"b_i of a camera", v.bcam, 6LL * B->nc, 6, "camera"},
      {"A_ii of a cuboid", v.Hcub, 81LL * B->no, 81, "cuboid"}, {"b_i of a cuboid", v.bcub, 9LL * B->no, 9, "cuboid"},
      {"A_jj of a point", v.Hll, 9LL * B->np, 9, "point"}, {"b_j of a point", v.bl, 3LL * B->np, 3, "point"},
      {"H_pl block of a projection edge (J_cam^T Omega J_point)", v.W, 18LL * E, 18, "projection edge (caller's index)"},
      {"J^T Omega J block of a camera-cuboid edge", v.ce_Hco, 54LL * B->n_cub, 54, "camera-cuboid edge"},
      {"J^T Omega J block of an odometry edge", v.oe_Hij, 36LL * B->n_odom, 36, "odometry edge"},
    };
    arrs.insert(arrs.end(), std::begin(sys), std::end(sys));
  }
  if (B->tm.n_solves > 0) {
    arrs.push_back({"pose increment x_p (solver order)", v.rhs, B->n_pose, 1, "column"});
    arrs.push_back({"landmark increment", v.xl, 3LL * B->np, 3, "point"});
  }
  std::vector<int> h(2 * arrs.size());
  for (size_t t = 0; t < arrs.size(); t++) { h[2 * t] = 0; h[2 * t + 1] = 0x7fffffff; }
  if ((rc = out.upload(h))) return rc;
  for (size_t t = 0; t < arrs.size(); t++) cs::ba_launch_scan_finite(arrs[t].p, arrs[t].n, out.p + 2 * t, B->st);
  BA_TRY(hipGetLastError());
  BA_TRY(hipMemcpyAsync(h.data(), out.p, sizeof(int) * h.size(), hipMemcpyDeviceToHost, B->st));
  BA_TRY(hipStreamSynchronize(B->st));
  long long total = 0;
  std::string rep;
  for (size_t t = 0; t < arrs.size(); t++) {
    if (!h[2 * t]) continue;
    total += h[2 * t];
    long long owner = (h[2 * t + 1] - 1) / arrs[t].per;
    if (arrs[t].per == 18 || (arrs[t].per == 1 && t == 0)) {      // projection edges: point-major slot -> the caller's edge index
      owner = (owner >= 0 && owner < B->slot_src_n) ? B->slot_src[(size_t)owner] : -1;
    }
    rep += std::string(arrs[t].name) + ": " + std::to_string(h[2 * t]) + " non-finite value(s), first in " + arrs[t].owner + " " + std::to_string(owner) + "\n";
  }
  if (report && report_cap > 0) { const size_t n = std::min(rep.size(), (size_t)report_cap - 1); std::memcpy(report, rep.data(), n); report[n] = 0; }
  if (n_bad_out) *n_bad_out = (int)std::min<long long>(total, 0x7fffffff);
  return CS_OK;
}
int cs_ba_check_finite(cs_ba* B, int* n_bad, char* report, int report_cap) {
  BA_GUARD_BEGIN
  return check_finite_impl(B, report, report_cap, n_bad);
  BA_GUARD_END("cs_ba_check_finite")
}
static bool debug_nan_enabled() { static const bool on = [] { const char* e = getenv("CS_BA_DEBUG_NAN"); return e && atoi(e) != 0; }(); return on; }
static void debug_nan_scan(cs_ba* B, const char* where) {
  if (!debug_nan_enabled()) return;
  char rep[2048]; int n = 0;
  if (check_finite_impl(B, rep, (int)sizeof(rep), &n) == CS_OK && n > 0) fprintf(stderr, "[cs_ba CS_BA_DEBUG_NAN] %s:\n%s", where, rep);
}

// cs_ba_dump / cs_ba_load: the whole problem description (current estimates included) as one flat binary file, so that a failing field
// case becomes a fixture -- the stand-in for OptimizableGraph::save / load (core/optimizable_graph.h:594-606), which the reference's own
// graph cannot use (its vendored g2o registers none of its types with the Factory).  Layout: magic "CSBA0002", 16 int32 counts / flags,
// then the arrays in the order written below, each raw little-endian.  Shard settings and external edges are not part of it.
namespace {
struct DumpHeader { char magic[8]; int v[16]; };
template <class T> bool wr(FILE* f, const std::vector<T>& a) { return a.empty() || fwrite(a.data(), sizeof(T), a.size(), f) == a.size(); }
template <class T> bool rd(FILE* f, std::vector<T>& a, size_t n) { a.resize(n); return n == 0 || fread(a.data(), sizeof(T), n, f) == n; }
}  // namespace
static int cs_ba_dump_impl(cs_ba* B, const char* path) {
  if (!B || !path) return CS_ERR_INVALID_ARG;
  BA_TRY(hipSetDevice(B->device));
  BA_TRY(hipStreamSynchronize(B->st));
  const int np_e = B->n_proj, n3 = (int)B->u3_cam.size(), n4 = (int)B->up_cam.size(), n6 = B->n_odom;
  std::vector<double> cams(7 * (size_t)B->nc), cubs(10 * (size_t)B->no), pts(3 * (size_t)B->np), uv(2 * (size_t)np_e), info(4 * (size_t)np_e), intr(4 * (size_t)np_e), hub(B->have_huber ? np_e : 0);
  auto d2h = [&](std::vector<double>& h, const double* d) -> int { if (!h.empty()) BA_TRY(hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost)); return CS_OK; };
  int rc;
  if ((rc = d2h(cams, B->cams.p)) || (rc = d2h(cubs, B->cubes.p)) || (rc = d2h(pts, B->points.p)) || (rc = d2h(uv, B->raw_uv.p)) || (rc = d2h(hub, B->raw_huber.p))) return rc;
  // (records that were never stored -- every edge carries the reference record -- are written out here)
  if (B->raw_info_virtual) { for (int k = 0; k < np_e; k++) std::memcpy(&info[4 * (size_t)k], B->uni8, 32); } else if ((rc = d2h(info, B->raw_info.p))) return rc;
  if (B->raw_intr_virtual) { for (int k = 0; k < np_e; k++) std::memcpy(&intr[4 * (size_t)k], B->uni8 + 4, 32); } else if ((rc = d2h(intr, B->raw_intr.p))) return rc;
  FILE* f = fopen(path, "wb");
  if (!f) { cs_set_error_ba(std::string("cs_ba_dump: cannot open ") + path); return CS_ERR_INVALID_ARG; }
  DumpHeader H{};
  std::memcpy(H.magic, "CSBA0002", 8);
  const int counts[16] = {B->nc, B->no, B->np, B->cuboids_first, np_e, B->have_huber ? 1 : 0, (int)B->rk_proj.size(), n3, (int)B->rk_cub3.size(), n4, (int)B->rk_cproj.size(), n6, (int)B->rk_odom.size(), 0, 0, 0};
  std::memcpy(H.v, counts, sizeof(counts));
  bool ok = fwrite(&H, sizeof(H), 1, f) == 1;
  ok = ok && wr(f, cams) && wr(f, B->cam_fixed) && wr(f, cubs) && wr(f, B->cub_fixed) && wr(f, pts) && wr(f, B->pt_fixed);
  ok = ok && wr(f, B->e_pt) && wr(f, B->e_cam) && wr(f, uv) && wr(f, info) && wr(f, intr) && wr(f, hub) && wr(f, B->rk_proj);
  ok = ok && wr(f, B->u3_cam) && wr(f, B->u3_cub) && wr(f, B->h_ce_meas) && wr(f, B->h_ce_info) && wr(f, B->rk_cub3) && wr(f, B->rd_cub3);
  ok = ok && wr(f, B->up_cam) && wr(f, B->up_cub) && wr(f, B->h_pe_meas) && wr(f, B->h_pe_info) && wr(f, B->h_pe_K) && wr(f, B->rk_cproj) && wr(f, B->rd_cproj);
  ok = ok && wr(f, B->oe_i) && wr(f, B->oe_j) && wr(f, B->h_oe_meas) && wr(f, B->h_oe_info) && wr(f, B->rk_odom) && wr(f, B->rd_odom);
  ok = (fclose(f) == 0) && ok;
  if (!ok) { cs_set_error_ba(std::string("cs_ba_dump: write to ") + path + " failed"); return CS_ERR_INVALID_ARG; }
  return CS_OK;
}
int cs_ba_dump(cs_ba* B, const char* path) {
  BA_GUARD_BEGIN
  return cs_ba_dump_impl(B, path);
  BA_GUARD_END("cs_ba_dump")
}
static int cs_ba_load_impl(const char* path, int device, cs_ba** out) {
  if (!path || !out) return CS_ERR_INVALID_ARG;
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) { cs_set_error_ba(std::string("cs_ba_load: cannot open ") + path); return CS_ERR_INVALID_ARG; }
  struct Close { FILE* f; ~Close() { fclose(f); } } cl{f};
  DumpHeader H;
  if (fread(&H, sizeof(H), 1, f) != 1 || std::memcmp(H.magic, "CSBA0002", 8) != 0) { cs_set_error_ba("cs_ba_load: not a cs_ba dump (magic CSBA0002)"); return CS_ERR_INVALID_ARG; }
  for (int i = 0; i < 13; i++) if (H.v[i] < 0) { cs_set_error_ba("cs_ba_load: corrupt header"); return CS_ERR_INVALID_ARG; }
  const int nc = H.v[0], no = H.v[1], np = H.v[2], cf = H.v[3], npe = H.v[4], hh = H.v[5], nrk = H.v[6], n3 = H.v[7], nrk3 = H.v[8], n4 = H.v[9], nrk4 = H.v[10], n6 = H.v[11], nrk6 = H.v[12];
  std::vector<double> cams, cubs, pts, uv, info, intr, hub, m10, i81, rd3, m4, i16, k9, rd4, m7, i36, rd6;
  std::vector<int> camf, cubf, ptf, ept, ecam, rkp, c3, o3, rk3, c4, o4, rk4, oi, oj, rk6;
  bool ok = rd(f, cams, 7 * (size_t)nc) && rd(f, camf, nc) && rd(f, cubs, 10 * (size_t)no) && rd(f, cubf, no) && rd(f, pts, 3 * (size_t)np) && rd(f, ptf, np);
  ok = ok && rd(f, ept, npe) && rd(f, ecam, npe) && rd(f, uv, 2 * (size_t)npe) && rd(f, info, 4 * (size_t)npe) && rd(f, intr, 4 * (size_t)npe) && rd(f, hub, hh ? npe : 0) && rd(f, rkp, nrk);
  ok = ok && rd(f, c3, n3) && rd(f, o3, n3) && rd(f, m10, 10 * (size_t)n3) && rd(f, i81, 81 * (size_t)n3) && rd(f, rk3, nrk3) && rd(f, rd3, nrk3);
  ok = ok && rd(f, c4, n4) && rd(f, o4, n4) && rd(f, m4, 4 * (size_t)n4) && rd(f, i16, 16 * (size_t)n4) && rd(f, k9, 9 * (size_t)n4) && rd(f, rk4, nrk4) && rd(f, rd4, nrk4);
  ok = ok && rd(f, oi, n6) && rd(f, oj, n6) && rd(f, m7, 7 * (size_t)n6) && rd(f, i36, 36 * (size_t)n6) && rd(f, rk6, nrk6) && rd(f, rd6, nrk6);
  if (!ok || (nrk && nrk != npe) || (nrk3 && nrk3 != n3) || (nrk4 && nrk4 != n4) || (nrk6 && nrk6 != n6)) { cs_set_error_ba("cs_ba_load: truncated or inconsistent file"); return CS_ERR_INVALID_ARG; }
  cs_ba* B = nullptr;
  int rc = cs_ba_create(device, &B); if (rc) return rc;
  struct Guard { cs_ba* b; ~Guard() { if (b) cs_ba_destroy(b); } } g{B};
  if ((rc = cs_ba_set_vertices(B, cams.data(), camf.data(), nc, cubs.data(), cubf.data(), no, pts.data(), ptf.data(), np, cf))) return rc;
  // (the setters normalise quaternions; the dumped ones are normalised already and must come back with their exact bits)
  if (nc) BA_TRY(hipMemcpy(B->cams.p, cams.data(), 56 * (size_t)nc, hipMemcpyHostToDevice));
  if (npe && (rc = cs_ba_set_edges_proj(B, npe, ept.data(), ecam.data(), uv.data(), info.data(), intr.data(), hh ? hub.data() : nullptr))) return rc;
  if (nrk && (rc = cs_ba_set_robust_kernels(B, CS_EDGE_PROJ, npe, rkp.data(), hub.data()))) return rc;
  if (n3 && (rc = cs_ba_set_edges_cuboid(B, n3, c3.data(), o3.data(), m10.data(), i81.data()))) return rc;
  if (nrk3 && (rc = cs_ba_set_robust_kernels(B, CS_EDGE_CUBOID, n3, rk3.data(), rd3.data()))) return rc;
  if (n4 && (rc = cs_ba_set_edges_cuboid_proj(B, n4, c4.data(), o4.data(), m4.data(), i16.data(), k9.data()))) return rc;
  if (nrk4 && (rc = cs_ba_set_robust_kernels(B, CS_EDGE_CUBOID_PROJ, n4, rk4.data(), rd4.data()))) return rc;
  if (n6 && (rc = cs_ba_set_edges_odom(B, n6, oi.data(), oj.data(), m7.data(), i36.data()))) return rc;
  if (n6) B->h_oe_meas = m7;
  if (nrk6 && (rc = cs_ba_set_robust_kernels(B, CS_EDGE_ODOM, n6, rk6.data(), rd6.data()))) return rc;
  *out = B;
  g.b = nullptr;
  return CS_OK;
}
int cs_ba_load(const char* path, int device, cs_ba** out) {
  BA_GUARD_BEGIN
  return cs_ba_load_impl(path, device, out);
  BA_GUARD_END("cs_ba_load")
}
