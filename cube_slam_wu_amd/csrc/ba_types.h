// ba_types.h -- device data layout of the bundle-adjustment path (shared by ba_kernels.hip / ba_host.cpp).
// See DESIGN.md "Path B: data layout in HBM".
#pragma once
#include "cs_se3.h"
#include "cs_robust.h"

namespace cs {

// one small upload of the structure phase, deferred to a batched launch (ba_launch_multi_copy): src in pinned host memory
struct BaCopyItem { void* dst; const void* src; size_t bytes; };

struct BaView {
  // ---- vertices (estimates) -------------------------------------------------------------------------
  double* cams;      // nc x 7   world-to-camera SE3Quat (VertexSE3Expmap)
  double* points;    // np x 3   (VertexSBAPointXYZ, marginalised)
  double* cubes;     // no x 10  (VertexCuboid: pose 7 + half sizes 3)
  const int* cam_col;  // nc: scalar column of the vertex in the reduced (pose) system, -1 = fixed
  const int* cub_col;  // no
  const int* pt_free;  // np: 1 = optimised, 0 = fixed
  int nc, np, no, n_pose;
  int elim_max_slots;     // observing cameras of the widest free cuboid (sizes ba_cub_elim_kernel's dynamic LDS; <= BA_ELIM_MAX_SLOTS)
  // The system the solver factorises has n_red unknowns: n_pose (cameras and cuboids, g2o's reduced system) or, with elim = 1, the
  // cameras only -- the free cuboids are then eliminated like landmarks (S_cc -= H_co D_oo^-1 H_co^T, block_solver.hpp:385-431
  // applied to the 9 x 9 blocks too) and their increments live behind the reduced system's in `rhs` (cub_col >= n_red).
  int n_red, elim;
  const int* cubS_ptr; const int* cubS_cam;   // no + 1; slot -> camera: the distinct free cameras observing a free cuboid, by column
  const int* ce_slot;                         // cuboid edge -> slot (-1: fixed camera / fixed cuboid)
  const int* slotE_ptr; const int* slotE_idx; // slot -> its cuboid edges, in edge order
  const int* cub_tile; const int* cub_coef;   // per cuboid: first partial block of its slot pairs (a <= b), first partial vector
  const int* cub_mine;                        // no: 1 = this rank eliminates the cuboid (sharded BA: it holds all of its edges)
  double* cub_M;                              // 54 per slot: H_co summed over the slot's edges (6 x 9)
  double* cub_Dinv;                           // 81 per cuboid: (H_oo + lambda I)^-1
  int* elim_fail;                             // set when a damped 9 x 9 block is not positive definite
  // ---- projection edges (EdgeSE3ProjectXYZ) ---------------------------------------------------------
  // point-major copy: edges of one landmark are contiguous, sorted by pose column (the CCS column order of
  // block_solver.hpp:398-431); camera-major copy: edges of one camera are contiguous.
  int n_proj;
  const int* pm_pt; const int* pm_cam; const double* pm_uv; const double* pm_info; const double* pm_intr; const double* pm_huber;
  // every projection edge with the same information matrix / the same intrinsics (one camera, one sigma: the usual case): 4 doubles each, read
  // instead of the 32-byte per-edge records (64 of an edge's 88 bytes, in every pass over the edges); nullptr: the per-edge arrays are read
  const double* info_u; const double* intr_u;
  const int* pt_ptr;   // np + 1
  const int* cm_pm;    // n_proj: index into the point-major arrays
  const int* cm_pt; const double* cm_uv; const double* cm_info; const double* cm_intr; const double* cm_huber;
  const int* cam_ptr;  // nc + 1
  // robust kernels (cs_robust.h).  Projection edges: pm_huber / cm_huber hold RobustKernel::delta(); pm_rk / cm_rk the kernel kind, or
  // nullptr when every projection edge has Huber or none (then delta > 0 means Huber -- the common case keeps its 8-byte record)
  const int* pm_rk; const int* cm_rk;
  // ---- cuboid edges and odometry edges (EdgeSE3Expmap): numeric Jacobians ------------------------------
  // camera-cuboid edges 0 .. n_cub3 - 1 are EdgeSE3Cuboid (9-dim, ce_meas / ce_info), edges n_cub3 .. n_cub - 1 are
  // EdgeSE3CuboidProj (4-dim bounding-box error, pe_meas 4 / pe_info 16 / pe_K 9 per edge, indexed k - n_cub3); both
  // kinds share the index lists, the output blocks and the vertex adjacency
  int n_cub3; const double* pe_meas; const double* pe_info; const double* pe_K;
  int n_cub; const int* ce_cam; const int* ce_cub; const double* ce_meas; const double* ce_info; const int* ce_active;
  const int* ce_rk; const double* ce_rdelta;   // n_cub: kernel kind / delta of the camera-cuboid edges (nullptr: none has a kernel)
  const int* oe_rk; const double* oe_rdelta;   // n_odom
  double* ce_Hcc; double* ce_Hoo; double* ce_Hco; double* ce_bc; double* ce_bo;   // 36, 81, 54, 6, 9 per edge
  int n_odom; const int* oe_i; const int* oe_j; const double* oe_meas; const double* oe_info; const int* oe_active;
  double* oe_Hii; double* oe_Hjj; double* oe_Hij; double* oe_bi; double* oe_bj;   // 36, 36, 36, 6, 6 per edge
  // pose-vertex adjacency of those edges (CSR), for a deterministic gather-accumulate
  const int* cam_ce_ptr; const int* cam_ce_idx;     // cuboid edges of a camera
  const int* cam_oei_ptr; const int* cam_oei_idx;   // odometry edges where the camera is vertex 0
  const int* cam_oej_ptr; const int* cam_oej_idx;   // ... vertex 1
  const int* cub_ce_ptr; const int* cub_ce_idx;     // cuboid edges of a cuboid
  // ---- linear system --------------------------------------------------------------------------------
  double* Hcam; double* bcam;   // nc x 36, nc x 6   (A_ii, b_i of the cameras)
  double* Hcub; double* bcub;   // no x 81, no x 9
  double* Hll; double* bl;      // np x 9,  np x 3
  double* W;                    // n_proj x 18  Hpl block of each projection edge (6x3, point-major order)
  double* WD;                   // n_proj x 18  W * Dinv
  double* Dinv;                 // np x 9       (Hll + lambda I)^-1
  double* dbl;                  // np x 3       Dinv * b_l
  double* S;                    // reduced system, LOWER triangle only.  band_ld == 0: dense, element (r, c), r >= c, at
                                // S[r * n_pose + c]; band_ld > 0: band storage, element (r, c) at S[c * band_ld + (r - c)],
                                // band_ld = bandwidth + 1 (the pose vertices are ordered by reverse Cuthill-McKee)
  int band_ld;
  int lam_lo, lam_hi;           // sharded BA: lambda goes to the pose diagonals of the columns [lam_lo, lam_hi) only -- every column
                                // gets it from exactly one rank (the partial systems are summed)
  double* rhs;                  // n_pose       b_schur, overwritten by x_p
  double* xl;                   // np x 3       landmark increments
  // Schur structure: one entry per (landmark, ordered camera pair i1 <= i2), grouped by block pair
  int n_pairs; const int* pair_ptr; const int* pair_i1; const int* pair_i2;   // columns of the block
  const int* ent_a; const int* ent_b;                                          // point-major edge ids
  // Fused Schur schedule (ba_schur_fused_kernel): the landmarks that take part in the Schur complement, grouped by the set of
  // cameras that see them and cut into segments of at most BA_SEG_LM landmarks.  One wavefront multiplies a segment's
  // W_j D_j^-1 W_j^T (6k x 6k, k <= BA_FUSED_KMAX cameras) on the matrix cores, the segment's landmarks being the contraction
  // dimension, and writes k (k + 1) / 2 partial 6 x 6 blocks + k partial 6-vectors of W D^-1 b_l; per destination block / camera
  // the partials are then summed in a fixed order (gpair_* / gcam_*: the destination schedule).
  int fused;                    // 0 = pair-major path (ba_prep / ba_wd / ba_schur kernels)
  int fuse_lin;                 // 1: the landmark side of the projection edges is linearised inside the Schur kernels (ba_lin_schur_kernel): ba_launch_linearize leaves ba_lin_pt_kernel out
  int n_seg; int seg_class[5];   // segments [0, seg_class[0]): k <= 2, then k <= 5, k <= 7, k <= 10, k <= 13 (one 16-row tile more per class), the rest: long tracks, k <= BA_LONG_KMAX
  const int* seg_ptr; const int* seg_k; const int* seg_tile; const int* seg_slot; const int* run_lm;
  const int* run_e0; const int* seg_cam;      // per run entry: pt_ptr[run_lm]; per segment slot (seg_slot[seg] + a): the slot's camera
  double* part_tiles;           // 36 per partial block
  double* part_coef;            // 6 per (segment, camera slot)
  int n_gpairs; const int* gpair_ptr; const int* gpair_i1; const int* gpair_i2; const int* gtile;
  const int* gcam_ptr; const int* gslot;   // nc + 1; partial-vector ids of a camera
  // chi2 partial sums
  double* chi_partial;
};
// A trial's prologue (lambda into device memory, status words and [S | rhs] cleared) handed to ba_launch_reduce: with the cuboid elimination on
// a side stream the prologue runs THERE, in front of it, and the landmark segments' kernel takes lambda by value -- nothing on the main stream
// reads what the prologue writes before the streams join, so its launch leaves the trial's chain.
struct BaSidePrologue { double* d_lam; double lam0, lam1; int* info24; int* elim_fail; double* S; size_t n_clear; };
enum { BA_SEG_LM = 32, BA_FUSED_KMAX = 13, BA_LONG_KMAX = 64, BA_ELIM_MAX_SLOTS = 64 };   // (6 k + 1 <= 80 rows = five tiles: 15 accumulator tiles per wavefront)

CS_HD double* ba_S_at(const BaView& v, int r, int c) {  // requires r >= c (and r - c < band_ld in band mode)
  return v.band_ld ? v.S + (size_t)c * v.band_ld + (r - c) : v.S + (size_t)r * v.n_red + c;
}

}  // namespace cs
