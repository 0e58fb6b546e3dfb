// detect_types.h -- device data layout of the proposal sweep (shared by detect_kernels.hip and
// detect_host.cpp).  See DESIGN.md "Path A: data layout in HBM".
#pragma once
#include <cstdint>

#include "cs_geom.h"

namespace cs {

// One (frame, box, height sample) of detect_cuboid(): the unit the reference's loops L2..L5 run over
// (box_proposal_detail.cpp:200-705).  All offsets index the pooled SoA arrays in DetectDeviceView.
struct JobDesc {
  BoxGeom g;
  int map_w;             // width of the distance map (right_expan - left_expan, :248)
  int m;                 // merged line segments inside the ROI (:292)
  int Y;                 // yaw samples (:184)
  int T;                 // top-edge samples (:219)
  int RP;                // roll x pitch samples (:344-355)
  int down_expand;       // height sample (:202)
  int line_off;          // -> mid_x / mid_y / line_angle
  int yaw_off;           // -> yaw / yaw_cos / yaw_sin
  int top_off;           // -> top_x
  int rp_off;            // -> rp (RpPose records)
  int vp_off;            // -> vp / bound: first (rp, yaw) entry of this job
  int frame, box, hid;
  long long map_off;     // -> maps (float index)
  long long slot_off;    // first candidate slot; slot = slot_off + ((rp*Y + yaw)*T + top)*2 + (config-1)
  double diag;           // obj_diaglength_expan (:207)
};

struct SweepParams {
  double vp12_thre_rad, vp3_thre_rad;  // 15 deg, 10 deg (:102-103)
  double short_thre;                   // 20 px (:104)
  double short_sq_bound;               // sqrt(x) < short_thre  <=>  x < short_sq_bound (see build_corners)
  int consider_config_1, consider_config_2;
};

// Candidate flag bits (one int per slot): low 2 bits = vp_1_position (0 = rejected), bit 2 = a
// negative half size after the 3D lift (the reference drops those only after normalisation, :766).
// Between compact_kernel and score_kernel the bits from CAND_JOB_SHIFT up carry the proposal's job index (score_kernel needs it
// for every lane and would otherwise search slot_prefix); score_kernel strips them again.
enum { CAND_VP_MASK = 3, CAND_NEG_SCALE = 4, CAND_JOB_SHIFT = 3 };

struct DetectDeviceView {
  const JobDesc* jobs;
  int n_jobs;
  const long long* slot_prefix;  // n_jobs + 1 (== jobs[j].slot_off, total at the end)
  const int* vp_prefix;          // n_jobs + 1
  const float* maps;
  const double* mid_x; const double* mid_y; const double* line_angle;
  const double* yaw; const double* yaw_cos; const double* yaw_sin;
  const int* top_x;
  const RpPose* rp;
  const double* invK;            // 9 per frame
  // sweep intermediates
  double* vp;                    // 6 per (job, rp, yaw): vp1.x vp1.y vp2.x vp2.y vp3.x vp3.y
  double* bound;                 // 6 per (job, rp, yaw): the 3x2 VP support angles (NaN = none)
  double* bound3;                // lean path: 2 per (job, rp) slot -- the third VP's support angles (they do not depend on yaw)
  // per-slot outputs
  int* flag;
  // per-job valid counts and compacted outputs
  int* job_valid;                // n_jobs
  long long* job_cbase;          // n_jobs + 1, exclusive scan of job_valid
  long long* c_slot;             // compacted (valid proposals in the reference's row order): slot id
  double* c_dist; double* c_angle; double* c_skew;   // written by score_kernel
  int* c_flag;
  // capacity layout (candidate_compact_kernel, the lean path): a job's compacted rows start at slot_prefix[j] (a multiple of 256) instead of
  // at the exclusive scan of the counts -- no pass over all jobs between the corner construction and the scorer.  blk_info: [0] = number of
  // blocks of 256 rows that hold valid rows (counted up by the jobs' workgroups as they finish; zeroed before the launch), then from [2] on
  // one (block index, valid rows) pair per such block: the scorer's work list.  null = the scanned (dense) layout.
  int* blk_info;
};

// Ranking stage on the device (fuse_normalize_scores_v2 + the skew-weighted final ranking).
struct RankParams {
  double w_angle;        // weight_vp_angle 0.8 (box_proposal_detail.cpp:109)
  double w_skew;         // weight_skew_error 1.5 (:110)
  double nominal_skew;   // nominal_skew_ratio
  double max_cut_skew;   // max_cut_skew
  int kmax;              // max_cuboid_num (<= RANK_KMAX on the device path)
  double short_sq_bound; // SweepParams::short_sq_bound: the winners' corners are rebuilt (slot_corners16)
};
enum { RANK_KMAX = 8 };

// One winner of the final ranking of a box.
struct RankWinner {
  long long slot;        // global proposal slot, -1 = none
  double normalized_error, dist_err, angle_err;
  int flag;              // vp_1_position
  int pad;
  double corners[16];
};

struct RankView {
  const int* box_job0;   // n_boxes: first job of the box (its height samples are consecutive jobs)
  const int* box_njobs;  // n_boxes
  int n_boxes;
  RankWinner* winners;   // n_boxes * kmax
  int* win_count;        // n_boxes
  int* fallback;         // n_boxes: 1 = a tie made the host ordering matter; redo this box on the host
  long long* last_slot;  // n_boxes or null: slot of the last proposal fuse_normalize_scores_v2 keeps for the box's last height sample
                         // (-1 = none) -- with roll/pitch sampling the next box of the frame starts from that proposal's camera yaw
};

// Lean roll/pitch path: the compacted columns of the boxes the ranking flagged (a tie that can reach the output or the carried
// proposal) are copied to a side pool before the next round reuses the per-round arrays; the host ranks them exactly afterwards.
struct RpSaveView {
  const int* fallback; const int* box_job0; const int* box_njobs; int n_boxes;
  unsigned long long* pool_used; long long pool_cap;
  long long* box_base;             // per box of the round: first pool entry, -1 = not saved (not flagged, or the pool is full)
  double* p_dist; double* p_angle; double* p_skew; int* p_flag; long long* p_slot;
};

struct RpCarryView {
  int n_frames, NT, YCAP;
  const int* prev_box_of_frame;     // [n_frames] box (index into the previous round's boxes) or -1
  const long long* prev_last_slot;  // previous round, per box
  const JobDesc* prev_jobs; const int* prev_box_job0; const int* prev_box_njobs;
  const int* job0_of_frame; const int* njobs_of_frame;   // this round: first job of the frame (-1: none), its height samples
  const int* tab_count;             // [n_frames * NT] yaw samples of list i of frame f
  int* cur_idx;                     // [n_frames] list in force
};

}  // namespace cs
