/* cubeslam_hip.h -- C ABI of the MI355X-native Cube SLAM hot path (libcubeslam_hip.so).
 *
 * Path A: detect_3d_cuboid::detect_cuboid() -- the cuboid proposal sampler/scorer.
 *   Replaces  void detect_3d_cuboid::detect_cuboid(const cv::Mat&, const Eigen::Matrix4d&,
 *             const Eigen::MatrixXd&, Eigen::MatrixXd, std::vector<ObjectSet>&)
 *             (detect_3d_cuboid/include/detect_3d_cuboid/detect_3d_cuboid.h:88-92,
 *              detect_3d_cuboid/src/box_proposal_detail.cpp:65-861)
 *   plus      set_calibration / set_cam_pose (box_proposal_detail.cpp:38-56), which become the K
 *             and T_wc arguments of every call.
 * Path B (g2o bundle adjustment) is declared further down.
 *
 * Conventions: all matrices are row-major C arrays; pixel coordinates are 0-based; every function
 * returns 0 on success or a negative cs_status; no C++ or torch types cross this boundary.
 * A cs_detector owns one HIP stream; calls on one handle must not overlap (the reference class is
 * not re-entrant either: it mutates its cam_pose member, box_proposal_detail.cpp:374,734).
 */
#ifndef CUBESLAM_HIP_H
#define CUBESLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cs_status {
  CS_OK = 0,
  CS_ERR_INVALID_ARG = -1,
  CS_ERR_HIP = -2,          /* a HIP runtime call failed; cs_last_error() has the text            */
  CS_ERR_NO_DEVICE = -3,    /* no gfx950 device visible: the library never falls back to the CPU */
  CS_ERR_CAPACITY = -4,
  CS_ERR_NOT_RUN = -5
} cs_status;

const char* cs_last_error(void);
int cs_device_count(void);
/* 1 when the library was built with -DCS_DIAG (make DIAG=1): only that build honours CS_DETECT_SKIP=<kernel names>, a timing
 * experiment that leaves kernels out of the sweep and makes the results meaningless.  The default build returns 0 and has no
 * switch that changes results; bench.py refuses to print a line from a diagnostic build. */
int cs_diag_build(void);

/* ------------------------------------------------------------------ Path A: detect_cuboid ----- */

/* Public flags of class detect_3d_cuboid (detect_3d_cuboid.h:95-117) + the constants hard-coded in
 * detect_cuboid() (box_proposal_detail.cpp:102-110,184,288-290).  cs_detect_default_params() fills
 * the reference's values.                                                                        */
typedef struct cs_detect_params {
  int consider_config_1;              /* detect_3d_cuboid.h:108 */
  int consider_config_2;              /* :109 */
  int whether_sample_cam_roll_pitch;  /* :110 */
  int whether_sample_bbox_height;     /* :111 */
  int max_cuboid_num;                 /* :113 */
  double nominal_skew_ratio;          /* :114 */
  double max_cut_skew;                /* :115 */
  double yaw_range_deg;               /* 45  (box_proposal_detail.cpp:184) */
  double yaw_step_deg;                /* 6   (box_proposal_detail.cpp:184) */
  double vp12_edge_angle_thre;        /* 15  (:102) */
  double vp3_edge_angle_thre;         /* 10  (:103) */
  double shorted_edge_thre;           /* 20  (:104) */
  double weight_vp_angle;             /* 0.8 (:109) */
  double weight_skew_error;           /* 1.5 (:110) */
  double pre_merge_dist_thre;         /* 20  (:288) */
  double pre_merge_angle_thre;        /* 5   (:289) */
  double edge_length_threshold;       /* 30  (:290) */
  int host_threads;                   /* worker threads for the host stages; 0 = hardware concurrency */
} cs_detect_params;

void cs_detect_default_params(cs_detect_params* p);

/* POD mirror of class cuboid (detect_3d_cuboid.h:20-41). */
typedef struct cs_cuboid {
  double pos[3];
  double scale[3];                   /* half sizes */
  double rotY;
  double box_config_type[2];         /* configuration id, vp1 left(1)/right(2) */
  int32_t box_corners_2d[16];        /* 2x8 row-major, row 0 = x */
  double box_corners_3d_world[24];   /* 3x8 row-major */
  double rect_detect_2d[4];
  double edge_distance_error;
  double edge_angle_error;
  double normalized_error;
  double skew_ratio;
  double down_expand_height;
  double camera_roll_delta;
  double camera_pitch_delta;
} cs_cuboid;

/* Region of the image whose distance transform the scorer reads for (box, height sample k)
 * (box_proposal_detail.cpp:242-248,320).  A distance map handed to this library is the float32
 * height x width result of cv::distanceTransform(255 - cv::Canny(gray(roi),80,200), DIST_L2, 3)
 * (:324-327).  The reference indexes it without bounds checks and, when a cuboid corner sits on the
 * ROI's far edge, reads one row / one column past it (object_3d_util.cpp:651 with the inclusive test
 * at :241); this library defines those reads as 0.0f.                                             */
typedef struct cs_roi {
  int left, top, width, height;
  int down_expand;                   /* 0 / half / full height expansion of this sample (:160-172) */
} cs_roi;

/* Host-only helper: ROIs of the 1..3 height samples of one box [x y w h prob]; returns their count. */
int cs_box_rois(const double box5[5], int img_w, int img_h, int whether_sample_bbox_height, cs_roi out[3]);

/* Host-only helper: the ZYX Euler angles (roll, pitch, yaw) set_cam_pose() stores in cam_pose.euler_angle
 * (box_proposal_detail.cpp:49-50 via matrix_utils.cpp:38-49); callers read cam_pose_raw.euler_angle back
 * after detect_cuboid() (object_slam/src/main_obj.cpp:664).                                          */
int cs_cam_euler_zyx(const double T_wc[16], double euler3[3]);

typedef struct cs_detector cs_detector;
typedef struct cs_batch cs_batch;

/* One frame worth of detect_cuboid() arguments. */
typedef struct cs_frame_desc {
  const double* K;                   /* 3x3  (set_calibration)                                    */
  const double* T_wc;                /* 4x4  camera-to-world (transToWolrd)                        */
  int img_w, img_h;                  /* rgb_img.cols / rows                                       */
  const double* boxes;               /* n_boxes x 5: x y w h prob (obj_bbox_coors)                */
  int n_boxes;
  const double* lines;               /* n_lines x 4: x1 y1 x2 y2 (all_lines_raw)                  */
  int n_lines;
  const float* const* dist_maps;     /* [n_boxes*3]: map of (box i, height sample k) at [3*i+k],  */
                                     /* height*width floats of the cs_box_rois() ROI; host memory  */
} cs_frame_desc;

int cs_detector_create(const cs_detect_params* params, int device, cs_detector** out);
void cs_detector_destroy(cs_detector* d);

/* Drop-in single call: what the adapter's detect_cuboid() executes.  out: n_boxes x max_cuboid_num
 * records (box-major), out_counts[n_boxes] = cuboids returned per box (the size of each ObjectSet). */
int cs_detect_cuboids(cs_detector* d, const cs_frame_desc* frame, cs_cuboid* out, int* out_counts);

/* ---- distance-map front end --------------------------------------------------------
 * What detect_cuboid() does before the sweep (box_proposal_detail.cpp:84, :320-327): cvtColor(BGR2GRAY), then per
 * (box, height sample) cv::Canny(gray_img(object_bbox), im_canny, 80, 200) and
 * cv::distanceTransform(255 - im_canny, dist_map, CV_DIST_L2, 3).  OpenCV is a third-party dependency of the
 * reference; these entry points implement the published algorithms of its imgproc module (3x3 Sobel, L1 magnitude,
 * 4-sector non-maximum suppression, hysteresis; two-pass 3x3 chamfer in 16.16 fixed point), integer arithmetic
 * throughout.  The adapter may keep calling OpenCV instead.                                     */
int cs_bgr_to_gray(const unsigned char* bgr, int n_pixels, unsigned char* gray);               /* host helper      */
/* out_maps[k]: rois[k].height x rois[k].width floats (host memory), the dist_map of ROI k of the gray image.     */
int cs_edge_distance_maps(cs_detector* d, const unsigned char* gray, int img_w, int img_h, const cs_roi* rois, int n_rois, float* const* out_maps);
/* The same over several images of one size: ROI k belongs to image roi_image[k].  out_maps may be NULL (maps stay on
 * the device and are discarded: timing); kernel_ms, if not NULL, receives the device time of the two kernels.      */
int cs_edge_distance_maps_multi(cs_detector* d, const unsigned char* const* grays, int n_images, int img_w, int img_h, const cs_roi* rois, const int* roi_image,
                                int n_rois, float* const* out_maps, double* kernel_ms);
/* cs_batch_create() for image input: grays[f] is frame f's 8-bit gray image (img_h x img_w, all frames one size); the
 * frames' dist_maps are ignored and every (box, height sample) map is produced in HBM by the kernels above.          */
int cs_batch_create_gray(cs_detector* d, const cs_frame_desc* frames, const unsigned char* const* grays, int n_frames, cs_batch** out);
/* New images for the same frame descriptions (the caller of detect_cuboid hands over an image per call, box_proposal_detail.cpp:84,320-327;
 * a batch of a fixed rig keeps its boxes' layout): grays[f] as above.  Asynchronous: the upload runs on a copy stream of the batch into its
 * second image buffer -- beside a front end / sweep that is still running --, images that are contiguous in host memory go up as ONE copy
 * (pinned host memory gives the copy engine its full rate); the next cs_batch_submit / cs_batch_run waits for the OLDEST queued upload and
 * queues the front end over it in front of its sweep.  Up to two uploads may be queued (one per image buffer; a third returns
 * CS_ERR_INVALID_ARG): queueing the upload after next before each submit keeps the copy engine busy without a gap.  The host images must
 * stay valid until cs_batch_refill_wait() returns (it waits for every queued upload).  */
int cs_batch_refill_gray(cs_detector* d, cs_batch* batch, const unsigned char* const* grays);
int cs_batch_refill_wait(cs_batch* batch);
/* cs_detect_cuboids() with the frame's dist_maps computed here from the gray image (frame->dist_maps is ignored). */
int cs_detect_cuboids_gray(cs_detector* d, const cs_frame_desc* frame, const unsigned char* gray, cs_cuboid* out, int* out_counts);

/* ---- line-segment producer ------------------------------------------------------------------
 * Replaces  void line_lbd_detect::detect_filter_lines(const cv::Mat& gray_img, cv::Mat& linesmat_out)
 *           (line_lbd/include/line_lbd/line_lbd_allclass.h:23-79, line_lbd/class/line_lbd_allclass.cpp:199-235) with use_LSD = false:
 *           the EDLines detector of line_lbd/libs/binary_descriptor.cpp (BinaryDescriptor::detect :421-590, OctaveKeyLines :796-1148,
 *           EDLineDetector::EdgeDrawing / EDline :1583-2905), one octave, as the graph driver configures it
 *           (object_slam/src/main_obj.cpp:502-505,593; length_thres = its line_length_thres, 15 there, 50 by default).
 * gray: img_h x img_w 8-bit (cvtColor BGR2GRAY of the input: cs_bgr_to_gray).  lines4: cap x 4 floats x1 y1 x2 y2 (the CV_32F
 * rows of linesmat_out, start / end in the reference's order); *n_lines = rows written.  Per-pixel stages on the device, the
 * routing / fitting / validation chain on the host.                                                                        */
int cs_detect_lines_gray(cs_detector* d, const unsigned char* gray, int img_w, int img_h, double length_thres, float* lines4, int cap, int* n_lines);
/* The same for n_images gray images of one size: the per-pixel stages of all of them on the device, the sequential halves side by
 * side on the detector's worker pool.  lines4[i]: image i's segments (cap rows of 4 floats), n_lines[i]: their number.           */
int cs_detect_lines_batch(cs_detector* d, const unsigned char* const* grays, int n_images, int img_w, int img_h, double length_thres,
                          float* const* lines4, int cap, int* n_lines);
/* Timing (ms) of this detector's last line-detection call: the device kernels, the host stage (wall), the whole call.            */
int cs_detect_lines_last_timing(cs_detector* d, double* device_ms, double* host_ms, double* total_ms);
/* The same producer with use_LSD = true (line_lbd_allclass.cpp:130-150,199-215): LSDDetector::detectImpl, one octave
 * (line_lbd/libs/LSDDetector.cpp:55-105,154-260 -- end points clamped into the image, segments hugging a border dropped) over the
 * vendored OpenCV-3 detector LineSegmentDetectorImpl::flsd with LSD_REFINE_ADV and its default parameters
 * (line_lbd/libs/lsd.cpp:185-187,402-1148), then filter_lines (length > length_thres).  This is the detector whose output the
 * reference ships for its bundled frame (detect_3d_cuboid/data/edge_detection/LSD/0000_edge.txt).  Same arguments and layout as
 * cs_detect_lines_gray / _batch / _last_timing.  Blur, resize and the gradient / level-line-angle planes on the device; region
 * growing, rectangle fitting and the a-contrario validation (sequential in the pixel-visit order) on the host.                   */
int cs_detect_lsd_gray(cs_detector* d, const unsigned char* gray, int img_w, int img_h, double length_thres, float* lines4, int cap, int* n_lines);
int cs_detect_lsd_batch(cs_detector* d, const unsigned char* const* grays, int n_images, int img_w, int img_h, double length_thres,
                        float* const* lines4, int cap, int* n_lines);
int cs_detect_lsd_last_timing(cs_detector* d, double* device_ms, double* host_ms, double* total_ms);

/* Batched form for throughput: cs_batch_create() copies the frames' inputs into HBM (maps, lines,
 * boxes, cameras); cs_batch_run() is the hot path proper -- resident inputs in, cuboids out.
 * out / out_counts are laid out frame-major with stride max_boxes = max over frames of n_boxes:
 * out[(f*max_boxes + i)*max_cuboid_num + k], out_counts[f*max_boxes + i].                         */
int cs_batch_create(cs_detector* d, const cs_frame_desc* frames, int n_frames, cs_batch** out);
int cs_batch_max_boxes(const cs_batch* b);
int cs_batch_run(cs_detector* d, cs_batch* b, cs_cuboid* out, int* out_counts);
/* cs_batch_run() in two halves.  cs_batch_submit packs the batch on the host and queues the whole sweep on the detector's streams
 * without waiting for it; cs_batch_collect waits, writes `out` / `out_counts` (which must stay valid until then) and the timing.  A
 * throughput caller that owns several batches keeps the next one queued while the current one is on the device, so the host stages
 * of one batch overlap the sweep of another.  One outstanding submit per batch; the batches of one detector are submitted and
 * collected by one thread, in submission order.  (Configurations that need host round trips inside the sweep -- roll/pitch
 * sampling, debug retention, chunked pipeline -- run to completion inside cs_batch_submit; collect is then a no-op.)            */
int cs_batch_submit(cs_detector* d, cs_batch* b, cs_cuboid* out, int* out_counts);
int cs_batch_collect(cs_detector* d, cs_batch* b);
void cs_batch_destroy(cs_batch* b);

/* Timing of the last cs_batch_run() (milliseconds; kernel times from hipEvents on the detector's
 * stream, host stages from a monotonic clock).                                                    */
typedef struct cs_detect_timing {
  double setup_host_ms, h2d_ms, vp_kernel_ms, cand_kernel_ms, compact_ms, d2h_ms, rank_host_ms, finalize_ms, total_ms;
  long long n_jobs, n_slots, n_valid;
  long long cand_kernel_bytes;       /* algorithmic bytes of the candidate kernel     */
  int cand_kernel_launches;
  int n_fallback_boxes;              /* boxes whose ranking hit a tie and was redone on the host  */
  int n_redo_frames;                 /* roll/pitch sampling, lean path: frames redone round by round because of such a box */
  double rank_kernel_ms;
  double line_setup_ms;              /* line_setup_kernel (ROI filter + merge_break_lines on the device)  */
  double score_kernel_ms;            /* score_kernel: distance + angle errors of the valid proposals       */
  long long score_kernel_bytes;      /* its algorithmic bytes                                   */
} cs_detect_timing;
int cs_batch_last_timing(const cs_batch* b, cs_detect_timing* t);

/* Bit 0: retain every valid proposal (rows, corners, kept ids) on the host during cs_batch_run() for the
 * cs_batch_debug_* getters (off by default: it costs the device-to-host copy of every proposal).
 * Bit 1: force the ranking stage onto the host (the exact std::partial_sort path that otherwise only handles
 * ties and roll/pitch sampling).  Bit 2: force the line setup (ROI filter + merge_break_lines) onto the host.
 * Bit 3: take the general round-based code path even where the lean single-pass path applies. */
int cs_batch_set_debug(cs_batch* b, int enable);

/* Without roll/pitch sampling the boxes are independent and cs_batch_run() can cut the batch into n_chunks
 * chunks that flow through a two-slot pipeline (host packing / record writing of one chunk overlaps the sweep of
 * the next).  Default 1: on MI355X the sweep's kernels are long enough that one pass wins. */
int cs_batch_set_pipeline_chunks(cs_batch* b, int n_chunks);

/* Stage-by-stage inspection after cs_batch_run(), for parity tests.  (frame, box, k) names one
 * (box, height sample) job.  Rows follow all_configs_error_one_objH (box_proposal_detail.cpp:677-690):
 * [config, vp1 position, yaw, top sample id, dist error, angle error, down expand, roll, pitch];
 * corners are the 2x8 box_corners_2d_float (:628-630), row-major.  Pass NULL to skip an output;
 * cap = capacity in candidates.  Returns the number of valid candidates, or a negative cs_status. */
int cs_batch_debug_candidates(cs_batch* b, int frame, int box, int k, int cap, double* rows9, double* corners16);
/* good_proposal_ids / normalized_score of fuse_normalize_scores_v2 (object_3d_util.cpp:726-837). */
int cs_batch_debug_kept(cs_batch* b, int frame, int box, int k, int cap, int* keep_ids, double* scores);


/* ------------------------------------------------------------------ Path B: g2o bundle adjustment -- */
/* Replaces, for graphs made of VertexSE3Expmap / VertexSBAPointXYZ / VertexCuboid and EdgeSE3ProjectXYZ /
 * EdgeSE3Cuboid / EdgeSE3Expmap, what g2o::BlockSolver + OptimizationAlgorithmLevenberg do per iteration:
 *   g2o::Solver virtuals  object_slam/Thirdparty/g2o/g2o/core/solver.h:53-132
 *                         (buildStructure, buildSystem, setLambda, restoreDiagonal, solve, x(), b())
 *   their implementation  core/block_solver.hpp:142-295 (structure), :501-560 (buildSystem),
 *                         :563-604 (lambda), :353-486 (Schur complement solve)
 *   the edge virtuals the solver drives: computeError / linearizeOplus / constructQuadraticForm
 *                         core/optimizable_graph.h:382-525, core/base_binary_edge.hpp:54-205
 *   and the LM driver     core/optimization_algorithm_levenberg.cpp:61-189, core/sparse_optimizer.cpp:354-435.
 * The problem is described in flat arrays (a C++ adapter packs them from activeVertices()/activeEdges(),
 * see INTEGRATION.md).  Poses are 7 doubles x y z qx qy qz qw (SE3Quat::toVector, types/se3quat.h:151-163);
 * a camera vertex stores world-to-camera; a cuboid is its object-to-world pose + 3 half sizes
 * (g2o::cuboid::toVector, object_slam/include/object_slam/g2o_Object.h:145-151).
 * Vertex ids, which fix the ordering of the system like g2o's sort-by-id (sparse_optimizer.cpp:166-190):
 * cuboids_first ? (cuboids, then cameras) : (cameras, then cuboids); points follow and are marginalised
 * (setMarginalized(true)), so the pose block is solved through the Schur complement.                  */
typedef struct cs_ba cs_ba;

int cs_ba_create(int device, cs_ba** out);
void cs_ba_destroy(cs_ba* ba);
int cs_ba_set_vertices(cs_ba* ba, const double* cams7, const int* cam_fixed, int n_cams,
                       const double* cuboids10, const int* cub_fixed, int n_cuboids,
                       const double* points3, const int* pt_fixed, int n_points, int cuboids_first);
/* New estimates for the same graph (after g2o-side update()/pop()): no structure rebuild.  NULL = keep. */
int cs_ba_set_estimates(cs_ba* ba, const double* cams7, const double* cuboids10, const double* points3);
/* EdgeSE3ProjectXYZ (types/types_six_dof_expmap.h:145-174): vertex 0 = point, vertex 1 = camera;
 * info4 = 2x2 information, intr4 = fx fy cx cy, huber[k] <= 0 means no robust kernel (NULL: none). */
int cs_ba_set_edges_proj(cs_ba* ba, int n, const int* point, const int* cam, const double* uv2,
                         const double* info4, const double* intr4, const double* huber);
/* EdgeSE3Cuboid (g2o_Object.h:235-260): vertex 0 = camera, vertex 1 = cuboid; meas10 = cuboid in the
 * camera frame, info81 = 9x9 information.                                                          */
int cs_ba_set_edges_cuboid(cs_ba* ba, int n, const int* cam, const int* cuboid, const double* meas10, const double* info81);
/* EdgeSE3Expmap (types_six_dof_expmap.h:83-99): error = log(meas * T_i * T_j^-1); info36 = 6x6.      */
/* EdgeSE3CuboidProj (object_slam/include/object_slam/g2o_Object.h:264-293): 4-dim error = bounding rectangle
 * (centre x, centre y, width, height) of the cuboid's 8 projected corners (cuboid::projectOntoImageBbox :181-197) minus
 * the measured one; vertex 0 = camera, vertex 1 = cuboid; numeric Jacobians like EdgeSE3Cuboid.  meas4: n x 4,
 * info16: n x 16 (row-major 4 x 4), K9: n x 9 (the edge's public Kalib member, row-major).  In g2o's edge order these
 * follow the EdgeSE3Cuboid edges. */
int cs_ba_set_edges_cuboid_proj(cs_ba* ba, int n, const int* cam, const int* cub, const double* meas4, const double* info16, const double* K9);
int cs_ba_set_edges_odom(cs_ba* ba, int n, const int* cam_i, const int* cam_j, const double* meas7, const double* info36);
/* Robust kernels on any edge class (OptimizableGraph::Edge::setRobustKernel, core/optimizable_graph.h:419-423): the classes the
 * reference registers in core/robust_kernel_impl.cpp:169-174, evaluated as :78-165 evaluate them -- rho(e) enters chi2
 * (sparse_optimizer.cpp:100-114), rho'(e) weights Omega and -Omega e in the quadratic form (base_binary_edge.hpp:88-111,
 * base_unary_edge.hpp:55-63, base_edge.h:96-102; rho'' is unused there).  delta[k] = RobustKernel::delta() (for DCS: phi; for Tukey:
 * _deltaSqr = delta^2, _invDeltaSqr = 1 / delta^2 as single-precision members, as in the reference).  Huber's delta^2 is a
 * single-precision member too (robust_kernel_impl.h:86) and is used as such.  Call after the class's edges are set; n = the class's
 * edge count; kind == NULL removes the class's kernels (n is then ignored).  cs_ba_set_edges_proj's `huber` argument is shorthand for
 * (CS_RK_HUBER, huber[k]) where huber[k] > 0.  Edges appended later carry no kernel (projection edges: their `huber` value).     */
enum cs_robust_kernel { CS_RK_NONE = 0, CS_RK_HUBER = 1, CS_RK_PSEUDO_HUBER = 2, CS_RK_CAUCHY = 3, CS_RK_SATURATED = 4, CS_RK_DCS = 5, CS_RK_TUKEY = 6 };
enum cs_edge_class { CS_EDGE_PROJ = 0, CS_EDGE_CUBOID = 1, CS_EDGE_CUBOID_PROJ = 2, CS_EDGE_ODOM = 3 };
int cs_ba_set_robust_kernels(cs_ba* ba, int edge_class, int n, const int* kind, const double* delta);

/* External (host-evaluated) edges: the CPU path for edge types the library does not evaluate -- what g2o's BlockSolver does for
 * EVERY edge, kept for the ones that have no kernel here.  The caller runs such an edge through its own virtuals (computeError,
 * linearizeOplus(JacobianWorkspace&), constructQuadraticForm: core/optimizable_graph.h:394-454; the numeric default of linearizeOplus:
 * core/base_binary_edge.hpp:130-205, core/base_unary_edge.hpp:82-123; the quadratic form incl. robust kernel: base_binary_edge.hpp:54-120,
 * base_unary_edge.hpp:42-72) and hands over what those accumulate (cube_slam_wu_amd/adapters/block_solver_hip.h does exactly this):
 *   cs_ba_set_external_edges   the coupling pattern (structure: ordering, bandwidth).  Edge k joins vertex (class_i[k], idx_i[k]) and
 *                              (class_j[k], idx_j[k]); idx_j[k] < 0 = unary.  A binary edge may join cameras and cuboids; a (marginalised)
 *                              point takes unary terms only.  Indices are the caller's per-class vertex indices.
 *   cs_ba_set_external_terms   the terms at the CURRENT estimates, to be refreshed before every cs_ba_build_system(): per vertex the sum
 *                              over its external edges of A_ii (cam36: n_cams x 36, cub81, pt9; row-major, symmetric) and b_i (cam6,
 *                              cub9, pt3); a NULL pair = the class has none.  Hij81: n x 81, the off-diagonal block A_i^T Omega A_j of
 *                              binary edge k, row-major dim_i x dim_j in the first dim_i * dim_j entries.  chi2: the edges' summed
 *                              (robustified) chi2, which cs_ba_compute_errors() adds to the device edges'.  Terms of fixed vertices are ignored.
 *   cs_ba_set_external_chi2    the chi2 share alone (after an update, when only the error is needed).
 *   cs_ba_set_external_callback  for cs_ba_optimize(), whose LM loop runs inside the library: fn(ctx, ba, want_system) is called before
 *                              every linearisation (want_system = 1: read the state with cs_ba_get_state, call cs_ba_set_external_terms)
 *                              and after every trial's update (want_system = 0: call cs_ba_set_external_chi2 ONLY -- a rejected trial
 *                              re-solves the system built before it); it returns 0 on success.  Each call is a host round trip of the
 *                              state: a fallback for a handful of unusual edges, not a fast path.
 * Not available on a sharded handle.  A binary external edge on a cuboid keeps the cuboids in the reduced system (cs_ba_reduced_size). */
enum cs_vertex_class { CS_VERTEX_CAM = 0, CS_VERTEX_CUBOID = 1, CS_VERTEX_POINT = 2 };
typedef int (*cs_external_fn)(void* ctx, cs_ba* ba, int want_system);
int cs_ba_set_external_edges(cs_ba* ba, int n, const int* class_i, const int* idx_i, const int* class_j, const int* idx_j);
int cs_ba_set_external_terms(cs_ba* ba, const double* cam36, const double* cam6, const double* cub81, const double* cub9, const double* pt9, const double* pt3,
                             const double* Hij81, double chi2);
int cs_ba_set_external_chi2(cs_ba* ba, double chi2);
int cs_ba_set_external_callback(cs_ba* ba, cs_external_fn fn, void* ctx);

/* The g2o::Solver / SparseOptimizer steps, one call each (all state stays in HBM):                   */
int cs_ba_compute_errors(cs_ba* ba, double* robust_chi2);   /* computeActiveErrors + activeRobustChi2 */
int cs_ba_build_system(cs_ba* ba);                           /* Solver::buildSystem (needs current errors) */
int cs_ba_solve(cs_ba* ba, double lambda, int* positive_definite); /* setLambda + solve + restoreDiagonal */
/* Growing graphs (the reference adds a frame and calls optimize(5): main_obj.cpp:802-803; g2o's seam is Solver::updateStructure,
 * core/solver.h:62).  The new vertices / edges are appended behind the existing ones (indices continue); estimates that live on the
 * device -- possibly optimised there -- are kept; the structure phase runs again on the next solve.  Huber deltas must be given for
 * all projection edges or for none.                                                                                               */
int cs_ba_append_vertices(cs_ba* ba, const double* cams7, const int* cam_fixed, int n_cams, const double* cuboids10, const int* cub_fixed, int n_cuboids,
                          const double* points3, const int* pt_fixed, int n_points);
int cs_ba_append_edges_proj(cs_ba* ba, int n, const int* pt, const int* cam, const double* uv, const double* info4, const double* intr4, const double* huber);
int cs_ba_append_edges_cuboid(cs_ba* ba, int n, const int* cam, const int* cub, const double* meas10, const double* info81);
int cs_ba_append_edges_cuboid_proj(cs_ba* ba, int n, const int* cam, const int* cub, const double* meas4, const double* info16, const double* K9);
int cs_ba_append_edges_odom(cs_ba* ba, int n, const int* cam_i, const int* cam_j, const double* meas7, const double* info36);
int cs_ba_update(cs_ba* ba);                                 /* SparseOptimizer::update(x)              */
int cs_ba_push(cs_ba* ba);                                   /* SparseOptimizer::push / pop / discardTop */
int cs_ba_pop(cs_ba* ba);

/* SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg; returns (in *iterations_done)
 * what g2o returns.  History arrays (may be NULL) receive chi2 / lambda / LM trials of every iteration. */
int cs_ba_optimize(cs_ba* ba, int iterations, int* iterations_done, double* chi2_hist, double* lambda_hist, int* trials_hist, int hist_cap);

/* Sharded BA over the GPUs of one node (one process per GPU).  Every rank describes the FULL problem
 * (cs_ba_set_vertices / set_edges_*) and then calls cs_ba_set_shard(rank, n_ranks) (or cs_ba_comm_init).
 *
 * Separator mode (the banded reduced system, every rank's share wide enough; cs_ba_shard_info reports it): rank r owns the
 * columns [cut_r, cut_r+1) of the band in solver order -- its first >= bandwidth columns are the separator Z_r, the rest its
 * interior -- and every landmark (with all its projection edges), cuboid and odometry edge whose lowest column falls into that
 * range.  Residuals, Jacobians, landmark blocks and Schur products are rank-local and land in the rank's own columns and in the
 * diagonal block of the next separator; each rank factorises ITS interior only, the ranks exchange the separators' Schur
 * complements (one all-gather of 3 w^2 + 2 w doubles per rank, w = separator width ~ bandwidth: ~0.35 MB at KITTI shape), every
 * rank solves the small separator system, back-substitutes its interior, and one all-reduce of the solution vector (8 n_pose bytes)
 * hands every rank all increments (block_solver.hpp:385-485 is what a shard builds and solves).  A third, three-double all-reduce
 * carries [chi2, LM scale term, "a factorisation failed"].
 *
 * Fallback (dense reduced system, or shares narrower than the band): landmarks go to the rank of the camera subsequence
 * [rank*Nc/R, (rank+1)*Nc/R) of their first observing camera, cuboid / odometry edges to their camera's; the ranks' partial
 * reduced systems [S | b_schur] are summed with ONE all-reduce per damped solve and the reduced solve is replicated.
 *
 * `fn` performs an in-place all-reduce of n doubles at `data` (device memory if on_device != 0, else host), op 0 = SUM, 1 = MAX;
 * it must return 0 on success and have completed when it returns (an all-gather is issued through it as the sum of zero-padded
 * buffers).  With n_ranks == 1 (or fn == NULL and no communicator) this is cs_ba_optimize.                                   */
typedef int (*cs_allreduce_fn)(void* ctx, void* data, size_t n_doubles, int on_device, int op);
int cs_ba_set_shard(cs_ba* ba, int rank, int n_ranks);
/* RCCL (librccl, the collectives run over xGMI): the library issues ncclAllReduce itself, on the handle's own stream, queued
 * behind the kernels that produce the data -- no host round trip, no caller code in the loop.  Rank 0 calls
 * cs_ba_comm_unique_id() (ncclGetUniqueId) and hands the 128 bytes to every rank by whatever means the application has (MPI,
 * torch.distributed, a file); every rank then calls cs_ba_comm_init(), which creates the communicator (ncclCommInitRank: a
 * collective call) and sets the shard like cs_ba_set_shard().  cs_ba_optimize_sharded(..., fn = NULL, ...) then uses it: per LM
 * trial the collectives above, queued on the stream, and one host synchronisation.  One communicator handle per process.    */
int cs_ba_comm_unique_id(unsigned char id128[128]);
int cs_ba_comm_init(cs_ba* ba, int rank, int n_ranks, const unsigned char id128[128]);
int cs_ba_optimize_sharded(cs_ba* ba, int iterations, cs_allreduce_fn fn, void* ctx, int* iterations_done,
                           double* chi2_hist, double* lambda_hist, int* trials_hist, int hist_cap);
/* Host-only: the rank that owns each landmark under the FALLBACK rule (no GPU needed; used by the CPU multi-process test). */
int cs_ba_shard_landmark_owners(int n_ranks, int n_cams, int n_points, int n_proj, const int* e_pt, const int* e_cam, int* owner_out);
/* The rule in force for this handle (after the structure phase): owner of every landmark; how the sharded solve is organised --
 * sep_mode 1 = separator mode, n_sep / w_max = size of the separator system / widest separator, bytes_per_trial = what this rank
 * contributes to the collectives of one LM trial, bytes_per_trial_allreduce = what the all-reduce of [S | b] would move instead,
 * interior_n = unknowns this rank factorises.  Any pointer may be NULL.                                                          */
int cs_ba_get_landmark_owners(cs_ba* ba, int* owner_out);
/* Accumulated stage times (ms) of the separator-mode solves: interior factorisation, separator message, gather (through a callback:
 * including the host round trip), separator system, interior back-substitution -- they partition cs_ba_timing.factor_ms. */
int cs_ba_shard_timing(cs_ba* ba, double out5[5]);
int cs_ba_shard_info(cs_ba* ba, int* sep_mode, int* n_sep, int* w_max, long long* bytes_per_trial, long long* bytes_per_trial_allreduce, int* interior_n);

int cs_ba_get_state(cs_ba* ba, double* cams7, double* cuboids10, double* points3);
int cs_ba_sizes(cs_ba* ba, int* size_pose, int* size_landmarks);
/* How the reduced (pose) system is solved: band_ld > 0 = reverse Cuthill-McKee ordering + banded Cholesky with
 * band_ld = bandwidth + 1 (one persistent kernel, `team` workgroups); band_ld == 0 = dense rocSOLVER potrf/potrs
 * (graphs whose bandwidth exceeds half the system, systems under 128 unknowns). */
int cs_ba_solver_layout(cs_ba* ba, int* band_ld, int* team);
/* The system the solver factorises.  g2o's reduced system holds cameras and cuboids (only the points are marginalised); since a cuboid
 * is coupled to its observing cameras only, the library may eliminate the cuboids' 9 x 9 blocks as well -- the same exact block
 * elimination as for the landmarks, i.e. a different elimination order of the same Cholesky factorisation -- when that makes the
 * banded factorisation cheaper (C4: 10 494 -> 5 994 unknowns, bandwidth 182 -> 119).  n_reduced = unknowns of the factorised system;
 * cuboids_eliminated = 1 if it holds the cameras only.  x() / b() / cs_ba_sizes keep g2o's layout either way.
 * CS_BA_KEEP_CUBOIDS=1 (environment) keeps g2o's system.                                                              */
int cs_ba_reduced_size(cs_ba* ba, int* n_reduced, int* cuboids_eliminated);
/* Which factorisation the reduced system takes (the place of g2o's LinearSolverDense / LinearSolverEigen, solvers/linear_solver_dense.h:65-113,
 * solvers/linear_solver_eigen.h:94-232): CS_BA_PATH_BAND (reverse Cuthill-McKee band, persistent banded Cholesky), CS_BA_PATH_SPARSE (minimum-degree
 * block ordering, symbolic factorisation on the host, level-scheduled sparse Cholesky on the device: graphs the ordering cannot band -- 2-D
 * covisibility meshes, many loop closures), CS_BA_PATH_DENSE (rocSOLVER potrf / potrs: small systems, graphs that fill in anyway).  *bandwidth:
 * the band's (0 otherwise); *sparse_fill: values of the sparse factor / those of the dense triangle (0 otherwise).  CS_BA_FORCE_DENSE=1,
 * CS_BA_SPARSE=0 / 1 (never / whenever the plan fits) override the choice.                                                             */
typedef enum { CS_BA_PATH_DENSE = 0, CS_BA_PATH_BAND = 1, CS_BA_PATH_SPARSE = 2 } cs_ba_solver_path_kind;
int cs_ba_solver_path(cs_ba* ba, int* path, int* bandwidth, double* sparse_fill);
/* How a CS_BA_PATH_BAND system is eliminated (round 5): *block_cyclic_reduction = 1: odd-even (nested dissection) order over blocks of 128
 * unknowns, ceil(log2(n / 128 + 1)) levels of one 128-column factorisation each -- bcr_kernels.hip; bandwidth <= 128 and the levels cheaper than
 * the persistent banded kernels' chain of 32-column steps (C4: 6 levels against 53 steps); 0: those kernels (wider bands, short systems).  *levels:
 * the number of levels (0 otherwise).  CS_BAND_BCR=0 / 2 (environment): never / whenever the band allows it.                          */
int cs_ba_band_order(cs_ba* ba, int* block_cyclic_reduction, int* levels);
/* How the Schur complement S -= sum_j W_j D_j^-1 W_j^T (block_solver.hpp:385-431) is formed.  fused = 1: landmarks grouped by
 * camera set, one wavefront per segment of <= 32 landmarks, the product on the matrix cores (v_mfma_f64_16x16x4_f64) with the
 * landmarks as contraction dimension, n_partial_blocks partial 6x6 blocks summed per destination in a fixed order; fused = 0
 * (some landmark is seen by more than 64 cameras, or more than a quarter of them by more than 13, or CS_BA_SCHUR_PAIRS=1): one wavefront
 * per covisible camera pair.  Tracks of 14 .. 64 views ride in the fused schedule through a plain multiply-add kernel.
 * n_blocks = 6x6 blocks of S the landmarks touch.                                                                  */
int cs_ba_schur_layout(cs_ba* ba, int* fused, int* n_segments, int* n_partial_blocks, int* n_blocks);
/* Inspection for tests: one 64-bit hash per index table the structure phase (BlockSolver::buildStructure, block_solver.hpp:142-295)
 * leaves on the device, in a fixed order; n_tables = how many there are (out holds min(cap, n_tables)).  Two builds of the same graph
 * -- threaded and sequential (CS_BA_STRUCT_THREADS=1), appended frame by frame and set up at once -- agree table by table.          */
int cs_ba_structure_digest(cs_ba* ba, unsigned long long* out, int cap, int* n_tables);
/* Inspection for parity tests (host copies, caller-sized): dense Hpp (size_pose^2, no lambda), Hll (9 per
 * free point in point order), Hpl (18 per projection edge in the caller's edge order), b, x.            */
int cs_ba_get_system(cs_ba* ba, double* Hpp_dense, double* Hll9, double* Hpl18, double* b, double* x);
/* Solver::computeMarginals (core/block_solver.hpp:488-499 -> LinearSolver::solvePattern on _Hpp, core/marginal_covariance_cholesky.cpp:154-222):
 * blocks of the inverse of the pose-pose Hessian as cs_ba_build_system left it (no lambda, no Schur complement -- what the reference's call
 * factorises).  Pair k = rows of vertex (class_i[k], idx_i[k]) x columns of vertex (class_j[k], idx_j[k]) (cs_vertex_class: cameras 6, cuboids
 * 9; free vertices only), written row-major one after the other into `out`.  *positive_definite = 0 when H_pp cannot be factorised (g2o
 * returns false there).  A query call (dense rocSOLVER factorisation of H_pp), not a per-iteration path.                                  */
int cs_ba_pose_marginals(cs_ba* ba, int n_pairs, const int* class_i, const int* idx_i, const int* class_j, const int* idx_j, double* out, int* positive_definite);

/* The damped reduced system as the solver is about to factorise it (the Schur-complement build of block_solver.hpp:373-439 at `lambda`
 * on the current linearisation, no factorisation): S_dense n_red x n_red symmetric (n_red from cs_ba_reduced_size), rhs n_red, and the
 * column of every camera / cuboid in solver order (-1: fixed; a cuboid column >= n_red: eliminated, not part of S).  NULL = skip.   */
int cs_ba_get_reduced_system(cs_ba* ba, double lambda, double* S_dense, double* rhs, int* cam_col, int* cub_col);

/* A_ii of every vertex after cs_ba_build_system(): what g2o keeps mapped into its vertices (BaseVertex::mapHessianMemory,
 * core/base_vertex.hpp:52-54, mapped by BlockSolver::buildStructure block_solver.hpp:185,191) and what
 * OptimizationAlgorithmLevenberg::computeLambdaInit() reads through v->hessian(j, j)
 * (optimization_algorithm_levenberg.cpp:166-180).  Caller's vertex order; cam36: n_cams x 36, cub81: n_cuboids x 81,
 * pt9: n_points x 9 (NULL = skip); fixed vertices read as zero.                                                    */
int cs_ba_get_vertex_hessians(cs_ba* ba, double* cam36, double* cub81, double* pt9);

typedef struct cs_ba_timing {
  double errors_ms, linearize_ms, reduce_ms, schur_ms, factor_ms, backsub_ms, update_ms, total_ms;
  long long n_linearizations, n_solves;
  long long linearize_bytes;        /* algorithmic bytes of one linearisation + Schur build */
  long long schur_entries;          /* sum over landmarks of k_j (k_j + 1) / 2                           */
} cs_ba_timing;
int cs_ba_last_timing(cs_ba* ba, cs_ba_timing* t);
/* The stage split above (errors / linearize / reduce / factor / backsub _ms) is g2o's G2OBatchStatistics (core/batch_stats.h:48-62) and, like
 * there (SparseOptimizer::setComputeBatchStatistics, core/sparse_optimizer.cpp:379-397), it is OFF until asked for: every phase mark is an
 * event on the handle's stream (~6 us of dispatch gap each, eight per LM trial), and with the marks on, cs_ba_optimize also gives up queueing the
 * next iteration's linearisation behind a trial before the trial's verdict is known.  total_ms, the counters and the byte figures are always kept. */
int cs_ba_set_stage_timing(cs_ba* ba, int on);
/* OptimizationAlgorithmLevenberg's two properties (core/optimization_algorithm_levenberg.cpp:50-51; setUserLambdaInit :196-199,
 * setMaxTrialsAfterFailure :191-194): user_lambda_init > 0 is the first iteration's lambda instead of tau * max |H_jj| (computeLambdaInit
 * :166-180; <= 0: computed, the default), max_trials_after_failure (>= 1, default 10) bounds the trials of one iteration (:149) -- an iteration
 * that uses them all ends the run as in :151.  Applies to cs_ba_optimize calls that follow. */
int cs_ba_set_lm_params(cs_ba* ba, double user_lambda_init, int max_trials_after_failure);

/* Debug / repro aids (what g2o offers through its debug builds and its text IO).
 * cs_ba_check_finite: scans for NaN / Inf where g2o's debug builds look for them -- the edges' errors (SparseOptimizer::
 * computeActiveErrors, core/sparse_optimizer.cpp:78-86) and Jacobians (BlockSolver::buildSystem, core/block_solver.hpp:533-544) --
 * through what they turn into on the device: every edge's squared error, every block of the linear system (after cs_ba_build_system),
 * the increments (after a solve) and the estimates.  *n_bad = number of non-finite values (0 = clean); report (may be NULL) receives one
 * line per offending array naming the first offending vertex / edge in the caller's indices.  With CS_BA_DEBUG_NAN=1 in the
 * environment the library runs the scan itself after every linearisation, solve and update and prints the report to stderr.
 * cs_ba_dump / cs_ba_load: the complete problem (vertices with their CURRENT estimates and fixed flags, all four edge lists, robust
 * kernels) as one flat binary file, so that a failing field case becomes a fixture -- the stand-in for OptimizableGraph::save / load
 * (core/optimizable_graph.h:594-606), which the reference's own graph cannot use (no type of it is registered with g2o's Factory).
 * Shard settings and external edges are not stored.                                                                               */
int cs_ba_check_finite(cs_ba* ba, int* n_bad, char* report, int report_cap);
int cs_ba_dump(cs_ba* ba, const char* path);
int cs_ba_load(const char* path, int device, cs_ba** out);

#ifdef __cplusplus
}
#endif
#endif /* CUBESLAM_HIP_H */
