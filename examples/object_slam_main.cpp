// object_slam_main -- the reference's graph driver (object_slam/src/main_obj.cpp:479-841, offline mode :682-722, result
// files :305-336) on top of the C ABI.
//
// Inputs are the reference's own text tables (read with the semantics of read_all_number_txt, matrix_utils.cpp:209-244):
//   truth_cam_poses.txt        time x y z qx qy qz qw        only row 0 is used: the fixed first camera (:523)
//   pop_cam_poses_saved.txt    time x y z qx qy qz qw        camera pose the saved detections are expressed against (:706-708)
//   detect_cuboids_saved.txt   frame x y z yaw sx sy sz err  at most one cuboid per frame, on the frame's ground plane (:692-705)
// One object, perfect association, as in the reference.  Every frame adds a camera vertex (constant-motion initial guess
// from the two previous optimised cameras, :545-564), its EdgeSE3Cuboid (information (2 q)^2 I9 with
// q = (1 - err + 0.5) / 2, :727-780) and an EdgeSE3Expmap to the previous camera (identity information, :786-800), then
// runs optimize(5) over the whole graph so far (:805-806) -- here cs_ba_optimize, i.e. the Levenberg loop with the
// linearisation, Schur reduction and solve on the device.  Outputs follow :305-336: output_cam_poses.txt (final camera
// poses, camera-to-world) and output_obj_poses.txt (the cuboid after every frame as toMinimalVector).
//
// With --online the driver follows the reference's online branch instead (:585-680): every frame's colour image is read
// (binary PPM here; the reference reads the JPEG with cv::imread), converted with cs_bgr_to_gray (cvtColor) and handed to
// cs_detect_cuboids_gray together with the frame's first 2D box (filter_2d_obj_txts/NNNN_yolo2_0.15.txt, 1-based, :619-621)
// and its line segments (one x1 y1 x2 y2 row per segment; the reference gets them from line_lbd, :590-597 -- cs_detect_lines_gray here).  The camera pose
// handed to the detector is the current estimate for frame 0 and the first frame's pose with roll/pitch sampling afterwards
// (:623-629); with sampling the measurement is re-expressed in the sampled camera frame (:660-668).
//
//   g++ -O2 -I include -I cube_slam_wu_amd/csrc examples/object_slam_main.cpp -L cube_slam_wu_amd -lcubeslam_hip ... -o build_tmp/object_slam_main
//   build_tmp/object_slam_main <data_dir> <out_dir> [digits]
//   build_tmp/object_slam_main --online <data_dir> <ppm_dir> <segments_dir | detect> <out_dir> [digits]
//   build_tmp/object_slam_main --g2o <in.g2o> <out.g2o> [iterations = 5] [digits = 17]
// Either run also writes <out_dir>/graph.g2o: the whole graph (final estimates, every edge) in g2o's text format -- examples/g2o_text.h
// (OptimizableGraph::save, core/optimizable_graph.h:594-606, with the classes' own write() field order).  --g2o reads such a file back
// (OptimizableGraph::load), runs cs_ba_optimize over it (0 iterations: no device is touched) and saves the result.
// (segments_dir: one NNNN.txt per frame; the literal `detect`: the segments come from cs_detect_lines_gray, the reference's
// EDLines producer -- the whole online branch then runs image in, trajectory and object out, on the device)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "cs_se3.h"
#include "cubeslam_hip.h"
#include <algorithm>

#include "g2o_text.h"

using cs::Cube;
using cs::Pose;

static bool read_all_number_txt(const std::string& name, int cols, std::vector<double>& out, int& rows) {
  std::ifstream f(name.c_str());
  if (!f) { std::cout << "ERROR!!! Cannot read txt file " << name << std::endl; return false; }
  rows = 0;
  out.clear();
  std::string line;
  while (std::getline(f, line)) {
    if (line.empty()) continue;
    std::stringstream ss(line);
    std::vector<double> row(cols, 0.0);
    double t;
    int c = 0;
    while (ss >> t) { if (c < cols) row[c] = t; c++; }
    out.insert(out.end(), row.begin(), row.end());
    rows++;
  }
  return true;
}

// SE3Quat(Vector7d) (se3quat.h:84-100): x y z qx qy qz qw, rotation normalised with w >= 0
static Pose pose_from_vector7(const double* v) {
  Pose p = cs::pose_load(v);
  cs::pose_normalize(p);
  return p;
}

using g2o_text::cuboid_from_minimal;     // g2o::cuboid::fromMinimalVector / toMinimalVector (g2o_Object.h:37-42, 137-143): examples/g2o_text.h
using g2o_text::cuboid_to_minimal;

// transform_to / transform_from (g2o_Object.h:117-133): the pose moves, the half sizes stay
static Cube cuboid_transform_to(const Cube& c, const Pose& Twc) { Cube r = c; r.pose = cs::pose_mul(cs::pose_inv(Twc), c.pose); return r; }
static Cube cuboid_transform_from(const Cube& c, const Pose& Twc) { Cube r = c; r.pose = cs::pose_mul(Twc, c.pose); return r; }

// binary PPM (P6, 8 bit), returned in OpenCV's BGR order
static bool read_ppm_as_bgr(const std::string& name, std::vector<unsigned char>& bgr, int& w, int& h) {
  std::ifstream f(name.c_str(), std::ios::binary);
  std::string magic;
  int maxv = 0;
  if (!(f >> magic >> w >> h >> maxv) || magic != "P6" || maxv != 255) return false;
  f.get();
  bgr.resize((size_t)w * h * 3);
  f.read(reinterpret_cast<char*>(bgr.data()), (std::streamsize)bgr.size());
  if (!f) return false;
  for (size_t p = 0; p < (size_t)w * h; p++) std::swap(bgr[3 * p], bgr[3 * p + 2]);
  return true;
}

// SE3Quat::to_homogeneous_matrix (se3quat.h:332-340), row-major 4 x 4
static void pose_to_matrix(const Pose& p, double* T) {
  double R[9];
  cs::pose_rotmat(p, R);
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T[4 * r + c] = R[3 * r + c]; T[4 * r + 3] = p.t[r]; }
  T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

// SE3Quat(R, t) (se3quat.h:65-69) with R = euler_zyx_to_rot(roll, pitch, yaw) (matrix_utils.cpp:84-99)
static Pose pose_from_euler_zyx(double roll, double pitch, double yaw, const double* t) {
  const double cp = std::cos(pitch), sp = std::sin(pitch), sr = std::sin(roll), cr = std::cos(roll), sy = std::sin(yaw), cy = std::cos(yaw);
  const double R[9] = {cp * cy, (sr * sp * cy) - (cr * sy), (cr * sp * cy) + (sr * sy),
                       cp * sy, (sr * sp * sy) + (cr * cy), (cr * sp * sy) - (sr * cy),
                       -sp, sr * cp, cr * cp};
  Pose p;
  cs::quat_from_rotmat(R, p);
  for (int d = 0; d < 3; d++) p.t[d] = t[d];
  cs::pose_normalize(p);
  return p;
}

#define CHECK(call) do { int st_ = (call); if (st_ != 0) { std::fprintf(stderr, "%s failed: %d\n", #call, st_); return 1; } } while (0)

static int run_g2o_file(const char* in, const char* out, int iterations, int digits) {
  g2o_text::Graph g;
  std::string warnings;
  if (!g2o_text::load(in, g, &warnings)) { std::fprintf(stderr, "cannot read %s\n", in); return 1; }
  if (!warnings.empty()) std::cerr << warnings;
  std::cout << "loaded " << g.cam_id.size() << " cameras, " << g.cub_id.size() << " cuboids, " << g.ce_cam.size() << " camera-cuboid edges, " << g.oe_i.size() << " odometry edges";
  if (!g.pt_id.empty() || !g.pe_pt.empty()) std::cout << ", " << g.pt_id.size() << " points, " << g.pe_pt.size() << " camera-point edges";
  std::cout << std::endl;
  if (iterations > 0) {
    // (the handle goes with the scope, whichever CHECK leaves it)
    struct BaGuard { cs_ba* p = nullptr; ~BaGuard() { if (p) cs_ba_destroy(p); } } ba;
    CHECK(cs_ba_create(0, &ba.p));
    // Column order of the reduced system = g2o's: pose vertices by id (sparse_optimizer.cpp:174-187).  The C ABI orders a CLASS at a time, so
    // a file whose cuboid ids all lie below (above) its camera ids is ordered exactly as g2o would; interleaved ids keep the cameras first
    // (the LM run is then the same problem in another elimination order: same minimum, trial sequence not guaranteed).
    const bool cuboids_first = !g.cub_id.empty() && !g.cam_id.empty() && *std::max_element(g.cub_id.begin(), g.cub_id.end()) < *std::min_element(g.cam_id.begin(), g.cam_id.end());
    CHECK(cs_ba_set_vertices(ba.p, g.cam_Tcw.data(), g.cam_fixed.data(), (int)g.cam_id.size(), g.cuboids.data(), g.cub_fixed.data(), (int)g.cub_id.size(),
                             g.points.data(), g.pt_fixed.data(), (int)g.pt_id.size(), cuboids_first));
    if (!g.pe_pt.empty()) CHECK(cs_ba_set_edges_proj(ba.p, (int)g.pe_pt.size(), g.pe_pt.data(), g.pe_cam.data(), g.pe_uv.data(), g.pe_info.data(), g.pe_intr.data(), g.pe_huber.data()));
    if (!g.ce_cam.empty()) CHECK(cs_ba_set_edges_cuboid(ba.p, (int)g.ce_cam.size(), g.ce_cam.data(), g.ce_cub.data(), g.ce_meas.data(), g.ce_info.data()));
    if (!g.oe_i.empty()) CHECK(cs_ba_set_edges_odom(ba.p, (int)g.oe_i.size(), g.oe_i.data(), g.oe_j.data(), g.oe_meas.data(), g.oe_info.data()));
    int done = 0;
    CHECK(cs_ba_optimize(ba.p, iterations, &done, nullptr, nullptr, nullptr, 0));
    CHECK(cs_ba_get_state(ba.p, g.cam_Tcw.data(), g.cuboids.data(), g.points.empty() ? nullptr : g.points.data()));
    std::cout << "LM iterations: " << done << std::endl;
  }
  if (!g2o_text::save(out, g, digits)) { std::fprintf(stderr, "cannot write %s\n", out); return 1; }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "--g2o") {
    if (argc < 4) { std::fprintf(stderr, "usage: %s --g2o <in.g2o> <out.g2o> [iterations] [digits]\n", argv[0]); return 2; }
    return run_g2o_file(argv[2], argv[3], argc > 4 ? std::atoi(argv[4]) : 5, argc > 5 ? std::atoi(argv[5]) : 17);
  }
  const bool online_detect_mode = argc > 1 && std::string(argv[1]) == "--online";
  if ((!online_detect_mode && argc < 3) || (online_detect_mode && argc < 6)) {
    std::fprintf(stderr, "usage: %s <data_dir> <out_dir> [digits]\n       %s --online <data_dir> <ppm_dir> <segments_dir | detect> <out_dir> [digits]\n", argv[0], argv[0]);
    return 2;
  }
  const int a0 = online_detect_mode ? 2 : 1;
  const std::string base_folder = std::string(argv[a0]) + "/";
  const std::string ppm_folder = online_detect_mode ? std::string(argv[3]) + "/" : "", seg_folder = online_detect_mode ? std::string(argv[4]) + "/" : "";
  const bool detect_lines = online_detect_mode && std::string(argv[4]) == "detect";
  const std::string out_folder = std::string(argv[online_detect_mode ? 5 : 2]) + "/";
  const int a_digits = online_detect_mode ? 6 : 3;
  const int digits = argc > a_digits ? std::atoi(argv[a_digits]) : 6;      // Eigen's stream precision is 6 significant digits

  std::vector<double> pred_frame_objects, init_frame_poses, truth_frame_poses;
  int n_obs = 0, n_init = 0, total_frame_number = 0;
  if (!online_detect_mode) {
    if (!read_all_number_txt(base_folder + "detect_cuboids_saved.txt", 9, pred_frame_objects, n_obs)) return 1;
    if (!read_all_number_txt(base_folder + "pop_cam_poses_saved.txt", 8, init_frame_poses, n_init)) return 1;
  }
  if (!read_all_number_txt(base_folder + "truth_cam_poses.txt", 8, truth_frame_poses, total_frame_number)) return 1;
  std::cout << "read data size:  " << n_obs << "  " << n_init << "  " << total_frame_number << std::endl;
  if (total_frame_number < 1 || (!online_detect_mode && n_init < total_frame_number)) { std::fprintf(stderr, "pose tables too short\n"); return 1; }

  // the detector of the online branch (main_obj.cpp:491-500): no height sampling, nominal_skew_ratio 2; roll/pitch sampling
  // is switched per frame (:623), here one detector for each setting
  const double calib[9] = {535.4, 0, 320.1, 0, 539.2, 247.6, 0, 0, 1};
  cs_detector* detect_cuboid_obj[2] = {nullptr, nullptr};
  if (online_detect_mode)
    for (int sample = 0; sample < 2; sample++) {
      cs_detect_params prm;
      cs_detect_default_params(&prm);
      prm.whether_sample_bbox_height = 0;
      prm.nominal_skew_ratio = 2;
      prm.whether_sample_cam_roll_pitch = sample;
      CHECK(cs_detector_create(&prm, 0, &detect_cuboid_obj[sample]));
    }

  const Pose fixed_init_cam_pose_Twc = pose_from_vector7(&truth_frame_poses[1]);

  std::vector<double> cam_Tcw;                       // 7 per frame: optimised world-to-camera (the vertex estimates)
  std::vector<int> cam_fixed;
  std::vector<int> ce_cam, ce_cub;                   // EdgeSE3Cuboid list
  std::vector<double> ce_meas, ce_info;
  std::vector<int> oe_i, oe_j;                       // EdgeSE3Expmap list
  std::vector<double> oe_meas, oe_info;
  double cube10[10] = {0, 0, 0, 0, 0, 0, 1, 0, 0, 0};
  std::vector<double> cube_history;                  // 9 per frame
  int offline_cube_obs_row_id = 0, total_iterations = 0;
  cs_ba* ba = nullptr;

  for (int frame_index = 0; frame_index < total_frame_number; frame_index++) {
    Pose curr_cam_pose_Twc, odom_val = {{0, 0, 0}, 0, 0, 0, 1};
    if (frame_index == 0) {
      curr_cam_pose_Twc = fixed_init_cam_pose_Twc;
    } else {
      const Pose prev_pose_Tcw = cs::pose_load(&cam_Tcw[7 * (frame_index - 1)]);
      if (frame_index > 1)                           // constant-motion model from the third frame on
        odom_val = cs::pose_mul(prev_pose_Tcw, cs::pose_inv(cs::pose_load(&cam_Tcw[7 * (frame_index - 2)])));
      curr_cam_pose_Twc = cs::pose_inv(cs::pose_mul(odom_val, prev_pose_Tcw));
    }

    bool has_detected_cuboid = false;
    Cube cube_local_meas = {{{0, 0, 0}, 0, 0, 0, 1}, {0, 0, 0}};
    double proposal_error = 0;
    if (online_detect_mode) {
      char frame_index_c[16];
      std::snprintf(frame_index_c, sizeof(frame_index_c), "%04d", frame_index);
      std::vector<double> raw_2d_objs, all_lines_raw;
      int n_objs = 0, n_lines = 0;
      if (!read_all_number_txt(base_folder + "filter_2d_obj_txts/" + frame_index_c + "_yolo2_0.15.txt", 5, raw_2d_objs, n_objs)) return 1;
      if (n_objs > 0) {
        std::vector<unsigned char> bgr;
        int img_w = 0, img_h = 0;
        if (!read_ppm_as_bgr(ppm_folder + frame_index_c + ".ppm", bgr, img_w, img_h)) { std::fprintf(stderr, "cannot read image of frame %d\n", frame_index); return 1; }
        std::vector<unsigned char> gray((size_t)img_w * img_h);
        CHECK(cs_bgr_to_gray(bgr.data(), img_w * img_h, gray.data()));
        if (detect_lines) {      // line_lbd_obj.detect_filter_lines(raw_rgb_img, all_lines_mat) with line_length_thres = 15 (:502-505,593), float rows copied into doubles (:596-599)
          std::vector<float> seg(4 * 20000);
          CHECK(cs_detect_lines_gray(detect_cuboid_obj[0], gray.data(), img_w, img_h, 15.0, seg.data(), 20000, &n_lines));
          all_lines_raw.assign(seg.begin(), seg.begin() + 4 * (size_t)n_lines);
        } else if (!read_all_number_txt(seg_folder + frame_index_c + ".txt", 4, all_lines_raw, n_lines)) return 1;
        raw_2d_objs[0] -= 1; raw_2d_objs[1] -= 1;               // the box file is 1-based (:621); this data has one landmark, the first box is it
        const int sample = frame_index != 0;                     // roll/pitch sampling from the second frame on (:623)
        double transToWolrd[16];
        pose_to_matrix(sample ? fixed_init_cam_pose_Twc : curr_cam_pose_Twc, transToWolrd);
        cs_frame_desc fr{};
        fr.K = calib; fr.T_wc = transToWolrd; fr.img_w = img_w; fr.img_h = img_h;
        fr.boxes = raw_2d_objs.data(); fr.n_boxes = 1; fr.lines = all_lines_raw.data(); fr.n_lines = n_lines; fr.dist_maps = nullptr;
        cs_cuboid detected_cube;
        int count = 0;
        CHECK(cs_detect_cuboids_gray(detect_cuboid_obj[sample], &fr, gray.data(), &detected_cube, &count));
        has_detected_cuboid = count > 0;
        if (has_detected_cuboid) {
          const double cube_pose[9] = {detected_cube.pos[0], detected_cube.pos[1], detected_cube.pos[2], 0, 0, detected_cube.rotY,
                                       detected_cube.scale[0], detected_cube.scale[1], detected_cube.scale[2]};   // x y z, roll, pitch, yaw, half sizes
          const Cube cube_ground_value = cuboid_from_minimal(cube_pose);
          cube_local_meas = cuboid_transform_to(cube_ground_value, curr_cam_pose_Twc);     // the measurement lives in the camera's frame (:658)
          if (sample) {   // with sampling: the frame of the camera pose the winning proposal was built with (:660-668)
            double new_camera_eulers[3];
            CHECK(cs_cam_euler_zyx(transToWolrd, new_camera_eulers));
            new_camera_eulers[0] += detected_cube.camera_roll_delta; new_camera_eulers[1] += detected_cube.camera_pitch_delta;
            const double trans[3] = {transToWolrd[3], transToWolrd[7], transToWolrd[11]};
            cube_local_meas = cuboid_transform_to(cube_ground_value, pose_from_euler_zyx(new_camera_eulers[0], new_camera_eulers[1], new_camera_eulers[2], trans));
          }
          proposal_error = detected_cube.normalized_error;
        }
      }
    } else if (offline_cube_obs_row_id < n_obs) {
      const double* m = &pred_frame_objects[9 * offline_cube_obs_row_id];
      has_detected_cuboid = (int)m[0] == frame_index;
      if (has_detected_cuboid) {
        const double cube_pose[9] = {m[1], m[2], m[3], 0, 0, m[4], m[5], m[6], m[7]};   // x y z, roll, pitch, yaw, half sizes
        const Pose cam_val_Twc = pose_from_vector7(&init_frame_poses[8 * frame_index + 1]);
        cube_local_meas = cuboid_transform_to(cuboid_from_minimal(cube_pose), cam_val_Twc);
        proposal_error = m[8];
        offline_cube_obs_row_id++;
      }
    }
    if (frame_index == 0) cs::cube_store(cuboid_transform_from(cube_local_meas, curr_cam_pose_Twc), cube10);

    double tcw[7];
    cs::pose_store(cs::pose_inv(curr_cam_pose_Twc), tcw);
    cam_Tcw.insert(cam_Tcw.end(), tcw, tcw + 7);
    cam_fixed.push_back(frame_index == 0);

    if (has_detected_cuboid) {
      const double meas_quality = (1 - proposal_error + 0.5) / 2;
      const double inv_sigma = 1.0 * 2.0 * meas_quality;
      double meas[10], info[81] = {0};
      cs::cube_store(cube_local_meas, meas);
      for (int d = 0; d < 9; d++) info[d * 9 + d] = inv_sigma * inv_sigma;
      ce_cam.push_back(frame_index); ce_cub.push_back(0);
      ce_meas.insert(ce_meas.end(), meas, meas + 10);
      ce_info.insert(ce_info.end(), info, info + 81);
    }
    if (frame_index > 0) {
      double meas[7], info[36] = {0};
      cs::pose_store(odom_val, meas);
      for (int d = 0; d < 6; d++) info[d * 6 + d] = 1.0;
      oe_i.push_back(frame_index - 1); oe_j.push_back(frame_index);
      oe_meas.insert(oe_meas.end(), meas, meas + 7);
      oe_info.insert(oe_info.end(), info, info + 36);
    }

    // graph.initializeOptimization(); graph.optimize(5);  -- cuboid id 0 sorts before the cameras (ids frame + 1)
    // one solver handle for the whole run, like the reference's one graph (main_obj.cpp:510-520): every frame hands it the grown
    // vertex / edge lists, and the structure phase is redone when they change
    const int cub_fixed = 0;
    if (!ba) CHECK(cs_ba_create(0, &ba));
    CHECK(cs_ba_set_vertices(ba, cam_Tcw.data(), cam_fixed.data(), frame_index + 1, cube10, &cub_fixed, 1, nullptr, nullptr, 0, 1));
    if (!ce_cam.empty()) CHECK(cs_ba_set_edges_cuboid(ba, (int)ce_cam.size(), ce_cam.data(), ce_cub.data(), ce_meas.data(), ce_info.data()));
    if (!oe_i.empty()) CHECK(cs_ba_set_edges_odom(ba, (int)oe_i.size(), oe_i.data(), oe_j.data(), oe_meas.data(), oe_info.data()));
    int iterations_done = 0;
    CHECK(cs_ba_optimize(ba, 5, &iterations_done, nullptr, nullptr, nullptr, 0));
    CHECK(cs_ba_get_state(ba, cam_Tcw.data(), cube10, nullptr));
    total_iterations += iterations_done;

    double minimal[9];
    cuboid_to_minimal(cs::cube_load(cube10), minimal);
    cube_history.insert(cube_history.end(), minimal, minimal + 9);
  }
  for (int sample = 0; sample < 2; sample++) if (detect_cuboid_obj[sample]) cs_detector_destroy(detect_cuboid_obj[sample]);
  if (ba) cs_ba_destroy(ba);
  std::cout << "+++++++++++++Finish all optimization!+++++++++++++  LM iterations: " << total_iterations << std::endl;

  {   // the graph itself, g2o text (vertex ids as the reference assigns them: the object 0, camera of frame f: f + 1, main_obj.cpp:529-575)
    g2o_text::Graph g;
    g.cub_id.push_back(0); g.cub_fixed.push_back(0); g.cuboids.assign(cube10, cube10 + 10);
    for (int i = 0; i < total_frame_number; i++) g.cam_id.push_back(i + 1);
    g.cam_Tcw = cam_Tcw; g.cam_fixed = cam_fixed;
    g.ce_cam = ce_cam; g.ce_cub = ce_cub; g.ce_meas = ce_meas; g.ce_info = ce_info;
    g.oe_i = oe_i; g.oe_j = oe_j; g.oe_meas = oe_meas; g.oe_info = oe_info;
    if (!g2o_text::save(out_folder + "graph.g2o", g, 17)) { std::fprintf(stderr, "cannot write %sgraph.g2o\n", out_folder.c_str()); return 1; }
  }
  {
    const std::string path = out_folder + "output_cam_poses.txt";
    FILE* f = std::fopen(path.c_str(), "w");
    if (!f) { std::fprintf(stderr, "cannot write %s\n", path.c_str()); return 1; }
    std::fprintf(f, "# timestamp tx ty tz qx qy qz qw\n");
    for (int i = 0; i < total_frame_number; i++) {
      double twc[7];
      cs::pose_store(cs::pose_inv(cs::pose_load(&cam_Tcw[7 * i])), twc);
      std::fprintf(f, "%.9f  ", truth_frame_poses[8 * i]);
      for (int d = 0; d < 7; d++) std::fprintf(f, "%s%.*g", d ? " " : "", digits, twc[d]);
      std::fprintf(f, "\n");
    }
    std::fclose(f);
  }
  {
    const std::string path = out_folder + "output_obj_poses.txt";
    FILE* f = std::fopen(path.c_str(), "w");
    if (!f) { std::fprintf(stderr, "cannot write %s\n", path.c_str()); return 1; }
    for (int j = 0; j < total_frame_number; j++) {
      for (int d = 0; d < 9; d++) std::fprintf(f, "%s%.*g", d ? " " : "", digits, cube_history[9 * j + d]);
      std::fprintf(f, " \n");
    }
    std::fclose(f);
  }
  return 0;
}
