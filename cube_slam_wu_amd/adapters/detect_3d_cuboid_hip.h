// detect_3d_cuboid_hip.h -- header-only drop-in for the reference's `class detect_3d_cuboid`
// (detect_3d_cuboid/include/detect_3d_cuboid/detect_3d_cuboid.h:74-118) on top of libcubeslam_hip.so.
//
// Same class name, same public fields, same set_calibration / set_cam_pose / detect_cuboid signatures, so
// detect_3d_cuboid/src/main.cpp and object_slam/src/main_obj.cpp compile against it unchanged; include this
// header *instead of* detect_3d_cuboid/detect_3d_cuboid.h and link -lcubeslam_hip in place of the reference's
// libdetect_3d_cuboid.  It needs Eigen and OpenCV (for cv::Canny / cv::distanceTransform, whose exact output is
// an input of the scorer) and therefore is not compiled in the build container, which has neither; it is kept
// thin on purpose: all arithmetic of the path lives behind the C ABI.
#pragma once

#include <Eigen/Core>
#include <Eigen/Dense>
#include <opencv2/core/core.hpp>
#include <opencv2/imgproc/imgproc.hpp>

#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "cubeslam_hip.h"

// class cuboid / ObjectSet exactly as in the reference header (:20-42)
class cuboid {
 public:
  Eigen::Vector3d pos;
  Eigen::Vector3d scale;
  double rotY;
  Eigen::Vector2d box_config_type;
  Eigen::Matrix2Xi box_corners_2d;
  Eigen::Matrix3Xd box_corners_3d_world;
  Eigen::Vector4d rect_detect_2d;
  double edge_distance_error;
  double edge_angle_error;
  double normalized_error;
  double skew_ratio;
  double down_expand_height;
  double camera_roll_delta;
  double camera_pitch_delta;
};
typedef std::vector<cuboid*> ObjectSet;

struct cam_pose_infos {  // :59-71 (only what callers read back: main_obj.cpp:647-678 reads euler_angle)
  Eigen::Matrix4d transToWolrd;
  Eigen::Matrix3d Kalib;
  Eigen::Vector3d euler_angle;
  double camera_yaw;
};

class detect_3d_cuboid {
 public:
  cam_pose_infos cam_pose;
  cam_pose_infos cam_pose_raw;

  detect_3d_cuboid() {}
  ~detect_3d_cuboid() { for (auto* d : dets_) if (d) cs_detector_destroy(d); }

  void set_calibration(const Eigen::Matrix3d& Kalib) { cam_pose.Kalib = Kalib; }
  void set_cam_pose(const Eigen::Matrix4d& transToWolrd) {
    cam_pose.transToWolrd = transToWolrd;
    double T[16], e[3];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) T[4 * i + j] = transToWolrd(i, j);
    cs_cam_euler_zyx(T, e);
    cam_pose.euler_angle = Eigen::Vector3d(e[0], e[1], e[2]);
    cam_pose.camera_yaw = e[2];
  }

  void detect_cuboid(const cv::Mat& rgb_img, const Eigen::Matrix4d& transToWolrd, const Eigen::MatrixXd& obj_bbox_coors,
                     Eigen::MatrixXd edges, std::vector<ObjectSet>& all_object_cuboids) {
    const int n_boxes_in = (int)obj_bbox_coors.rows();
    // (the reference's detect_cuboid is void and has no error channel: it prints and leaves empty object sets.  Same here: no throw.)
    if (!ensure_detector()) { all_object_cuboids.assign((size_t)n_boxes_in, ObjectSet()); return; }
    set_cam_pose(transToWolrd);
    cam_pose_raw = cam_pose;  // :79
    cv::Mat gray;
    if (rgb_img.channels() == 3) cv::cvtColor(rgb_img, gray, cv::COLOR_BGR2GRAY); else gray = rgb_img;  // :82-86
    const int n = (int)obj_bbox_coors.rows(), m = (int)edges.rows();
    all_object_cuboids.resize(n);
    // Eigen is column-major, the C ABI row-major
    std::vector<double> K(9), T(16), boxes(5 * (size_t)n), lines(4 * (size_t)m);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) K[3 * i + j] = cam_pose.Kalib(i, j);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) T[4 * i + j] = transToWolrd(i, j);
    for (int i = 0; i < n; i++) for (int j = 0; j < 5; j++) boxes[5 * i + j] = obj_bbox_coors(i, j);
    for (int i = 0; i < m; i++) for (int j = 0; j < 4; j++) lines[4 * i + j] = edges(i, j);
    // per (box, height sample): Canny + distance transform on the ROI, exactly as :320-327
    std::vector<cv::Mat> maps(3 * (size_t)n);
    std::vector<const float*> map_ptr(3 * (size_t)n, nullptr);
    for (int i = 0; i < n; i++) {
      cs_roi roi[3];
      int nh = cs_box_rois(&boxes[5 * i], rgb_img.cols, rgb_img.rows, whether_sample_bbox_height, roi);
      for (int k = 0; k < nh; k++) {
        cv::Mat canny, dist;
        cv::Canny(gray(cv::Rect(roi[k].left, roi[k].top, roi[k].width, roi[k].height)), canny, 80, 200);
        cv::distanceTransform(255 - canny, dist, cv::DIST_L2, 3);
        maps[3 * i + k] = dist.isContinuous() ? dist : dist.clone();
        map_ptr[3 * i + k] = maps[3 * i + k].ptr<float>();
      }
    }
    cs_frame_desc fr;
    fr.K = K.data(); fr.T_wc = T.data(); fr.img_w = rgb_img.cols; fr.img_h = rgb_img.rows;
    fr.boxes = boxes.data(); fr.n_boxes = n; fr.lines = lines.data(); fr.n_lines = m; fr.dist_maps = map_ptr.data();
    std::vector<cs_cuboid> out((size_t)n * max_cuboid_num);
    std::vector<int> counts(n);
    int rc = cs_detect_cuboids(det_, &fr, out.data(), counts.data());
    if (rc != CS_OK) {
      std::cerr << "detect_3d_cuboid (HIP): cs_detect_cuboids: " << cs_last_error() << std::endl;
      for (int i = 0; i < n; i++) all_object_cuboids[i].clear();
      return;
    }
    for (int i = 0; i < n; i++) {
      all_object_cuboids[i].clear();
      for (int k = 0; k < counts[i]; k++) {
        const cs_cuboid& c = out[(size_t)i * max_cuboid_num + k];
        cuboid* o = new cuboid();  // caller owns, as in the reference (:740)
        o->pos = Eigen::Vector3d(c.pos[0], c.pos[1], c.pos[2]);
        o->scale = Eigen::Vector3d(c.scale[0], c.scale[1], c.scale[2]);
        o->rotY = c.rotY;
        o->box_config_type = Eigen::Vector2d(c.box_config_type[0], c.box_config_type[1]);
        o->box_corners_2d.resize(2, 8);
        o->box_corners_3d_world.resize(3, 8);
        for (int j = 0; j < 8; j++) {
          o->box_corners_2d(0, j) = c.box_corners_2d[j]; o->box_corners_2d(1, j) = c.box_corners_2d[8 + j];
          for (int r = 0; r < 3; r++) o->box_corners_3d_world(r, j) = c.box_corners_3d_world[8 * r + j];
        }
        o->rect_detect_2d = Eigen::Vector4d(c.rect_detect_2d[0], c.rect_detect_2d[1], c.rect_detect_2d[2], c.rect_detect_2d[3]);
        o->edge_distance_error = c.edge_distance_error; o->edge_angle_error = c.edge_angle_error;
        o->normalized_error = c.normalized_error; o->skew_ratio = c.skew_ratio; o->down_expand_height = c.down_expand_height;
        o->camera_roll_delta = c.camera_roll_delta; o->camera_pitch_delta = c.camera_pitch_delta;
        all_object_cuboids[i].push_back(o);
      }
    }
  }

  // public flags, same names and defaults as the reference (:95-117); plotting flags are accepted and ignored
  bool whether_plot_detail_images = true;    // (the reference's default, :95; nothing is plotted here whatever it says)
  bool whether_plot_final_images = false;
  bool whether_save_final_images = false;
  cv::Mat cuboids_2d_img;
  bool print_details = false;
  bool consider_config_1 = true;
  bool consider_config_2 = true;
  bool whether_sample_cam_roll_pitch = true;
  bool whether_sample_bbox_height = false;
  int max_cuboid_num = 1;
  double nominal_skew_ratio = 1;
  double max_cut_skew = 3;

 private:
  // One detector per sampling mode: main_obj.cpp:623 toggles whether_sample_cam_roll_pitch between frame 0 and the rest, and a
  // detector owns streams, events and a worker pool -- it must not be rebuilt per call.  The other flags are compared field by
  // field (a memcmp would read the struct's padding).
  cs_detector* dets_[2] = {nullptr, nullptr};
  cs_detect_params last_[2] = {};
  cs_detector* det_ = nullptr;   // the one the current call uses
  static bool same_params(const cs_detect_params& a, const cs_detect_params& b) {
    return a.consider_config_1 == b.consider_config_1 && a.consider_config_2 == b.consider_config_2 &&
           a.whether_sample_cam_roll_pitch == b.whether_sample_cam_roll_pitch && a.whether_sample_bbox_height == b.whether_sample_bbox_height &&
           a.max_cuboid_num == b.max_cuboid_num && a.nominal_skew_ratio == b.nominal_skew_ratio && a.max_cut_skew == b.max_cut_skew &&
           a.yaw_range_deg == b.yaw_range_deg && a.yaw_step_deg == b.yaw_step_deg && a.vp12_edge_angle_thre == b.vp12_edge_angle_thre &&
           a.vp3_edge_angle_thre == b.vp3_edge_angle_thre && a.shorted_edge_thre == b.shorted_edge_thre && a.weight_vp_angle == b.weight_vp_angle &&
           a.weight_skew_error == b.weight_skew_error && a.pre_merge_dist_thre == b.pre_merge_dist_thre && a.pre_merge_angle_thre == b.pre_merge_angle_thre &&
           a.edge_length_threshold == b.edge_length_threshold && a.host_threads == b.host_threads;
  }
  bool ensure_detector() {
    cs_detect_params p{};
    cs_detect_default_params(&p);
    p.consider_config_1 = consider_config_1; p.consider_config_2 = consider_config_2;
    p.whether_sample_cam_roll_pitch = whether_sample_cam_roll_pitch; p.whether_sample_bbox_height = whether_sample_bbox_height;
    p.max_cuboid_num = max_cuboid_num; p.nominal_skew_ratio = nominal_skew_ratio; p.max_cut_skew = max_cut_skew;
    const int slot = whether_sample_cam_roll_pitch ? 1 : 0;
    if (dets_[slot] && same_params(p, last_[slot])) { det_ = dets_[slot]; return true; }
    if (dets_[slot]) cs_detector_destroy(dets_[slot]);
    dets_[slot] = nullptr;
    if (cs_detector_create(&p, 0, &dets_[slot]) != CS_OK) {
      std::cerr << "detect_3d_cuboid (HIP): cs_detector_create: " << cs_last_error() << std::endl;
      dets_[slot] = nullptr; det_ = nullptr;
      return false;
    }
    last_[slot] = p;
    det_ = dets_[slot];
    return true;
  }
};
