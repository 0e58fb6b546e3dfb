// Translation unit of tests/test_adapters_compile.py: the three adapters, instantiated the way the reference's drivers use them
// (detect_3d_cuboid/src/main.cpp:52-73, object_slam/src/main_obj.cpp:502-519,593,633).  g++ -fsyntax-only; see README.md.
#include "detect_3d_cuboid_hip.h"
#include "line_lbd_hip.h"
#include "block_solver_hip.h"

void use_detect(const cv::Mat& rgb, const Eigen::Matrix4d& T, const Eigen::MatrixXd& boxes, const Eigen::MatrixXd& edges) {
  detect_3d_cuboid det;
  Eigen::Matrix3d K;
  det.whether_plot_detail_images = false;
  det.whether_plot_final_images = false;
  det.print_details = false;
  det.set_calibration(K);
  det.whether_sample_bbox_height = false;
  det.nominal_skew_ratio = 2;
  det.whether_save_final_images = true;
  std::vector<ObjectSet> all;
  det.whether_sample_cam_roll_pitch = true;
  det.detect_cuboid(rgb, T, boxes, edges, all);
  double yaw = det.cam_pose_raw.euler_angle(0) + all[0][0]->camera_roll_delta + all[0][0]->pos(2);
  (void)yaw;
}
void use_lines(const cv::Mat& gray) {
  line_lbd_detect line_lbd_obj;
  line_lbd_obj.use_LSD = false;
  line_lbd_obj.line_length_thres = 15;
  cv::Mat all_lines_mat;
  line_lbd_obj.detect_filter_lines(gray, all_lines_mat);
}
g2o::Solver* use_solver() {
  g2o::Solver* solver_ptr = new cubeslam::BlockSolverHIP();      // would not compile if a pure virtual of g2o::Solver were left open
  solver_ptr->setLambda(1.0, true);
  return solver_ptr;
}
