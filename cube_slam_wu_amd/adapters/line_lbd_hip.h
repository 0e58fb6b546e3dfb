// line_lbd_hip.h -- header-only drop-in for the part of the reference's `class line_lbd_detect` the hot path's caller uses
// (line_lbd/include/line_lbd/line_lbd_allclass.h:23-79): construction, the public flags use_LSD / line_length_thres, and
//     void detect_filter_lines(const cv::Mat& gray_img, cv::Mat& linesmat_out);          // n x 4 CV_32F: x1 y1 x2 y2
// as called by object_slam/src/main_obj.cpp:502-505,593.  Include this header instead of line_lbd/line_lbd_allclass.h and link
// -lcubeslam_hip in place of the reference's libline_lbd_lib.  Both detector branches are served: EDLines (use_LSD = false, the one
// the graph driver selects) by cs_detect_lines_gray, LSD (use_LSD = true, the one whose output the reference ships for its bundled
// frame) by cs_detect_lsd_gray; descriptors and matching are not on the hot path and stay with the reference library.  Needs OpenCV
// for cv::Mat / cvtColor; tests/test_adapters_compile.py compiles it against declaration-only stubs.
#pragma once

#include <opencv2/core/core.hpp>
#include <opencv2/imgproc/imgproc.hpp>

#include <iostream>
#include <string>
#include <vector>

#include "cubeslam_hip.h"

class line_lbd_detect {
 public:
  explicit line_lbd_detect(int numoctaves = 1, float octaveratio = 1) : numoctaves_(numoctaves), octaveratio_(octaveratio), use_LSD(false), line_length_thres(50) {}
  ~line_lbd_detect() { if (det_) cs_detector_destroy(det_); }
  line_lbd_detect(const line_lbd_detect&) = delete;
  line_lbd_detect& operator=(const line_lbd_detect&) = delete;

  int numoctaves_;
  float octaveratio_;
  bool use_LSD;              // line_lbd_allclass.cpp:130-150: LSD when set, EDLines otherwise
  float line_length_thres;   // line_lbd_allclass.cpp:147: 50 by default, 15 in the graph driver

  void detect_filter_lines(const cv::Mat& img, cv::Mat& linesmat_out) {
    // (the reference's detect_filter_lines is void and has no error channel; a failure here prints and hands back no segments)
    linesmat_out.create(0, 4, CV_32FC1);
    if (numoctaves_ != 1) { std::cerr << "line_lbd_detect (HIP): one octave only, as the graph driver uses it (main_obj.cpp:502)" << std::endl; return; }
    if (!det_ && cs_detector_create(nullptr, 0, &det_) != CS_OK) { std::cerr << "line_lbd_detect (HIP): cs_detector_create: " << cs_last_error() << std::endl; det_ = nullptr; return; }
    cv::Mat gray;
    if (img.channels() != 1) cv::cvtColor(img, gray, cv::COLOR_BGR2GRAY);     // BinaryDescriptor::detectImpl, binary_descriptor.cpp:489-495
    else gray = img;
    if (!gray.isContinuous()) gray = gray.clone();
    std::vector<float> seg(4 * (size_t)kCap);
    int n = 0;
    const int rc = use_LSD ? cs_detect_lsd_gray(det_, gray.data, gray.cols, gray.rows, (double)line_length_thres, seg.data(), kCap, &n)
                           : cs_detect_lines_gray(det_, gray.data, gray.cols, gray.rows, (double)line_length_thres, seg.data(), kCap, &n);
    if (rc != CS_OK) { std::cerr << "line_lbd_detect (HIP): " << (use_LSD ? "cs_detect_lsd_gray: " : "cs_detect_lines_gray: ") << cs_last_error() << std::endl; return; }
    linesmat_out.create(n, 4, CV_32FC1);                                        // keylines_to_mat, line_lbd_allclass.cpp:33-43
    for (int j = 0; j < n; j++) for (int c = 0; c < 4; c++) linesmat_out.at<float>(j, c) = seg[4 * (size_t)j + c];
  }

 private:
  enum { kCap = 20000 };
  cs_detector* det_ = nullptr;
};
