// detect_main -- the reference's stand-alone driver (detect_3d_cuboid/src/main.cpp:29-76) on top of the C ABI.
//
// Same hard-coded inputs as the reference: calibration (:37-40), camera-to-world pose (:42-46), one 2D box given in
// 1-based MATLAB coordinates and shifted to 0-based (:48-50), sampling switches off (:67-68); the segments come from
// data/edge_detection/LSD/0000_edge.txt through a restatement of read_all_number_txt (matrix_utils.cpp:209-244:
// one row per non-empty line, whitespace-separated numbers, columns beyond the given width ignored, missing ones zero).
// The image is read as a binary PGM of the gray image (the reference reads a JPEG with cv::imread and converts with
// cvtColor; neither a JPEG decoder nor OpenCV is part of this repository -- tests/golden holds the converted image), and
// the Canny + distance-transform front end runs on the device (cs_detect_cuboids_gray).
//
//   g++ -O2 -I include examples/detect_main.cpp -L cube_slam_wu_amd -lcubeslam_hip -Wl,-rpath,$PWD/cube_slam_wu_amd -o build_tmp/detect_main
//   build_tmp/detect_main <gray.pgm> <edge.txt>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "cubeslam_hip.h"

// read_all_number_txt (matrix_utils.cpp:209-244) for a cols-wide matrix of doubles
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

static bool read_pgm(const std::string& name, std::vector<unsigned char>& px, int& w, int& h) {
  std::ifstream f(name.c_str(), std::ios::binary);
  std::string magic;
  int maxv = 0;
  if (!(f >> magic >> w >> h >> maxv) || magic != "P5" || maxv != 255) return false;
  f.get();
  px.resize((size_t)w * h);
  f.read(reinterpret_cast<char*>(px.data()), (std::streamsize)px.size());
  return (bool)f;
}

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s <gray.pgm> <edge.txt>\n", argv[0]); return 2; }
  const double Kalib[9] = {529.5000, 0, 365.0000, 0, 529.5000, 265.0000, 0, 0, 1.0000};
  const double transToWolrd[16] = {1, 0.0011, 0.0004, 0, 0, -0.3376, 0.9413, 0, 0.0011, -0.9413, -0.3376, 1.35, 0, 0, 0, 1};
  double obj_bbox_coors[5] = {188, 189, 201, 311, 0.8800};   // [x y w h prob]
  obj_bbox_coors[0] -= 1; obj_bbox_coors[1] -= 1;            // change matlab coordinate to c++, minus 1

  std::vector<unsigned char> gray;
  int img_w = 0, img_h = 0;
  if (!read_pgm(argv[1], gray, img_w, img_h)) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  std::vector<double> all_lines_raw;
  int n_lines = 0;
  if (!read_all_number_txt(argv[2], 4, all_lines_raw, n_lines)) return 1;

  cs_detect_params prm;
  cs_detect_default_params(&prm);
  prm.whether_sample_bbox_height = 0;
  prm.whether_sample_cam_roll_pitch = 0;
  cs_detector* det = nullptr;
  if (cs_detector_create(&prm, 0, &det) != CS_OK) { std::fprintf(stderr, "cs_detector_create: %s\n", cs_last_error()); return 1; }
  cs_frame_desc fr{};
  fr.K = Kalib; fr.T_wc = transToWolrd; fr.img_w = img_w; fr.img_h = img_h;
  fr.boxes = obj_bbox_coors; fr.n_boxes = 1; fr.lines = all_lines_raw.data(); fr.n_lines = n_lines; fr.dist_maps = nullptr;
  std::vector<cs_cuboid> out((size_t)prm.max_cuboid_num);
  int count = 0;
  if (cs_detect_cuboids_gray(det, &fr, gray.data(), out.data(), &count) != CS_OK) { std::fprintf(stderr, "cs_detect_cuboids_gray: %s\n", cs_last_error()); return 1; }
  std::printf("segments %d  cuboids %d\n", n_lines, count);
  for (int k = 0; k < count; k++) {
    const cs_cuboid& c = out[k];
    std::printf("pos %.17g %.17g %.17g\nscale %.17g %.17g %.17g\nrotY %.17g\nconfig %d %d\nnormalized_error %.17g\nskew_ratio %.17g\n", c.pos[0], c.pos[1], c.pos[2], c.scale[0],
                c.scale[1], c.scale[2], c.rotY, (int)c.box_config_type[0], (int)c.box_config_type[1], c.normalized_error, c.skew_ratio);
    std::printf("corners2d");
    for (int q = 0; q < 16; q++) std::printf(" %d", c.box_corners_2d[q]);
    std::printf("\n");
  }
  cs_detector_destroy(det);
  return 0;
}
