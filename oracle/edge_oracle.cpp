// edge_oracle.cpp -- CPU restatement of the distance-map front end of detect_cuboid() (TEST INFRASTRUCTURE ONLY:
// nothing under cube_slam_wu_amd/ includes, links or calls this file).
//
// Reference call site: detect_3d_cuboid/src/box_proposal_detail.cpp:84 (cvtColor BGR2GRAY), :320-327
//   cv::Canny(gray_img(object_bbox), im_canny, 80, 200);  cv::distanceTransform(255 - im_canny, dist_map, CV_DIST_L2, 3);
// The arithmetic lives in OpenCV, a third-party dependency that is NOT under /root/reference and is not installed in
// this image (find_package(OpenCV REQUIRED), version unpinned; legacy opencv/cv.h headers => 2.4 / 3.x).  What follows
// restates the published algorithms of modules/imgproc/src/{color,canny,distransform}.cpp of those versions
// (non-SIMD paths):
//   * BGR2GRAY: (B*1868 + G*9617 + R*4899 + 8192) >> 14                                   (color.cpp, yuv_shift = 14)
//   * Canny(low, high, aperture 3, L2gradient = false): 3x3 Sobel (CV_16S, BORDER_REPLICATE; a ROI view is NOT
//     isolated: the filter reads the parent image's pixels around the ROI), L1 magnitude, non-maximum suppression
//     in 4 sectors with the fixed-point tangent test TG22 = 13573 = round(tan(22.5 deg) * 2^15), thresholds
//     floor(low) / floor(high) with strict '>' tests, hysteresis = 8-connected components of the surviving
//     pixels that contain a pixel above the high threshold
//   * distanceTransform(src, DIST_L2, 3): two-pass 3x3 chamfer in 16.16 fixed point, a = round(0.955f * 65536)
//     = 62587, b = round(1.3693f * 65536) = 89738, zero pixels of src are the features, border value
//     UINT_MAX - b, output float(t * 2^-16)
// PARITY UNPINNED for this row: the reference has no fixture for im_canny / dist_map and OpenCV cannot run here, so
// the restatement is checked only against hand-computed cases (tests/test_edge_oracle.py).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {

void oracle_bgr_to_gray(const uint8_t* bgr, int n_pixels, uint8_t* gray) {
  for (int i = 0; i < n_pixels; i++) gray[i] = (uint8_t)((bgr[3 * i] * 1868 + bgr[3 * i + 1] * 9617 + bgr[3 * i + 2] * 4899 + (1 << 13)) >> 14);
}

// Canny of the ROI (l, t, w, h) of a gray image: out[h*w] = 0 / 255.
void oracle_canny_roi(const uint8_t* gray, int img_w, int img_h, int l, int t, int w, int h, int low, int high, uint8_t* out) {
  if (low > high) std::swap(low, high);
  auto px = [&](int x, int y) -> int {  // BORDER_REPLICATE at the image border; inside the image the ROI is not isolated
    x = std::min(std::max(x, 0), img_w - 1); y = std::min(std::max(y, 0), img_h - 1);
    return gray[(size_t)y * img_w + x];
  };
  std::vector<int> dx((size_t)w * h), dy((size_t)w * h), mag((size_t)w * h);
  for (int i = 0; i < h; i++)
    for (int j = 0; j < w; j++) {
      const int x = l + j, y = t + i;
      const int gx = (px(x + 1, y - 1) + 2 * px(x + 1, y) + px(x + 1, y + 1)) - (px(x - 1, y - 1) + 2 * px(x - 1, y) + px(x - 1, y + 1));
      const int gy = (px(x - 1, y + 1) + 2 * px(x, y + 1) + px(x + 1, y + 1)) - (px(x - 1, y - 1) + 2 * px(x, y - 1) + px(x + 1, y - 1));
      dx[(size_t)i * w + j] = gx; dy[(size_t)i * w + j] = gy; mag[(size_t)i * w + j] = std::abs(gx) + std::abs(gy);
    }
  auto M = [&](int i, int j) -> int { return (i < 0 || i >= h || j < 0 || j >= w) ? 0 : mag[(size_t)i * w + j]; };
  // map: 0 = might belong to an edge, 1 = not an edge, 2 = edge
  std::vector<uint8_t> map((size_t)w * h, 1);
  std::vector<int> stack;
  const int TG22 = 13573;
  for (int i = 0; i < h; i++)
    for (int j = 0; j < w; j++) {
      const int m = M(i, j);
      if (m <= low) continue;
      const int xs = dx[(size_t)i * w + j], ys = dy[(size_t)i * w + j];
      const int x = std::abs(xs);
      const long long y = (long long)std::abs(ys) << 15;       // int in OpenCV; |ys| <= 1020 so it fits either way
      const long long tg22x = (long long)x * TG22;
      bool keep;
      if (y < tg22x) keep = m > M(i, j - 1) && m >= M(i, j + 1);
      else {
        const long long tg67x = tg22x + ((long long)x << 16);
        if (y > tg67x) keep = m > M(i - 1, j) && m >= M(i + 1, j);
        else { const int s = (xs ^ ys) < 0 ? -1 : 1; keep = m > M(i - 1, j - s) && m > M(i + 1, j + s); }
      }
      if (!keep) continue;
      if (m > high) { map[(size_t)i * w + j] = 2; stack.push_back(i * w + j); }
      else map[(size_t)i * w + j] = 0;
    }
  while (!stack.empty()) {
    const int p = stack.back(); stack.pop_back();
    const int i = p / w, j = p % w;
    for (int di = -1; di <= 1; di++)
      for (int dj = -1; dj <= 1; dj++) {
        const int a = i + di, b = j + dj;
        if ((di || dj) && a >= 0 && a < h && b >= 0 && b < w && map[(size_t)a * w + b] == 0) { map[(size_t)a * w + b] = 2; stack.push_back(a * w + b); }
      }
  }
  for (size_t p = 0; p < (size_t)w * h; p++) out[p] = (uint8_t)(map[p] == 2 ? 255 : 0);
}

// distanceTransform(src, dist, DIST_L2, 3) with src = 255 - edges: the edge pixels (edges255 != 0) are the features.
void oracle_dist_l2_3x3(const uint8_t* edges255, int w, int h, float* out) {
  const unsigned HV = 62587u, DIAG = 89738u, DMAX = 0xffffffffu - DIAG;
  const float scale = 1.f / (1 << 16);
  const int step = w + 2;
  std::vector<unsigned> tmp((size_t)step * (h + 2), DMAX);
  for (int i = 0; i < h; i++) {
    unsigned* r = &tmp[(size_t)(i + 1) * step + 1];
    for (int j = 0; j < w; j++) {
      if (edges255[(size_t)i * w + j]) { r[j] = 0; continue; }
      unsigned t0 = r[j - step - 1] + DIAG, t = r[j - step] + HV;
      if (t0 > t) t0 = t;
      t = r[j - step + 1] + DIAG; if (t0 > t) t0 = t;
      t = r[j - 1] + HV; if (t0 > t) t0 = t;
      r[j] = (t0 > DMAX) ? DMAX : t0;   // saturate: keeps every later sum below 2^32
    }
  }
  for (int i = h - 1; i >= 0; i--) {
    unsigned* r = &tmp[(size_t)(i + 1) * step + 1];
    for (int j = w - 1; j >= 0; j--) {
      unsigned t0 = r[j];
      if (t0 > HV) {
        unsigned t = r[j + step + 1] + DIAG; if (t0 > t) t0 = t;
        t = r[j + step] + HV; if (t0 > t) t0 = t;
        t = r[j + step - 1] + DIAG; if (t0 > t) t0 = t;
        t = r[j + 1] + HV; if (t0 > t) t0 = t;
        r[j] = t0;
      }
      t0 = (t0 > DMAX) ? DMAX : t0;
      out[(size_t)i * w + j] = (float)t0 * scale;
    }
  }
}

// the map detect_cuboid() hands to the scorer for one (box, height sample)
void oracle_edge_distance_map(const uint8_t* gray, int img_w, int img_h, int l, int t, int w, int h, int low, int high, float* out) {
  std::vector<uint8_t> e((size_t)w * h);
  oracle_canny_roi(gray, img_w, img_h, l, t, w, h, low, high, e.data());
  oracle_dist_l2_3x3(e.data(), w, h, out);
}

}  // extern "C"
