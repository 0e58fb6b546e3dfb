// lsd_oracle.cpp -- CPU restatement (TEST INFRASTRUCTURE ONLY: tests/, smoke() and bench.py's cpu_baseline leg may use it, the
// product never does) of the LSD branch of the reference's segment producer:
//   line_lbd_detect::detect_filter_lines with use_LSD = true          line_lbd/class/line_lbd_allclass.cpp:130-150,200-215
//   LSDDetector::detectImpl (one octave)                              line_lbd/libs/LSDDetector.cpp:55-75,154-260
//   LineSegmentDetectorImpl::detect / flsd (LSD_REFINE_ADV, defaults) line_lbd/libs/lsd.cpp:185-187,402-1148
// The detector is OpenCV 3's LSD as the reference vendors it, quirks included and kept: the seed list is walked in raster order (the
// gradient-sorted linked list built at :612-633 is never followed, :478-479 index the array), rect_nfa's slopes are INTEGER
// divisions and its second slopes mix tailp->p.x into a y difference (:1065-1073), region angles are accumulated in float.
// Three OpenCV library calls sit in front of it and are restated from their published algorithms (parity of those three is pinned
// only through the end result -- the reference's bundled detect_3d_cuboid/data/edge_detection/LSD/0000_edge.txt, see
// tests/test_lsd_oracle.py):
//   GaussianBlur(CV_64F, 7 x 7, sigma 0.75, BORDER_REFLECT_101): getGaussianKernel in double, generic row filter (taps left to right),
//                symmetric column filter (centre tap, then the pairs outward);
//   resize(scale 0.8, INTER_LINEAR) on CV_64F: pixel-centre mapping, coefficients rounded to float, horizontal pass then vertical;
//   fastAtan2(float y, float x): the degree-valued polynomial (0.9997878412794807, -0.3258083974640975, 0.1555786518463281,
//                -0.04432655554792128) x 180/pi on min/max with (float)DBL_EPSILON in the denominator, quadrant fix-ups.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

const double kPi = 3.1415926535897932384626433832795;   // CV_PI
const double NOTDEF = -1024.0;
const double M_3_2_PI = 4.71238898038, M_2__PI = 6.28318530718;   // lsd.cpp:53-58
const double DEG_TO_RADS = kPi / 180;
const double RELATIVE_ERROR_FACTOR = 100.0;

float fast_atan2(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / kPi), p3 = -0.3258083974640975f * (float)(180 / kPi),
                     p5 = 0.1555786518463281f * (float)(180 / kPi), p7 = -0.04432655554792128f * (float)(180 / kPi);
  const float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
  return p;
}

inline double dist_sq(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }
inline double dist(double x1, double y1, double x2, double y2) { return std::sqrt(dist_sq(x1, y1, x2, y2)); }
inline double angle_diff_signed(double a, double b) {
  double d = a - b;
  while (d <= -kPi) d += M_2__PI;
  while (d > kPi) d -= M_2__PI;
  return d;
}
inline double angle_diff(double a, double b) { return std::fabs(angle_diff_signed(a, b)); }
inline bool double_equal(double a, double b) {
  if (a == b) return true;
  const double ad = std::fabs(a - b), aa = std::fabs(a), bb = std::fabs(b);
  double mx = aa > bb ? aa : bb;
  if (mx < DBL_MIN) mx = DBL_MIN;
  return (ad / mx) <= (RELATIVE_ERROR_FACTOR * DBL_EPSILON);
}
inline double log_gamma_windschitl(double x) { return 0.918938533204673 + (x - 0.5) * std::log(x) - x + 0.5 * x * std::log(x * std::sinh(1 / x) + 1 / (810.0 * std::pow(x, 6.0))); }
inline double log_gamma_lanczos(double x) {
  static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
  double a = (x + 0.5) * std::log(x + 5.5) - (x + 5.5), b = 0;
  for (int n = 0; n < 7; ++n) { a -= std::log(x + double(n)); b += q[n] * std::pow(x, double(n)); }
  return a + std::log(b);
}
inline double log_gamma(double x) { return x > 15.0 ? log_gamma_windschitl(x) : log_gamma_lanczos(x); }

struct RegionPoint { int x, y; unsigned char* used; double angle, modgrad; };
struct Rect { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };
struct Edge { int px, py; bool taken; };

struct Lsd {
  // defaults of createLineSegmentDetector(LSD_REFINE_ADV) (lsd.cpp:185-187)
  const double SCALE = 0.8, SIGMA_SCALE = 0.6, QUANT = 2.0, ANG_TH = 22.5, LOG_EPS = 0, DENSITY_TH = 0.7;
  const int N_BINS = 1024;
  int W = 0, H = 0;          // of the scaled image
  double LOG_NT = 0;
  std::vector<double> img, angles, modgrad;
  std::vector<unsigned char> used;

  bool is_aligned(int address, double theta, double prec) const {
    if (address < 0) return false;
    const double a = angles[address];
    if (a == NOTDEF) return false;
    double n_theta = theta - a;
    if (n_theta < 0) n_theta = -n_theta;
    if (n_theta > M_3_2_PI) {
      n_theta -= M_2__PI;
      if (n_theta < 0) n_theta = -n_theta;
    }
    return n_theta <= prec;
  }

  void scale_image(const unsigned char* gray, int w0, int h0) {
    // ---- GaussianBlur
    const double sigma = SIGMA_SCALE / SCALE, sprec = 3;
    const unsigned hk = (unsigned)std::ceil(sigma * std::sqrt(2 * sprec * std::log(10.0)));
    const int ks = 1 + 2 * (int)hk;
    std::vector<double> k(ks);
    {
      const double scale2X = -0.5 / (sigma * sigma);
      double sum = 0;
      for (int i = 0; i < ks; i++) { const double x = i - (ks - 1) * 0.5; k[i] = std::exp(scale2X * x * x); sum += k[i]; }
      sum = 1. / sum;
      for (int i = 0; i < ks; i++) k[i] *= sum;
    }
    std::vector<double> src((size_t)w0 * h0), rowp((size_t)w0 * h0), blur((size_t)w0 * h0);
    for (size_t i = 0; i < src.size(); i++) src[i] = (double)gray[i];
    const int r = ks / 2;
    for (int y = 0; y < h0; y++)
      for (int x = 0; x < w0; x++) {
        double s0 = k[0] * src[(size_t)y * w0 + reflect101(x - r, w0)];
        for (int t = 1; t < ks; t++) s0 += k[t] * src[(size_t)y * w0 + reflect101(x - r + t, w0)];
        rowp[(size_t)y * w0 + x] = s0;
      }
    for (int y = 0; y < h0; y++)
      for (int x = 0; x < w0; x++) {
        double s0 = k[r] * rowp[(size_t)y * w0 + x];
        for (int t = 1; t <= r; t++) s0 += k[r + t] * (rowp[(size_t)reflect101(y + t, h0) * w0 + x] + rowp[(size_t)reflect101(y - t, h0) * w0 + x]);
        blur[(size_t)y * w0 + x] = s0;
      }
    // ---- resize by SCALE, bilinear
    W = (int)std::lrint(w0 * SCALE); H = (int)std::lrint(h0 * SCALE);
    const double inv_x = SCALE, inv_y = SCALE, scale_x = 1. / inv_x, scale_y = 1. / inv_y;
    std::vector<int> xo(W), yo(H);
    std::vector<float> xa(2 * (size_t)W), ya(2 * (size_t)H);
    for (int dx = 0; dx < W; dx++) {
      float fx = (float)((dx + 0.5) * scale_x - 0.5);
      int sx = (int)std::floor(fx);
      fx -= sx;
      if (sx < 0) { fx = 0; sx = 0; }
      if (sx >= w0 - 1) { fx = 0; sx = w0 - 1; }
      xo[dx] = sx; xa[2 * dx] = 1.f - fx; xa[2 * dx + 1] = fx;
    }
    for (int dy = 0; dy < H; dy++) {
      float fy = (float)((dy + 0.5) * scale_y - 0.5);
      int sy = (int)std::floor(fy);
      fy -= sy;
      yo[dy] = sy; ya[2 * dy] = 1.f - fy; ya[2 * dy + 1] = fy;
    }
    img.assign((size_t)W * H, 0.0);
    std::vector<double> r0(W), r1(W);
    auto hrow = [&](int sy, std::vector<double>& out) {
      sy = std::min(std::max(sy, 0), h0 - 1);
      const double* S = &blur[(size_t)sy * w0];
      for (int dx = 0; dx < W; dx++) {
        const int sx = xo[dx];
        const int sx1 = std::min(sx + 1, w0 - 1);
        out[dx] = S[sx] * xa[2 * dx] + S[sx1] * xa[2 * dx + 1];
      }
    };
    for (int dy = 0; dy < H; dy++) {
      hrow(yo[dy], r0); hrow(yo[dy] + 1, r1);
      const float b0 = ya[2 * dy], b1 = ya[2 * dy + 1];
      for (int dx = 0; dx < W; dx++) img[(size_t)dy * W + dx] = r0[dx] * b0 + r1[dx] * b1;
    }
  }

  void ll_angle(double threshold) {
    angles.assign((size_t)W * H, 0.0); modgrad.assign((size_t)W * H, 0.0);
    for (int x = 0; x < W; x++) angles[(size_t)(H - 1) * W + x] = NOTDEF;
    for (int y = 0; y < H; y++) angles[(size_t)y * W + W - 1] = NOTDEF;
    for (int y = 0; y < H - 1; ++y)
      for (int addr = y * W, addr_end = addr + W - 1; addr < addr_end; ++addr) {
        const double DA = img[addr + W + 1] - img[addr], BC = img[addr + 1] - img[addr + W];
        const double gx = DA + BC, gy = DA - BC;
        const double norm = std::sqrt((gx * gx + gy * gy) / 4);
        modgrad[addr] = norm;
        if (norm <= threshold) angles[addr] = NOTDEF;
        else angles[addr] = fast_atan2(float(gx), float(-gy)) * DEG_TO_RADS;
      }
  }

  void region_grow(int sx, int sy, std::vector<RegionPoint>& reg, int& reg_size, double& reg_angle, double prec) {
    reg_size = 1;
    reg[0].x = sx; reg[0].y = sy;
    int addr = sx + sy * W;
    reg[0].used = &used[addr];
    reg_angle = angles[addr];
    reg[0].angle = reg_angle;
    reg[0].modgrad = modgrad[addr];
    float sumdx = float(std::cos(reg_angle)), sumdy = float(std::sin(reg_angle));
    *reg[0].used = 1;
    for (int i = 0; i < reg_size; ++i) {
      const RegionPoint rp = reg[i];
      const int xx_min = std::max(rp.x - 1, 0), xx_max = std::min(rp.x + 1, W - 1);
      const int yy_min = std::max(rp.y - 1, 0), yy_max = std::min(rp.y + 1, H - 1);
      for (int yy = yy_min; yy <= yy_max; ++yy) {
        int c_addr = xx_min + yy * W;
        for (int xx = xx_min; xx <= xx_max; ++xx, ++c_addr) {
          if (used[c_addr] != 1 && is_aligned(c_addr, reg_angle, prec)) {
            used[c_addr] = 1;
            RegionPoint& q = reg[reg_size];
            q.x = xx; q.y = yy; q.used = &used[c_addr]; q.modgrad = modgrad[c_addr];
            const double angle = angles[c_addr];
            q.angle = angle;
            ++reg_size;
            sumdx += std::cos(float(angle));     // float overloads (lsd.cpp:682-683)
            sumdy += std::sin(float(angle));
            reg_angle = fast_atan2(sumdy, sumdx) * DEG_TO_RADS;
          }
        }
      }
    }
  }

  double get_theta(const std::vector<RegionPoint>& reg, int reg_size, double x, double y, double reg_angle, double prec) const {
    double Ixx = 0, Iyy = 0, Ixy = 0;
    for (int i = 0; i < reg_size; ++i) {
      const double regx = reg[i].x, regy = reg[i].y, weight = reg[i].modgrad;
      const double dx = regx - x, dy = regy - y;
      Ixx += dy * dy * weight; Iyy += dx * dx * weight; Ixy -= dx * dy * weight;
    }
    const double lambda = 0.5 * (Ixx + Iyy - std::sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (std::fabs(Ixx) > std::fabs(Iyy)) ? double(fast_atan2(float(lambda - Ixx), float(Ixy))) : double(fast_atan2(float(Ixy), float(lambda - Iyy)));
    theta *= DEG_TO_RADS;
    if (angle_diff(theta, reg_angle) > prec) theta += kPi;
    return theta;
  }

  void region2rect(const std::vector<RegionPoint>& reg, int reg_size, double reg_angle, double prec, double p, Rect& rec) const {
    double x = 0, y = 0, sum = 0;
    for (int i = 0; i < reg_size; ++i) { const double w = reg[i].modgrad; x += double(reg[i].x) * w; y += double(reg[i].y) * w; sum += w; }
    x /= sum; y /= sum;
    const double theta = get_theta(reg, reg_size, x, y, reg_angle, prec);
    const double dx = std::cos(theta), dy = std::sin(theta);
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
    for (int i = 0; i < reg_size; ++i) {
      const double regdx = double(reg[i].x) - x, regdy = double(reg[i].y) - y;
      const double l = regdx * dx + regdy * dy, w = -regdx * dy + regdy * dx;
      if (l > l_max) l_max = l; else if (l < l_min) l_min = l;
      if (w > w_max) w_max = w; else if (w < w_min) w_min = w;
    }
    rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy; rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
    rec.width = w_max - w_min; rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy; rec.prec = prec; rec.p = p;
    if (rec.width < 1.0) rec.width = 1.0;
  }

  bool reduce_region_radius(std::vector<RegionPoint>& reg, int& reg_size, double reg_angle, double prec, double p, Rect& rec, double density, double density_th) {
    const double xc = double(reg[0].x), yc = double(reg[0].y);
    const double r1 = dist_sq(xc, yc, rec.x1, rec.y1), r2 = dist_sq(xc, yc, rec.x2, rec.y2);
    double radSq = r1 > r2 ? r1 : r2;
    while (density < density_th) {
      radSq *= 0.75 * 0.75;
      for (int i = 0; i < reg_size; ++i)
        if (dist_sq(xc, yc, double(reg[i].x), double(reg[i].y)) > radSq) {
          *(reg[i].used) = 0;
          std::swap(reg[i], reg[reg_size - 1]);
          --reg_size;
          --i;
        }
      if (reg_size < 2) return false;
      region2rect(reg, reg_size, reg_angle, prec, p, rec);
      density = double(reg_size) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    }
    return true;
  }

  bool refine(std::vector<RegionPoint>& reg, int& reg_size, double reg_angle, double prec, double p, Rect& rec, double density_th) {
    double density = double(reg_size) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density >= density_th) return true;
    const double xc = double(reg[0].x), yc = double(reg[0].y), ang_c = reg[0].angle;
    double sum = 0, s_sum = 0;
    int n = 0;
    for (int i = 0; i < reg_size; ++i) {
      *(reg[i].used) = 0;
      if (dist(xc, yc, reg[i].x, reg[i].y) < rec.width) {
        const double ang_d = angle_diff_signed(reg[i].angle, ang_c);
        sum += ang_d; s_sum += ang_d * ang_d; ++n;
      }
    }
    const double mean_angle = sum / double(n);
    const double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean_angle * sum) / double(n) + mean_angle * mean_angle);
    region_grow(reg[0].x, reg[0].y, reg, reg_size, reg_angle, tau);
    if (reg_size < 2) return false;
    region2rect(reg, reg_size, reg_angle, prec, p, rec);
    density = double(reg_size) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density < density_th) return reduce_region_radius(reg, reg_size, reg_angle, prec, p, rec, density, density_th);
    return true;
  }

  double nfa(int n, int k, double p) const {
    if (n == 0 || k == 0) return -LOG_NT;
    if (n == k) return -LOG_NT - double(n) * std::log10(p);
    const double p_term = p / (1 - p);
    const double log1term = (double(n) + 1) - log_gamma(double(k) + 1) - log_gamma(double(n - k) + 1) + double(k) * std::log(p) + double(n - k) * std::log(1.0 - p);
    double term = std::exp(log1term);
    if (double_equal(term, 0)) {
      if (k > n * p) return -log1term / M_LN10 - LOG_NT;
      return -LOG_NT;
    }
    double bin_tail = term;
    const double tolerance = 0.1;
    for (int i = k + 1; i <= n; ++i) {
      const double bin_term = double(n - i + 1) / double(i);
      const double mult_term = bin_term * p_term;
      term *= mult_term;
      bin_tail += term;
      if (bin_term < 1) {
        const double err = term * ((1 - std::pow(mult_term, double(n - i + 1))) / (1 - mult_term) - 1);
        if (err < tolerance * std::fabs(-std::log10(bin_tail) - LOG_NT) * bin_tail) break;
      }
    }
    return -std::log10(bin_tail) - LOG_NT;
  }

  double rect_nfa(const Rect& rec) const {
    int total_pts = 0, alg_pts = 0;
    const double half_width = rec.width / 2.0, dyhw = rec.dy * half_width, dxhw = rec.dx * half_width;
    Edge o[4];
    o[0] = Edge{int(rec.x1 - dyhw), int(rec.y1 + dxhw), false};
    o[1] = Edge{int(rec.x2 - dyhw), int(rec.y2 + dxhw), false};
    o[2] = Edge{int(rec.x2 + dyhw), int(rec.y2 - dxhw), false};
    o[3] = Edge{int(rec.x1 + dyhw), int(rec.y1 - dxhw), false};
    std::sort(o, o + 4, [](const Edge& a, const Edge& b) { return a.px == b.px ? a.py < b.py : a.px < b.px; });
    Edge* min_y = &o[0];
    Edge* max_y = &o[0];
    for (int i = 1; i < 4; ++i) {
      if (min_y->py > o[i].py) min_y = &o[i];
      if (max_y->py < o[i].py) max_y = &o[i];
    }
    min_y->taken = true;
    Edge* leftmost = nullptr;
    for (int i = 0; i < 4; ++i) if (!o[i].taken) { if (!leftmost) leftmost = &o[i]; else if (leftmost->px > o[i].px) leftmost = &o[i]; }
    leftmost->taken = true;
    Edge* rightmost = nullptr;
    for (int i = 0; i < 4; ++i) if (!o[i].taken) { if (!rightmost) rightmost = &o[i]; else if (rightmost->px < o[i].px) rightmost = &o[i]; }
    rightmost->taken = true;
    Edge* tailp = nullptr;
    for (int i = 0; i < 4; ++i) if (!o[i].taken) { if (!tailp) tailp = &o[i]; else if (tailp->px > o[i].px) tailp = &o[i]; }
    tailp->taken = true;
    // integer divisions, and tailp->p.x where a y was meant: as the reference has them (lsd.cpp:1065-1073)
    const double flstep = (min_y->py != leftmost->py) ? (min_y->px - leftmost->px) / (min_y->py - leftmost->py) : 0;
    const double slstep = (leftmost->py != tailp->px) ? (leftmost->px - tailp->px) / (leftmost->py - tailp->px) : 0;
    const double frstep = (min_y->py != rightmost->py) ? (min_y->px - rightmost->px) / (min_y->py - rightmost->py) : 0;
    const double srstep = (rightmost->py != tailp->px) ? (rightmost->px - tailp->px) / (rightmost->py - tailp->px) : 0;
    double lstep = flstep, rstep = frstep;
    double left_x = min_y->px, right_x = min_y->px;
    const int min_iter = min_y->py, max_iter = max_y->py;
    for (int y = min_iter; y <= max_iter; ++y) {
      if (y < 0 || y >= H) continue;
      int adx = y * W + int(left_x);
      for (int x = int(left_x); x <= int(right_x); ++x, ++adx) {
        if (x < 0 || x >= W) continue;
        ++total_pts;
        if (is_aligned(adx, rec.theta, rec.prec)) ++alg_pts;
      }
      if (y >= leftmost->py) lstep = slstep;
      if (y >= rightmost->py) rstep = srstep;
      left_x += lstep;
      right_x += rstep;
    }
    return nfa(total_pts, alg_pts, rec.p);
  }

  double rect_improve(Rect& rec) const {
    const double delta = 0.5, delta_2 = delta / 2.0;
    double log_nfa = rect_nfa(rec);
    if (log_nfa > LOG_EPS) return log_nfa;
    Rect r = rec;
    for (int n = 0; n < 5; ++n) {
      r.p /= 2; r.prec = r.p * kPi;
      const double v = rect_nfa(r);
      if (v > log_nfa) { log_nfa = v; rec = r; }
    }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (int n = 0; n < 5; ++n)
      if ((r.width - delta) >= 0.5) {
        r.width -= delta;
        const double v = rect_nfa(r);
        if (v > log_nfa) { rec = r; log_nfa = v; }
      }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (int n = 0; n < 5; ++n)
      if ((r.width - delta) >= 0.5) {
        r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2; r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2; r.width -= delta;
        const double v = rect_nfa(r);
        if (v > log_nfa) { rec = r; log_nfa = v; }
      }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (int n = 0; n < 5; ++n)
      if ((r.width - delta) >= 0.5) {
        r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2; r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2; r.width -= delta;
        const double v = rect_nfa(r);
        if (v > log_nfa) { rec = r; log_nfa = v; }
      }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (int n = 0; n < 5; ++n)
      if ((r.width - delta) >= 0.5) {
        r.p /= 2; r.prec = r.p * kPi;
        const double v = rect_nfa(r);
        if (v > log_nfa) { rec = r; log_nfa = v; }
      }
    return log_nfa;
  }

  void flsd(const unsigned char* gray, int w0, int h0, std::vector<float>& lines) {
    const double prec = kPi * ANG_TH / 180, p = ANG_TH / 180, rho = QUANT / std::sin(prec);
    scale_image(gray, w0, h0);
    ll_angle(rho);
    LOG_NT = 5 * (std::log10(double(W)) + std::log10(double(H))) / 2 + std::log10(11.0);
    const int min_reg_size = int(-LOG_NT / std::log10(p));
    used.assign((size_t)W * H, 0);
    std::vector<RegionPoint> reg((size_t)W * H);
    for (int y = 0; y < H - 1; ++y)            // the seed list in the order it was filled: raster order over the defined area
      for (int x = 0; x < W - 1; ++x) {
        const int adx = x + y * W;
        if (used[adx] != 0 || angles[adx] == NOTDEF) continue;
        int reg_size;
        double reg_angle;
        region_grow(x, y, reg, reg_size, reg_angle, prec);
        if (reg_size < min_reg_size) continue;
        Rect rec;
        region2rect(reg, reg_size, reg_angle, prec, p, rec);
        if (!refine(reg, reg_size, reg_angle, prec, p, rec, DENSITY_TH)) continue;
        const double log_nfa = rect_improve(rec);
        if (log_nfa <= LOG_EPS) continue;
        rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
        rec.x1 /= SCALE; rec.y1 /= SCALE; rec.x2 /= SCALE; rec.y2 /= SCALE; rec.width /= SCALE;
        lines.push_back(float(rec.x1)); lines.push_back(float(rec.y1)); lines.push_back(float(rec.x2)); lines.push_back(float(rec.y2));
      }
  }
};

}  // namespace

// gray: img_h x img_w, 8 bit.  out4: cap rows of x1 y1 x2 y2.  Returns the number of segments (-1: more than cap).
// Post-processing of LSDDetector::detectImpl (end points clamped into the image, segments hugging a border dropped, length) and
// line_lbd_detect::filter_lines (length > length_thres).
extern "C" int lsd_oracle_detect(const unsigned char* gray, int img_w, int img_h, double length_thres, float* out4, int cap) {
  Lsd L;
  std::vector<float> raw;
  L.flsd(gray, img_w, img_h, raw);
  const float pre_boundary_thre = 10;
  int n = 0;
  for (size_t k = 0; k + 3 < raw.size(); k += 4) {
    float e[4] = {raw[k], raw[k + 1], raw[k + 2], raw[k + 3]};
    for (int q = 0; q < 4; q += 2) {           // checkLineExtremes (LSDDetector.cpp:75-105)
      if (e[q] < 0) e[q] = 0;
      if (e[q] >= img_w) e[q] = (float)img_w - 1.0f;
      if (e[q + 1] < 0) e[q + 1] = 0;
      if (e[q + 1] >= img_h) e[q + 1] = (float)img_h - 1.0f;
    }
    const float sx = e[0], sy = e[1], ex = e[2], ey = e[3];
    if (((sx < pre_boundary_thre) && (ex < pre_boundary_thre)) || ((sx > img_w - pre_boundary_thre) && (ex > img_w - pre_boundary_thre)) ||
        ((sy < pre_boundary_thre) && (ey < pre_boundary_thre)) || ((sy > img_h - pre_boundary_thre) && (ey > img_h - pre_boundary_thre)))
      continue;
    const float len = (float)std::sqrt(std::pow(e[0] - e[2], 2) + std::pow(e[1] - e[3], 2));
    if (!(len > (float)length_thres)) continue;
    if (n >= cap) return -1;
    out4[4 * n] = sx; out4[4 * n + 1] = sy; out4[4 * n + 2] = ex; out4[4 * n + 3] = ey;
    n++;
  }
  return n;
}
