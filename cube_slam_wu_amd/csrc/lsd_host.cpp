// lsd_host.cpp -- the LSD branch of the line-segment producer behind the C ABI:
//   line_lbd_detect::detect_filter_lines(gray, lines) with use_LSD = true     line_lbd/class/line_lbd_allclass.cpp:130-150,199-235
//   LSDDetector::detectImpl, one octave (clamp, border filter, length)         line_lbd/libs/LSDDetector.cpp:55-105,154-260
//   LineSegmentDetectorImpl::flsd, LSD_REFINE_ADV with the default parameters  line_lbd/libs/lsd.cpp:402-1148
// The per-pixel stages (double Gaussian blur, resize by 0.8, gradient modulus and level-line angle) run on the device
// (csrc/lsd_kernels.hip) and come back as two double planes per image; what depends on the order pixels are visited in stays here:
//   region growing (:644-692)      seeds in raster order (the reference fills a gradient-sorted list and then never follows it),
//                                  8-neighbourhood, the region angle re-estimated from float cos / sin sums after every pixel
//   rectangle fit  (:694-802)      modulus-weighted centre, inertia axis through fastAtan2, extents
//   refinement     (:804-905)      angle tolerance from the spread near the seed, then the radius cut until density >= 0.7
//   validation     (:907-1096)     rect_improve's five search loops over rect_nfa -- with its integer slopes and the x-for-y slip in
//                                  the second slopes -- and the binomial tail of cs_nfa.h (first term n + 1, as :1107 has it)
// The arithmetic follows the reference operation for operation; tests/test_lines_gpu.py holds the result to the CPU restatement
// (itself pinned on the reference's saved segments) bit for bit.  No CPU fallback for the device stages.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/cubeslam_hip.h"
#include "batch_gate.h"
#include "cs_fast_atan.h"
#include "cs_nfa.h"

void cs_set_error_ba(const std::string& s);
extern "C" void* cs_internal_detector_stream(cs_detector* d);
extern "C" int cs_internal_detector_device(cs_detector* d);
extern "C" void** cs_internal_detector_lsd_slot(cs_detector* d, void (*deleter)(void*));
extern "C" void* cs_internal_detector_lines_mutex(cs_detector* d);
extern "C" void cs_internal_detector_parallel(cs_detector* d, int n, void (*fn)(int, void*), void* ctx);
extern "C" void cs_internal_detector_parallel_long(cs_detector* d, int n, void (*fn)(int, void*), void* ctx);

namespace cs {
struct LsdGauss { double k[7]; };
struct LsdScaleTab { const int* xo; const float* xa; const int* yo; const float* ya; };
void launch_lsd_maps(const unsigned char* gray, int W, int H, int Ws, int Hs, const LsdGauss& G, const LsdScaleTab& T, double rho, double* blur, char* out, size_t out_stride, hipStream_t st, int n_images);
}  // namespace cs

namespace {

#define LSD_TRY(expr)                                                          \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      cs_set_error_ba(std::string(#expr) + ": " + hipGetErrorString(_e));      \
      return CS_ERR_HIP;                                                       \
    }                                                                          \
  } while (0)

// createLineSegmentDetector(LSD_REFINE_ADV) defaults (lsd.cpp:185-187) and the constants of :53-64
constexpr double kScale = 0.8, kSigmaScale = 0.6, kQuant = 2.0, kAngTh = 22.5, kLogEps = 0.0, kDensityTh = 0.7;
constexpr double kPi = 3.1415926535897932384626433832795, k3PiHalf = 4.71238898038, k2Pi = 6.28318530718;
constexpr double kDegToRad = kPi / 180;
constexpr double kGrowMargin = 0.2 * kDegToRad;   // (Field::grow)

inline float atan2_deg(float y, float x) { return cs::fast_atan2_deg(y, x); }      // OpenCV's fastAtan2 (cs_fast_atan.h)

inline double sq_dist(double ax, double ay, double bx, double by) { return (bx - ax) * (bx - ax) + (by - ay) * (by - ay); }
inline double signed_angle_gap(double a, double b) {
  double d = a - b;
  while (d <= -kPi) d += k2Pi;
  while (d > kPi) d -= k2Pi;
  return d;
}

struct Pixel { int x, y; double angle, weight; };       // a region member: position, level-line angle, gradient modulus
struct Box {                                             // the rectangle of a region (struct rect, :89-97)
  double x1, y1, x2, y2, width, cx, cy, theta, ux, uy, tol, p;
};

// One scaled image's planes plus the state region growing threads through it.  The level-line angle arrives as the float fastAtan2
// returned (degrees; kUndefinedDeg where the gradient is below the threshold): the reference's double is that float times pi / 180,
// one IEEE multiplication, redone here wherever the angle is used -- half the bytes to copy back and to miss the cache on.
constexpr float kUndefinedDeg = -1024.f;
enum : unsigned char { ST_TAKEN = 1, ST_UNIT = 2, ST_UNDEF = 4 };
struct Field {
  int W, H;
  const float* deg;
  const double* weight;
  unsigned char* state = nullptr;   // ST_TAKEN: the pixel belongs to a region; ST_UNIT: its cos / sin are in `unit`; ST_UNDEF: no angle (all of these live in the worker thread's reusable scratch, see lsd_host_stage)
  float* unit = nullptr;            // per pixel: cos((float)angle), sin((float)angle), filled the first time a region tests the pixel
  Pixel* region = nullptr;
  int count = 0;                  // members of `region` in use
  double log_nt = 0;

  double ang(int at) const { return (double)deg[at] * kDegToRad; }

  static bool within(double theta, double a, double tol) {      // isAligned :924-947 for a pixel that has an angle
    double gap = theta - a;
    if (gap < 0) gap = -gap;
    if (gap > k3PiHalf) {
      gap -= k2Pi;
      if (gap < 0) gap = -gap;
    }
    return gap <= tol;
  }
  bool aligned(int at, double theta, double tol) const {
    if (at < 0) return false;
    const float d = deg[at];
    if (d == kUndefinedDeg) return false;
    return within(theta, (double)d * kDegToRad, tol);
  }

  // isAligned over a run of a row, without branches (the loop is a plain count: the compiler vectorises it)
  static int count_aligned(const float* d, int n, double theta, double tol) {
    int hits = 0;
    for (int i = 0; i < n; i++) {
      const float di = d[i];
      const double gap = std::fabs(theta - (double)di * kDegToRad), folded = std::fabs(gap - k2Pi);
      hits += (int)((di != kUndefinedDeg) & ((gap > k3PiHalf ? folded : gap) <= tol));
    }
    return hits;
  }

  // The same test for one (theta, tol) and many pixels (rect_nfa scans ~700 k of them per image): as a function of the stored float
  // the test passes on at most two intervals -- around theta and around theta -+ 2 pi, where the reference folds the difference --
  // because every step from the float to the comparison (the multiplication by pi / 180, the subtraction, the fold) is monotone in
  // IEEE arithmetic on either side of the interval's centre.  The intervals' end points are found by bisection over the float bit
  // patterns WITH the reference's own arithmetic, inside a quarter turn of the centre where nothing else can pass (tol < 90 degrees);
  // a pixel then costs two range comparisons on the stored float.
  struct AlignedSet {
    float lo[2], hi[2];
    bool exact = false;       // false: intervals not established, use count_aligned
    int count(const float* d, int n) const {
      int hits = 0;
      for (int i = 0; i < n; i++) hits += (int)(((d[i] >= lo[0]) & (d[i] <= hi[0])) | ((d[i] >= lo[1]) & (d[i] <= hi[1])));
      return hits;
    }
  };
  static AlignedSet aligned_set(double theta, double tol) {
    AlignedSet S;
    S.lo[0] = S.lo[1] = 1.f; S.hi[0] = S.hi[1] = 0.f;      // empty
    if (!(tol > 0) || !(tol < kPi / 2 - 0.01) || !(theta >= -k2Pi) || !(theta <= 2 * k2Pi)) return S;
    auto pass = [&](float d) { return within(theta, (double)d * kDegToRad, tol); };
    auto bits = [](float f) { unsigned u; std::memcpy(&u, &f, 4); return u; };
    auto flt = [](unsigned u) { float f; std::memcpy(&f, &u, 4); return f; };
    auto clampd = [](double deg) { return (float)std::min(360.0, std::max(0.0, deg)); };
    int n = 0;
    const double centres[3] = {theta, theta - k2Pi, theta + k2Pi};
    for (double c : centres) {
      if (c < -tol || c > k2Pi + tol) continue;
      // a float that passes, next to the centre (the centre itself may lie just outside [0, 360] or be rounded across the end point)
      float mid = clampd(c / kDegToRad);
      if (!pass(mid)) continue;                                  // (an interval that does not reach into [0, 360]; or a tolerance of a few ulps: see `exact`)
      if (n == 2) return S;                                      // (cannot happen for tol < 90 degrees; stay with the plain loop)
      const float left_end = clampd((c - kPi / 2) / kDegToRad), right_end = clampd((c + kPi / 2) / kDegToRad);
      // smallest passing float in [left_end, mid], largest in [mid, right_end]: a few steps from where c -+ tol lands; bisection if not
      const unsigned bl = bits(left_end), bm = bits(mid), br = bits(right_end);
      unsigned lo_b, hi_b;
      if (pass(left_end)) lo_b = bl;
      else {
        unsigned e = std::min(bm, std::max(bl + 1, bits(clampd((c - tol) / kDegToRad))));
        int steps = 0;
        if (pass(flt(e))) { while (steps < 32 && e > bl + 1 && pass(flt(e - 1))) { e--; steps++; } }
        else { while (steps < 32 && !pass(flt(e))) { e++; steps++; } }          // (e <= bm, and mid passes)
        if (steps == 32) { unsigned a = bl, b = bm; while (b - a > 1) { const unsigned m = a + (b - a) / 2; if (pass(flt(m))) b = m; else a = m; } e = b; }
        lo_b = e;
      }
      if (pass(right_end)) hi_b = br;
      else {
        unsigned e = std::max(bm, std::min(br - 1, bits(clampd((c + tol) / kDegToRad))));
        int steps = 0;
        if (pass(flt(e))) { while (steps < 32 && e < br - 1 && pass(flt(e + 1))) { e++; steps++; } }
        else { while (steps < 32 && !pass(flt(e))) { e--; steps++; } }          // (e >= bm, and mid passes)
        if (steps == 32) { unsigned a = bm, b = br; while (b - a > 1) { const unsigned m = a + (b - a) / 2; if (pass(flt(m))) a = m; else b = m; } e = a; }
        hi_b = e;
      }
      S.lo[n] = flt(lo_b); S.hi[n] = flt(hi_b);
      n++;
    }
    // a tolerance so small that no float next to the centre passes would leave the set empty although a pixel might pass: such
    // tolerances (below 1e-5 rad) do not occur (p >= 0.125 / 1024), and the plain loop takes them if they ever did
    S.exact = tol > 1e-5;
    return S;
  }

  // :644-692.  Returns the region angle.
  // The reference re-estimates the region angle after EVERY pixel it adds (theta = fastAtan2(sum sin, sum cos)) and tests the next
  // neighbour against it: a chain of ~60 cycles of dependent arithmetic per pixel (division, polynomial, fix-ups) that the rest of
  // the loop waits for.  The sums are kept exactly as the reference keeps them, but a neighbour's test is first decided from the
  // sums themselves: with v = (sum cos, sum sin) and u = (cos a, sin a), cos(angle between them) = v.u / |v|, compared (squared, no
  // root) against cos(tol -+ kGrowMargin).  fastAtan2 is within 0.0096 degrees of the true angle of v (every quotient, every
  // branch: tools/microbench/lsd_atan_bound.cpp runs through all of them), u within 3e-5 degrees of a, so outside a band of 0.2 degrees around
  // the tolerance the reference's comparison cannot come out differently; inside the band (about one test in a thousand) -- and
  // whenever the sums nearly cancel, or the tolerance is not well inside (0, 90) degrees -- the reference's own arithmetic decides.
  // A pixel's cos / sin (the reference's float calls, needed exactly for the sums) are computed once and kept: a pixel is tested
  // from about three neighbours before it joins a region.  Which of the eight neighbours are free AND have an angle comes from the
  // state bytes of the three rows at once (away from the image border), so the loop runs over candidates only -- in the reference's order.
  double grow(int sx, int sy, double tol) {
    const int at0 = sx + sy * W;
    double theta = ang(at0);
    region[0] = Pixel{sx, sy, theta, weight[at0]};
    count = 1;
    state[at0] |= ST_TAKEN;
    float sum_c = (float)std::cos(theta), sum_s = (float)std::sin(theta);
    const bool fast = tol - kGrowMargin > 0 && tol + kGrowMargin < kPi / 2;
    const double k_in = fast ? std::cos(tol - kGrowMargin) : 0, k_out = fast ? std::cos(tol + kGrowMargin) : 0;
    const double k_in2 = k_in * k_in * (1 + 1e-9), k_out2 = k_out * k_out * (1 - 1e-9);
    bool theta_current = true;          // theta is the estimate that belongs to the sums (the seed's own angle before the first addition)
    bool sums_usable = false;           // at least one pixel added, the tolerance inside (0, 90) degrees, the sums not nearly cancelling
    double vc = 0, vs = 0, in_thr = 0, out_thr = 0;
    auto candidate = [&](int x, int y, int at) {
      const double a = ang(at);
      float ca, sa;
      if (state[at] & ST_UNIT) { ca = unit[2 * at]; sa = unit[2 * at + 1]; }
      else { ca = std::cos((float)a); sa = std::sin((float)a); unit[2 * at] = ca; unit[2 * at + 1] = sa; state[at] |= ST_UNIT; }
      int verdict = -1;                 // 1 aligned, 0 not, -1 the reference's arithmetic decides
      if (sums_usable) {                // (vc, vs, the two thresholds: functions of the sums, refreshed when a pixel is added)
        const double dot = vc * ca + vs * sa, d2 = dot * dot;
        if (dot > 0 && d2 >= in_thr) verdict = 1;
        else if (dot <= 0 || d2 <= out_thr) verdict = 0;
      }
      if (verdict < 0) {
        if (!theta_current) { theta = (double)atan2_deg(sum_s, sum_c) * kDegToRad; theta_current = true; }
        verdict = within(theta, a, tol) ? 1 : 0;
      }
      if (!verdict) return;
      state[at] |= ST_TAKEN;
      region[count++] = Pixel{x, y, a, weight[at]};
      sum_c += ca;
      sum_s += sa;
      theta_current = false;
      vc = sum_c; vs = sum_s;
      const double n2 = vc * vc + vs * vs;
      sums_usable = fast && n2 > 0.25;
      in_thr = k_in2 * n2; out_thr = k_out2 * n2;
    };
    for (int i = 0; i < count; i++) {
      const int px = region[i].x, py = region[i].y;
      if (px > 0 && px < W - 1 && py > 0 && py < H - 1) {
        // bit 3 r + c of m: the neighbour in row py - 1 + r, column px - 1 + c is neither taken nor without an angle
        unsigned m = 0;
        const unsigned char* row = state + (py - 1) * W + (px - 1);
#pragma unroll
        for (int r = 0; r < 3; r++, row += W) {
          unsigned v;
          std::memcpy(&v, row, 4);      // (three bytes of the row and one beyond: the scratch is padded)
          v &= 0x050505u;
          const unsigned free3 = ~(v | (v >> 2)) & 0x010101u;
          m |= (((free3 * 0x10204u) >> 16) & 7u) << (3 * r);      // bits 0, 8, 16 -> 0, 1, 2 (the partial products land on distinct bits)
        }
        while (m) {
          const int bit = __builtin_ctz(m);
          m &= m - 1;
          const int r = (bit * 11) >> 5, c = bit - 3 * r;      // bit / 3 for bit < 9
          candidate(px - 1 + c, py - 1 + r, (py - 1 + r) * W + px - 1 + c);
        }
      } else {
        const int xa = std::max(px - 1, 0), xb = std::min(px + 1, W - 1), ya = std::max(py - 1, 0), yb = std::min(py + 1, H - 1);
        for (int y = ya; y <= yb; y++)
          for (int x = xa, at = xa + y * W; x <= xb; x++, at++)
            if (!(state[at] & (ST_TAKEN | ST_UNDEF))) candidate(x, y, at);
      }
    }
    if (!theta_current) theta = (double)atan2_deg(sum_s, sum_c) * kDegToRad;
    return theta;
  }

  // region2rect + get_theta :694-802
  void fit(double region_theta, double tol, double p, Box& b) const {
    double cx = 0, cy = 0, wsum = 0;
    for (int i = 0; i < count; i++) { const double w = region[i].weight; cx += (double)region[i].x * w; cy += (double)region[i].y * w; wsum += w; }
    cx /= wsum; cy /= wsum;
    double ixx = 0, iyy = 0, ixy = 0;
    for (int i = 0; i < count; i++) {
      const double w = region[i].weight, dx = (double)region[i].x - cx, dy = (double)region[i].y - cy;
      ixx += dy * dy * w; iyy += dx * dx * w; ixy -= dx * dy * w;
    }
    const double small_eig = 0.5 * (ixx + iyy - std::sqrt((ixx - iyy) * (ixx - iyy) + 4.0 * ixy * ixy));
    double theta = (std::fabs(ixx) > std::fabs(iyy)) ? (double)atan2_deg((float)(small_eig - ixx), (float)ixy) : (double)atan2_deg((float)ixy, (float)(small_eig - iyy));
    theta *= kDegToRad;
    if (std::fabs(signed_angle_gap(theta, region_theta)) > tol) theta += kPi;
    const double ux = std::cos(theta), uy = std::sin(theta);
    double lo_l = 0, hi_l = 0, lo_w = 0, hi_w = 0;
    for (int i = 0; i < count; i++) {
      const double dx = (double)region[i].x - cx, dy = (double)region[i].y - cy;
      const double along = dx * ux + dy * uy, across = -dx * uy + dy * ux;
      if (along > hi_l) hi_l = along; else if (along < lo_l) lo_l = along;
      if (across > hi_w) hi_w = across; else if (across < lo_w) lo_w = across;
    }
    b.x1 = cx + lo_l * ux; b.y1 = cy + lo_l * uy; b.x2 = cx + hi_l * ux; b.y2 = cy + hi_l * uy;
    b.width = hi_w - lo_w; b.cx = cx; b.cy = cy; b.theta = theta; b.ux = ux; b.uy = uy; b.tol = tol; b.p = p;
    if (b.width < 1.0) b.width = 1.0;
  }

  double density(const Box& b) const { return (double)count / (std::sqrt(sq_dist(b.x1, b.y1, b.x2, b.y2)) * b.width); }

  // refine + reduce_region_radius :804-905.  False: the region is given up.
  bool refine(double region_theta, double tol, double p, Box& b) {
    double dens = density(b);
    if (dens >= kDensityTh) return true;
    // the angle spread of the members within one width of the seed sets a new tolerance, and the region is grown again from the seed
    const double sx = (double)region[0].x, sy = (double)region[0].y, seed_angle = region[0].angle;
    double s1 = 0, s2 = 0;
    int near = 0;
    for (int i = 0; i < count; i++) {
      state[region[i].x + region[i].y * W] &= (unsigned char)~ST_TAKEN;
      if (std::sqrt(sq_dist(sx, sy, (double)region[i].x, (double)region[i].y)) < b.width) {
        const double g = signed_angle_gap(region[i].angle, seed_angle);
        s1 += g; s2 += g * g; near++;
      }
    }
    const double mean = s1 / (double)near;
    const double tau = 2.0 * std::sqrt((s2 - 2.0 * mean * s1) / (double)near + mean * mean);
    region_theta = grow(region[0].x, region[0].y, tau);
    if (count < 2) return false;
    fit(region_theta, tol, p, b);
    dens = density(b);
    if (dens >= kDensityTh) return true;
    // still too sparse: keep only what lies within a shrinking radius of the seed
    const double r1 = sq_dist(sx, sy, b.x1, b.y1), r2 = sq_dist(sx, sy, b.x2, b.y2);
    double rad_sq = r1 > r2 ? r1 : r2;
    while (dens < kDensityTh) {
      rad_sq *= 0.75 * 0.75;
      for (int i = 0; i < count; i++)
        if (sq_dist(sx, sy, (double)region[i].x, (double)region[i].y) > rad_sq) {
          state[region[i].x + region[i].y * W] &= (unsigned char)~ST_TAKEN;
          std::swap(region[i], region[count - 1]);
          count--;
          i--;
        }
      if (count < 2) return false;
      fit(region_theta, tol, p, b);
      dens = density(b);
    }
    return true;
  }

  // rect_nfa :1008-1096: aligned / total pixels inside the rectangle, scanned row by row between two stepped borders
  double box_score(const Box& b) const {
    struct Corner { int x, y; bool used; };
    const double hw = b.width / 2.0, oy = b.uy * hw, ox = b.ux * hw;
    Corner c[4] = {{(int)(b.x1 - oy), (int)(b.y1 + ox), false}, {(int)(b.x2 - oy), (int)(b.y2 + ox), false}, {(int)(b.x2 + oy), (int)(b.y2 - ox), false}, {(int)(b.x1 + oy), (int)(b.y1 - ox), false}};
    std::sort(c, c + 4, [](const Corner& l, const Corner& r) { return l.x == r.x ? l.y < r.y : l.x < r.x; });
    Corner* top = &c[0];
    Corner* bottom = &c[0];
    for (int i = 1; i < 4; i++) {
      if (top->y > c[i].y) top = &c[i];
      if (bottom->y < c[i].y) bottom = &c[i];
    }
    top->used = true;
    auto pick = [&](bool want_right) {
      Corner* best = nullptr;
      for (int i = 0; i < 4; i++) {
        if (c[i].used) continue;
        if (!best || (want_right ? best->x < c[i].x : best->x > c[i].x)) best = &c[i];
      }
      best->used = true;
      return best;
    };
    const Corner* left = pick(false);
    const Corner* right = pick(true);
    const Corner* tail = pick(false);
    // int / int, and tail->x where tail->y was meant: as the reference computes them
    const double l_first = (top->y != left->y) ? (top->x - left->x) / (top->y - left->y) : 0;
    const double l_second = (left->y != tail->x) ? (left->x - tail->x) / (left->y - tail->x) : 0;
    const double r_first = (top->y != right->y) ? (top->x - right->x) / (top->y - right->y) : 0;
    const double r_second = (right->y != tail->x) ? (right->x - tail->x) / (right->y - tail->x) : 0;
    double l_step = l_first, r_step = r_first, xl = top->x, xr = top->x;
    int total = 0, hits = 0;
    const AlignedSet set = aligned_set(b.theta, b.tol);
    for (int y = top->y; y <= bottom->y; y++) {
      if (y < 0 || y >= H) continue;                     // (the borders do not advance on skipped rows either)
      const int x0 = std::max((int)xl, 0), x1 = std::min((int)xr, W - 1);      // (the reference walks (int)xl .. (int)xr and skips what lies outside the image)
      if (x1 >= x0) {
        total += x1 - x0 + 1;
        hits += set.exact ? set.count(deg + (size_t)y * W + x0, x1 - x0 + 1) : count_aligned(deg + (size_t)y * W + x0, x1 - x0 + 1, b.theta, b.tol);
      }
      if (y >= left->y) l_step = l_second;
      if (y >= right->y) r_step = r_second;
      xl += l_step;
      xr += r_step;
    }
    return cs::minus_log10_nfa(total, hits, b.p, log_nt, false);
  }

  // rect_improve :907-1006
  double improve(Box& b) const {
    const double step = 0.5, half_step = step / 2.0;
    double best = box_score(b);
    if (best > kLogEps) return best;
    auto consider = [&](const Box& trial) {
      const double v = box_score(trial);
      if (v > best) { best = v; b = trial; }
    };
    Box t = b;
    for (int n = 0; n < 5; n++) { t.p /= 2; t.tol = t.p * kPi; consider(t); }                                   // finer precision
    if (best > kLogEps) return best;
    t = b;
    for (int n = 0; n < 5; n++) if (t.width - step >= 0.5) { t.width -= step; consider(t); }                      // narrower
    if (best > kLogEps) return best;
    for (int side = 0; side < 2; side++) {                                                                       // narrower from one side, then the other
      t = b;
      for (int n = 0; n < 5; n++)
        if (t.width - step >= 0.5) {
          if (side == 0) { t.x1 += -t.uy * half_step; t.y1 += t.ux * half_step; t.x2 += -t.uy * half_step; t.y2 += t.ux * half_step; }
          else { t.x1 -= -t.uy * half_step; t.y1 -= t.ux * half_step; t.x2 -= -t.uy * half_step; t.y2 -= t.ux * half_step; }
          t.width -= step;
          consider(t);
        }
      if (best > kLogEps) return best;
    }
    t = b;
    for (int n = 0; n < 5; n++) if (t.width - step >= 0.5) { t.p /= 2; t.tol = t.p * kPi; consider(t); }          // finer precision again
    return best;
  }
};

// the sequential half of one image: segments in the reference's order, post-processed as LSDDetector::detectImpl and filter_lines do
int lsd_host_stage(int img_w, int img_h, int Ws, int Hs, const float* deg, const double* weight, double length_thres, float* lines4, int cap, int* n_lines) {
  *n_lines = 0;
  // a region can grow to the whole image, so its array is image-sized (8 MB at KITTI size): kept per worker thread across calls instead of
  // being allocated and page-faulted in per image
  static thread_local std::vector<unsigned char> tl_state;
  static thread_local std::vector<Pixel> tl_region;
  static thread_local std::vector<float> tl_unit;
  const size_t Ns = (size_t)Ws * Hs;
  if (tl_state.size() < Ns + 16) tl_state.resize(Ns + 16, ST_UNDEF);
  {
    unsigned char* st = tl_state.data();
    for (size_t at = 0; at < Ns; at++) st[at] = deg[at] == kUndefinedDeg ? ST_UNDEF : 0;
    for (size_t at = Ns; at < Ns + 16; at++) st[at] = ST_UNDEF;
  }
  if (tl_region.size() < (size_t)Ws * Hs) tl_region.resize((size_t)Ws * Hs);
  if (tl_unit.size() < 2 * (size_t)Ws * Hs) tl_unit.resize(2 * (size_t)Ws * Hs);
  Field F;
  F.W = Ws; F.H = Hs; F.deg = deg; F.weight = weight;
  F.state = tl_state.data(); F.region = tl_region.data(); F.unit = tl_unit.data();
  const double tol = kPi * kAngTh / 180, p = kAngTh / 180;
  F.log_nt = 5 * (std::log10((double)Ws) + std::log10((double)Hs)) / 2 + std::log10(11.0);
  const int min_region = (int)(-F.log_nt / std::log10(p));
  const float border = 10;                                     // pre_boundary_thre, LSDDetector.cpp:219
  int n = 0;
  // the seeds in raster order (the last column and the last row have no angle, so walking every pixel is walking the reference's
  // (Hs - 1) x (Ws - 1) area): eight state bytes at a time, re-read after every region because growing it takes pixels further on
  for (size_t g0 = 0; g0 < Ns; g0 += 8)
    for (int from = 0; from < 8;) {
      unsigned long long v;
      std::memcpy(&v, F.state + g0, 8);
      v &= 0x0505050505050505ull;
      unsigned long long free8 = ~(v | (v >> 2)) & 0x0101010101010101ull;
      free8 &= ~0ull << (8 * from);
      if (!free8) break;
      const int k = __builtin_ctzll(free8) >> 3;
      from = k + 1;
      const int at = (int)g0 + k, y = at / Ws, x = at - y * Ws;
      const double theta = F.grow(x, y, tol);
      if (F.count < min_region) continue;
      Box b;
      F.fit(theta, tol, p, b);
      if (!F.refine(theta, tol, p, b)) continue;
      if (!(F.improve(b) > kLogEps)) continue;
      // back to the input image's frame (:528-539), to float, clamped into the image (checkLineExtremes)
      float e[4] = {(float)((b.x1 + 0.5) / kScale), (float)((b.y1 + 0.5) / kScale), (float)((b.x2 + 0.5) / kScale), (float)((b.y2 + 0.5) / kScale)};
      for (int q = 0; q < 4; q++) {
        const int lim = (q & 1) ? img_h : img_w;
        if (e[q] < 0) e[q] = 0;
        if (e[q] >= lim) e[q] = (float)lim - 1.0f;
      }
      if ((e[0] < border && e[2] < border) || (e[0] > img_w - border && e[2] > img_w - border) || (e[1] < border && e[3] < border) || (e[1] > img_h - border && e[3] > img_h - border)) continue;
      const float len = (float)std::sqrt(std::pow(e[0] - e[2], 2) + std::pow(e[1] - e[3], 2));
      if (!(len > (float)length_thres)) continue;
      if (n >= cap) { cs_set_error_ba("cs_detect_lsd_gray: more segments than `cap`"); return CS_ERR_CAPACITY; }
      std::memcpy(lines4 + 4 * (size_t)n, e, sizeof(e));
      n++;
    }
  *n_lines = n;
  return CS_OK;
}

// Resident scratch of a detector's LSD producer (grows only)
struct LsdScratch {
  unsigned char* d_gray = nullptr; double* d_blur = nullptr; char* d_out = nullptr; char* d_tab = nullptr;
  char* h_out = nullptr;            // pinned: per image [modulus (double) | angle (float degrees)]
  unsigned char* h_in = nullptr;    // pinned: the batch's images side by side (one upload)
  size_t cap_in = 0, cap_out = 0;   // input pixels x images; bytes of the planes
  cs::ChunkEvents chunks;
  int tab_w = 0, tab_h = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double device_ms = 0, host_ms = 0, total_ms = 0;
};
void lsd_scratch_free(void* p) {
  LsdScratch* S = (LsdScratch*)p;
  if (!S) return;
  if (S->d_gray) (void)hipFree(S->d_gray);
  if (S->d_blur) (void)hipFree(S->d_blur);
  if (S->d_out) (void)hipFree(S->d_out);
  if (S->d_tab) (void)hipFree(S->d_tab);
  if (S->h_out) (void)hipHostFree(S->h_out);
  if (S->h_in) (void)hipHostFree(S->h_in);
  S->chunks.release();
  if (S->ev0) (void)hipEventDestroy(S->ev0);
  if (S->ev1) (void)hipEventDestroy(S->ev1);
  delete S;
}

double lsd_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

extern "C" int cs_detect_lsd_batch(cs_detector* d, const unsigned char* const* grays, int n_images, int img_w, int img_h, double length_thres, float* const* lines4, int cap,
                                   int* n_lines) {
  if (!d || n_images < 0 || (n_images && (!grays || !n_lines || (cap && !lines4))) || img_w < 8 || img_h < 8 || cap < 0) return CS_ERR_INVALID_ARG;
  for (int i = 0; i < n_images; i++) if (!grays[i] || (cap && !lines4[i])) return CS_ERR_INVALID_ARG;
  try {
    for (int i = 0; i < n_images; i++) n_lines[i] = 0;
    if (n_images == 0) return CS_OK;
    std::lock_guard<std::mutex> lk(*(std::mutex*)cs_internal_detector_lines_mutex(d));
    LSD_TRY(hipSetDevice(cs_internal_detector_device(d)));
    hipStream_t st = (hipStream_t)cs_internal_detector_stream(d);
    void** slot = cs_internal_detector_lsd_slot(d, lsd_scratch_free);
    if (!*slot) *slot = new LsdScratch();
    LsdScratch& S = *(LsdScratch*)*slot;
    const double t_begin = lsd_now_ms();
    // the scaled size: resize(..., Size(), 0.8, 0.8) rounds (saturate_cast<int>)
    const int Ws = (int)std::lrint(img_w * kScale), Hs = (int)std::lrint(img_h * kScale);
    const size_t N = (size_t)img_w * img_h, Ns = (size_t)Ws * Hs;
    const size_t out_stride = 8 * Ns + 4 * ((Ns + 1) & ~(size_t)1);       // per image: Ns doubles, Ns floats (padded to a multiple of 8 bytes)
    if (N * (size_t)n_images > S.cap_in) {
      if (S.d_gray) (void)hipFree(S.d_gray);
      if (S.d_blur) (void)hipFree(S.d_blur);
      if (S.h_in) (void)hipHostFree(S.h_in);
      S.d_gray = nullptr; S.d_blur = nullptr; S.h_in = nullptr; S.cap_in = 0;
      LSD_TRY(hipMalloc((void**)&S.d_gray, N * (size_t)n_images));
      LSD_TRY(hipHostMalloc((void**)&S.h_in, N * (size_t)n_images));
      LSD_TRY(hipMalloc((void**)&S.d_blur, N * (size_t)n_images * sizeof(double)));
      S.cap_in = N * (size_t)n_images;
    }
    if (out_stride * (size_t)n_images > S.cap_out) {
      if (S.d_out) (void)hipFree(S.d_out);
      if (S.h_out) (void)hipHostFree(S.h_out);
      S.d_out = nullptr; S.h_out = nullptr; S.cap_out = 0;
      LSD_TRY(hipMalloc((void**)&S.d_out, out_stride * (size_t)n_images));
      LSD_TRY(hipHostMalloc((void**)&S.h_out, out_stride * (size_t)n_images));
      S.cap_out = out_stride * (size_t)n_images;
    }
    if (!S.ev0) { LSD_TRY(hipEventCreate(&S.ev0)); LSD_TRY(hipEventCreate(&S.ev1)); }
    // resize's tables for this size: source offset and the two float weights per scaled column / row (pixel centres, INTER_LINEAR)
    const size_t tab_bytes = (size_t)(Ws + Hs) * (sizeof(int) + 2 * sizeof(float));
    if (S.tab_w != img_w || S.tab_h != img_h) {
      std::vector<char> host(tab_bytes);
      int* xo = (int*)host.data();
      int* yo = xo + Ws;
      float* xa = (float*)(yo + Hs);
      float* ya = xa + 2 * (size_t)Ws;
      const double step = 1.0 / kScale;
      for (int i = 0; i < Ws; i++) {
        float f = (float)((i + 0.5) * step - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= img_w - 1) { f = 0; s = img_w - 1; }
        xo[i] = s; xa[2 * i] = 1.f - f; xa[2 * i + 1] = f;
      }
      for (int i = 0; i < Hs; i++) {
        float f = (float)((i + 0.5) * step - 0.5);
        const int s = (int)std::floor(f);
        f -= s;
        yo[i] = s; ya[2 * i] = 1.f - f; ya[2 * i + 1] = f;
      }
      if (S.d_tab) (void)hipFree(S.d_tab);
      S.d_tab = nullptr; S.tab_w = S.tab_h = 0;
      LSD_TRY(hipMalloc((void**)&S.d_tab, tab_bytes));
      LSD_TRY(hipMemcpy(S.d_tab, host.data(), tab_bytes, hipMemcpyHostToDevice));
      S.tab_w = img_w; S.tab_h = img_h;
    }
    cs::LsdScaleTab T;
    T.xo = (const int*)S.d_tab; T.yo = T.xo + Ws; T.xa = (const float*)(T.yo + Hs); T.ya = T.xa + 2 * (size_t)Ws;
    // getGaussianKernel(7, 0.6 / 0.8, CV_64F): the window is 1 + 2 ceil(sigma sqrt(2 * 3 ln 10)) = 7 for these parameters (:452-456)
    cs::LsdGauss G;
    {
      const double sigma = kSigmaScale / kScale;
      if (1 + 2 * (int)std::ceil(sigma * std::sqrt(2 * 3 * std::log(10.0))) != 7) { cs_set_error_ba("cs_detect_lsd_batch: window"); return CS_ERR_INVALID_ARG; }
      double sum = 0;
      for (int i = 0; i < 7; i++) { const double x = i - 3.0; G.k[i] = std::exp(-0.5 / (sigma * sigma) * x * x); sum += G.k[i]; }
      sum = 1. / sum;
      for (int i = 0; i < 7; i++) G.k[i] *= sum;
    }
    const double rho = kQuant / std::sin(kPi * kAngTh / 180);
    struct Ctx {
      const char* h; size_t out_stride; int w, h0, Ws, Hs; size_t Ns; double thr; float* const* lines4; int cap; int* n_lines; std::vector<int> rc;
      cs::ChunkGate gate; int device; const hipEvent_t* done; int n_chunks, n_images;
      const unsigned char* const* grays; unsigned char* h_in; size_t N;
    } ctx{S.h_out, out_stride, img_w, img_h, Ws, Hs, Ns, length_thres, lines4, cap, n_lines, std::vector<int>(n_images, 0), {}, cs_internal_detector_device(d), nullptr, 0, n_images, grays, S.h_in, N};
    auto one = [](int i, void* vp) {
      Ctx& c = *(Ctx*)vp;
      const double* mod = (const double*)(c.h + c.out_stride * (size_t)i);
      try { c.rc[i] = lsd_host_stage(c.w, c.h0, c.Ws, c.Hs, (const float*)(mod + c.Ns), mod, c.thr, c.lines4 ? c.lines4[i] : nullptr, c.cap, &c.n_lines[i]); }
      catch (const std::exception&) { c.rc[i] = CS_ERR_CAPACITY; }
    };
    double t_host;
    if (n_images == 1) {
      LSD_TRY(hipMemcpyAsync(S.d_gray, grays[0], N, hipMemcpyHostToDevice, st));
      LSD_TRY(hipEventRecord(S.ev0, st));
      cs::launch_lsd_maps(S.d_gray, img_w, img_h, Ws, Hs, G, T, rho, S.d_blur, S.d_out, out_stride, st, 1);
      LSD_TRY(hipGetLastError());
      LSD_TRY(hipEventRecord(S.ev1, st));
      LSD_TRY(hipMemcpyAsync(S.h_out, S.d_out, out_stride, hipMemcpyDeviceToHost, st));
      LSD_TRY(hipStreamSynchronize(st));
      float ms = 0;
      LSD_TRY(hipEventElapsedTime(&ms, S.ev0, S.ev1));
      S.device_ms = ms;
      t_host = lsd_now_ms();
      one(0, &ctx);
    } else {
      // a batch: the images gathered into pinned memory by the pool (the caller's buffers are pageable: 64 staged copies otherwise), one
      // upload, then chunk by chunk [kernels | copy back | event] -- all queued before the pool starts on the first chunk (batch_gate.h)
      cs_internal_detector_parallel(d, n_images, [](int i, void* vp) { Ctx& c = *(Ctx*)vp; std::memcpy(c.h_in + c.N * (size_t)i, c.grays[i], c.N); }, &ctx);
      const int CH = cs::BATCH_CHUNK, n_chunks = (n_images + CH - 1) / CH;
      LSD_TRY(S.chunks.reserve(n_chunks));
      LSD_TRY(hipMemcpyAsync(S.d_gray, S.h_in, N * (size_t)n_images, hipMemcpyHostToDevice, st));
      for (int c = 0; c < n_chunks; c++) {
        const int i0 = c * CH, ni = std::min(CH, n_images - i0);
        LSD_TRY(hipEventRecord(S.chunks.k0[c], st));
        cs::launch_lsd_maps(S.d_gray + (size_t)i0 * N, img_w, img_h, Ws, Hs, G, T, rho, S.d_blur + (size_t)i0 * N, S.d_out + out_stride * (size_t)i0, out_stride, st, ni);
        LSD_TRY(hipGetLastError());
        LSD_TRY(hipEventRecord(S.chunks.k1[c], st));
        LSD_TRY(hipMemcpyAsync(S.h_out + out_stride * (size_t)i0, S.d_out + out_stride * (size_t)i0, out_stride * (size_t)ni, hipMemcpyDeviceToHost, st));
        LSD_TRY(hipEventRecord(S.chunks.done[c], st));
      }
      ctx.done = S.chunks.done.data(); ctx.n_chunks = n_chunks;
      t_host = lsd_now_ms();
      cs_internal_detector_parallel_long(d, n_images + 1, [](int t, void* vp) {
        Ctx& c = *(Ctx*)vp;
        if (t == 0) { c.gate.watch(c.device, c.done, c.n_chunks, cs::BATCH_CHUNK, c.n_images); return; }
        const int i = t - 1;
        if (!c.gate.wait_for(i)) { c.rc[i] = CS_ERR_HIP; return; }
        const double* mod = (const double*)(c.h + c.out_stride * (size_t)i);
        try { c.rc[i] = lsd_host_stage(c.w, c.h0, c.Ws, c.Hs, (const float*)(mod + c.Ns), mod, c.thr, c.lines4 ? c.lines4[i] : nullptr, c.cap, &c.n_lines[i]); }
        catch (const std::exception&) { c.rc[i] = CS_ERR_CAPACITY; }
      }, &ctx);
      LSD_TRY(hipStreamSynchronize(st));          // (every chunk's event has been waited for; this also surfaces a late error)
      double dev = 0;
      for (int c = 0; c < n_chunks; c++) { float ms = 0; LSD_TRY(hipEventElapsedTime(&ms, S.chunks.k0[c], S.chunks.k1[c])); dev += ms; }
      S.device_ms = dev;
    }
    S.host_ms = lsd_now_ms() - t_host; S.total_ms = lsd_now_ms() - t_begin;
    for (int r : ctx.rc) if (r) return r;
    return CS_OK;
  } catch (const std::exception& ex) {
    cs_set_error_ba(std::string("cs_detect_lsd_batch: ") + ex.what());
    return CS_ERR_CAPACITY;
  }
}

extern "C" int cs_detect_lsd_last_timing(cs_detector* d, double* device_ms, double* host_ms, double* total_ms) {
  if (!d) return CS_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(*(std::mutex*)cs_internal_detector_lines_mutex(d));
  void** slot = cs_internal_detector_lsd_slot(d, lsd_scratch_free);
  const LsdScratch* S = (const LsdScratch*)*slot;
  if (!S) return CS_ERR_NOT_RUN;
  if (device_ms) *device_ms = S->device_ms;
  if (host_ms) *host_ms = S->host_ms;
  if (total_ms) *total_ms = S->total_ms;
  return CS_OK;
}

extern "C" int cs_detect_lsd_gray(cs_detector* d, const unsigned char* gray, int img_w, int img_h, double length_thres, float* lines4, int cap, int* n_lines) {
  if (!d || !gray || img_w < 8 || img_h < 8 || cap < 0 || (cap && !lines4) || !n_lines) return CS_ERR_INVALID_ARG;
  float* l4 = lines4;
  return cs_detect_lsd_batch(d, &gray, 1, img_w, img_h, length_thres, &l4, cap, n_lines);
}
