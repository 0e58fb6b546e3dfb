// lines_host.cpp -- the line-segment producer behind the C ABI:
//   line_lbd_detect::detect_filter_lines(gray, lines)     line_lbd/class/line_lbd_allclass.cpp:199-235   (use_LSD = false: EDLines;
//                                                          the graph driver sets line_length_thres = 15, object_slam/src/main_obj.cpp:504-505)
// i.e. BinaryDescriptor::detect -> OctaveKeyLines -> EDLineDetector::EDline (libs/binary_descriptor.cpp:421-590, :796-1148, :1583-2905),
// one octave.  The per-pixel stages (Gaussian blur, Sobel, gradient / direction maps, anchors) run on the device
// (csrc/lines_kernels.hip); what is sequential by definition stays here:
//   smart routing   (EdgeDrawing :1672-2381)   anchors in column-major scan order, two routed walks per anchor, chains under 16 pixels dropped
//   line extraction (EDline :2383-2630)        least-squares fit of the first 15 pixels, extension with at most 3 consecutive outliers,
//                                              refits from running float normal equations (the reference's Mat_<float> members)
//   validation      (LineValidation_ :2793-2874, nfa descriptor.hpp:764-848), end points, start / end order (OctaveKeyLines :1074-1143)
// The arithmetic follows the reference operation for operation (same types: float normal-equation terms, double solve, float end points);
// tests/test_lines_gpu.py holds it to the CPU restatement of the same detector bit for bit.  No CPU fallback for the device stages.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/cubeslam_hip.h"
#include "batch_gate.h"
#include "cs_nfa.h"

void cs_set_error_ba(const std::string& s);
extern "C" void* cs_internal_detector_stream(cs_detector* d);
extern "C" int cs_internal_detector_device(cs_detector* d);
extern "C" void** cs_internal_detector_lines_slot(cs_detector* d, void (*deleter)(void*));
extern "C" void* cs_internal_detector_lines_mutex(cs_detector* d);
extern "C" void cs_internal_detector_parallel(cs_detector* d, int n, void (*fn)(int, void*), void* ctx);
extern "C" void cs_internal_detector_parallel_long(cs_detector* d, int n, void (*fn)(int, void*), void* ctx);

namespace cs {
struct LineMaps { unsigned char* p3; };
void launch_lines_maps(const unsigned char* gray, int W, int H, const LineMaps& m, const int k[3], int grad_thr, int anchor_thr, int scan, hipStream_t st, int n_images);
}  // namespace cs

namespace {

#define LN_TRY(expr)                                                           \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      cs_set_error_ba(std::string(#expr) + ": " + hipGetErrorString(_e));      \
      return CS_ERR_HIP;                                                       \
    }                                                                          \
  } while (0)

// EDLineDetector() :1515-1526 and BinaryDescriptor::Params() :110-117
struct EdParams { int grad_thr = 80, anchor_thr = 8, scan = 2, min_len = 15, try_time = 6, skip = 2, max_outlier = 3; double fit_err = 1.6; };

// Host copy of the device's packed map (pinned staging of the detector's scratch): one word per pixel = dx | (2 dy + anchor) << 16
// (lines_kernels.hip).  The reference's gImg_ and dirImg_ are functions of dx and dy, evaluated where they are read:
//   gImg_ = cvRound((|dx| + |dy| thresholded at gradienThreshold_ + 1) / 4)     binary_descriptor.cpp:1630-1640
//   dirImg_ = |dx| < |dy| ? Horizontal : Vertical
struct Maps {
  int W = 0, H = 0, grad_thr = 0;
  const unsigned char* p3 = nullptr;      // three bytes per pixel (lines_kernels.hip, LineMaps): dx | anchor << 11 | dy << 12; readable up to p3 + 3 N + 1
  unsigned word(size_t at) const { unsigned w; std::memcpy(&w, p3 + 3 * at, 4); return w; }      // (the fourth byte is the next pixel's, or the pad)
  int dx(size_t at) const { return (int)(word(at) << 21) >> 21; }
  int dy(size_t at) const { return (int)(word(at) << 9) >> 21; }
  bool anchor(size_t at) const { return (p3[3 * at + 1] >> 3) & 1; }
  short g(size_t at) const {
    const int s = std::abs(dx(at)) + std::abs(dy(at)), v = s > grad_thr + 1 ? s : 0, q = v >> 2, r = v & 3;
    return (short)(q + ((r == 3) || (r == 2 && (q & 1))));        // round half to even
  }
  bool horizontal(size_t at) const { return std::abs(dx(at)) < std::abs(dy(at)); }
  bool horizontal(unsigned x, unsigned y) const { return horizontal((size_t)y * W + x); }
};

// the 32-bit packing of the CPU restatement's map (dx | (2 dy + anchor) << 16) in the device's three bytes (tools/hostonly, tests)
inline void maps_pack3(const int* pk, size_t n, std::vector<unsigned char>& out) {
  out.assign(3 * n + 4, 0);
  for (size_t i = 0; i < n; i++) {
    const unsigned wd = (((unsigned)(pk[i] >> 16) & 0xfffu) << 11) | ((unsigned)pk[i] & 0x7ffu);
    out[3 * i] = (unsigned char)wd; out[3 * i + 1] = (unsigned char)(wd >> 8); out[3 * i + 2] = (unsigned char)(wd >> 16);
  }
}

struct Chain { std::vector<unsigned> x, y; };

// ---- smart routing ---------------------------------------------------------------------------------------------------------
// A walk advances one pixel per step in one of four headings.  Per heading: the three candidate neighbours in the order the
// reference compares them (first, straight, third) and the image sides that end the walk.
enum Heading { UP = 1, RIGHT = 2, DOWN = 3, LEFT = 4 };
struct Step { int dx[3], dy[3]; };
static const Step kStep[5] = {
    {{0, 0, 0}, {0, 0, 0}},
    {{+1, 0, -1}, {-1, -1, -1}},   // up:    up-right, up, up-left
    {{+1, +1, +1}, {-1, 0, +1}},   // right: up-right, right, down-right
    {{+1, 0, -1}, {+1, +1, +1}},   // down:  down-right, down, down-left
    {{-1, -1, -1}, {-1, 0, +1}},   // left:  up-left, left, down-left
};

struct Router {
  const Maps& M;
  std::vector<unsigned char> taken;
  unsigned last_x = 0, last_y = 0;   // where the previous step stood (they outlive a walk, as in the reference)
  explicit Router(const Maps& m) : M(m), taken((size_t)m.W * m.H, 0) {}

  void walk(unsigned x, unsigned y, int heading, Chain& out) {
    const unsigned W = M.W, H = M.H;
    for (;;) {
      const size_t at = (size_t)y * W + x;
      if (!(M.g(at) > 0) || taken[at]) return;
      taken[at] = 1;
      out.x.push_back(x); out.y.push_back(y);
      int go;
      if (M.horizontal(at)) go = (heading == UP || heading == DOWN) ? (x > last_x ? RIGHT : LEFT) : heading;     // horizontal pixel: left or right
      else                  go = (heading == RIGHT || heading == LEFT) ? (y > last_y ? DOWN : UP) : heading;     // vertical pixel: up or down
      last_x = x; last_y = y;
      const bool at_border = go == RIGHT ? (x == W - 1 || y == 0 || y == H - 1)
                           : go == LEFT  ? (x == 0 || y == 0 || y == H - 1)
                           : go == DOWN  ? (x == 0 || x == W - 1 || y == H - 1)
                                         : (x == 0 || x == W - 1 || y == 0);
      if (at_border) return;
      const Step& s = kStep[go];
      unsigned char gv[3];   // the reference compares the gradients as unsigned char (values above 255 wrap)
      for (int q = 0; q < 3; q++) gv[q] = (unsigned char)M.g((size_t)(y + s.dy[q]) * W + (x + s.dx[q]));
      const int pick = (gv[0] >= gv[1] && gv[0] >= gv[2]) ? 0 : ((gv[2] >= gv[1] && gv[2] >= gv[0]) ? 2 : 1);
      x += s.dx[pick]; y += s.dy[pick];
      heading = go;
    }
  }
};

// ---- running least squares: value = a * coord + b with float normal-equation terms (Mat_<float> ATA, ATV) ---------------------
struct NormalEq {
  float a00 = 0, a01 = 0, a11 = 0, v0 = 0, v1 = 0;
  // the terms of a block of pixels: sums in double, stored as float (cv::gemm on CV_32F)
  static NormalEq of(const unsigned* coord, const unsigned* value, int n) {
    double s00 = 0, s01 = 0, t0 = 0, t1 = 0;
    for (int i = 0; i < n; i++) { const double c = (double)coord[i], v = (double)value[i]; s00 += c * c; s01 += c; t0 += c * v; t1 += v; }
    NormalEq r; r.a00 = (float)s00; r.a01 = (float)s01; r.a11 = (float)(double)n; r.v0 = (float)t0; r.v1 = (float)t1;
    return r;
  }
  void add(const NormalEq& o) { a00 = a00 + o.a00; a01 = a01 + o.a01; a11 = a11 + o.a11; v0 = v0 + o.v0; v1 = v1 + o.v1; }
  void solve(double& a, double& b) const {
    const double det_inv = 1.0 / ((double)a00 * (double)a11 - (double)a01 * (double)a01);
    a = det_inv * ((double)a11 * (double)v0 - (double)a01 * (double)v1);
    b = det_inv * ((double)a00 * (double)v1 - (double)a01 * (double)v0);
  }
};

// number of false alarms (descriptor.hpp:650-848): cs_nfa.h, with the log-gamma first term
double minus_log10_nfa(int n, int k, double p, double logNT) { return cs::minus_log10_nfa(n, k, p, logNT, true); }

struct Segment { float x1, y1, x2, y2, direction; };

// ---- line extraction from the chains ------------------------------------------------------------------------------------------
struct Extractor {
  const Maps& M;
  const EdParams& P;
  double logNT;
  std::vector<unsigned> lx, ly;     // pixels of the line being grown
  Extractor(const Maps& m, const EdParams& p) : M(m), P(p), logNT(2.0 * (std::log10((double)m.W) + std::log10((double)m.H))) {}

  // gradient orientation statistics + border test + NFA (LineValidation_)
  bool accept(const double w[3], float& direction) const {
    const int n = (int)lx.size();
    int sum_gx = 0, sum_gy = 0;
    std::vector<double> level(n);
    for (int i = 0; i < n; i++) {
      const size_t at = (size_t)ly[i] * M.W + lx[i];
      const int gx = M.dx(at), gy = M.dy(at);
      sum_gx += gx; sum_gy += gy;
      level[i] = std::atan2(-(double)gx, (double)gy);
    }
    if (sum_gx == 0 && sum_gy == 0) return false;
    const double ax = std::fabs(w[1]), ay = std::fabs(w[0]);
    if (sum_gx > 0 && sum_gy >= 0) direction = (float)std::atan2(-ay, ax);
    if (sum_gx <= 0 && sum_gy > 0) direction = (float)std::atan2(ay, ax);
    if (sum_gx < 0 && sum_gy <= 0) direction = (float)std::atan2(ay, -ax);
    if (sum_gx >= 0 && sum_gy < 0) direction = (float)std::atan2(-ay, -ax);
    const double ad = std::fabs(direction);
    if ((ad < 0.15 || M_PI - ad < 0.15) && (std::fabs(w[2]) < 10 || std::fabs((unsigned)M.H - std::fabs(w[2])) < 10)) return false;     // along the top / bottom border
    if (std::fabs(ad - M_PI * 0.5) < 0.15 && (std::fabs(w[2]) < 10 || std::fabs((unsigned)M.W - std::fabs(w[2])) < 10)) return false;    // along the left / right border
    int aligned = 0;
    for (int i = 0; i < n; i++) {
      const double d = std::fabs(direction - level[i]);
      if (std::fabs(2 * M_PI - d) < 0.392699 || d < 0.392699) aligned++;
    }
    return minus_log10_nfa(n, aligned, 0.125, logNT) > 0;
  }

  void run(const Chain& c, std::vector<Segment>& out) {
    const unsigned* cx = c.x.data(); const unsigned* cy = c.y.data();
    unsigned s = 0;
    const unsigned e = (unsigned)c.x.size();
    const unsigned L = (unsigned)P.min_len;
    while (e > s + L) {
      // an initial segment: slide along the chain until 15 consecutive pixels fit a line
      NormalEq eq;
      double a = 0, b = 0, err = 0;
      bool hor = false;
      while (e > s + L) {
        hor = M.horizontal(cx[s], cy[s]);
        const unsigned* coord = (hor ? cx : cy) + s; const unsigned* value = (hor ? cy : cx) + s;
        eq = NormalEq::of(coord, value, (int)L);
        eq.solve(a, b);
        double sq = 0;
        for (unsigned i = 0; i < L; i++) { const double r = (double)value[i] - (double)coord[i] * a - b; sq += r * r; }
        err = std::sqrt(sq);
        if (err <= P.fit_err) break;
        s += P.skip;
      }
      if (err > P.fit_err) break;
      // grow it.  The orientation of the first pixel decides the parametrisation (y = a x + b or x = a y + b) for the whole line.
      lx.assign(cx + s, cx + s + L); ly.assign(cy + s, cy + s + L);
      s += L;
      double norm = 0;
      size_t grown_from = 0;      // first pixel the latest attempt added
      for (int attempt = 1;; attempt++) {
        if (attempt > 1) {
          // refit with the pixels the previous attempt added: their terms join the running (float) sums.  The parametrisation
          // test reads the line's first pixel (LeastSquaresLineFit_ :2733 / :2762).
          const bool h2 = M.horizontal(lx[0], ly[0]);
          const size_t n_new = lx.size() - grown_from;
          eq.add(NormalEq::of((h2 ? lx.data() : ly.data()) + grown_from, (h2 ? ly.data() : lx.data()) + grown_from, (int)n_new));
          eq.solve(a, b);
        }
        norm = 1 / std::sqrt(a * a + 1);
        grown_from = lx.size();
        int outliers = 0;
        while (e > s) {
          const double d = (hor ? std::fabs(a * cx[s] - cy[s] + b) : std::fabs(cx[s] - a * cy[s] - b)) * norm;
          lx.push_back(cx[s]); ly.push_back(cy[s]); s++;
          if (d > P.fit_err) { if (++outliers > P.max_outlier) break; }
          else outliers = 0;
        }
        lx.resize(lx.size() - outliers); ly.resize(ly.size() - outliers); s -= outliers;   // trailing outliers go back to the chain
        if (!(lx.size() > grown_from && attempt < P.try_time)) break;
      }
      double w[3];   // w0 x + w1 y + w2 = 0, w0^2 + w1^2 = 1
      if (hor) { w[0] = a * norm; w[1] = -1 * norm; w[2] = b * norm; }
      else { w[0] = 1 * norm; w[1] = -a * norm; w[2] = -b * norm; }
      float direction = 0;
      if (accept(w, direction)) {
        const double p1 = w[1] * w[1], p2 = w[0] * w[0], p3 = w[0] * w[1], p4 = w[2] * w[0], p5 = w[2] * w[1];
        Segment sg;
        unsigned X = lx.front(), Y = ly.front();
        sg.x1 = (float)(p1 * X - p3 * Y - p4); sg.y1 = (float)(p2 * Y - p3 * X - p5);
        X = lx.back(); Y = ly.back();
        sg.x2 = (float)(p1 * X - p3 * Y - p4); sg.y2 = (float)(p2 * Y - p3 * X - p5);
        sg.direction = direction;
        out.push_back(sg);
      }
    }
  }
};

// Resident scratch of a detector's line producer: device maps and pinned host copies for `cap_images` images of `cap_pixels` pixels
// (grows only).  A call used to pay three hipMalloc / hipFree pairs and five copies into pageable vectors.
struct LinesScratch {
  unsigned char* d_gray = nullptr; unsigned char* d_pk = nullptr;
  unsigned char* h_pin = nullptr;  // per image: the packed map, 3 bytes per pixel (+ 4 bytes of pad behind the last image)
  unsigned char* h_in = nullptr;   // pinned: the batch's images side by side (one upload)
  size_t cap = 0;                  // pixels x images
  cs::ChunkEvents chunks;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double device_ms = 0, host_ms = 0, total_ms = 0;
  int n_images = 0;
};
void lines_scratch_free(void* p) {
  LinesScratch* S = (LinesScratch*)p;
  if (!S) return;
  if (S->d_gray) (void)hipFree(S->d_gray);
  if (S->d_pk) (void)hipFree(S->d_pk);
  if (S->h_pin) (void)hipHostFree(S->h_pin);
  if (S->h_in) (void)hipHostFree(S->h_in);
  S->chunks.release();
  if (S->ev0) (void)hipEventDestroy(S->ev0);
  if (S->ev1) (void)hipEventDestroy(S->ev1);
  delete S;
}

// the sequential half of one image: routing, fitting, validation, end points (all of it independent of the other images)
int lines_host_stage(const Maps& M, const EdParams& P, double length_thres, float* lines4, int cap, int* n_lines) {
  const int img_w = M.W, img_h = M.H;
  const size_t N = (size_t)img_w * img_h;
  *n_lines = 0;
  // anchors in the reference's scan order: columns outermost (:1641-1643)
  Router R(M);
  Extractor X(M, P);
  std::vector<Segment> segs;
  Chain first, second, chain;
  // The reference visits the scan grid column by column; on a row-major map that is one cache miss per probe (117 k probes, 5 KB apart, at KITTI's
  // size: 0.34 of the stage's 0.48 ms).  The grid is read ROW by row instead and its anchors dealt to their columns (a counting sort, stable: a
  // column's anchors stay in row order) -- the list below IS the reference's visiting order.
  std::vector<unsigned> col_at((size_t)img_w + 1, 0u), grid_at;
  grid_at.reserve(4096);
  for (int y = 1; y < img_h - 1; y += P.scan) {
    const unsigned char* row = M.p3 + 3 * (size_t)y * img_w + 1;      // (the anchor flag: bit 3 of a pixel's second byte)
    for (int x = 1; x < img_w - 1; x += P.scan)
      if ((row[3 * x] >> 3) & 1) { grid_at.push_back((unsigned)((size_t)y * img_w + x)); col_at[(size_t)x + 1]++; }
  }
  if (grid_at.size() > N / 5) { cs_set_error_ba("cs_detect_lines_gray: more anchors than the reference's arrays hold"); return CS_ERR_CAPACITY; }
  for (int x = 0; x < img_w; x++) col_at[(size_t)x + 1] += col_at[x];
  std::vector<unsigned> anchors(grid_at.size());
  for (const unsigned at : grid_at) anchors[col_at[at % (unsigned)img_w]++] = at;
  for (const unsigned at_u : anchors) {
    {
      const size_t at = at_u;
      const int x = (int)(at_u % (unsigned)img_w), y = (int)(at_u / (unsigned)img_w);
      if (R.taken[at]) continue;
      first.x.clear(); first.y.clear(); second.x.clear(); second.y.clear();
      const bool hor = M.horizontal(at);
      R.walk(x, y, hor ? RIGHT : DOWN, first);
      R.taken[at] = 0;                                 // the anchor opens the second walk too
      R.walk(x, y, hor ? LEFT : UP, second);
      if ((int)(first.x.size() + second.x.size()) < P.min_len + 1) continue;     // too short: dropped, its pixels stay taken
      chain.x.assign(first.x.rbegin(), first.x.rend()); chain.y.assign(first.y.rbegin(), first.y.rend());
      chain.x.insert(chain.x.end(), second.x.begin() + 1, second.x.end()); chain.y.insert(chain.y.end(), second.y.begin() + 1, second.y.end());
      X.run(chain, segs);
    }
  }
  // OctaveKeyLines :862-875 (length), :1074-1143 (which end is the start), filter_lines: length > threshold
  int n = 0;
  for (const Segment& s : segs) {
    const float ddx = std::fabs(s.x1 - s.x2), ddy = std::fabs(s.y1 - s.y2);
    const float len = std::sqrt(ddx * ddx + ddy * ddy);
    if (!(len > (float)length_thres)) continue;
    const float ex = s.x2 - s.x1, ey = s.y2 - s.y1, dir = s.direction;
    bool flip = false;
    if (dir >= -0.75 * M_PI && dir < -0.25 * M_PI) flip = ey > 0;
    if (dir >= -0.25 * M_PI && dir < 0.25 * M_PI) flip = flip || ex < 0;
    if (dir >= 0.25 * M_PI && dir < 0.75 * M_PI) flip = flip || ey < 0;
    if ((dir >= 0.75 * M_PI && dir < M_PI) || (dir >= -M_PI && dir < -0.75 * M_PI)) flip = flip || ex > 0;
    if (n >= cap) { cs_set_error_ba("cs_detect_lines_gray: more segments than `cap`"); return CS_ERR_CAPACITY; }
    float* o = lines4 + 4 * (size_t)n;
    if (flip) { o[0] = s.x2; o[1] = s.y2; o[2] = s.x1; o[3] = s.y1; } else { o[0] = s.x1; o[1] = s.y1; o[2] = s.x2; o[3] = s.y2; }
    n++;
  }
  *n_lines = n;
  return CS_OK;
}

double ln_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

// n_images gray images of one size: the per-pixel stages of all of them on the device (one kernel per image, queued back to back, the
// maps copied into pinned staging), then the sequential halves side by side on the detector's worker pool.  lines4[i] receives image
// i's segments (cap rows of 4), n_lines[i] their number.
extern "C" int cs_detect_lines_batch(cs_detector* d, const unsigned char* const* grays, int n_images, int img_w, int img_h, double length_thres, float* const* lines4, int cap,
                                     int* n_lines) {
  if (!d || n_images < 0 || (n_images && (!grays || !n_lines || (cap && !lines4))) || img_w < 3 || img_h < 3 || cap < 0) return CS_ERR_INVALID_ARG;
  for (int i = 0; i < n_images; i++) if (!grays[i] || (cap && !lines4[i])) return CS_ERR_INVALID_ARG;
  try {
    for (int i = 0; i < n_images; i++) n_lines[i] = 0;
    if (n_images == 0) return CS_OK;
    std::lock_guard<std::mutex> lk(*(std::mutex*)cs_internal_detector_lines_mutex(d));
    LN_TRY(hipSetDevice(cs_internal_detector_device(d)));
    hipStream_t st = (hipStream_t)cs_internal_detector_stream(d);
    void** slot = cs_internal_detector_lines_slot(d, lines_scratch_free);
    if (!*slot) *slot = new LinesScratch();
    LinesScratch& S = *(LinesScratch*)*slot;
    const double t_begin = ln_now_ms();
    const EdParams P;
    const size_t N = (size_t)img_w * img_h, need = N * (size_t)n_images;
    if (need > S.cap) {
      if (S.d_gray) (void)hipFree(S.d_gray);
      if (S.d_pk) (void)hipFree(S.d_pk);
      if (S.h_pin) (void)hipHostFree(S.h_pin);
      if (S.h_in) (void)hipHostFree(S.h_in);
      S.d_gray = nullptr; S.d_pk = nullptr; S.h_pin = nullptr; S.h_in = nullptr; S.cap = 0;
      LN_TRY(hipMalloc((void**)&S.d_gray, need));
      LN_TRY(hipMalloc((void**)&S.d_pk, 3 * need + 4));
      LN_TRY(hipHostMalloc((void**)&S.h_pin, 3 * need + 4));
      LN_TRY(hipHostMalloc((void**)&S.h_in, need));
      S.cap = need;
    }
    if (!S.ev0) { LN_TRY(hipEventCreate(&S.ev0)); LN_TRY(hipEventCreate(&S.ev1)); }
    // getGaussianKernel(5, 1.0, CV_32F) rounded to 8-bit fixed point, as createSeparableLinearFilter does for 8-bit images
    int k[3];
    {
      float cf[5]; double sum = 0;
      for (int i = 0; i < 5; i++) { const double x = i - 2.0; cf[i] = (float)std::exp(-0.5 * x * x); sum += cf[i]; }
      sum = 1. / sum;
      for (int i = 0; i < 3; i++) k[i] = (int)std::nearbyint((double)(float)(cf[i] * sum) * 256.0);
    }
    const cs::LineMaps dm{S.d_pk};
    struct Ctx {
      const unsigned char* h_pk; int W, H; size_t N; const EdParams* P; double thr; float* const* lines4; int cap; int* n_lines; std::vector<int> rc;
      cs::ChunkGate gate; int device; const hipEvent_t* done; int n_chunks, n_images;
      const unsigned char* const* grays; unsigned char* h_in;
    } ctx{S.h_pin, img_w, img_h, N, &P, length_thres, lines4, cap, n_lines, std::vector<int>(n_images, 0), {}, cs_internal_detector_device(d), nullptr, 0, n_images, grays, S.h_in};
    auto one = [](int i, void* vp) {
      Ctx& c = *(Ctx*)vp;
      Maps M; M.W = c.W; M.H = c.H; M.grad_thr = c.P->grad_thr; M.p3 = c.h_pk + 3 * c.N * (size_t)i;
      try { c.rc[i] = lines_host_stage(M, *c.P, c.thr, c.lines4 ? c.lines4[i] : nullptr, c.cap, &c.n_lines[i]); }
      catch (const std::exception&) { c.rc[i] = CS_ERR_CAPACITY; }
    };
    double t_host;
    if (n_images == 1) {
      LN_TRY(hipMemcpyAsync(S.d_gray, grays[0], N, hipMemcpyHostToDevice, st));
      LN_TRY(hipEventRecord(S.ev0, st));
      cs::launch_lines_maps(S.d_gray, img_w, img_h, dm, k, P.grad_thr, P.anchor_thr, P.scan, st, 1);
      LN_TRY(hipGetLastError());
      LN_TRY(hipEventRecord(S.ev1, st));
      LN_TRY(hipMemcpyAsync(S.h_pin, S.d_pk, 3 * N, hipMemcpyDeviceToHost, st));
      LN_TRY(hipStreamSynchronize(st));
      float ms = 0;
      LN_TRY(hipEventElapsedTime(&ms, S.ev0, S.ev1));
      S.device_ms = ms;
      t_host = ln_now_ms();
      one(0, &ctx);
    } else {
      // a batch: the images gathered into pinned memory by the pool (the caller's buffers are pageable), one upload, then chunk by chunk
      // [kernel | copy back | event], all queued before the pool starts on the first chunk's images (batch_gate.h)
      cs_internal_detector_parallel(d, n_images, [](int i, void* vp) { Ctx& c = *(Ctx*)vp; std::memcpy(c.h_in + c.N * (size_t)i, c.grays[i], c.N); }, &ctx);
      const int CH = cs::BATCH_CHUNK, n_chunks = (n_images + CH - 1) / CH;
      LN_TRY(S.chunks.reserve(n_chunks));
      LN_TRY(hipMemcpyAsync(S.d_gray, S.h_in, N * (size_t)n_images, hipMemcpyHostToDevice, st));
      for (int c = 0; c < n_chunks; c++) {
        const int i0 = c * CH, ni = std::min(CH, n_images - i0);
        const cs::LineMaps mi{dm.p3 + 3 * (size_t)i0 * N};
        LN_TRY(hipEventRecord(S.chunks.k0[c], st));
        cs::launch_lines_maps(S.d_gray + (size_t)i0 * N, img_w, img_h, mi, k, P.grad_thr, P.anchor_thr, P.scan, st, ni);
        LN_TRY(hipGetLastError());
        LN_TRY(hipEventRecord(S.chunks.k1[c], st));
        LN_TRY(hipMemcpyAsync(S.h_pin + 3 * N * (size_t)i0, S.d_pk + 3 * N * (size_t)i0, 3 * N * (size_t)ni, hipMemcpyDeviceToHost, st));
        LN_TRY(hipEventRecord(S.chunks.done[c], st));
      }
      ctx.done = S.chunks.done.data(); ctx.n_chunks = n_chunks;
      t_host = ln_now_ms();
      cs_internal_detector_parallel_long(d, n_images + 1, [](int t, void* vp) {
        Ctx& c = *(Ctx*)vp;
        if (t == 0) { c.gate.watch(c.device, c.done, c.n_chunks, cs::BATCH_CHUNK, c.n_images); return; }
        const int i = t - 1;
        if (!c.gate.wait_for(i)) { c.rc[i] = CS_ERR_HIP; return; }
        Maps M; M.W = c.W; M.H = c.H; M.grad_thr = c.P->grad_thr; M.p3 = c.h_pk + 3 * c.N * (size_t)i;
        try { c.rc[i] = lines_host_stage(M, *c.P, c.thr, c.lines4 ? c.lines4[i] : nullptr, c.cap, &c.n_lines[i]); }
        catch (const std::exception&) { c.rc[i] = CS_ERR_CAPACITY; }
      }, &ctx);
      LN_TRY(hipStreamSynchronize(st));
      double dev = 0;
      for (int c = 0; c < n_chunks; c++) { float ms = 0; LN_TRY(hipEventElapsedTime(&ms, S.chunks.k0[c], S.chunks.k1[c])); dev += ms; }
      S.device_ms = dev;
    }
    S.host_ms = ln_now_ms() - t_host; S.total_ms = ln_now_ms() - t_begin; S.n_images = n_images;
    for (int r : ctx.rc) if (r) return r;
    return CS_OK;
  } catch (const std::exception& ex) {
    cs_set_error_ba(std::string("cs_detect_lines_batch: ") + ex.what());
    return CS_ERR_CAPACITY;
  }
}

// timing of the last cs_detect_lines_batch / cs_detect_lines_gray of this detector: the device kernels, the host stage (wall), the whole call
extern "C" int cs_detect_lines_last_timing(cs_detector* d, double* device_ms, double* host_ms, double* total_ms) {
  if (!d) return CS_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(*(std::mutex*)cs_internal_detector_lines_mutex(d));
  void** slot = cs_internal_detector_lines_slot(d, lines_scratch_free);
  const LinesScratch* S = (const LinesScratch*)*slot;
  if (!S) return CS_ERR_NOT_RUN;
  if (device_ms) *device_ms = S->device_ms;
  if (host_ms) *host_ms = S->host_ms;
  if (total_ms) *total_ms = S->total_ms;
  return CS_OK;
}

extern "C" int cs_detect_lines_gray(cs_detector* d, const unsigned char* gray, int img_w, int img_h, double length_thres, float* lines4, int cap, int* n_lines) {
  if (!d || !gray || img_w < 3 || img_h < 3 || cap < 0 || (cap && !lines4) || !n_lines) return CS_ERR_INVALID_ARG;
  float* l4 = lines4;
  return cs_detect_lines_batch(d, &gray, 1, img_w, img_h, length_thres, &l4, cap, n_lines);
}
