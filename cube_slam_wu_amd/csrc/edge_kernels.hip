// edge_kernels.hip -- the distance-map front end of detect_cuboid() on the device:
//   cv::Canny(gray_img(object_bbox), im_canny, 80, 200); cv::distanceTransform(255 - im_canny, dist_map, CV_DIST_L2, 3)
// (detect_3d_cuboid/src/box_proposal_detail.cpp:320-327).  OpenCV is a third-party dependency of the reference; the
// kernels implement its published algorithms (3x3 Sobel + L1 magnitude + 4-sector non-maximum suppression with the
// fixed-point tangent test + hysteresis as connected components; two-pass 3x3 chamfer in 16.16 fixed point), all in
// integer arithmetic (bit-reproducible on any host).  One workgroup per ROI.
#include <hip/hip_runtime.h>
#include "cs_hip_util.h"

#include <algorithm>
#include <cstdint>
#include <cstdlib>

namespace cs {

struct EdgeRoi {
  int l, t, w, h;            // ROI inside the gray image
  long long img_off;         // -> first pixel of the ROI's image in the gray pool
  long long cls_off;         // -> class bytes (w * h)
  long long map_off;         // -> output floats (w * h)
};

__device__ __forceinline__ int edge_px(const unsigned char* __restrict__ g, int W, int H, int x, int y) {
  x = min(max(x, 0), W - 1); y = min(max(y, 0), H - 1);   // BORDER_REPLICATE at the image border; the ROI is not isolated
  return g[(size_t)y * W + x];
}
__device__ __forceinline__ void edge_sobel(const unsigned char* __restrict__ g, int W, int H, int x, int y, int& gx, int& gy) {
  const int a = edge_px(g, W, H, x - 1, y - 1), b = edge_px(g, W, H, x, y - 1), c = edge_px(g, W, H, x + 1, y - 1);
  const int d = edge_px(g, W, H, x - 1, y), f = edge_px(g, W, H, x + 1, y);
  const int p = edge_px(g, W, H, x - 1, y + 1), q = edge_px(g, W, H, x, y + 1), r = edge_px(g, W, H, x + 1, y + 1);
  gx = (c + 2 * f + r) - (a + 2 * d + p);
  gy = (p + 2 * q + r) - (a + 2 * b + c);
}
// L1 gradient magnitude at ROI pixel (i, j); 0 outside the ROI (the magnitude buffer of cv::Canny is zero-padded)
__device__ __forceinline__ int edge_mag(const unsigned char* __restrict__ g, int W, int H, const EdgeRoi& R, int i, int j) {
  if (i < 0 || i >= R.h || j < 0 || j >= R.w) return 0;
  int gx, gy;
  edge_sobel(g, W, H, R.l + j, R.t + i, gx, gy);
  return abs(gx) + abs(gy);
}

// class per pixel: 0 = may belong to an edge, 1 = not an edge, 2 = edge.  Pass 1: one Sobel per pixel, L1 magnitude and
// the suppression sector packed into the (not yet used) output buffer; pass 2: non-maximum suppression + thresholds
// from the packed values; then hysteresis.  Waves walk rows, lanes walk columns (no divisions).
__device__ void edge_hyst_global(const struct EdgeRoi& R, unsigned char* cls, unsigned* pk, int n, int* n_front, int& changed);
enum { CANNY_BAND = 16, CANNY_TILE_W = 448 };   // LDS band of the fused Sobel + suppression: 18 x 450 x 2 bytes = 16 KB
__global__ __launch_bounds__(256) void edge_canny_kernel(const unsigned char* __restrict__ gray, int W, int H, const EdgeRoi* __restrict__ rois, unsigned char* cls_pool,
                                                         float* map_pool, int low, int high, int fuse_hyst) {
  __shared__ int n_front[2];
  __shared__ int changed;
  __shared__ unsigned short canny_lds[(CANNY_BAND + 2) * (CANNY_TILE_W + 2)];
  const EdgeRoi R = rois[blockIdx.x];
  gray += R.img_off;
  unsigned char* cls = cls_pool + R.cls_off;
  unsigned* pk = reinterpret_cast<unsigned*>(map_pool + R.map_off);   // magnitude (bits 0-15) | sector (16-17): 0 horizontal, 1 vertical, 2 diagonal same sign, 3 diagonal opposite sign
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int TG22 = 13573;
  if (R.w <= CANNY_TILE_W) {
    // ROIs up to CANNY_TILE_W columns: bands of CANNY_BAND rows through LDS.  The band's magnitudes (11 bits) and sectors (2 bits)
    // sit in LDS as 16-bit words with a zero frame (cv::Canny's zero-padded magnitude buffer), one halo row above and below;
    // the suppression reads its neighbours there.  Nothing but the class bytes goes to memory (the wide path below writes and
    // re-reads a 4-byte word per pixel: that traffic, 3 GB per 8000 ROIs through L2, was most of the kernel).
    // (gridDim.y > 1: a call of a few ROIs -- the bands are independent, each loads its own halo rows: workgroup y takes the bands
    // y, y + gridDim.y, ...; with one workgroup per ROI a KITTI frame's 8 boxes were 8 workgroups working through ~10 bands each, 198 us)
    const int ldw = R.w + 2;
    for (int b0 = (int)blockIdx.y * CANNY_BAND; b0 < R.h; b0 += CANNY_BAND * (int)gridDim.y) {
      const int rows = min(CANNY_BAND, R.h - b0);
      __syncthreads();                                   // the previous band's readers are done
      for (int r = wv; r < rows + 2; r += 4) {           // LDS row r = ROI row b0 - 1 + r
        const int i = b0 - 1 + r;
        for (int c = lane; c < ldw; c += 64) {           // LDS column c = ROI column c - 1
          const int j = c - 1;
          unsigned short pv = 0;
          if (i >= 0 && i < R.h && j >= 0 && j < R.w) {
            int xs, ys;
            edge_sobel(gray, W, H, R.l + j, R.t + i, xs, ys);
            const int x = abs(xs), y = abs(ys) << 15;
            const int tg22x = x * TG22;
            unsigned sector;
            if (y < tg22x) sector = 0;
            else if (y > tg22x + (x << 16)) sector = 1;
            else sector = ((xs ^ ys) < 0) ? 3u : 2u;
            pv = (unsigned short)((unsigned)(abs(xs) + abs(ys)) | (sector << 11));     // |gx| + |gy| <= 2040 < 2^11
          }
          canny_lds[r * ldw + c] = pv;
        }
      }
      __syncthreads();
      for (int r = wv; r < rows; r += 4) {
        const int i = b0 + r;
        const unsigned short* up = canny_lds + r * ldw, *mid = up + ldw, *dn = mid + ldw;   // LDS rows of ROI rows i - 1, i, i + 1
        for (int j = lane; j < R.w; j += 64) {
          const unsigned v = mid[j + 1];
          const int m = (int)(v & 0x7ffu);
          unsigned char c = 1;
          if (m > low) {
            const unsigned sector = v >> 11;
            bool keep;
            if (sector == 0) keep = m > (int)(mid[j] & 0x7ffu) && m >= (int)(mid[j + 2] & 0x7ffu);
            else if (sector == 1) keep = m > (int)(up[j + 1] & 0x7ffu) && m >= (int)(dn[j + 1] & 0x7ffu);
            else { const int sg = (sector == 3) ? -1 : 1; keep = m > (int)(up[j + 1 - sg] & 0x7ffu) && m > (int)(dn[j + 1 + sg] & 0x7ffu); }
            if (keep) c = (m > high) ? 2 : 0;
          }
          cls[i * R.w + j] = c;
        }
      }
    }
  } else {
  if (blockIdx.y) return;      // (the wide path is not split)
  for (int i = wv; i < R.h; i += 4)
    for (int j = lane; j < R.w; j += 64) {
      int xs, ys;
      edge_sobel(gray, W, H, R.l + j, R.t + i, xs, ys);
      const int x = abs(xs), y = abs(ys) << 15;
      const int tg22x = x * TG22;
      unsigned sector;
      if (y < tg22x) sector = 0;
      else if (y > tg22x + (x << 16)) sector = 1;
      else sector = ((xs ^ ys) < 0) ? 3u : 2u;
      pk[i * R.w + j] = (unsigned)(abs(xs) + abs(ys)) | (sector << 16);
    }
  __syncthreads();
  auto mag = [&](int i, int j) -> int { return (i < 0 || i >= R.h || j < 0 || j >= R.w) ? 0 : (int)(pk[i * R.w + j] & 0xffffu); };   // zero-padded like cv::Canny's buffer
  for (int i = wv; i < R.h; i += 4)
    for (int j = lane; j < R.w; j += 64) {
      const unsigned v = pk[i * R.w + j];
      const int m = (int)(v & 0xffffu);
      unsigned char c = 1;
      if (m > low) {
        const unsigned sector = v >> 16;
        bool keep;
        if (sector == 0) keep = m > mag(i, j - 1) && m >= mag(i, j + 1);
        else if (sector == 1) keep = m > mag(i - 1, j) && m >= mag(i + 1, j);
        else { const int s = (sector == 3) ? -1 : 1; keep = m > mag(i - 1, j - s) && m > mag(i + 1, j + s); }
        if (keep) c = (m > high) ? 2 : 0;
      }
      cls[i * R.w + j] = c;
    }
  }
  // large batches: the hysteresis right here on the class bytes in memory (the device is full of ROIs, what counts is how many of them
  // a CU holds); small calls run edge_hyst_kernel instead (LDS-resident, a third of the latency, a quarter of the workgroups per CU)
  if (fuse_hyst) {
    __syncthreads();
    edge_hyst_global(R, cls, pk, R.w * R.h, n_front, changed);
  }
}

// ---- round 6: Canny of a whole ROI in one workgroup with the hysteresis on bit planes in LDS (large batches) ---------------------------
// What the fused form above spends: nine clamped byte loads and ~110 instructions per pixel for the Sobel taps, and a breadth-first growth on
// the class bytes in memory (rounds of atomics and barriers: 0.8 of its 1.5 ms per 8000 ROIs).  Here
//   * a band's gray rows are staged in LDS once (clamped at the image border: BORDER_REPLICATE, the ROI is not isolated) and a lane computes
//     four neighbouring Sobel responses from six words of it (column sums t + 2m + b and differences b - t shared between the four);
//   * the suppression's classes leave as bytes (as before) AND as two bit planes in LDS -- strong (class 2) and weak (class 0), one bit per
//     pixel, a wave ballot per 64 pixels;
//   * the hysteresis is sweeps over the words of the weak plane: a word that still has weak pixels gathers the strong bits around it (its
//     own row and the rows above / below, shifted by one either way, with the neighbouring words' edge bits), and every RUN of weak pixels
//     that touches one joins at once -- the carry of an addition walks up a run, the same on the bit-reversed word walks down.  32 pixels
//     per instruction, no atomics (a word is written by its owner only; a reader that sees the new value merely joins a sweep early),
//     one barrier per sweep; the joined pixels' bytes are patched in memory as they join.
// The result is the set of weak pixels connected to a strong one (8-neighbourhood) -- the same fixed point whatever the order.
enum { CB_BAND = 8, HYST_WAVE_REPEATS = 8 };     // (bands of 8 rows: 16-row bands cost a workgroup per CU in LDS, 1.18 -> 1.08 ms per 8000 ROIs)
// e / d for small e (< 2^16) by a multiplication: inv = cb_inv(d)  (d = 1 has no 32-bit reciprocal of this form: the high product of e and
// 0xffffffff is e - 1 for e > 0, so it gets its own arm)
__device__ __forceinline__ unsigned cb_inv(int d) { return d <= 1 ? 0u : 0xffffffffu / (unsigned)d + 1u; }
__device__ __forceinline__ int cb_div(int e, unsigned inv) { return inv ? (int)__umulhi((unsigned)e, inv) : e; }
__device__ __forceinline__ int cb_ms_stride(int w) { return (w + 2 + 3) & ~3; }           // 16-bit words per row of the magnitude band
__device__ __forceinline__ int cb_gray_stride(int w) { return (w + 9 + 3) & ~3; }         // bytes per row of the gray band (ROI columns -2 ...)
__host__ __device__ __forceinline__ int cb_ms_words(int max_w) { return (((CB_BAND + 2) * ((max_w + 2 + 3) & ~3)) + 1) >> 1; }
__host__ __device__ __forceinline__ int cb_gray_words(int max_w) { return ((CB_BAND + 4) * ((max_w + 9 + 3) & ~3)) >> 2; }
__global__ __launch_bounds__(256) void edge_canny_bits_kernel(const unsigned char* __restrict__ gray, int W, int H, const EdgeRoi* __restrict__ rois, unsigned char* cls_pool, int low, int high,
                                                              int max_w) {
  extern __shared__ unsigned cb_lds[];
  const EdgeRoi R = rois[blockIdx.x];
  gray += R.img_off;
  unsigned char* cls = cls_pool + R.cls_off;
  const int w = R.w, h = R.h;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int MS = cb_ms_stride(w), GW = cb_gray_stride(w), Wd = (w + 31) >> 5;
  unsigned short* ms = reinterpret_cast<unsigned short*>(cb_lds);
  unsigned* gw32 = cb_lds + cb_ms_words(max_w);
  unsigned* Sp = gw32 + cb_gray_words(max_w);          // strong plane: row i of the ROI at words [(i + 1) Wd, (i + 2) Wd); rows 0 and h + 1 stay empty
  unsigned* Kp = Sp + (h + 2) * Wd;                    // weak plane
  const int TG22 = 13573;
  for (int e = tid; e < Wd; e += 256) { Sp[e] = 0; Sp[(h + 1) * Wd + e] = 0; Kp[e] = 0; Kp[(h + 1) * Wd + e] = 0; }
  const int GQ = GW >> 2, MQ = MS >> 2;
  const unsigned inv_gq = cb_inv(GQ), inv_mq = cb_inv(MQ);
  for (int b0 = 0; b0 < h; b0 += CB_BAND) {
    const int rows = min(CB_BAND, h - b0);
    __syncthreads();                                   // the previous band's readers are done
    // ---- gray rows b0 - 2 .. b0 + rows + 1, ROI columns -2 .. GW - 3, four bytes per lane
    for (int e = tid; e < (rows + 4) * GQ; e += 256) {
      const int g = cb_div(e, inv_gq), q = e - g * GQ;
      const int y = min(max(R.t + b0 - 2 + g, 0), H - 1);
      const unsigned char* src = gray + (size_t)y * W;
      const int x0 = R.l - 2 + 4 * q;
      const unsigned v = (unsigned)src[min(max(x0, 0), W - 1)] | ((unsigned)src[min(max(x0 + 1, 0), W - 1)] << 8) | ((unsigned)src[min(max(x0 + 2, 0), W - 1)] << 16) |
                         ((unsigned)src[min(max(x0 + 3, 0), W - 1)] << 24);
      gw32[g * GQ + q] = v;
    }
    __syncthreads();
    // ---- Sobel + L1 magnitude + sector of band rows -1 .. rows (ROI rows b0 - 1 + r), four columns per lane
    for (int e = tid; e < (rows + 2) * MQ; e += 256) {
      const int r = cb_div(e, inv_mq), q = e - r * MQ;
      const int i = b0 - 1 + r, c0 = 4 * q;            // magnitude columns c0 .. c0 + 3 = ROI columns c0 - 1 .. c0 + 2
      unsigned out[2] = {0u, 0u};
      if (i >= 0 && i < h) {
        const unsigned* g0 = gw32 + r * GQ + q;        // gray rows i - 1, i, i + 1 from byte column c0 on (ROI column c0 - 2)
        const unsigned t0 = g0[0], t1 = g0[1], m0 = g0[GQ], m1 = g0[GQ + 1], d0 = g0[2 * GQ], d1 = g0[2 * GQ + 1];
        int S6[6], D6[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const int tt = (int)(((k < 4 ? t0 : t1) >> (8 * (k & 3))) & 0xffu), mm = (int)(((k < 4 ? m0 : m1) >> (8 * (k & 3))) & 0xffu), bb = (int)(((k < 4 ? d0 : d1) >> (8 * (k & 3))) & 0xffu);
          S6[k] = tt + 2 * mm + bb; D6[k] = bb - tt;
        }
#pragma unroll
        for (int p = 0; p < 4; p++) {
          const int c = c0 + p;
          unsigned pv = 0;
          if (c >= 1 && c <= w) {
            const int xs = S6[p + 2] - S6[p], ys = D6[p] + 2 * D6[p + 1] + D6[p + 2];
            const int x = abs(xs), y = abs(ys) << 15;
            const int tg22x = x * TG22;
            unsigned sector;
            if (y < tg22x) sector = 0;
            else if (y > tg22x + (x << 16)) sector = 1;
            else sector = ((xs ^ ys) < 0) ? 3u : 2u;
            pv = (unsigned)(abs(xs) + abs(ys)) | (sector << 11);
          }
          out[p >> 1] |= pv << (16 * (p & 1));
        }
      }
      unsigned* dst = reinterpret_cast<unsigned*>(ms + r * MS + c0);
      dst[0] = out[0]; dst[1] = out[1];
    }
    __syncthreads();
    // ---- non-maximum suppression + thresholds: class bytes to memory, strong / weak bits to the planes
    for (int r = wv; r < rows; r += 4) {
      const int i = b0 + r;
      const unsigned short* mid = ms + (r + 1) * MS;      // band row r + 1 = ROI row i (row 0 of the band is the halo above)
      for (int j0 = 0; j0 < w; j0 += 64) {
        const int j = j0 + lane;
        unsigned char c = 1;
        if (j < w) {
          // branch-free: both neighbours of the pixel's sector are read whatever its magnitude (a wave always holds pixels of every sector,
          // so the three-way branch ran all its arms anyway)
          const unsigned v = mid[j + 1];
          const int m = (int)(v & 0x7ffu);
          const unsigned sector = v >> 11;
          const int step = sector == 0 ? 1 : (sector == 1 ? MS : (sector == 2 ? MS + 1 : MS - 1));      // towards the second neighbour; the first is opposite
          const int ma = (int)(mid[j + 1 - step] & 0x7ffu), mb = (int)(mid[j + 1 + step] & 0x7ffu);
          const bool keep = m > low && m > ma && (sector < 2 ? m >= mb : m > mb);
          c = keep ? (m > high ? 2 : 0) : 1;
        }
        const unsigned long long bs = __ballot(j < w && c == 2), bk = __ballot(j < w && c == 0);
        if (lane < 2) {
          const int kw = (j0 >> 5) + lane;
          if (kw < Wd) { Sp[(i + 1) * Wd + kw] = (unsigned)(bs >> (32 * lane)); Kp[(i + 1) * Wd + kw] = (unsigned)(bk >> (32 * lane)); }
        }
      }
    }
  }
  // ---- hysteresis on the planes
  const int nwords = h * Wd;
  const unsigned inv_wd = cb_inv(Wd);
  for (;;) {
    __syncthreads();
    int ch = 0;
    for (int e0 = 0; e0 < nwords; e0 += 256) {         // (wave-uniform trip count: the inner repeats vote)
      const int e = e0 + tid;
      const bool in = e < nwords;
      const int i = cb_div(in ? e : 0, inv_wd), kw = (in ? e : 0) - i * Wd;
      const int at = (i + 1) * Wd + kw;
      unsigned k = in ? Kp[at] : 0u;
      if (!__any(k != 0)) continue;
      const unsigned k_in = k;
      // a wave's 64 words are ~64 / Wd whole rows: repeating the step lets a chain cross several of them within one sweep (the wave
      // reads what its own lanes have just written: LDS operations of a wave stay in order)
      for (int rep = 0; rep < HYST_WAVE_REPEATS; rep++) {
        unsigned f = 0;
        if (k) {
          unsigned nb = 0, s_mid = 0;
#pragma unroll
          for (int dr = -1; dr <= 1; dr++) {
            const unsigned* rw = Sp + at + dr * Wd;
            const unsigned sv = rw[0];
            const unsigned l = kw > 0 ? rw[-1] >> 31 : 0u, rr = kw + 1 < Wd ? rw[1] << 31 : 0u;
            nb |= sv | (sv << 1) | (sv >> 1) | l | rr;
            if (dr == 0) s_mid = sv;
          }
          const unsigned jn = k & nb;
          if (jn) {
            f = ((k + jn) ^ k) | jn;                                      // up the runs: the carry walks from a touched bit to the run's end
            const unsigned kr = __brev(k), jr = __brev(jn);
            f |= __brev(((kr + jr) ^ kr) | jr);                           // ... and down
            f &= k;
            Sp[at] = s_mid | f;
            k &= ~f;
          }
        }
        if (!__any(f != 0)) break;
        ch = 1;
      }
      if (k != k_in) Kp[at] = k;
    }
    if (!__syncthreads_or(ch)) break;
  }
  // ---- what the distance transform needs: the strong plane, a row of ceil(w / 8) bytes per ROI row (bit j & 7 of byte j >> 3 = pixel j is an
  // edge) in the ROI's class-byte region -- an eighth of the class bytes, and neither the suppression nor the hysteresis writes a byte per pixel
  const int RS = (w + 7) >> 3;
  const unsigned inv_rs = cb_inv(RS);
  for (int e = tid; e < h * RS; e += 256) {
    const int i = cb_div(e, inv_rs), b = e - i * RS;
    cls[e] = (unsigned char)(Sp[(i + 1) * Wd + (b >> 2)] >> (8 * (b & 3)));
  }
}

// ---- hysteresis ----------------------------------------------------------------------------------------------------------------
// = the 8-connected components of the surviving pixels (class 0 or 2) that contain a strong one (class 2); every pixel of such a
// component becomes 2.  The result does not depend on the order the pixels are visited in.
//
// Path for ROIs too large for LDS: class bytes in memory, breadth-first growth with the frontier lists in the (free) map scratch.
__device__ void edge_hyst_global(const EdgeRoi& R, unsigned char* cls, unsigned* pk, int n, int* n_front, int& changed) {
  // breadth-first growth from the strong pixels: a frontier pixel claims its weak neighbours (atomic OR on the class word: exactly
  // one claimant) and they form the next frontier
  int* lists[2] = {reinterpret_cast<int*>(pk), reinterpret_cast<int*>(pk) + n / 2};
  const int cap = n / 2;
  bool overflow = false;
  if (threadIdx.x < 2) n_front[threadIdx.x] = 0;
  __syncthreads();
  for (int p = threadIdx.x; p < n; p += 256)
    if (cls[p] == 2) { int k = atomicAdd(&n_front[0], 1); if (k < cap) lists[0][k] = p; }
  __syncthreads();
  int curl = 0;
  for (int it = 0; it < n; it++) {
    const int ncur = n_front[curl];
    if (ncur > cap) overflow = true;
    if (ncur == 0) break;
    __syncthreads();
    if (threadIdx.x == 0) n_front[curl ^ 1] = 0;
    __syncthreads();
    const int* cur = lists[curl];
    int* nxt = lists[curl ^ 1];
    for (int q = threadIdx.x; q < min(ncur, cap); q += 256) {
      const int p = cur[q];
      const int i = p / R.w, j = p - i * R.w;
      for (int di = -1; di <= 1; di++)
        for (int dj = -1; dj <= 1; dj++) {
          const int a = i + di, b = j + dj;
          if (!(di | dj) || a < 0 || a >= R.h || b < 0 || b >= R.w) continue;
          unsigned char* cp = cls + a * R.w + b;
          if (*cp != 0) continue;
          const unsigned long long ad = reinterpret_cast<unsigned long long>(cp);
          const unsigned sh = (unsigned)(ad & 3) * 8;
          const unsigned old = atomicOr(reinterpret_cast<unsigned*>(ad & ~3ull), 2u << sh);
          if (((old >> sh) & 0xffu) == 0) { int k = atomicAdd(&n_front[curl ^ 1], 1); if (k < cap) nxt[k] = a * R.w + b; }
        }
    }
    __syncthreads();
    curl ^= 1;
  }
  if (!overflow) return;
  // a frontier outgrew its list (more than half of the ROI at once): finish by relaxation sweeps over all pixels
  for (int it = 0; it < n; it++) {
    __syncthreads();
    if (threadIdx.x == 0) changed = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < n; p += 256) {
      if (cls[p] != 0) continue;
      const int i = p / R.w, j = p - i * R.w;
      bool hit = false;
      for (int di = -1; di <= 1 && !hit; di++)
        for (int dj = -1; dj <= 1; dj++) {
          const int a = i + di, b = j + dj;
          if ((di | dj) && a >= 0 && a < R.h && b >= 0 && b < R.w && cls[a * R.w + b] == 2) { hit = true; break; }
        }
      if (hit) { cls[p] = 2; changed = 1; }
    }
    __syncthreads();
    if (!changed) break;
  }
}

// One workgroup per ROI, after edge_canny_kernel.  A chain of weak pixels is followed one pixel per round, so the rounds are a chain of
// dependent steps (tens to a few hundred per ROI) and with the class bytes and the frontier lists in memory every round cost two or
// three memory round trips: the per-ROI latency of the front end (0.5 ms of 0.9) and a third of its batch time.  Here the ROI's class
// bytes (<= lds_px of them) and both frontier lists live in LDS: a round is two workgroup barriers and LDS atomics, and a thread follows
// the chain it opens by itself, so rounds are spent on branches only.  Round one is a sweep (every weak pixel looks for a strong
// neighbour), so the many strong pixels never have to be listed; a frontier that outgrows its list falls back to relaxation sweeps, still
// in LDS.  Launched per size class (lo_px < pixels <= lds_px; take_rest: also the ROIs beyond every class, on the memory path).
enum { HYST_LIST = 1024, HYST_BULK_SWEEPS = 6 };
__device__ __forceinline__ int hyst_row_words(int w) { return (w + 2 + 3) >> 2; }              // a framed row, rounded up to whole words
__global__ __launch_bounds__(256) void edge_hyst_kernel(const EdgeRoi* __restrict__ rois, unsigned char* cls_pool, float* map_pool, int lo_px, int lds_px, int list_cap, int take_rest) {
  extern __shared__ unsigned hyst_lds[];      // [lds_px / 4 words: the ROI inside a frame of class 1, rows of SW words][2 x HYST_LIST byte indices]
  __shared__ int n_front[2];
  __shared__ int changed;
  const EdgeRoi R = rois[blockIdx.x];
  unsigned char* cls = cls_pool + R.cls_off;
  const int n = R.w * R.h;
  const int SW = hyst_row_words(R.w), SB = 4 * SW, nwords = SW * (R.h + 2);      // pixel (i, j) at byte (i + 1) SB + j + 1
  const bool fits = 4 * nwords <= lds_px;
  if (4 * nwords <= lo_px || (!fits && !take_rest)) return;   // another launch's size class
  if (!fits) { edge_hyst_global(R, cls, reinterpret_cast<unsigned*>(map_pool + R.map_off), n, n_front, changed); return; }
  unsigned char* c8 = reinterpret_cast<unsigned char*>(hyst_lds);
  int* lists[2] = {reinterpret_cast<int*>(hyst_lds + lds_px / 4), reinterpret_cast<int*>(hyst_lds + lds_px / 4) + HYST_LIST};
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < nwords; e += 256) hyst_lds[e] = 0x01010101u;
  if (threadIdx.x < 2) n_front[threadIdx.x] = 0;
  if (threadIdx.x == 0) changed = 0;
  __syncthreads();
  {
    // the class bytes, a wave per row, 64 columns per step, eight independent loads in flight per thread
    const int chunks = (R.w + 63) >> 6, rows = (R.h - wv + 3) >> 2, T = rows * chunks;
    for (int t0 = 0; t0 < T; t0 += 8) {
      unsigned char vb[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int t = t0 + u, rr = t / chunks, i = wv + 4 * rr, j = lane + 64 * (t - rr * chunks);
        vb[u] = (t < T && j < R.w) ? cls[i * R.w + j] : (unsigned char)1;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int t = t0 + u, rr = t / chunks, i = wv + 4 * rr, j = lane + 64 * (t - rr * chunks);
        if (t < T && j < R.w) c8[(i + 1) * SB + j + 1] = vb[u];
      }
    }
  }
  __syncthreads();
  // Classes in LDS: 0 weak, 1 not an edge, 2 strong, 3 = a weak pixel that joined here (bit 1 = "edge"); only the 3s go back to memory.
  //
  // Bulk: sweeps, a word (4 pixels) per lane per step, all in bit operations on nine words -- the "edge" bits of the word, its two
  // neighbours in the row and the same three of the rows above and below, spread one byte left and right; a zero byte with an edge bit
  // around it joins.  No divergence, no atomics (a word is written by one thread; a neighbour that sees the new value already only joins
  // a sweep early).  The last sweep queues what it joined.
  auto sweep = [&](bool queue) {
    int joined = 0;
    for (int r = 1 + wv; r <= R.h; r += 4)
      for (int kw = lane; kw < SW; kw += 64) {
        const unsigned* row = hyst_lds + r * SW;
        const unsigned wd = row[kw];
        const unsigned z = ~(wd | (wd >> 1)) & 0x01010101u;                      // 1 in the zero bytes
        if (!z) continue;
        unsigned nbr = 0;
#pragma unroll
        for (int dr = -1; dr <= 1; dr++) {
          const unsigned* rw = row + dr * SW;
          const unsigned c = (rw[kw] >> 1) & 0x01010101u;
          const unsigned l = kw > 0 ? (rw[kw - 1] >> 25) & 1u : 0u, rr = kw + 1 < SW ? (rw[kw + 1] >> 1) & 1u : 0u;
          const unsigned side = (c << 8) | l | (c >> 8) | (rr << 24);
          nbr |= dr ? (side | c) : side;
        }
        const unsigned join = z & nbr;
        if (!join) continue;
        hyst_lds[r * SW + kw] = wd | (join * 3u);
        joined += __popc(join);
        if (queue) {
#pragma unroll
          for (int q = 0; q < 4; q++)
            if ((join >> (8 * q)) & 1u) { const int kk = atomicAdd(&n_front[0], 1); if (kk < list_cap) lists[0][kk] = 4 * (r * SW + kw) + q; }
        }
      }
    return joined;
  };
  // a few bulk sweeps while they still join many pixels, then one that queues its (few) pixels for the chain following below
  for (int it = 0; it < HYST_BULK_SWEEPS; it++) {
    const int j = sweep(false);
    if (j) atomicAdd(&changed, j);
    __syncthreads();
    const int total = changed;
    __syncthreads();
    if (threadIdx.x == 0) changed = 0;
    __syncthreads();
    if (total <= list_cap / 4) break;
  }
  sweep(true);
  __syncthreads();
  const int off[8] = {-SB - 1, -SB, -SB + 1, -1, 1, SB - 1, SB, SB + 1};
  bool overflow = n_front[0] > list_cap;
  int curl = 0;
  while (!overflow) {
    const int ncur = n_front[curl];
    if (ncur == 0) break;
    __syncthreads();
    if (threadIdx.x == 0) n_front[curl ^ 1] = 0;
    __syncthreads();
    const int* cur = lists[curl];
    int* nxt = lists[curl ^ 1];
    for (int q = threadIdx.x; q < ncur; q += 256) {
      // a thread follows the chain it opens: the first neighbour it claims is its next pixel, further ones (branches) go to the next
      // frontier.  The eight neighbour bytes are requested together (the frame makes every address valid): one LDS round trip per pixel.
      int at = cur[q];
      while (at >= 0) {
        unsigned char nb[8];
#pragma unroll
        for (int k = 0; k < 8; k++) nb[k] = c8[at + off[k]];
        int follow = -1;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          if (nb[k] != 0) continue;
          const int a2 = at + off[k];
          const unsigned sh = (unsigned)(a2 & 3) * 8;
          const unsigned old = atomicOr(&hyst_lds[a2 >> 2], 3u << sh);       // exactly one claimant per pixel
          if (((old >> sh) & 0xffu) != 0) continue;
          if (follow < 0) follow = a2;
          else { const int kk = atomicAdd(&n_front[curl ^ 1], 1); if (kk < list_cap) nxt[kk] = a2; }
        }
        at = follow;
      }
    }
    __syncthreads();
    curl ^= 1;
    overflow = n_front[curl] > list_cap;
  }
  if (overflow) {
    // (entries beyond the list were claimed but not queued: what they would have reached is found by sweeping)
    for (int it = 0; it < n; it++) {
      __syncthreads();
      if (threadIdx.x == 0) changed = 0;
      __syncthreads();
      if (sweep(false)) changed = 1;
      __syncthreads();
      if (!changed) break;
    }
  }
  __syncthreads();
  for (int r = 1 + wv; r <= R.h; r += 4)
    for (int kw = lane; kw < SW; kw += 64) {
      const unsigned wd = hyst_lds[r * SW + kw];
      if (!(wd & (wd >> 1) & 0x01010101u)) continue;                    // no byte with both low bits set
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int j = 4 * kw + q - 1;
        if (((wd >> (8 * q)) & 3u) == 3u && j >= 0 && j < R.w) cls[(r - 1) * R.w + j] = 2;
      }
    }
}

// inclusive prefix minimum over the 64 lanes of a wave, on DPP: shifts by 1, 2, 4, 8 inside each row of 16 lanes (a lane without
// a source keeps the identity), then lane 15 of rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2 / 3.  Six VALU
// instructions with their data path in the ALU -- the rows of the distance transform are a dependent chain of these scans,
// and the shuffle version went through the LDS crossbar six times per scan.
__device__ __forceinline__ int wave_prefix_min(int v) {
  const int ID = 0x7fffffff;
  v = min(v, __builtin_amdgcn_update_dpp(ID, v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = min(v, __builtin_amdgcn_update_dpp(ID, v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = min(v, __builtin_amdgcn_update_dpp(ID, v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = min(v, __builtin_amdgcn_update_dpp(ID, v, 0x118, 0xf, 0xf, false));   // row_shr:8
  v = min(v, __builtin_amdgcn_update_dpp(ID, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
  v = min(v, __builtin_amdgcn_update_dpp(ID, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
  return v;
}

// two-pass 3x3 chamfer distance (distanceTransform(255 - canny, DIST_L2, 3)), ONE WAVE per ROI: the rows are a
// dependent chain, the batch supplies the parallelism (thousands of ROIs), and a wave needs no barriers.  The
// recurrence along a row, d[j] = min(t[j], d[j-1] + a), is a prefix minimum of t[k] - k a.
// Working values: OpenCV's are 16.16 fixed point in unsigned 32 bits, saturated at DIST_MAX = UINT_MAX - b ("no feature
// reachable").  Every finite distance inside an image is below 2^30, so the kernel carries 2^30 for "unreachable" in
// signed 32 bits (nothing overflows, sums stay ordered) and writes DIST_MAX's float for it; finite values are identical.
#ifndef CS_DT_AHEAD
#define CS_DT_AHEAD 4
#endif
enum { DT_CHUNKS = 6, DT_AHEAD = CS_DT_AHEAD };   // ROI widths up to 384 columns take the register-pipelined instances, 4 rows of operands in flight
struct DtArgs { const unsigned char* cls; int* tmp; int* row_lds; int row_cap, w, h; };
// pixel (row, column jc) is an edge: class byte 2, or -- BITS: the strong plane edge_canny_bits_kernel leaves, ceil(w / 8) bytes per row -- its bit
template <bool BITS>
__device__ __forceinline__ int dt_class(const unsigned char* __restrict__ cls, int w, int row, int jc) {
  if (BITS) return (int)((cls[row * ((w + 7) >> 3) + (jc >> 3)] >> (jc & 7)) & 1u) << 1;
  return cls[row * w + jc];
}

// The two passes for a ROI of exactly NC chunks of 64 columns.  Per row: the global operands (class bytes on the way down, the
// forward distances on the way up) were requested one row ahead; the NC scans are independent of each other and of the
// carry, which only enters the last minimum -- so a row costs one LDS round trip (neighbour row), one scan latency and NC
// carry steps, not NC times all of it.
template <int NC, bool BITS>
__device__ __forceinline__ void edge_dt_rows(const DtArgs& A) {
  const int HV = 62587, DIAG = 89738, INF = 1 << 30;
  const float scale = 1.f / (1 << 16);
  const float far = (float)(0xffffffffu - 89738u) * scale;     // what OpenCV writes where no feature is reachable
  const int w = A.w, h = A.h, lane = threadIdx.x;
  const unsigned char* cls = A.cls;
  int* tmp = A.tmp;
  int* nb = A.row_lds + 1;                      // nb[-1 .. w]: the neighbour row
  int* cur = A.row_lds + (A.row_cap + 2) + 1;   // the row being produced (becomes the neighbour row of the next one)
  constexpr int D = DT_AHEAD;                   // rows whose global operands are in flight ahead of the row being computed
  int pre[D][NC];
  // ---- forward: top-left to bottom-right
  for (int j = lane; j < 2 * (A.row_cap + 2); j += 64) A.row_lds[j] = INF;
#pragma unroll
  for (int u = 0; u < D; u++)
#pragma unroll
    for (int c = 0; c < NC; c++) { const int j = c * 64 + lane; pre[u][c] = (u < h) ? dt_class<BITS>(cls, w, u, min(j, w - 1)) : 0; }
  for (int i0 = 0; i0 < h; i0 += D) {
#pragma unroll
    for (int u = 0; u < D; u++) {
      const int i = i0 + u;
      if (i < h) {
        int now[NC], d[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) now[c] = pre[u][c];
        // (branch-free up to the stores: lanes past the ROI's last column read its last column's cells and are masked by a select --
        // with `if (j < w)` around the loads every chunk of every row cost several EXEC-mask branches, most of a row's latency)
        if (i + D < h) {
#pragma unroll
          for (int c = 0; c < NC; c++) { const int j = c * 64 + lane; pre[u][c] = dt_class<BITS>(cls, w, i + D, min(j, w - 1)); }
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
          const int j = c * 64 + lane, jc = min(j, w - 1);
          int t = min(min(nb[jc - 1] + DIAG, nb[jc] + HV), nb[jc + 1] + DIAG);
          t = (now[c] == 2) ? 0 : t;
          const int a = (j < w) ? t - j * HV : 0x7fffffff;
          const int s = wave_prefix_min(a);
          d[c] = (s == 0x7fffffff) ? INF : s + j * HV;
        }
        int carry = INF;                            // d at the column left of the running chunk
#pragma unroll
        for (int c = 0; c < NC; c++) {
          const int j0 = c * 64, j = j0 + lane;
          const int v = min(min(d[c], carry + (lane + 1) * HV), INF);
          if (j < w) { tmp[i * w + j] = v; cur[j] = v; }
          carry = __builtin_amdgcn_readlane(v, min(63, max(0, w - 1 - j0)));
        }
        int* sw = nb; nb = cur; cur = sw;           // the row just produced becomes the neighbour row (border cells stay INF)
      }
    }
  }
  __syncthreads();   // (one wave) the forward values written by other lanes are read back below
  // ---- backward: bottom-right to top-left (mirrored scan), writing the floats
  for (int j = lane; j < 2 * (A.row_cap + 2); j += 64) A.row_lds[j] = INF;
#pragma unroll
  for (int u = 0; u < D; u++)
#pragma unroll
    for (int c = 0; c < NC; c++) { const int jr = c * 64 + 63 - lane; pre[u][c] = (h - 1 - u >= 0) ? tmp[(h - 1 - u) * w + min(jr, w - 1)] : 0; }
  for (int i0 = h - 1; i0 >= 0; i0 -= D) {
#pragma unroll
    for (int u = 0; u < D; u++) {
      const int i = i0 - u;
      if (i >= 0) {
        int now[NC], d[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) now[c] = pre[u][c];
        if (i - D >= 0) {
#pragma unroll
          for (int c = 0; c < NC; c++) { const int jr = c * 64 + 63 - lane; pre[u][c] = tmp[(i - D) * w + min(jr, w - 1)]; }
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
          const int jr = c * 64 + 63 - lane, jc = min(jr, w - 1);        // lane 0 takes the chunk's rightmost column: scan order = right to left
          const int t0 = min(min(now[c], nb[jc + 1] + DIAG), min(nb[jc] + HV, nb[jc - 1] + DIAG));
          const int a = (jr < w) ? t0 + jr * HV : 0x7fffffff;             // d[j] = min over k >= j of u[k] + (k - j) HV
          const int s = wave_prefix_min(a);
          d[c] = (s == 0x7fffffff) ? INF : s - jr * HV;
        }
        int carry = INF;                            // d at the column right of the running chunk
#pragma unroll
        for (int cc = 0; cc < NC; cc++) {
          const int c = NC - 1 - cc, j0 = c * 64, jr = j0 + 63 - lane;
          const int chunk_right = min(w - 1, j0 + 63);
          const int v = min(min(d[c], carry + (chunk_right - jr + 1) * HV), INF);
          if (jr < w) { cur[jr] = v; reinterpret_cast<float*>(tmp)[i * w + jr] = (v >= INF) ? far : (float)(unsigned)v * scale; }
          carry = __builtin_amdgcn_readlane(v, 63);   // the chunk's leftmost column
        }
        int* sw = nb; nb = cur; cur = sw;
      }
    }
  }
}

// any width: the plain loop over chunks (no prefetch, chunk after chunk)
template <bool BITS>
__device__ __forceinline__ void edge_dt_rows_wide(const DtArgs& A) {
  const int HV = 62587, DIAG = 89738, INF = 1 << 30;
  const float scale = 1.f / (1 << 16);
  const float far = (float)(0xffffffffu - 89738u) * scale;
  const int w = A.w, h = A.h, lane = threadIdx.x;
  const unsigned char* cls = A.cls;
  int* tmp = A.tmp;
  int* nb = A.row_lds + 1;
  int* cur = A.row_lds + (A.row_cap + 2) + 1;
  for (int j = lane; j < 2 * (A.row_cap + 2); j += 64) A.row_lds[j] = INF;
  for (int i = 0; i < h; i++) {
    int carry = INF;
    for (int j0 = 0; j0 < w; j0 += 64) {
      const int j = j0 + lane;
      int a = 0x7fffffff;
      if (j < w) {
        int t = 0;
        if (dt_class<BITS>(cls, w, i, j) != 2) t = min(min(nb[j - 1] + DIAG, nb[j] + HV), nb[j + 1] + DIAG);
        a = t - j * HV;
      }
      int d = wave_prefix_min(a);
      d = (d == 0x7fffffff) ? INF : d + j * HV;
      d = min(min(d, carry + (lane + 1) * HV), INF);
      if (j < w) { tmp[i * w + j] = d; cur[j] = d; }
      carry = __builtin_amdgcn_readlane(d, min(63, w - 1 - j0));
    }
    int* sw = nb; nb = cur; cur = sw;
  }
  __syncthreads();
  for (int j = lane; j < 2 * (A.row_cap + 2); j += 64) A.row_lds[j] = INF;
  const int nchunk = (w + 63) / 64;
  for (int i = h - 1; i >= 0; i--) {
    int carry = INF;
    for (int cch = nchunk - 1; cch >= 0; cch--) {
      const int j0 = cch * 64;
      const int jr = j0 + 63 - lane;
      int a = 0x7fffffff;
      if (jr < w) {
        const int t0 = min(min(tmp[i * w + jr], nb[jr + 1] + DIAG), min(nb[jr] + HV, nb[jr - 1] + DIAG));
        a = t0 + jr * HV;
      }
      int d = wave_prefix_min(a);
      d = (d == 0x7fffffff) ? INF : d - jr * HV;
      const int chunk_right = min(w - 1, j0 + 63);
      d = min(min(d, carry + (chunk_right - jr + 1) * HV), INF);
      if (jr < w) { cur[jr] = d; reinterpret_cast<float*>(tmp)[i * w + jr] = (d >= INF) ? far : (float)(unsigned)d * scale; }
      carry = __builtin_amdgcn_readlane(d, 63);
    }
    int* sw = nb; nb = cur; cur = sw;
  }
}

template <bool BITS>
__device__ __forceinline__ void edge_dt_dispatch(const DtArgs& A, int w) {
  switch ((w + 63) / 64) {
    case 0: break;
    case 1: edge_dt_rows<1, BITS>(A); break;
    case 2: edge_dt_rows<2, BITS>(A); break;
    case 3: edge_dt_rows<3, BITS>(A); break;
    case 4: edge_dt_rows<4, BITS>(A); break;
    case 5: edge_dt_rows<5, BITS>(A); break;
    case 6: edge_dt_rows<6, BITS>(A); break;
    default: edge_dt_rows_wide<BITS>(A); break;
  }
}
__global__ __launch_bounds__(64) void edge_dt_kernel(const EdgeRoi* __restrict__ rois, const unsigned char* __restrict__ cls_pool, float* map_pool, int row_cap, int bits) {
  extern __shared__ int row_lds_i[];        // two rows of row_cap + 2 values (border cells at both ends): neighbour row, current row
  const EdgeRoi R = rois[blockIdx.x];
  const DtArgs A{cls_pool + R.cls_off, reinterpret_cast<int*>(map_pool + R.map_off), row_lds_i, row_cap, R.w, R.h};
  if (bits) edge_dt_dispatch<true>(A, R.w); else edge_dt_dispatch<false>(A, R.w);
}

// A handful of tables between PINNED host memory and device memory in ONE launch (either direction; the host side is addressed through the
// unified address space).  Why not one hipMemcpyAsync each: ten small copies are ten trips through a copy-engine ring (~0.3 ms of latency in
// front of a sweep, ~0.15 ms behind it), and a ring that holds a bulk upload (cs_batch_refill_gray) makes every one of them wait for it.
struct CopySeg { const void* src; void* dst; unsigned long long bytes; };
struct CopySegs { CopySeg s[16]; int n; };
__global__ __launch_bounds__(256) void multi_copy_kernel(CopySegs segs) {
  const CopySeg sg = segs.s[blockIdx.y];
  const size_t n16 = ((reinterpret_cast<uintptr_t>(sg.src) | reinterpret_cast<uintptr_t>(sg.dst)) & 15) ? 0 : sg.bytes / 16;
  const uint4* s16 = static_cast<const uint4*>(sg.src);
  uint4* d16 = static_cast<uint4*>(sg.dst);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) d16[i] = s16[i];
  const unsigned char* sb = static_cast<const unsigned char*>(sg.src);
  unsigned char* db = static_cast<unsigned char*>(sg.dst);
  for (size_t i = 16 * n16 + (size_t)blockIdx.x * 256 + threadIdx.x; i < sg.bytes; i += (size_t)gridDim.x * 256) db[i] = sb[i];
}
void launch_multi_copy(const CopySegs& segs, hipStream_t st) {
  if (segs.n <= 0) return;
  unsigned long long mx = 0;
  for (int i = 0; i < segs.n; i++) mx = segs.s[i].bytes > mx ? segs.s[i].bytes : mx;
  const int bx = (int)std::min<unsigned long long>(32, std::max<unsigned long long>(1, mx / (16 * 256 * 4)));
  hipLaunchKernelGGL(multi_copy_kernel, dim3(bx, segs.n), dim3(256), 0, st, segs);
}
// (Tried in round 6: the batch as four chunks of ROIs, the distance transform of chunk c on a second stream beside the Canny of chunk c + 1 --
// a latency chain beside an issue-bound kernel.  Side by side both crawl: 2.44 ms per 8000 ROIs against 1.84 one after the other.)
void launch_edge_maps(const unsigned char* gray, int W, int H, const EdgeRoi* rois, int n_rois, unsigned char* cls_pool, float* map_pool, int max_w, long long max_px, int low, int high,
                      hipStream_t st) {   // max_px: the largest framed size of an ROI, 4 ceil((w + 2) / 4) (h + 2)
  if (n_rois <= 0) return;
  // CS_EDGE_HYST (tests / measurements): "lds", "fused" or "bits" forces one hysteresis path; by default calls that cannot fill the device take the LDS kernel
  static const int force = [] { const char* e = getenv("CS_EDGE_HYST"); return !e ? 0 : (e[0] == 'l' ? 1 : (e[0] == 'f' ? 2 : (e[0] == 'b' ? 3 : 0))); }();
  const bool fused = force ? force >= 2 : n_rois > 1024;
  // large batches whose ROIs fit: the whole Canny of an ROI in one workgroup, hysteresis on bit planes in LDS (edge_canny_bits_kernel).
  // LDS: the magnitude band, the gray band and two planes of (h + 2) ceil(w / 32) words -- bounded through the largest framed ROI
  // (max_px = 4 ceil((w + 2) / 4) (h + 2) >= w (h + 2)): w (h + 2) / 32 + (h + 2) words each.  CS_EDGE_HYST=fused keeps the memory form (tests).
  const size_t bits_lds = 4 * ((size_t)cb_ms_words(max_w) + (size_t)cb_gray_words(max_w) + 2 * ((size_t)max_px / 32 + (size_t)H + 2 + 8));
  if ((force == 3 || (!force && fused)) && max_w <= CANNY_TILE_W && bits_lds <= 64 * 1024) {
    const size_t dt_lds = 2 * (size_t)(max_w + 2) * sizeof(unsigned);
    hipLaunchKernelGGL(edge_canny_bits_kernel, dim3(n_rois), dim3(256), bits_lds, st, gray, W, H, rois, cls_pool, low, high, max_w);
    hipLaunchKernelGGL(edge_dt_kernel, dim3(n_rois), dim3(64), dt_lds, st, rois, cls_pool, map_pool, max_w, 1);
    return;
  }
  hipLaunchKernelGGL(edge_canny_kernel, dim3(n_rois, (!fused && n_rois <= 64) ? 8 : 1), dim3(256), 0, st, gray, W, H, rois, cls_pool, map_pool, low, high, fused ? 1 : 0);
  if (!fused) {
    // the ROI's class bytes in LDS: sized for the largest ROI of the call, at most 64 KB (two workgroups per CU with the lists; larger ROIs take
    // the memory path inside the kernel).  CS_EDGE_HYST_LIST (tests): a smaller frontier capacity, CS_EDGE_HYST_LDS: the cap in bytes.
    static const int list_cap = [] { const char* e = getenv("CS_EDGE_HYST_LIST"); const int v = e ? atoi(e) : HYST_LIST; return std::min(std::max(v, 1), (int)HYST_LIST); }();
    static const long long lds_cap = [] { const char* e = getenv("CS_EDGE_HYST_LDS"); const long long v = e ? atoll(e) : 64 * 1024; return std::min<long long>(std::max<long long>(v, 8), 96 * 1024); }();
    static DynLdsOnce hyst_lds_once;     // per device (cs_hip_util.h)
    const bool big_lds_ok = hyst_lds_once.set(reinterpret_cast<const void*>(edge_hyst_kernel), 96 * 1024 + 2 * HYST_LIST * (int)sizeof(int));
    // two size classes, so that ordinary ROIs do not reserve the LDS of the largest one (4 workgroups per CU under 32 KB)
    // (refused: stay inside the default 64 KB -- larger ROIs take the memory path inside the kernel)
    const long long cap_px = big_lds_ok ? lds_cap : std::min<long long>(lds_cap, 64 * 1024 - 2 * HYST_LIST * (long long)sizeof(int));
    const int big_px = (int)((std::min<long long>(std::max<long long>(max_px, 8), cap_px) + 3) & ~3LL);
    // (a call of a few ROIs -- one frame -- cannot fill the device either way: ONE launch sized for its largest ROI instead of two in a row,
    // 85 + 206 us for a KITTI frame's 8 boxes)
    const int small_px = n_rois <= 64 ? big_px : std::min(big_px, 32 * 1024);
    hipLaunchKernelGGL(edge_hyst_kernel, dim3(n_rois), dim3(256), (size_t)small_px + 2 * HYST_LIST * sizeof(int), st, rois, cls_pool, map_pool, 0, small_px, list_cap, max_px > small_px ? 0 : 1);
    if (max_px > small_px)
      hipLaunchKernelGGL(edge_hyst_kernel, dim3(n_rois), dim3(256), (size_t)big_px + 2 * HYST_LIST * sizeof(int), st, rois, cls_pool, map_pool, small_px, big_px, list_cap, 1);
  }
  hipLaunchKernelGGL(edge_dt_kernel, dim3(n_rois), dim3(64), 2 * (size_t)(max_w + 2) * sizeof(unsigned), st, rois, cls_pool, map_pool, max_w, 0);
}

}  // namespace cs
