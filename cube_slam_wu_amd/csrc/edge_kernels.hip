// edge_kernels.hip -- the distance-map front end of detect_cuboid() on the device:
//   cv::Canny(gray_img(object_bbox), im_canny, 80, 200); cv::distanceTransform(255 - im_canny, dist_map, CV_DIST_L2, 3)
// (detect_3d_cuboid/src/box_proposal_detail.cpp:320-327).  OpenCV is a third-party dependency of the reference; the
// kernels implement its published algorithms (3x3 Sobel + L1 magnitude + 4-sector non-maximum suppression with the
// fixed-point tangent test + hysteresis as connected components; two-pass 3x3 chamfer in 16.16 fixed point), all in
// integer arithmetic (bit-reproducible on any host).  One workgroup per ROI.
#include <hip/hip_runtime.h>

#include <cstdint>

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

// class per pixel: 0 = may belong to an edge, 1 = not an edge, 2 = edge; then hysteresis; then 0 / 255
__global__ __launch_bounds__(256) void edge_canny_kernel(const unsigned char* __restrict__ gray, int W, int H, const EdgeRoi* __restrict__ rois, unsigned char* cls_pool,
                                                         int low, int high) {
  enum { cand_cap = 8192 };           // weak pixels listed in LDS; a ROI with more of them sweeps all its pixels instead
  __shared__ int cand[cand_cap];
  __shared__ int n_cand;
  __shared__ int changed;
  const EdgeRoi R = rois[blockIdx.x];
  gray += R.img_off;
  unsigned char* cls = cls_pool + R.cls_off;
  const int n = R.w * R.h;
  if (threadIdx.x == 0) n_cand = 0;
  __syncthreads();
  const int TG22 = 13573;
  for (int p = threadIdx.x; p < n; p += 256) {
    const int i = p / R.w, j = p - i * R.w;
    int xs, ys;
    edge_sobel(gray, W, H, R.l + j, R.t + i, xs, ys);
    const int m = abs(xs) + abs(ys);
    unsigned char c = 1;
    if (m > low) {
      const int x = abs(xs), y = abs(ys) << 15;
      const int tg22x = x * TG22;
      bool keep;
      if (y < tg22x) keep = m > edge_mag(gray, W, H, R, i, j - 1) && m >= edge_mag(gray, W, H, R, i, j + 1);
      else {
        const int tg67x = tg22x + (x << 16);
        if (y > tg67x) keep = m > edge_mag(gray, W, H, R, i - 1, j) && m >= edge_mag(gray, W, H, R, i + 1, j);
        else { const int s = (xs ^ ys) < 0 ? -1 : 1; keep = m > edge_mag(gray, W, H, R, i - 1, j - s) && m > edge_mag(gray, W, H, R, i + 1, j + s); }
      }
      if (keep) {
        c = (m > high) ? 2 : 0;
        if (c == 0) { int k = atomicAdd(&n_cand, 1); if (k < cand_cap) cand[k] = p; }
      }
    }
    cls[p] = c;
  }
  __syncthreads();
  // hysteresis: a weak pixel next to an edge pixel becomes an edge pixel, until nothing changes (the fixed point is the
  // set of 8-connected components that contain a strong pixel -- independent of the visiting order)
  const int nc = min(n_cand, cand_cap);
  const bool overflow = n_cand > cand_cap;     // more weak pixels than the list holds: sweep every pixel instead
  for (int it = 0; it < n; it++) {
    if (threadIdx.x == 0) changed = 0;
    __syncthreads();
    const int cnt = overflow ? n : nc;
    for (int q = threadIdx.x; q < cnt; q += 256) {
      const int p = overflow ? q : cand[q];
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
    __syncthreads();
  }
}

// inclusive prefix minimum over the 256 threads of a workgroup (one value each); ws: 4 int64 of LDS
__device__ __forceinline__ long long block_prefix_min(long long v, long long* ws) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { long long u = __shfl_up(v, o); if (lane >= o) v = u < v ? u : v; }
  if (lane == 63) ws[wv] = v;
  __syncthreads();
  long long pre = 0x7fffffffffffffffLL;
  for (int q = 0; q < wv; q++) pre = ws[q] < pre ? ws[q] : pre;
  __syncthreads();
  return pre < v ? pre : v;
}

// two-pass 3x3 chamfer distance (distanceTransform(255 - canny, DIST_L2, 3)); the 16.16 fixed-point working values live
// in the output buffer itself and are turned into floats by the backward pass
__global__ __launch_bounds__(256) void edge_dt_kernel(const EdgeRoi* __restrict__ rois, const unsigned char* __restrict__ cls_pool, float* map_pool, int row_cap) {
  extern __shared__ unsigned row_lds[];     // neighbouring row: row_cap + 2 values (border cells at both ends)
  __shared__ long long ws[4];
  const EdgeRoi R = rois[blockIdx.x];
  const unsigned char* cls = cls_pool + R.cls_off;
  unsigned* tmp = reinterpret_cast<unsigned*>(map_pool + R.map_off);
  const unsigned HV = 62587u, DIAG = 89738u, DMAX = 0xffffffffu - DIAG;
  const float scale = 1.f / (1 << 16);
  const int w = R.w, h = R.h;
  unsigned* nb = row_lds + 1;               // nb[-1 .. w]
  // ---- forward: top-left to bottom-right
  for (int j = threadIdx.x; j < w + 2; j += 256) row_lds[j] = DMAX;
  __syncthreads();
  for (int i = 0; i < h; i++) {
    long long carry = (long long)DMAX;      // d[j0 - 1] of the running chunk, as "value at column j0 - 1"
    for (int j0 = 0; j0 < w; j0 += 256) {
      const int j = j0 + threadIdx.x;
      long long a = 0x7fffffffffffffffLL;
      unsigned t = DMAX;
      if (j < w) {
        if (cls[i * w + j] == 2) t = 0;
        else {
          unsigned t0 = nb[j - 1] + DIAG, u = nb[j] + HV;
          if (t0 > u) t0 = u;
          u = nb[j + 1] + DIAG; if (t0 > u) t0 = u;
          t = t0;
        }
        a = (long long)t - (long long)j * HV;
      }
      // d[j] = min over k <= j of t[k] + (j - k) HV, and of the value carried in from the left
      long long pm = block_prefix_min(a, ws);
      long long d = pm + (long long)j * HV;
      const long long from_left = carry + (long long)(j - j0 + 1) * HV;
      if (from_left < d) d = from_left;
      if (d > (long long)DMAX) d = DMAX;
      __shared__ long long last;
      if (j < w && (threadIdx.x == 255 || j == w - 1)) last = d;
      __syncthreads();
      if (j < w) tmp[i * w + j] = (unsigned)d;
      carry = last;
      __syncthreads();
    }
    for (int j = threadIdx.x; j < w; j += 256) nb[j] = tmp[i * w + j];
    __syncthreads();
  }
  // ---- backward: bottom-right to top-left (mirrored scan), writing the floats
  for (int j = threadIdx.x; j < w + 2; j += 256) row_lds[j] = DMAX;
  __syncthreads();
  for (int i = h - 1; i >= 0; i--) {
    long long carry = (long long)DMAX;      // d[j + 1] to the right of the running chunk
    const int nchunk = (w + 255) / 256;
    for (int cch = nchunk - 1; cch >= 0; cch--) {
      const int j0 = cch * 256;
      const int jr = j0 + 255 - (int)threadIdx.x;   // thread 0 takes the chunk's rightmost column: scan order = right to left
      long long a = 0x7fffffffffffffffLL;
      if (jr < w) {
        unsigned t0 = tmp[i * w + jr];
        unsigned u = nb[jr + 1] + DIAG; if (t0 > u) t0 = u;
        u = nb[jr] + HV; if (t0 > u) t0 = u;
        u = nb[jr - 1] + DIAG; if (t0 > u) t0 = u;
        a = (long long)t0 + (long long)jr * HV;     // d[j] = min over k >= j of u[k] + (k - j) HV
      }
      long long pm = block_prefix_min(a, ws);
      long long d = pm - (long long)jr * HV;
      const int chunk_right = min(w - 1, j0 + 255);
      const long long from_right = carry + (long long)(chunk_right - jr + 1) * HV;
      if (from_right < d) d = from_right;
      if (d > (long long)DMAX) d = DMAX;
      __shared__ long long lastb;
      if (jr == j0) lastb = d;
      __syncthreads();
      if (jr < w) tmp[i * w + jr] = (unsigned)d;
      carry = lastb;
      __syncthreads();
    }
    for (int j = threadIdx.x; j < w; j += 256) {
      const unsigned v = tmp[i * w + j];
      nb[j] = v;
      reinterpret_cast<float*>(tmp)[i * w + j] = (float)v * scale;
    }
    __syncthreads();
  }
}

void launch_edge_maps(const unsigned char* gray, int W, int H, const EdgeRoi* rois, int n_rois, unsigned char* cls_pool, float* map_pool, int max_w, int low, int high,
                      hipStream_t st) {
  if (n_rois <= 0) return;
  hipLaunchKernelGGL(edge_canny_kernel, dim3(n_rois), dim3(256), 0, st, gray, W, H, rois, cls_pool, low, high);
  hipLaunchKernelGGL(edge_dt_kernel, dim3(n_rois), dim3(256), (size_t)(max_w + 2) * sizeof(unsigned), st, rois, cls_pool, map_pool, max_w);
}

}  // namespace cs
