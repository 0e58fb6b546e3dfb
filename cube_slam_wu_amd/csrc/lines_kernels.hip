// lines_kernels.hip -- device stages of the line-segment producer: the per-pixel half of the
// reference's EDLines detector,
//   BinaryDescriptor::OctaveKeyLines  line_lbd/libs/binary_descriptor.cpp:816-817   cv::GaussianBlur(image, blur, Size(5, 5), 1.0)
//   EDLineDetector::EdgeDrawing       :1622-1670   Sobel dx / dy (CV_16S), |dx| + |dy|, threshold (> gradienThreshold_ + 1), / 4 with
//                                                  rounding, direction map (|dx| < |dy| = horizontal), anchor scan (every second row
//                                                  and column, gradient >= both neighbours across the edge + anchorThreshold_).
// The arithmetic of the OpenCV calls (third party, not under /root/reference) is integer throughout: 8-bit fixed-point Gaussian
// (kernel rounded to round(256 k), (sum + 2^15) >> 16), exact Sobel, round-half-even division by 4; borders BORDER_REFLECT_101.
// One workgroup per 32 x 32 tile: the 40 x 40 gray patch the tile depends on is staged in LDS once (blur needs +-2, Sobel +-1, the
// anchor test +-1 of the gradient), so every gray pixel is read from HBM ~1.6 times and every output written once.
// The sequential half (smart routing, line fitting, validation) runs on the host: csrc/lines_host.cpp.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace cs {

// What goes back to the host, THREE bytes per pixel (round 6; a 32-bit word until then -- the copy back is the batch's longest stage): the Sobel
// derivatives (dxImg_, dyImg_; |.| <= 4 x 255 = 1020: eleven bits each) and the anchor flag, little endian,
//   bits 0..10  dx (two's complement)        bit 11  anchor        bits 12..22  dy (two's complement)        bit 23  0
// i.e. bits 11..22 are 2 dy + anchor as before.  The thresholded gradient / 4 (gImg_) and the direction map (dirImg_: |dx| < |dy| = horizontal) are
// functions of dx and dy that the host stage evaluates where it reads them (lines_host.cpp, Maps).  Image i's map starts at p3 + 3 N i.
struct LineMaps {
  unsigned char* p3;
};

__device__ __forceinline__ int lines_reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
  return p;
}
__device__ __forceinline__ int lines_div4_half_even(int v) {   // cvRound(v * 0.25) for v >= 0
  const int q = v >> 2, r = v & 3;
  return q + ((r == 3) || (r == 2 && (q & 1)));
}

enum { LT = 32, LG = LT + 8, LB = LT + 4, LS = LT + 2 };

// blockIdx.z = image of a batch: image i's gray at gray + i N, its packed map at m.p3 + 3 i N
__global__ __launch_bounds__(256) void lines_maps_kernel(const unsigned char* __restrict__ gray, int W, int H, LineMaps m, int k0, int k1, int k2,
                                                         int grad_thr, int anchor_thr, int scan) {
  {
    const size_t N = (size_t)W * H, img = blockIdx.z;
    gray += img * N;
    m.p3 += 3 * img * N;
  }
  __shared__ unsigned char sg[LG][LG + 4];     // gray, tile origin - 4
  __shared__ int rs[LG][LB];                   // row pass of the blur
  __shared__ unsigned char sb[LB][LB + 4];     // blurred, tile origin - 2
  __shared__ short sgr[LS][LS + 2];            // gImg, tile origin - 1
  __shared__ unsigned char sdir[LS][LS + 2];   // direction, tile origin - 1
  __shared__ short sdx[LS][LS + 2], sdy[LS][LS + 2];   // Sobel derivatives, tile origin - 1
  const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT, t = threadIdx.x;
  for (int e = t; e < LG * LG; e += 256) {
    const int r = e / LG, c = e - r * LG;
    sg[r][c] = gray[(size_t)lines_reflect101(y0 - 4 + r, H) * W + lines_reflect101(x0 - 4 + c, W)];
  }
  __syncthreads();
  for (int e = t; e < LG * LB; e += 256) {
    const int r = e / LB, c = e - r * LB;
    rs[r][c] = k0 * (sg[r][c] + sg[r][c + 4]) + k1 * (sg[r][c + 1] + sg[r][c + 3]) + k2 * sg[r][c + 2];
  }
  __syncthreads();
  for (int e = t; e < LB * LB; e += 256) {
    const int r = e / LB, c = e - r * LB;
    const int s = k0 * (rs[r][c] + rs[r + 4][c]) + k1 * (rs[r + 1][c] + rs[r + 3][c]) + k2 * rs[r + 2][c];
    sb[r][c] = (unsigned char)min(max((s + (1 << 15)) >> 16, 0), 255);
  }
  __syncthreads();
  for (int e = t; e < LS * LS; e += 256) {
    const int r = e / LS, c = e - r * LS;              // image pixel (x0 - 1 + c, y0 - 1 + r); blurred index (c + 1, r + 1)
    const int a = sb[r][c], b = sb[r][c + 1], cc = sb[r][c + 2], d = sb[r + 1][c], f = sb[r + 1][c + 2], p = sb[r + 2][c], q = sb[r + 2][c + 1], rr = sb[r + 2][c + 2];
    const int gx = (cc + 2 * f + rr) - (a + 2 * d + p), gy = (p + 2 * q + rr) - (a + 2 * b + cc);
    const int ax = abs(gx), ay = abs(gy), s = ax + ay;
    const int gq = lines_div4_half_even(s > grad_thr + 1 ? s : 0);
    sgr[r][c] = (short)gq;
    sdir[r][c] = ax < ay ? 255 : 0;
    sdx[r][c] = (short)gx; sdy[r][c] = (short)gy;
  }
  __syncthreads();
  for (int e = t; e < LT * LT; e += 256) {
    const int r = e / LT, c = e - r * LT;
    const int x = x0 + c, y = y0 + r;
    if (x >= W || y >= H) continue;
    unsigned char an = 0;
    // for (w = 1; w < W - 1; w += scan) for (h = 1; h < H - 1; h += scan)   (:1641-1668)
    if (x >= 1 && x < W - 1 && y >= 1 && y < H - 1 && (x - 1) % scan == 0 && (y - 1) % scan == 0) {
      const int g = sgr[r + 1][c + 1];
      const bool hor = sdir[r + 1][c + 1] == 255;
      const int n1 = hor ? sgr[r][c + 1] : sgr[r + 1][c], n2 = hor ? sgr[r + 2][c + 1] : sgr[r + 1][c + 2];
      an = (g >= n1 + anchor_thr && g >= n2 + anchor_thr) ? 1 : 0;
    }
    const unsigned wd = (((unsigned)(2 * (int)sdy[r + 1][c + 1] + an) & 0xfffu) << 11) | ((unsigned)(int)sdx[r + 1][c + 1] & 0x7ffu);
    unsigned char* o = m.p3 + 3 * ((size_t)y * W + x);
    o[0] = (unsigned char)wd; o[1] = (unsigned char)(wd >> 8); o[2] = (unsigned char)(wd >> 16);
  }
}

// n_images images laid out back to back (gray: N bytes each; m: image 0's planes, image i's at the strides lines_maps_kernel states)
void launch_lines_maps(const unsigned char* gray, int W, int H, const LineMaps& m, const int k[3], int grad_thr, int anchor_thr, int scan, hipStream_t st, int n_images) {
  for (int i0 = 0; i0 < n_images; i0 += 65535) {      // gridDim.z limit
    const int nz = n_images - i0 < 65535 ? n_images - i0 : 65535;
    const size_t N = (size_t)W * H;
    LineMaps mi{m.p3 + 3 * (size_t)i0 * N};
    hipLaunchKernelGGL(lines_maps_kernel, dim3((W + LT - 1) / LT, (H + LT - 1) / LT, nz), dim3(256), 0, st, gray + (size_t)i0 * N, W, H, mi, k[0], k[1], k[2], grad_thr, anchor_thr, scan);
  }
}

}  // namespace cs
