// lsd_kernels.hip -- device stages of the LSD branch of the line-segment producer (use_LSD = true):
//   LineSegmentDetectorImpl::flsd      line_lbd/libs/lsd.cpp:448-461   GaussianBlur(image CV_64F, 7 x 7, sigma 0.75), resize by 0.8
//   LineSegmentDetectorImpl::ll_angle  :545-610                        2 x 2 gradient, modulus, level-line angle, threshold rho
// Everything per pixel and independent: the blur as a separable 7-tap filter in double (row taps left to right, column taps from the
// centre outward -- the summation orders of OpenCV's generic row filter and symmetric column filter, third party; the CPU restatement
// of the same arithmetic is pinned on the reference's saved segments, tests/test_lsd_oracle.py), the bilinear resize with float weights (horizontal
// first), the gradient of the scaled image with the angle from the float polynomial arctangent OpenCV calls fastAtan2.  Compiled with
// -ffp-contract=off: each product and sum rounds once, as on the CPU.  Region growing and rectangle validation are sequential in
// the pixel-visit order and run on the host: csrc/lsd_host.cpp.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

namespace cs {

struct LsdGauss { double k[7]; };                  // getGaussianKernel(7, 0.75, CV_64F)
struct LsdScaleTab {                               // resize's per-column / per-row source offsets and weights (computed on the host)
  const int* xo; const float* xa;                  // W_s offsets, 2 W_s weights
  const int* yo; const float* ya;                  // H_s offsets, 2 H_s weights
};

#define LSD_NOTDEF (-1024.0)

__device__ __forceinline__ int lsd_reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
  return p;
}

// degrees in [0, 360]: the 7th-order odd polynomial on min / max, float throughout
__device__ __forceinline__ float lsd_fast_atan2(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795), p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795),
              p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795), p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
  const float ax = fabsf(x), ay = fabsf(y);
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

enum { BT = 32, BR = 3 };

// blockIdx.z = image.  One workgroup per 32 x 32 tile of the blurred image: the 38 x 38 gray patch in LDS, row pass into LDS, column pass out.
__global__ __launch_bounds__(256) void lsd_blur_kernel(const unsigned char* __restrict__ gray, int W, int H, LsdGauss G, double* __restrict__ blur) {
  const size_t N = (size_t)W * H;
  gray += blockIdx.z * N;
  blur += blockIdx.z * N;
  __shared__ unsigned char sg[BT + 2 * BR][BT + 2 * BR + 2];
  __shared__ double rp[BT + 2 * BR][BT + 1];
  const int x0 = blockIdx.x * BT, y0 = blockIdx.y * BT, t = threadIdx.x;
  for (int e = t; e < (BT + 2 * BR) * (BT + 2 * BR); e += 256) {
    const int r = e / (BT + 2 * BR), c = e - r * (BT + 2 * BR);
    sg[r][c] = gray[(size_t)lsd_reflect101(y0 - BR + r, H) * W + lsd_reflect101(x0 - BR + c, W)];
  }
  __syncthreads();
  for (int e = t; e < (BT + 2 * BR) * BT; e += 256) {
    const int r = e / BT, c = e - r * BT;
    double s = G.k[0] * (double)sg[r][c];
#pragma unroll
    for (int q = 1; q < 7; q++) s += G.k[q] * (double)sg[r][c + q];
    rp[r][c] = s;
  }
  __syncthreads();
  for (int e = t; e < BT * BT; e += 256) {
    const int r = e / BT, c = e - r * BT;
    const int x = x0 + c, y = y0 + r;
    if (x >= W || y >= H) continue;
    double s = G.k[BR] * rp[r + BR][c];
#pragma unroll
    for (int q = 1; q <= BR; q++) s += G.k[BR + q] * (rp[r + BR + q][c] + rp[r + BR - q][c]);
    blur[(size_t)y * W + x] = s;
  }
}

// One workgroup per 32 x 32 tile of the scaled image: 33 x 33 scaled values (the gradient looks one to the right and one down) in LDS,
// then angle and modulus.  out: per image (out_stride bytes apart) [modulus: Ws Hs doubles | angle: Ws Hs floats] -- the angle as the
// float fastAtan2 returns (degrees, LSD_NOTDEF where the modulus is below the threshold): the reference's double is that float times
// pi / 180, which the host stage forms where it needs it (lsd_host.cpp).
__global__ __launch_bounds__(256) void lsd_scale_grad_kernel(const double* __restrict__ blur, int W, int H, int Ws, int Hs, LsdScaleTab T, double rho, char* __restrict__ out, size_t out_stride) {
  const size_t N = (size_t)W * H, Ns = (size_t)Ws * Hs;
  blur += blockIdx.z * N;
  double* __restrict__ mod = reinterpret_cast<double*>(out + blockIdx.z * out_stride);
  float* __restrict__ ang = reinterpret_cast<float*>(mod + Ns);
  __shared__ double sc[BT + 1][BT + 2];
  const int x0 = blockIdx.x * BT, y0 = blockIdx.y * BT, t = threadIdx.x;
  for (int e = t; e < (BT + 1) * (BT + 1); e += 256) {
    const int r = e / (BT + 1), c = e - r * (BT + 1);
    const int dx = x0 + c, dy = y0 + r;
    double v = 0.0;
    if (dx < Ws && dy < Hs) {
      const int sx = T.xo[dx], sx1 = min(sx + 1, W - 1);
      const float a0 = T.xa[2 * dx], a1 = T.xa[2 * dx + 1], b0 = T.ya[2 * dy], b1 = T.ya[2 * dy + 1];
      const int sy0 = min(max(T.yo[dy], 0), H - 1), sy1 = min(max(T.yo[dy] + 1, 0), H - 1);
      const double* S0 = blur + (size_t)sy0 * W;
      const double* S1 = blur + (size_t)sy1 * W;
      const double h0 = S0[sx] * (double)a0 + S0[sx1] * (double)a1;
      const double h1 = S1[sx] * (double)a0 + S1[sx1] * (double)a1;
      v = h0 * (double)b0 + h1 * (double)b1;
    }
    sc[r][c] = v;
  }
  __syncthreads();
  for (int e = t; e < BT * BT; e += 256) {
    const int r = e / BT, c = e - r * BT;
    const int x = x0 + c, y = y0 + r;
    if (x >= Ws || y >= Hs) continue;
    float a = (float)LSD_NOTDEF;
    double m = 0.0;
    if (x < Ws - 1 && y < Hs - 1) {
      const double DA = sc[r + 1][c + 1] - sc[r][c], BC = sc[r][c + 1] - sc[r + 1][c];
      const double gx = DA + BC, gy = DA - BC;
      m = sqrt((gx * gx + gy * gy) / 4);
      if (m > rho) a = lsd_fast_atan2((float)gx, (float)(-gy));
    }
    ang[(size_t)y * Ws + x] = a;
    mod[(size_t)y * Ws + x] = m;
  }
}

void launch_lsd_maps(const unsigned char* gray, int W, int H, int Ws, int Hs, const LsdGauss& G, const LsdScaleTab& T, double rho, double* blur, char* out, size_t out_stride, hipStream_t st, int n_images) {
  for (int i0 = 0; i0 < n_images; i0 += 65535) {      // gridDim.z limit
    const int nz = n_images - i0 < 65535 ? n_images - i0 : 65535;
    const size_t N = (size_t)W * H;
    hipLaunchKernelGGL(lsd_blur_kernel, dim3((W + BT - 1) / BT, (H + BT - 1) / BT, nz), dim3(256), 0, st, gray + (size_t)i0 * N, W, H, G, blur + (size_t)i0 * N);
    hipLaunchKernelGGL(lsd_scale_grad_kernel, dim3((Ws + BT - 1) / BT, (Hs + BT - 1) / BT, nz), dim3(256), 0, st, blur + (size_t)i0 * N, W, H, Ws, Hs, T, rho, out + (size_t)i0 * out_stride, out_stride);
  }
}

}  // namespace cs
