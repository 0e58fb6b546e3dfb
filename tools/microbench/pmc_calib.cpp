// pmc_calib -- known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in the access patterns the
// library's kernels use (MI355X_MICROARCH.md calibrates only 16 B/lane streaming reads: FETCH_SIZE = 1/2 of the bytes).
// Every kernel touches a 1 GiB array (4x the 256 MiB Infinity Cache) exactly once, so the bytes that must cross the HBM
// interface are known:
//   calib_read16        16 B per lane, coalesced (the guide's reference pattern)                       1 GiB useful = 1 GiB of lines
//   calib_read8          8 B per lane, coalesced (doubles streamed by one lane per element)            1 GiB
//   calib_read4          4 B per lane, coalesced (floats: distance maps read row-wise)                 1 GiB
//   calib_read8_s144     8 B per lane at a 144-byte lane stride (a lane walks its own 6x3 W record: ba_schur_kernel,
//                        ba_backsub_kernel); the wave's 18 loads together cover its 64 records        1 GiB
//   calib_gather4        4 B gathers at pseudo-random addresses (score_kernel's map samples): every lane one 4-byte word from a
//                        different 128-byte line; useful 4 B, line traffic 64 or 128 B per gather      n x {4, 64, 128} B
//   calib_write16 / calib_write8_s144   the store counterparts (WRITE_SIZE)
// Usage: pmc_calib            (runs every kernel twice; profile with rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE)
// tools/pmc_calib.sh turns the two counter dumps into per-pattern factors  bytes = factor x counter x 1024.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void calib_read16(const double2* a, size_t n, double* sink) {
  double s = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { double2 v = a[i]; s += v.x + v.y; }
  if (s == 1.2345e300) sink[0] = s;
}
__global__ __launch_bounds__(256) void calib_read8(const double* a, size_t n, double* sink) {
  double s = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
  if (s == 1.2345e300) sink[0] = s;
}
__global__ __launch_bounds__(256) void calib_read4(const float* a, size_t n, double* sink) {
  float s = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
  if (s == 1.2345e30f) sink[0] = s;
}
// lane r reads the 18 doubles of record r one after the other (records are 144 B apart across the lanes of a load)
__global__ __launch_bounds__(256) void calib_read8_s144(const double* a, size_t n_rec, double* sink) {
  double s = 0;
  for (size_t r = blockIdx.x * 256ull + threadIdx.x; r < n_rec; r += (size_t)gridDim.x * 256) {
    const double* p = a + 18 * r;
#pragma unroll
    for (int q = 0; q < 18; q++) s += p[q];
  }
  if (s == 1.2345e300) sink[0] = s;
}
__global__ __launch_bounds__(256) void calib_gather4(const float* a, size_t n_lines, size_t n_gathers, double* sink) {
  float s = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n_gathers; i += (size_t)gridDim.x * 256) {
    // a permutation of the 128-byte lines (odd multiplier modulo a power of two): every line is touched exactly once
    const size_t line = (i * 2654435761ull + 12345ull) & (n_lines - 1);
    s += a[line * 32 + (i & 31)];
  }
  if (s == 1.2345e30f) sink[0] = s;
}
__global__ __launch_bounds__(256) void calib_write16(double2* a, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = double2{1.0, 2.0};
}
__global__ __launch_bounds__(256) void calib_write8_s144(double* a, size_t n_rec) {
  for (size_t r = blockIdx.x * 256ull + threadIdx.x; r < n_rec; r += (size_t)gridDim.x * 256) {
    double* p = a + 18 * r;
#pragma unroll
    for (int q = 0; q < 18; q++) p[q] = (double)q;
  }
}

int main() {
  const size_t bytes = 1ull << 30;
  void* buf; double* sink;
  CK(hipMalloc(&buf, bytes + 4096)); CK(hipMalloc((void**)&sink, 64));
  CK(hipMemset(buf, 0, bytes + 4096));
  const int grid = 256 * 16;
  const size_t n_rec = bytes / 144, n_lines = bytes / 128;   // n_lines = 2^23
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(calib_read16, dim3(grid), dim3(256), 0, 0, (const double2*)buf, bytes / 16, sink);
    hipLaunchKernelGGL(calib_read8, dim3(grid), dim3(256), 0, 0, (const double*)buf, bytes / 8, sink);
    hipLaunchKernelGGL(calib_read4, dim3(grid), dim3(256), 0, 0, (const float*)buf, bytes / 4, sink);
    hipLaunchKernelGGL(calib_read8_s144, dim3(grid), dim3(256), 0, 0, (const double*)buf, n_rec, sink);
    hipLaunchKernelGGL(calib_gather4, dim3(grid), dim3(256), 0, 0, (const float*)buf, n_lines, n_lines, sink);
    hipLaunchKernelGGL(calib_write16, dim3(grid), dim3(256), 0, 0, (double2*)buf, bytes / 16);
    hipLaunchKernelGGL(calib_write8_s144, dim3(grid), dim3(256), 0, 0, (double*)buf, n_rec);
    CK(hipDeviceSynchronize());
  }
  printf("bytes %zu records %zu lines %zu\n", bytes, n_rec, n_lines);
  return 0;
}
