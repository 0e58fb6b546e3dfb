// valu_rate -- issue cost of the FP64 / conversion / integer vector instructions the detect kernels are made of, in cycles per
// wavefront instruction per SIMD (development tool).  Eight independent chains per lane, 8 waves per SIMD, every SIMD of the device.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -x hip tools/microbench/valu_rate.cpp -o build_tmp/valu_rate && build_tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-result"

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  double a[8];
  for (int q = 0; q < 8; q++) a[q] = seed + q * 0.37 + threadIdx.x * 1e-3;
  int ai[8];
  for (int q = 0; q < 8; q++) ai[q] = (int)a[q] + q;
  float af[8];
  for (int q = 0; q < 8; q++) af[q] = (float)a[q];
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (OP == 0) a[q] = a[q] + 1.25;                                   // v_add_f64
      if (OP == 1) a[q] = a[q] * 1.0000001;                              // v_mul_f64
      if (OP == 2) a[q] = __builtin_fma(a[q], 1.0000001, 0.5);           // v_fma_f64
      if (OP == 3) { ai[q] = (int)a[q]; a[q] += (double)0.0; asm volatile("" : "+v"(ai[q])); asm volatile("" : "+v"(a[q])); }   // v_cvt_i32_f64 (+ nothing)
      if (OP == 4) a[q] = __builtin_amdgcn_rcp(a[q]);                    // v_rcp_f64
      if (OP == 5) a[q] = 1.5 / a[q];                                    // full IEEE division
      if (OP == 6) a[q] = __builtin_sqrt(a[q]);                          // IEEE sqrt
      if (OP == 7) af[q] = af[q] * 1.0000001f + 0.5f;                    // f32 mul + add (no contraction)
      if (OP == 8) ai[q] = ai[q] * 3 + q;                                // integer mad
      if (OP == 9) { a[q] = (a[q] < 2.0) ? a[q] + 1.0 : a[q]; }          // compare + select (+ add)
      if (OP == 10) { ai[q] = __mul24(ai[q], 5) + 1; }
    }
  }
  double s = 0;
  for (int q = 0; q < 8; q++) s += a[q] + ai[q] + af[q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
double run(double* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, d, 10, 1.5);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, d, iters, 1.5);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double* d; hipMalloc(&d, 256 * 8 * 256 * 8);
  int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const int iters = 4000;
  const char* names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_cvt_i32_f64", "v_rcp_f64", "f64 division (IEEE)", "f64 sqrt (IEEE)", "f32 mul + add", "i32 mul + add", "f64 cmp + cndmask x2 + add", "mul24 + add"};
  double ms[11] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters), run<5>(d, iters), run<6>(d, iters), run<7>(d, iters), run<8>(d, iters), run<9>(d, iters), run<10>(d, iters)};
  // wave-instructions per SIMD: 8 waves/SIMD (256 CUs x 8 blocks x 4 waves / 1024 SIMDs) x iters x 8 chains
  const double per_simd = 8.0 * iters * 8;
  printf("clock %d MHz (attribute); cycles per wavefront-instruction(-group) per SIMD at that clock:\n", clk_khz / 1000);
  for (int i = 0; i < 11; i++) printf("  %-28s %8.3f ms   %6.2f cycles\n", names[i], ms[i], ms[i] * 1e-3 * clk_khz * 1e3 / per_simd);
  return 0;
}
