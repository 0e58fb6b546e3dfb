// TEST INFRASTRUCTURE (tests/test_lines_host_cpu.py): the packed map the device hands to the EDLines host stage -- one word per
// pixel, dx | (2 dy + anchor) << 16 (csrc/lines_kernels.hip) -- computed by the CPU restatement's own blur / Sobel / gradient / anchor
// code, so that the product's sequential half (csrc/lines_host.cpp, linked from lines_host_check.cpp) can be held to the
// restatement WITHOUT a GPU.  A translation unit of its own (both sides keep their names in anonymous namespaces).
#include "../../oracle/edlines_oracle.cpp"

extern "C" void oracle_lines_packed_map(const uint8_t* gray, int w, int h, int* out) {
  std::vector<uint8_t> blur((size_t)w * h);
  gaussian_blur_5x5_sigma1(gray, w, h, blur.data());
  std::vector<short> dx, dy;
  sobel3(blur.data(), w, h, dx, dy);
  Detector D;                                    // (for its thresholds)
  const size_t N = (size_t)w * h;
  std::vector<short> g(N);
  std::vector<uint8_t> dir(N);
  for (size_t i = 0; i < N; i++) {
    const int ax = std::abs((int)dx[i]), ay = std::abs((int)dy[i]), s = ax + ay;
    g[i] = div4_round(s > D.gradienThreshold + 1 ? s : 0);
    dir[i] = ax < ay ? 255 : 0;
  }
  for (size_t i = 0; i < N; i++) out[i] = (int)(((unsigned)(2 * (int)dy[i]) << 16) | ((unsigned)(int)dx[i] & 0xffffu));
  for (int ww = 1; ww < w - 1; ww += D.scanIntervals)
    for (int hh = 1; hh < h - 1; hh += D.scanIntervals) {
      const int idx = hh * w + ww;
      const bool an = dir[idx] == 255 ? (g[idx] >= g[idx - w] + D.anchorThreshold && g[idx] >= g[idx + w] + D.anchorThreshold)
                                      : (g[idx] >= g[idx - 1] + D.anchorThreshold && g[idx] >= g[idx + 1] + D.anchorThreshold);
      if (an) out[idx] = (int)(((unsigned)(2 * (int)dy[idx] + 1) << 16) | ((unsigned)(int)dx[idx] & 0xffffu));
    }
}
