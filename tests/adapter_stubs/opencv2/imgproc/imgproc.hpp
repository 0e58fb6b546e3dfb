// compile-hygiene stand-in (tests/adapter_stubs/README.md)
#pragma once
#include "opencv2/core/core.hpp"
namespace cv {
enum { COLOR_BGR2GRAY = 6, DIST_L2 = 2 };
void cvtColor(const Mat& src, Mat& dst, int code);
void Canny(const Mat& image, Mat& edges, double threshold1, double threshold2);
void distanceTransform(const Mat& src, Mat& dst, int distanceType, int maskSize);
}  // namespace cv
