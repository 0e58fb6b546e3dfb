// compile-hygiene stand-in (tests/adapter_stubs/README.md): the slice of cv::Mat the adapters use
#pragma once
namespace cv {
struct Rect { Rect(int x, int y, int w, int h); };
class Mat {
 public:
  Mat();
  int rows, cols;
  unsigned char* data;
  int channels() const;
  bool isContinuous() const;
  Mat clone() const;
  Mat operator()(const Rect& roi) const;
  void create(int rows, int cols, int type);
  template <class T> T* ptr(int row = 0);
  template <class T> const T* ptr(int row = 0) const;
  template <class T> T& at(int i, int j);
};
Mat operator-(int s, const Mat& m);
}  // namespace cv
#define CV_32FC1 5
