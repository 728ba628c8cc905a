// cvlite.h -- the handful of OpenCV types the ORB front-end boundary mentions, for builds WITHOUT OpenCV
// (this image has none).  Layout-compatible with the real ones where the C ABI relies on it (cv::KeyPoint is
// 7 x 4 bytes).  When OpenCV is available compile with -DORBFE_HAVE_OPENCV and this file is not used.
#pragma once
#include <stdint.h>
#include <string.h>

#include <memory>
#include <vector>

namespace cv {

enum { CV_8U = 0, CV_8UC1 = 0, CV_32F = 5 };

template <typename T>
struct Point_ {
  T x, y;
  Point_() : x(0), y(0) {}
  Point_(T x_, T y_) : x(x_), y(y_) {}
  Point_& operator*=(T s) { x *= s; y *= s; return *this; }
};
typedef Point_<float> Point2f;
typedef Point_<int> Point2i;
typedef Point_<int> Point;

struct KeyPoint {
  Point2f pt;
  float size;
  float angle;
  float response;
  int octave;
  int class_id;
  KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
  KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1)
      : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint must be 28 bytes");

// Reference-counted 8-bit matrix with an optional ROI view (enough for images, descriptors, pyramid levels).
class Mat {
 public:
  int rows = 0, cols = 0;
  size_t step = 0;
  uint8_t* data = nullptr;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int /*type*/, void* ext, size_t st = 0) : rows(r), cols(c), step(st ? st : (size_t)c), data((uint8_t*)ext) {}
  void create(int r, int c, int /*type*/) {
    if (r == rows && c == cols && buf_ && step == (size_t)c) return;
    rows = r; cols = c; step = (size_t)c;
    buf_.reset(new uint8_t[(size_t)r * c + 1], std::default_delete<uint8_t[]>());
    data = buf_.get();
  }
  void release() { rows = cols = 0; step = 0; data = nullptr; buf_.reset(); }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int type() const { return CV_8UC1; }
  size_t step1() const { return step; }
  bool isContinuous() const { return step == (size_t)cols; }
  uint8_t* ptr(int r = 0) { return data + (size_t)r * step; }
  const uint8_t* ptr(int r = 0) const { return data + (size_t)r * step; }
  template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
  template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
  template <typename T> T& at(int r, int c) { return *reinterpret_cast<T*>(data + (size_t)r * step + c * sizeof(T)); }
  Mat row(int r) const { Mat m = *this; m.rows = 1; m.data = data + (size_t)r * step; return m; }
  Mat roi(int x, int y, int w, int h) const { Mat m = *this; m.rows = h; m.cols = w; m.data = data + (size_t)y * step + x; return m; }
  Mat clone() const {
    Mat m(rows, cols, CV_8U);
    for (int r = 0; r < rows; r++) memcpy(m.ptr(r), ptr(r), (size_t)cols);
    return m;
  }

 private:
  std::shared_ptr<uint8_t> buf_;
};

// The reference passes images as InputArray and receives descriptors through OutputArray.
class _InputArray {
 public:
  _InputArray() {}
  _InputArray(const Mat& m) : m_(m) {}
  Mat getMat() const { return m_; }
  bool empty() const { return m_.empty(); }
 private:
  Mat m_;
};
class _OutputArray {
 public:
  _OutputArray(Mat& m) : p_(&m) {}
  void create(int r, int c, int t) const { p_->create(r, c, t); }
  void release() const { p_->release(); }
  Mat getMat() const { return *p_; }
 private:
  Mat* p_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

}  // namespace cv
