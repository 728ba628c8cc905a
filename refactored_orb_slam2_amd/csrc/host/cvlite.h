// cvlite.h -- the handful of OpenCV types the ORB front-end boundary mentions, for builds WITHOUT OpenCV
// (this image has none).  Layout-compatible with the real ones where the C ABI relies on it (cv::KeyPoint is
// 7 x 4 bytes).  When OpenCV is available compile with -DORBFE_HAVE_OPENCV and this file is not used.
// The type codes are global macros, exactly as OpenCV defines them (CV_8UC1 is not a member of namespace cv), so the same
// source compiles against either.  The host sources restrict themselves to the subset below: Mat(rows, cols, type),
// create / release / empty / type / rows / cols / step / ptr / at<T>(r, c) / at<T>(i) / row / clone / operator()(Rect).
#pragma once
#include <stdint.h>
#include <string.h>

#include <memory>
#include <vector>

#ifndef CV_8U
#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_32FC1 5
#endif

namespace cv {

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

struct Rect {
  int x, y, width, height;
  Rect() : x(0), y(0), width(0), height(0) {}
  Rect(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {}
};

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

// Reference-counted single-channel matrix (CV_8U or CV_32F) with an optional ROI view: enough for images, descriptors,
// pyramid levels and the small float matrices (poses, points) the matcher reads.
class Mat {
 public:
  int rows = 0, cols = 0;
  size_t step = 0;   // bytes per row
  uint8_t* data = nullptr;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void* ext, size_t st = 0) : rows(r), cols(c), data((uint8_t*)ext), type_(type) {
    step = st ? st : (size_t)c * elemSize();
  }
  static Mat zeros(int r, int c, int type) {
    Mat m(r, c, type);
    memset(m.data, 0, (size_t)r * m.step);
    return m;
  }
  static Mat eye(int r, int c, int type) {
    Mat m = zeros(r, c, type);
    for (int i = 0; i < r && i < c; i++) {
      if (type == CV_32F) m.at<float>(i, i) = 1.f;
      else m.at<uint8_t>(i, i) = 1;
    }
    return m;
  }
  void create(int r, int c, int type) {
    if (r == rows && c == cols && type == type_ && buf_ && step == (size_t)c * elemSize()) return;
    rows = r; cols = c; type_ = type; step = (size_t)c * elemSize();
    buf_.reset(new uint8_t[(size_t)r * step + 4], std::default_delete<uint8_t[]>());
    data = buf_.get();
  }
  void release() { rows = cols = 0; step = 0; data = nullptr; buf_.reset(); }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int type() const { return type_; }
  size_t elemSize() const { return type_ == CV_32F ? 4 : 1; }
  size_t step1() const { return step / elemSize(); }
  bool isContinuous() const { return step == (size_t)cols * elemSize(); }
  uint8_t* ptr(int r = 0) { return data + (size_t)r * step; }
  const uint8_t* ptr(int r = 0) const { return data + (size_t)r * step; }
  template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
  template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
  template <typename T> T& at(int r, int c) { return *reinterpret_cast<T*>(data + (size_t)r * step + c * sizeof(T)); }
  template <typename T> const T& at(int r, int c) const { return *reinterpret_cast<const T*>(data + (size_t)r * step + c * sizeof(T)); }
  // single index: element i of a row or column vector
  template <typename T> T& at(int i) { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
  template <typename T> const T& at(int i) const { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
  Mat row(int r) const { Mat m = *this; m.rows = 1; m.data = data + (size_t)r * step; return m; }
  Mat operator()(const Rect& r) const {
    Mat m = *this;
    m.rows = r.height; m.cols = r.width; m.data = data + (size_t)r.y * step + (size_t)r.x * elemSize();
    return m;
  }
  Mat clone() const {
    Mat m(rows, cols, type_);
    for (int r = 0; r < rows; r++) memcpy(m.ptr(r), ptr(r), (size_t)cols * elemSize());
    return m;
  }

 private:
  int type_ = CV_8U;
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
