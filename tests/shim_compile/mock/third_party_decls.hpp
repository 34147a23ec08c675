// TEST DOUBLE -- declaration-only stand-ins for the slices of OpenCV / Eigen / Sophus that the shims touch, so that
// `g++ -fsyntax-only` can parse shim/*.cc and include/vieo_shim.hpp in an image that has none of those libraries.
// A syntax and type check only: nothing here is linked or run, and it is no parity evidence.
#pragma once
#include <cstddef>
#include <cstdint>
#include <list>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5

namespace cv {
struct Point2f {
  float x, y;
};
struct KeyPoint {
  Point2f pt;
  float size, angle, response;
  int octave, class_id;
};
struct Rect {
  Rect(int x_, int y_, int w_, int h_);
};
class _OutputArray;
class Mat {
 public:
  Mat();
  Mat(int rows_, int cols_, int type_);
  int rows, cols;
  unsigned char* data;
  struct Step {
    operator size_t() const;
  } step;
  template <class T> T& at(int r, int c);
  template <class T> const T& at(int r, int c) const;
  template <class T> T* ptr(int r = 0);
  template <class T> const T* ptr(int r = 0) const;
  bool empty() const;
  int type() const;
  Mat clone() const;
  Mat row(int r) const;
  Mat rowRange(int a, int b) const;
  Mat operator()(const Rect& roi) const;
  void copyTo(const _OutputArray& dst) const;
};
Mat operator*(const Mat& a, const Mat& b);
void vconcat(const Mat& a, const Mat& b, Mat& dst);
class _InputArray {
 public:
  _InputArray(const Mat& m);
  bool empty() const;
  Mat getMat() const;
};
class _OutputArray {
 public:
  _OutputArray(Mat& m);
  void release() const;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
}  // namespace cv

namespace Eigen {
template <class T, int R, int C>
class Matrix {
 public:
  Matrix();
  T& operator()(int r, int c);
  const T& operator()(int r, int c) const;
  T& operator()(int i);
  const T& operator()(int i) const;
  static Matrix Identity();
  static Matrix Zero();
  Matrix& operator+=(const Matrix& o);
};
template <class T>
class aligned_allocator : public std::allocator<T> {
 public:
  template <class U>
  struct rebind {
    typedef aligned_allocator<U> other;
  };
};
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<float, 3, 1> Vector3f;
template <class T>
class Quaternion {
 public:
  Quaternion(T w, T x, T y, T z);
  explicit Quaternion(const Matrix<T, 3, 3>& R);
  T w() const;
  T x() const;
  T y() const;
  T z() const;
};
typedef Quaternion<double> Quaterniond;
}  // namespace Eigen

namespace Sophus {
template <class T>
class SO3ex {
 public:
  SO3ex();
  explicit SO3ex(const Eigen::Quaternion<T>& q);
  const Eigen::Quaternion<T>& unit_quaternion() const;
};
typedef SO3ex<double> SO3exd;
template <class T>
class SE3 {
 public:
  Eigen::Matrix<T, 3, 3> rotationMatrix() const;
  Eigen::Matrix<T, 3, 1> translation() const;
};
typedef SE3<double> SE3d;
}  // namespace Sophus
