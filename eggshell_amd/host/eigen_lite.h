// eigen_lite.h -- the handful of Eigen types the eggshell hot-path API is
// written in, so the adapter in this directory keeps the reference's
// signatures (Vector3d, Matrix3d, VectorXd, MatrixXd, ArrayXb) without Eigen,
// which is not available in this image.  In the reference tree
// these aliases come from Eigen (utils.h:8-15) and this header is not needed:
// see INTEGRATION.md.  Storage is row-major; only what the path uses exists.
#ifndef EGS_EIGEN_LITE_H
#define EGS_EIGEN_LITE_H

#include <cmath>
#include <cstddef>
#include <vector>

struct Vector3d {
  double d[3];
  Vector3d() : d{0, 0, 0} {}
  Vector3d(double x, double y, double z) : d{x, y, z} {}
  static Vector3d Zero() { return Vector3d(); }
  double &operator()(int i) { return d[i]; }
  double operator()(int i) const { return d[i]; }
  double &operator[](int i) { return d[i]; }
  double operator[](int i) const { return d[i]; }
  Vector3d operator+(const Vector3d &o) const { return {d[0] + o.d[0], d[1] + o.d[1], d[2] + o.d[2]}; }
  Vector3d operator-(const Vector3d &o) const { return {d[0] - o.d[0], d[1] - o.d[1], d[2] - o.d[2]}; }
  Vector3d operator*(double s) const { return {d[0] * s, d[1] * s, d[2] * s}; }
  Vector3d operator/(double s) const { return {d[0] / s, d[1] / s, d[2] / s}; }
  double dot(const Vector3d &o) const { return (d[0] * o.d[0] + d[1] * o.d[1]) + d[2] * o.d[2]; }
  Vector3d cross(const Vector3d &o) const {
    return {d[1] * o.d[2] - d[2] * o.d[1], d[2] * o.d[0] - d[0] * o.d[2], d[0] * o.d[1] - d[1] * o.d[0]};
  }
  double norm() const { return std::sqrt(dot(*this)); }
};
inline Vector3d operator*(double s, const Vector3d &v) { return v * s; }

struct Matrix3d {
  double d[9];
  Matrix3d() : d{0, 0, 0, 0, 0, 0, 0, 0, 0} {}
  static Matrix3d Zero() { return Matrix3d(); }
  static Matrix3d Identity() {
    Matrix3d m;
    m.d[0] = m.d[4] = m.d[8] = 1.0;
    return m;
  }
  double &operator()(int r, int c) { return d[3 * r + c]; }
  double operator()(int r, int c) const { return d[3 * r + c]; }
  Vector3d col(int c) const { return {d[c], d[3 + c], d[6 + c]}; }
  Matrix3d transpose() const {
    Matrix3d t;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) t(r, c) = (*this)(c, r);
    return t;
  }
  Matrix3d operator*(const Matrix3d &o) const {
    Matrix3d m;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) m(r, c) = ((*this)(r, 0) * o(0, c) + (*this)(r, 1) * o(1, c)) + (*this)(r, 2) * o(2, c);
    return m;
  }
  Vector3d operator*(const Vector3d &v) const {
    return {(d[0] * v[0] + d[1] * v[1]) + d[2] * v[2], (d[3] * v[0] + d[4] * v[1]) + d[5] * v[2],
            (d[6] * v[0] + d[7] * v[1]) + d[8] * v[2]};
  }
  Matrix3d operator*(double s) const {
    Matrix3d m;
    for (int k = 0; k < 9; ++k) m.d[k] = d[k] * s;
    return m;
  }
  Matrix3d inverse() const {  // cofactors, as Eigen's fixed-size inverse
    const double *A = d;
    double c00 = A[4] * A[8] - A[5] * A[7], c10 = A[5] * A[6] - A[3] * A[8], c20 = A[3] * A[7] - A[4] * A[6];
    double id = 1.0 / ((A[0] * c00 + A[1] * c10) + A[2] * c20);
    Matrix3d m;
    m.d[0] = c00 * id; m.d[1] = (A[2] * A[7] - A[1] * A[8]) * id; m.d[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    m.d[3] = c10 * id; m.d[4] = (A[0] * A[8] - A[2] * A[6]) * id; m.d[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    m.d[6] = c20 * id; m.d[7] = (A[1] * A[6] - A[0] * A[7]) * id; m.d[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return m;
  }
};
inline Matrix3d operator*(double s, const Matrix3d &m) { return m * s; }

class VectorXd {
 public:
  VectorXd() {}
  explicit VectorXd(int n) : v_(n, 0.0) {}
  static VectorXd Zero(int n) { return VectorXd(n); }
  int size() const { return (int)v_.size(); }
  int rows() const { return size(); }
  void resize(int n) { v_.assign(n, 0.0); }
  double &operator()(int i) { return v_[i]; }
  double operator()(int i) const { return v_[i]; }
  double *data() { return v_.data(); }
  const double *data() const { return v_.data(); }
  VectorXd operator+(const VectorXd &o) const { VectorXd r(size()); for (int i = 0; i < size(); ++i) r(i) = v_[i] + o(i); return r; }
  VectorXd operator-(const VectorXd &o) const { VectorXd r(size()); for (int i = 0; i < size(); ++i) r(i) = v_[i] - o(i); return r; }
  VectorXd operator*(double s) const { VectorXd r(size()); for (int i = 0; i < size(); ++i) r(i) = v_[i] * s; return r; }
  double norm() const { double s = 0; for (double x : v_) s += x * x; return std::sqrt(s); }
 private:
  std::vector<double> v_;
};

class MatrixXd {
 public:
  MatrixXd() : r_(0), c_(0) {}
  MatrixXd(int r, int c) : r_(r), c_(c), v_((size_t)r * c, 0.0) {}
  static MatrixXd Zero(int r, int c) { return MatrixXd(r, c); }
  int rows() const { return r_; }
  int cols() const { return c_; }
  void resize(int r, int c) { r_ = r; c_ = c; v_.assign((size_t)r * c, 0.0); }
  double &operator()(int r, int c) { return v_[(size_t)r * c_ + c]; }
  double operator()(int r, int c) const { return v_[(size_t)r * c_ + c]; }
  double *data() { return v_.data(); }
  const double *data() const { return v_.data(); }
 private:
  int r_, c_;
  std::vector<double> v_;
};

class ArrayXb {
 public:
  ArrayXb() {}
  explicit ArrayXb(int n) : v_(n, 0) {}
  int size() const { return (int)v_.size(); }
  int rows() const { return size(); }
  void resize(int n) { v_.assign(n, 0); }
  unsigned char &operator()(int i) { return v_[i]; }
  bool operator()(int i) const { return v_[i] != 0; }
  const unsigned char *data() const { return v_.data(); }
  int count() const { int c = 0; for (auto b : v_) c += b != 0; return c; }
 private:
  std::vector<unsigned char> v_;
};

#endif
