// ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.  Tiny dense linear algebra + SO(3) helpers for the BA oracle.
// The reference uses Eigen 3.3.7 and Sophus (third-party, absent here); these restate the few
// closed-form operations it needs.  SO(3) functions follow common/so3_extra.h line by line.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

namespace vo {

typedef double V3[3];
typedef double M3[9];  // row-major

static inline void m3_mul(const double* A, const double* B, double* C) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  memcpy(C, t, sizeof(t));
}
static inline void m3_T(const double* A, double* C) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[i * 3 + j] = A[j * 3 + i];
  memcpy(C, t, sizeof(t));
}
static inline void m3_v(const double* A, const double* v, double* r) {
  double t[3];
  for (int i = 0; i < 3; i++) t[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
  memcpy(r, t, sizeof(t));
}
static inline void m3T_v(const double* A, const double* v, double* r) {
  double t[3];
  for (int i = 0; i < 3; i++) t[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
  memcpy(r, t, sizeof(t));
}
static inline void hat(const double* w, double* O) {  // so3_extra.h:108-112
  O[0] = 0, O[1] = -w[2], O[2] = w[1];
  O[3] = w[2], O[4] = 0, O[5] = -w[0];
  O[6] = -w[1], O[7] = w[0], O[8] = 0;
}
static inline void m3_identity(double* A) {
  memset(A, 0, 72);
  A[0] = A[4] = A[8] = 1;
}

// Unit quaternion (w, x, y, z)
struct Quat {
  double w = 1, x = 0, y = 0, z = 0;
};
static inline void quat_normalize(Quat& q) {
  double n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  q.w /= n, q.x /= n, q.y /= n, q.z /= n;
}
static inline Quat quat_mul(const Quat& a, const Quat& b) {  // Eigen quaternion product
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
static inline void quat_to_R(const Quat& q, double* R) {  // Eigen toRotationMatrix
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz), R[1] = txy - twz, R[2] = txz + twy;
  R[3] = txy + twz, R[4] = 1 - (txx + tzz), R[5] = tyz - twx;
  R[6] = txz - twy, R[7] = tyz + twx, R[8] = 1 - (txx + tyy);
}
static inline Quat R_to_quat(const double* R) {  // Eigen Quaternion(Matrix3) + normalize
  Quat q;
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (R[7] - R[5]) * t;
    q.y = (R[2] - R[6]) * t;
    q.z = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (R[k * 3 + j] - R[j * 3 + k]) * t;
    v[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    v[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    q.x = v[0], q.y = v[1], q.z = v[2];
  }
  quat_normalize(q);
  return q;
}

static const double SO3_SMALL_EPS = 1e-5;  // so3_extra.h:233

// SO3ex::exp (so3_extra.h:121-142)
static inline Quat so3_exp(const double* omega) {
  double theta = std::sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  double imag, real;
  if (theta < SO3_SMALL_EPS) {
    double theta_sq = theta * theta;
    imag = 0.5 - theta_sq / 48.;
    real = 1.0 - theta_sq / 8.;
  } else {
    double half = 0.5 * theta;
    imag = std::sin(half) / theta;
    real = std::cos(half);
  }
  Quat q;
  q.w = real, q.x = imag * omega[0], q.y = imag * omega[1], q.z = imag * omega[2];
  quat_normalize(q);  // SO3ex(QuaternionBase) ctor normalises (so3_extra.h:60-67)
  return q;
}

// SO3ex::log (so3_extra.h:150-190)
static inline void so3_log(const Quat& q, double* out) {
  double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  double w = q.w;
  double squared_w = w * w;
  double f;
  if (n < SO3_SMALL_EPS) {
    f = 2. / w - 2. / 3 * (n * n) / (w * squared_w);
  } else {
    if (std::fabs(w) < SO3_SMALL_EPS) {
      if (w > 0)
        f = M_PI / n;
      else
        f = -M_PI / n;
      double n2 = n * n, n4 = n2 * n2;
      f -= 2 * w / n2 - 2. / 3 * (w * squared_w) / n4;
    } else
      f = 2 * std::atan(n / w) / n;
  }
  out[0] = f * q.x, out[1] = f * q.y, out[2] = f * q.z;
}

// SO3ex::JacobianR (so3_extra.h:254-270)
static inline void so3_Jr(const double* w, double* J) {
  double theta = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double O[9], O2[9];
  m3_identity(J);
  if (theta < SO3_SMALL_EPS) {
    hat(w, O);
    m3_mul(O, O, O2);
    for (int i = 0; i < 9; i++) J[i] = J[i] - 0.5 * O[i] + O2[i] / 6.;
  } else {
    double k[3] = {w[0] / theta, w[1] / theta, w[2] / theta};
    hat(k, O);
    m3_mul(O, O, O2);
    double a = (1 - std::cos(theta)) / theta, b = 1 - std::sin(theta) / theta;
    for (int i = 0; i < 9; i++) J[i] = J[i] - a * O[i] + b * O2[i];
  }
}

// SO3ex::JacobianRInv (so3_extra.h:271-288)
static inline void so3_JrInv(const double* w, double* J) {
  double theta = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double O[9], O2[9];
  hat(w, O);
  m3_identity(J);
  if (theta < SO3_SMALL_EPS) {
    m3_mul(O, O, O2);
    for (int i = 0; i < 9; i++) J[i] = J[i] + 0.5 * O[i] + (1. / 12.) * O2[i];
  } else {
    double k[3] = {w[0] / theta, w[1] / theta, w[2] / theta};
    double K[9], K2[9];
    hat(k, K);
    m3_mul(K, K, K2);
    double c = 1.0 - (1.0 + std::cos(theta)) * theta / (2.0 * std::sin(theta));
    for (int i = 0; i < 9; i++) J[i] = J[i] + 0.5 * O[i] + c * K2[i];
  }
}

// Dense symmetric solve H x = b (n <= 64) by LDL^T without pivoting; false if a pivot is not
// positive (Eigen LDLT::isPositive()==false in linear_solver_dense.h:107-112).
static inline bool ldlt_solve(const double* H, const double* b, double* x, int n) {
  std::vector<double> L((size_t)n * n, 0.0), D(n, 0.0), y(n, 0.0);
  for (int j = 0; j < n; j++) {
    double d = H[j * n + j];
    for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k] * D[k];
    if (!(d > 0)) return false;
    D[j] = d;
    L[j * n + j] = 1;
    for (int i = j + 1; i < n; i++) {
      double s = H[i * n + j];
      for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k] * D[k];
      L[i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[i * n + k] * y[k];
    y[i] = s;
  }
  for (int i = 0; i < n; i++) y[i] /= D[i];
  for (int i = n - 1; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
    x[i] = s;
  }
  return true;
}

}  // namespace vo
