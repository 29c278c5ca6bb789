/* linalg.h -- tiny fixed-size helpers for the CPU oracle (test infrastructure).
 * Operation order is spelled out (no FMA contraction: build with
 * -ffp-contract=off) because the HIP kernels mirror it bit for bit. */
#ifndef ORC_LINALG_H
#define ORC_LINALG_H
#include <math.h>
#include <string.h>

static inline double dot3(const double *a, const double *b) {
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
static inline void cross3(const double *a, const double *b, double *o) {
  double x = a[1] * b[2] - a[2] * b[1];
  double y = a[2] * b[0] - a[0] * b[2];
  double z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
/* o = A(3x3 row-major) * v */
static inline void mat3_vec(const double *A, const double *v, double *o) {
  double x = (A[0] * v[0] + A[1] * v[1]) + A[2] * v[2];
  double y = (A[3] * v[0] + A[4] * v[1]) + A[5] * v[2];
  double z = (A[6] * v[0] + A[7] * v[1]) + A[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
/* O = A * B (3x3) */
static inline void mat3_mul(const double *A, const double *B, double *O) {
  double t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      t[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) +
                     A[3 * i + 2] * B[6 + j];
  memcpy(O, t, sizeof t);
}
static inline void mat3_T(const double *A, double *O) {
  double t[9] = {A[0], A[3], A[6], A[1], A[4], A[7], A[2], A[5], A[8]};
  memcpy(O, t, sizeof t);
}
/* utils.cc:16-24 */
static inline void cross_mat(const double *a, double *m) {
  m[0] = 0;     m[1] = -a[2]; m[2] = a[1];
  m[3] = a[2];  m[4] = 0;     m[5] = -a[0];
  m[6] = -a[1]; m[7] = a[0];  m[8] = 0;
}
/* 3x3 inverse by cofactors (what Eigen's fixed-size inverse does). */
static inline void mat3_inv(const double *A, double *O) {
  double c00 = A[4] * A[8] - A[5] * A[7];
  double c10 = A[5] * A[6] - A[3] * A[8];
  double c20 = A[3] * A[7] - A[4] * A[6];
  double det = (A[0] * c00 + A[1] * c10) + A[2] * c20;
  double id = 1.0 / det;
  double t[9];
  t[0] = c00 * id;
  t[1] = (A[2] * A[7] - A[1] * A[8]) * id;
  t[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  t[3] = c10 * id;
  t[4] = (A[0] * A[8] - A[2] * A[6]) * id;
  t[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  t[6] = c20 * id;
  t[7] = (A[1] * A[6] - A[0] * A[7]) * id;
  t[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  memcpy(O, t, sizeof t);
}
/* Eigen quaternion (w,x,y,z) -> rotation matrix (Quaternion::toRotationMatrix) */
static inline void quat_to_R(double w, double x, double y, double z, double *R) {
  double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w;
  double txx = tx * x, txy = ty * x, txz = tz * x;
  double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}
#endif
