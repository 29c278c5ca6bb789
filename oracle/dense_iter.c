/* dense_iter.c -- TEST INFRASTRUCTURE ONLY (see egs_oracle.h): the reference's iterations on an EXPLICIT dense
 * matrix, restated:
 *   BaseIteration(A, b, type, C, x_lo, x_hi)      sparse_iterations.cc:72-144
 *   GetResidualError(A, b, x, C, x_lo, x_hi)       sparse_iterations.cc:35-49
 *   MatrixSolveDiagonal / LowerTriangle / UpperTriangle (dense twins)   sparse_iterations_utils.cc:25-40, 110-128, 245-262
 *   ApplyProjection                                sparse_iterations_utils.cc:12-21
 * The splitting A = M - N: Jacobi M = diag, N = -(L + U); Gauss-Seidel M = lower triangle with the diagonal,
 * N = -strict upper; backward SOR M = strict upper + kSOR diag, N = -(strict lower) + (kSOR - 1) diag, kSOR = 1 / omega
 * (:91-110).  x0 = b (:124); stop at err <= tol or after max_iters sweeps (:128-141).
 * What is NOT restated: the spectral-radius gate (:113-121: EigenSolver on M^-1 N, CHECK(rho < 1) -> Panic); Eigen is
 * absent here.  A splitting that does not converge simply runs to the cap.
 * Operation order: N x + b as a row sum in increasing column order, then + b; the triangular solves exactly as the
 * reference's scalar loops (`substitutions += L(i, j) * x(j)`, j increasing resp. from i + 1 upwards). */
#include <math.h>
#include <stdlib.h>

#include "egs_oracle.h"

static double dproj(double x, int is_eq, double lo, double hi) {   /* utils.cc:12-21 */
  if (is_eq) return x;
  if (x < lo) return lo;
  if (x > hi) return hi;
  return x;
}

double orc_dense_residual(int n, const double *A, const double *b, const double *x, const uint8_t *C, const double *lo,
                          const double *hi) {
  double s_eq = 0.0, s_lo = 0.0, s_hi = 0.0, s_in = 0.0;
  for (int i = 0; i < n; ++i) {
    double w = 0.0;
    for (int j = 0; j < n; ++j) w = w + A[(size_t)i * n + j] * x[j];
    w = w - b[i];
    if (C[i]) s_eq += w * w;
    else {
      if (x[i] == lo[i] && w < 0) s_lo += w * w;
      if (x[i] == hi[i] && w > 0) s_hi += w * w;
      if (x[i] > lo[i] && x[i] < hi[i]) s_in += w * w;
    }
  }
  return sqrt(s_eq) + (sqrt(s_lo) + sqrt(s_hi) + sqrt(s_in));
}

/* method 0 Jacobi, 1 Gauss-Seidel (forward), 2 SOR (backward).  Returns the number of sweeps done. */
int orc_dense_iterate(int n, const double *A, const double *b, const uint8_t *C, const double *lo, const double *hi,
                      int method, double omega, int max_iters, double tol, double *x, double *residual_out) {
  if (n <= 0) { if (residual_out) *residual_out = 0.0; return 0; }
  const double ksor = 1.0 / omega;
  double *rhs = (double *)malloc(sizeof(double) * (size_t)n * 2);
  double *xn = rhs + n;
  for (int i = 0; i < n; ++i) x[i] = b[i];
  double err = orc_dense_residual(n, A, b, x, C, lo, hi);
  int it = 0;
  while (err > tol && it < max_iters) {
    for (int i = 0; i < n; ++i) {                 /* rhs = N x + b */
      double t = 0.0;
      if (method == 0) { for (int j = 0; j < n; ++j) if (j != i) t = t + (-A[(size_t)i * n + j]) * x[j]; }
      else if (method == 1) { for (int j = i + 1; j < n; ++j) t = t + (-A[(size_t)i * n + j]) * x[j]; }
      else {
        for (int j = 0; j < i; ++j) t = t + (-A[(size_t)i * n + j]) * x[j];
        t = t + ((ksor - 1.0) * A[(size_t)i * n + i]) * x[i];
      }
      rhs[i] = t + b[i];
    }
    if (method == 0) {
      for (int i = 0; i < n; ++i) xn[i] = dproj(1.0 / A[(size_t)i * n + i] * rhs[i], C[i], lo[i], hi[i]);
    } else if (method == 1) {
      for (int i = 0; i < n; ++i) {
        double sub = 0.0;
        for (int j = 0; j < i; ++j) sub += A[(size_t)i * n + j] * xn[j];
        xn[i] = dproj((rhs[i] - sub) / A[(size_t)i * n + i], C[i], lo[i], hi[i]);
      }
    } else {
      for (int i = n - 1; i >= 0; --i) {
        double sub = 0.0;
        for (int j = i + 1; j < n; ++j) sub += A[(size_t)i * n + j] * xn[j];
        xn[i] = dproj((rhs[i] - sub) / (ksor * A[(size_t)i * n + i]), C[i], lo[i], hi[i]);
      }
    }
    for (int i = 0; i < n; ++i) x[i] = xn[i];
    err = orc_dense_residual(n, A, b, x, C, lo, hi);
    ++it;
  }
  if (residual_out) *residual_out = err;
  free(rhs);
  return it;
}
