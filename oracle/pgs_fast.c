/* pgs_fast.c -- CPU ORACLE (test infrastructure): instantiates pgs_fast.inc
 * for fp64 and fp32.  See that file for the algorithm and its citations. */
#include <math.h>
#include <stdlib.h>

#include "egs_oracle.h"

#define REAL double
#define FMA fma
#define SQRT sqrt
#define NAME(x) f64_##x
#include "pgs_fast.inc"
#include "matvec_fast.inc"
#undef REAL
#undef FMA
#undef SQRT
#undef NAME

#define REAL float
#define FMA fmaf
#define SQRT sqrtf
#define NAME(x) f32_##x
#include "pgs_fast.inc"
#include "matvec_fast.inc"
#undef REAL
#undef FMA
#undef SQRT
#undef NAME

int orc_fast_iterate_f64(const orc_system *s, const double *rhs, double cfm,
                         int method, double omega, int max_iters, double tol,
                         int check_every, double *x, double *a_out,
                         double *residual_out) {
  return f64_iterate(s->n, s->m, s->Minv, s->body0, s->body1, s->J0, s->J1,
                     s->is_eq, s->lo, s->hi, rhs, cfm, method, omega,
                     max_iters, tol, check_every, x, a_out, residual_out);
}

int orc_fast_iterate_f32(int n, int m, const float *Minv, const int32_t *body0,
                         const int32_t *body1, const float *J0, const float *J1,
                         const uint8_t *is_eq, const float *lo, const float *hi,
                         const float *rhs, float cfm, int method, float omega,
                         int max_iters, float tol, int check_every, float *x,
                         float *a_out, float *residual_out) {
  return f32_iterate(n, m, Minv, body0, body1, J0, J1, is_eq, lo, hi, rhs, cfm,
                     method, omega, max_iters, tol, check_every, x, a_out,
                     residual_out);
}

/* parts: bit 0 = L, bit 1 = U, bit 2 = D (sums in the reference's order: Lx + Ux,
 * Ux + Dx, Lx + Dx, sparse_iterations_utils.cc:563-569, 606-622); 8 = J W J^T + eps I. */
void orc_fast_matvec_f64(const orc_system *s, const double *x, int parts, double eps,
                         double scale, double *y) {
  const int rows = 3 * s->m;
  if (parts == 8) { f64_matvec(s->n, s->m, s->Minv, s->body0, s->body1, s->J0, s->J1, x, 8, eps, scale, y); return; }
  double *tmp = (double *)malloc(sizeof(double) * (size_t)(rows + 1));
  int first = 1;
  for (int bit = 1; bit <= 4; bit <<= 1) {
    if (!(parts & bit)) continue;
    f64_matvec(s->n, s->m, s->Minv, s->body0, s->body1, s->J0, s->J1, x, bit, eps, scale, first ? y : tmp);
    if (!first) for (int k = 0; k < rows; ++k) y[k] = y[k] + tmp[k];
    first = 0;
  }
  if (first) for (int k = 0; k < rows; ++k) y[k] = 0;
  free(tmp);
}

void orc_fast_matvec_f32(int n, int m, const float *Minv, const int32_t *body0, const int32_t *body1,
                         const float *J0, const float *J1, const float *x, int parts, float eps,
                         float scale, float *y) {
  const int rows = 3 * m;
  if (parts == 8) { f32_matvec(n, m, Minv, body0, body1, J0, J1, x, 8, eps, scale, y); return; }
  float *tmp = (float *)malloc(sizeof(float) * (size_t)(rows + 1));
  int first = 1;
  for (int bit = 1; bit <= 4; bit <<= 1) {
    if (!(parts & bit)) continue;
    f32_matvec(n, m, Minv, body0, body1, J0, J1, x, bit, eps, scale, first ? y : tmp);
    if (!first) for (int k = 0; k < rows; ++k) y[k] = y[k] + tmp[k];
    first = 0;
  }
  if (first) for (int k = 0; k < rows; ++k) y[k] = 0;
  free(tmp);
}

/* element-wise w = A x - rhs from x and the accumulators orc_fast_iterate_f64 returned */
void orc_fast_wres_f64(const orc_system *s, const double *rhs, double cfm, const double *x, const double *a, double *w) {
  f64_wres(s->m, s->body0, s->body1, s->J0, s->J1, rhs, cfm, x, a, w);
}
