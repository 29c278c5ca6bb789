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
#undef REAL
#undef FMA
#undef SQRT
#undef NAME

#define REAL float
#define FMA fmaf
#define SQRT sqrtf
#define NAME(x) f32_##x
#include "pgs_fast.inc"
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
