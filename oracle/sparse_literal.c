/* sparse_literal.c -- CPU ORACLE (test infrastructure): literal restatement of
 * the reference's matrix-free O(m^2) algorithms, eggshell/
 * sparse_iterations_utils.cc and sparse_iterations.cc, on flat arrays.
 * The reference re-invokes the virtual ComputeJ for every (i,j) pair; the
 * blocks it gets back are the same values every time, so reading them from
 * the precomputed J0/J1 arrays is equivalent.  Everything else -- pair loops,
 * the if/else-if block selection, triangle conventions, projection -- follows
 * the reference line by line, including (optionally) quirk Q1. */
#include <stdlib.h>

#include "egs_oracle.h"
#include "linalg.h"

#define ROWS 3

/* out(3x3) += Ji(3x6) * W(6x6) * Jj(3x6)^T, evaluated as (Ji*W)*Jj^T. */
static void add_JWJt(const double *Ji, const double *W, const double *Jj,
                     double *out) {
  double T[18];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 6; ++c) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += Ji[6 * r + k] * W[6 * k + c];
      T[6 * r + c] = s;
    }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += T[6 * r + k] * Jj[6 * c + k];
      out[3 * r + c] += s;
    }
}

/* sparse_iterations_utils.cc:184-199 (same at :323-338, 453-463, 535-545,
 * 664-679): the (i,j) off-diagonal block. */
static void offdiag_block(const orc_system *s, int i, int j, double *blk) {
  for (int k = 0; k < 9; ++k) blk[k] = 0;
  const int i0 = s->body0[i], i1 = s->body1[i];
  const int j0 = s->body0[j], j1 = s->body1[j];
  const double *Ji0 = s->J0 + 18 * i, *Ji1 = s->J1 + 18 * i;
  const double *Jj0 = s->J0 + 18 * j, *Jj1 = s->J1 + 18 * j;
  if (i0 == j0 && i0 >= 0) {
    add_JWJt(Ji0, s->Minv + 36 * i0, Jj0, blk);
  } else if (i0 == j1 && i0 >= 0) {
    add_JWJt(Ji0, s->Minv + 36 * i0, Jj1, blk);
  }
  if (i1 == j0 && i1 >= 0) {
    add_JWJt(Ji1, s->Minv + 36 * i1, Jj0, blk);
  } else if (i1 == j1 && i1 >= 0) {
    add_JWJt(Ji1, s->Minv + 36 * i1, Jj1, blk);
  }
}

/* :206-217 etc.: the diagonal block, without epsilon. */
static void diag_block(const orc_system *s, int i, double *blk) {
  for (int k = 0; k < 9; ++k) blk[k] = 0;
  const int i0 = s->body0[i], i1 = s->body1[i];
  if (i0 >= 0) add_JWJt(s->J0 + 18 * i, s->Minv + 36 * i0, s->J0 + 18 * i, blk);
  if (i1 >= 0) add_JWJt(s->J1 + 18 * i, s->Minv + 36 * i1, s->J1 + 18 * i, blk);
}

static void add_blk_x(const double *blk, const double *x, double *out) {
  for (int r = 0; r < 3; ++r) {
    double t = 0;
    for (int c = 0; c < 3; ++c) t += blk[3 * r + c] * x[c];
    out[r] += t;
  }
}

/* sparse_iterations_utils.cc:427-493 */
void orc_lit_Lx(const orc_system *s, const double *x, double *out) {
  for (int k = 0; k < ROWS * s->m; ++k) out[k] = 0;
  for (int i = 0; i < s->m; ++i) {
    double blk[9];
    for (int j = 0; j < i; ++j) {
      offdiag_block(s, i, j, blk);
      add_blk_x(blk, x + 3 * j, out + 3 * i);
    }
    diag_block(s, i, blk);
    /* strictly lower triangle of the diagonal block (:484-486) */
    blk[0] = blk[1] = blk[2] = 0; blk[4] = blk[5] = 0; blk[8] = 0;
    add_blk_x(blk, x + 3 * i, out + 3 * i);
  }
}

/* sparse_iterations_utils.cc:495-561 */
void orc_lit_Ux(const orc_system *s, const double *x, double *out) {
  for (int k = 0; k < ROWS * s->m; ++k) out[k] = 0;
  for (int i = 0; i < s->m; ++i) {
    double blk[9];
    diag_block(s, i, blk);
    blk[0] = 0; blk[3] = blk[4] = 0; blk[6] = blk[7] = blk[8] = 0; /* :522-524 */
    add_blk_x(blk, x + 3 * i, out + 3 * i);
    for (int j = i + 1; j < s->m; ++j) {
      offdiag_block(s, i, j, blk);
      add_blk_x(blk, x + 3 * j, out + 3 * i); /* Q2: 3 rows per constraint */
    }
  }
}

/* sparse_iterations_utils.cc:571-603 */
void orc_lit_Dx(const orc_system *s, const double *x, double eps, double scale,
                double *out) {
  for (int i = 0; i < s->m; ++i) {
    double blk[9];
    diag_block(s, i, blk);
    for (int k = 0; k < 3; ++k)
      out[3 * i + k] = ((blk[4 * k] + eps) * scale) * x[3 * i + k]; /* :594-597 */
  }
}

/* sparse_iterations_utils.cc:624-695 */
void orc_lit_JMJtX(const orc_system *s, const double *x, double eps,
                   double *out) {
  for (int k = 0; k < ROWS * s->m; ++k) out[k] = 0;
  for (int i = 0; i < s->m; ++i) {
    double blk[9];
    for (int j = 0; j < s->m; ++j) {
      if (i == j) {
        diag_block(s, i, blk);
        blk[0] += eps; blk[4] += eps; blk[8] += eps; /* :660-663 */
      } else {
        offdiag_block(s, i, j, blk);
      }
      add_blk_x(blk, x + 3 * j, out + 3 * i);
    }
  }
}

/* sparse_iterations_utils.cc:12-21 */
static double apply_projection(double x, int C, double lo, double hi) {
  if (!C) {
    if (x < lo) return lo;
    else if (x > hi) return hi;
  }
  return x;
}

/* sparse_iterations_utils.cc:67-108 */
void orc_lit_solve_diag(const orc_system *s, const double *rhs, double eps,
                        double scale, double *x) {
  for (int i = 0; i < s->m; ++i) {
    double blk[9];
    diag_block(s, i, blk);
    for (int k = 0; k < 3; ++k) {
      double d = (blk[4 * k] + eps) * scale; /* :92-93 */
      double t = 1.0 / d * rhs[3 * i + k];   /* :95-97 */
      x[3 * i + k] = apply_projection(t, s->is_eq[3 * i + k], s->lo[3 * i + k],
                                      s->hi[3 * i + k]);
    }
  }
}

/* sparse_iterations_utils.cc:159-243.  Q1: in the reference the inner j loop
 * overwrites ct/c_lo/c_hi (:180), so for i>0 the projection at :229-235 uses
 * constraint i-1's type and bounds. quirks!=0 reproduces that. */
void orc_lit_solve_lower(const orc_system *s, const double *rhs, double eps,
                         double scale, int quirks, double *x) {
  for (int k = 0; k < ROWS * s->m; ++k) x[k] = 0;
  for (int i = 0; i < s->m; ++i) {
    double sub[3] = {0, 0, 0}, blk[9];
    for (int j = 0; j < i; ++j) {
      offdiag_block(s, i, j, blk);
      add_blk_x(blk, x + 3 * j, sub);
    }
    diag_block(s, i, blk);
    for (int k = 0; k < 3; ++k) blk[4 * k] = (blk[4 * k] + eps) * scale; /* :222-226 */
    const int pc = (quirks && i > 0) ? i - 1 : i; /* whose ct/lo/hi is live */
    for (int k = 0; k < 3; ++k) {
      for (int l = 0; l < k; ++l) sub[k] += blk[3 * k + l] * x[3 * i + l];
      x[3 * i + k] = apply_projection((rhs[3 * i + k] - sub[k]) / blk[4 * k],
                                      s->is_eq[3 * pc + k], s->lo[3 * pc + k],
                                      s->hi[3 * pc + k]);
    }
  }
}

/* sparse_iterations_utils.cc:292-373.  Q1 here uses constraint i+1's. */
void orc_lit_solve_upper(const orc_system *s, const double *rhs, double eps,
                         double scale, int quirks, double *x) {
  for (int k = 0; k < ROWS * s->m; ++k) x[k] = 0;
  for (int i = s->m - 1; i >= 0; --i) {
    double sub[3] = {0, 0, 0}, blk[9];
    for (int j = s->m - 1; j > i; --j) {
      offdiag_block(s, i, j, blk);
      add_blk_x(blk, x + 3 * j, sub);
    }
    diag_block(s, i, blk);
    for (int k = 0; k < 3; ++k) blk[4 * k] = (blk[4 * k] + eps) * scale; /* :355-359 */
    const int pc = (quirks && i < s->m - 1) ? i + 1 : i;
    for (int k = 2; k >= 0; --k) {
      for (int l = k + 1; l < 3; ++l) sub[k] += blk[3 * k + l] * x[3 * i + l];
      x[3 * i + k] = apply_projection((rhs[3 * i + k] - sub[k]) / blk[4 * k],
                                      s->is_eq[3 * pc + k], s->lo[3 * pc + k],
                                      s->hi[3 * pc + k]);
    }
  }
}

/* sparse_iterations.cc:51-69: four partial 2-norms, summed. */
static double residual_from_w(const orc_system *s, const double *w,
                              const double *x) {
  double e = 0, a = 0, b = 0, c = 0;
  for (int r = 0; r < ROWS * s->m; ++r) {
    if (s->is_eq[r]) e += w[r] * w[r];
    else {
      if (x[r] == s->lo[r] && w[r] < 0) a += w[r] * w[r];
      if (x[r] == s->hi[r] && w[r] > 0) b += w[r] * w[r];
      if (x[r] > s->lo[r] && x[r] < s->hi[r]) c += w[r] * w[r];
    }
  }
  return sqrt(e) + (sqrt(a) + sqrt(b) + sqrt(c));
}

double orc_lit_residual(const orc_system *s, const double *rhs, const double *x,
                        double cfm) {
  const int R = ROWS * s->m;
  double *w = (double *)malloc(sizeof(double) * (R > 0 ? R : 1));
  orc_lit_JMJtX(s, x, cfm, w);
  for (int r = 0; r < R; ++r) w[r] -= rhs[r];
  double res = residual_from_w(s, w, x);
  free(w);
  return res;
}

/* sparse_iterations.cc:148-226 */
int orc_lit_iterate(const orc_system *s, const double *rhs, double cfm,
                    int method, double omega, int max_iters, double tol,
                    int quirks, double *x, double *residual_out) {
  const int R = ROWS * s->m;
  if (s->m == 0) { if (residual_out) *residual_out = 0; return 0; }
  const double kSOR = 1.0 / omega;
  double *nx = (double *)malloc(sizeof(double) * R);
  double *t = (double *)malloc(sizeof(double) * R);
  double *it_rhs = (double *)malloc(sizeof(double) * R);
  for (int r = 0; r < R; ++r) x[r] = rhs[r]; /* :202, quirk Q7 */
  int i = 0;
  double err = orc_lit_residual(s, rhs, x, cfm);
  while ((tol <= 0 || err > tol) && i < max_iters) {
    if (method == 0) { /* JACOBI: LxUx, diag solve */
      orc_lit_Lx(s, x, nx);
      orc_lit_Ux(s, x, t);
      for (int r = 0; r < R; ++r) nx[r] += t[r];
    } else if (method == 1) { /* GS: Ux, lower solve */
      orc_lit_Ux(s, x, nx);
    } else { /* SOR backward: Lx + (1-k)(D+eps)x, upper solve with k*D */
      orc_lit_Lx(s, x, nx);
      orc_lit_Dx(s, x, cfm, 1.0 - kSOR, t);
      for (int r = 0; r < R; ++r) nx[r] += t[r];
    }
    for (int r = 0; r < R; ++r) it_rhs[r] = -1.0 * nx[r] + rhs[r]; /* :210-211 */
    if (method == 0) orc_lit_solve_diag(s, it_rhs, cfm, 1.0, x);
    else if (method == 1) orc_lit_solve_lower(s, it_rhs, cfm, 1.0, quirks, x);
    else orc_lit_solve_upper(s, it_rhs, cfm, kSOR, quirks, x);
    err = orc_lit_residual(s, rhs, x, cfm);
    ++i;
  }
  if (residual_out) *residual_out = err;
  free(nx); free(t); free(it_rhs);
  return i;
}

/* dense J Minv J^T + eps I (ensembles.cc:510), for the dense twins/tests. */
void orc_dense_JMJt(const orc_system *s, double eps, double *A) {
  const int R = ROWS * s->m;
  for (long k = 0; k < (long)R * R; ++k) A[k] = 0;
  for (int i = 0; i < s->m; ++i)
    for (int j = 0; j < s->m; ++j) {
      double blk[9];
      if (i == j) { diag_block(s, i, blk); blk[0] += eps; blk[4] += eps; blk[8] += eps; }
      else offdiag_block(s, i, j, blk);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
          A[(long)(3 * i + r) * R + 3 * j + c] = blk[3 * r + c];
    }
}
