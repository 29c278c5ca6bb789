/* lcp_dense.c -- CPU ORACLE (test infrastructure): restatement of the
 * reference's dense direct LCP, eggshell/lcp.cc: CheckMurtySolution (:20-103),
 * best-solution memory (:105-137), MurtyPrincipalPivot (:157-274) and
 * MixedConstraintsSolver (:276-336).  Eigen's ldlt()/inverse() are replaced by
 * an LDL^T with diagonal pivoting written here (same results to rounding). */
#include <stdlib.h>
#include <string.h>

#include "egs_oracle.h"
#include "linalg.h"

/* Solve A x = b, A symmetric (n x n, row-major, destroyed), LDL^T with
 * symmetric diagonal pivoting (largest |diagonal|), as Eigen::LDLT does. */
static void ldlt_solve_inplace(int n, double *A, double *b, int nrhs) {
  /* b is n x nrhs row-major, overwritten with the solution */
  if (n == 0) return;
  int *perm = (int *)malloc(sizeof(int) * n);
  for (int i = 0; i < n; ++i) perm[i] = i;
  double *tmp = (double *)malloc(sizeof(double) * (n > nrhs ? n : nrhs));
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = fabs(A[(long)k * n + k]);
    for (int i = k + 1; i < n; ++i) {
      double v = fabs(A[(long)i * n + i]);
      if (v > best) { best = v; piv = i; }
    }
    if (piv != k) { /* symmetric swap of rows/cols k and piv */
      for (int j = 0; j < n; ++j) {
        double t = A[(long)k * n + j]; A[(long)k * n + j] = A[(long)piv * n + j]; A[(long)piv * n + j] = t;
      }
      for (int i = 0; i < n; ++i) {
        double t = A[(long)i * n + k]; A[(long)i * n + k] = A[(long)i * n + piv]; A[(long)i * n + piv] = t;
      }
      for (int j = 0; j < nrhs; ++j) {
        double t = b[(long)k * nrhs + j]; b[(long)k * nrhs + j] = b[(long)piv * nrhs + j]; b[(long)piv * nrhs + j] = t;
      }
      int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    double d = A[(long)k * n + k];
    if (d == 0) continue;
    for (int i = k + 1; i < n; ++i) {
      double l = A[(long)i * n + k] / d;
      if (l == 0) { A[(long)i * n + k] = 0; continue; }
      for (int j = k + 1; j <= i; ++j) A[(long)i * n + j] -= l * A[(long)k * n + j];
      A[(long)i * n + k] = l;
    }
    /* keep the upper triangle mirrored for the next pivot search/swaps */
    for (int i = k + 1; i < n; ++i)
      for (int j = k + 1; j < i; ++j) A[(long)j * n + i] = A[(long)i * n + j];
    for (int i = k + 1; i < n; ++i) A[(long)k * n + i] = A[(long)i * n + k] * d;
  }
  /* forward: L y = b */
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < i; ++k) {
      double l = A[(long)i * n + k];
      if (l != 0)
        for (int j = 0; j < nrhs; ++j) b[(long)i * nrhs + j] -= l * b[(long)k * nrhs + j];
    }
  for (int i = 0; i < n; ++i) {
    double d = A[(long)i * n + i];
    for (int j = 0; j < nrhs; ++j) b[(long)i * nrhs + j] = (d != 0) ? b[(long)i * nrhs + j] / d : 0.0;
  }
  for (int i = n - 1; i >= 0; --i)
    for (int k = i + 1; k < n; ++k) {
      double l = A[(long)k * n + i];
      if (l != 0)
        for (int j = 0; j < nrhs; ++j) b[(long)i * nrhs + j] -= l * b[(long)k * nrhs + j];
    }
  /* undo permutation: solution row i belongs to original index perm[i] */
  for (int j = 0; j < nrhs; ++j) {
    for (int i = 0; i < n; ++i) tmp[perm[i]] = b[(long)i * nrhs + j];
    for (int i = 0; i < n; ++i) b[(long)i * nrhs + j] = tmp[i];
  }
  free(perm); free(tmp);
}

/* lcp.cc:20-103 */
int orc_check_murty(int dim, const double *A, const double *b, const double *x,
                    const double *w, uint8_t *S, double *C, const double *lo,
                    const double *hi, double err) {
  const double tol = fabs(err) > 1e-9 ? fabs(err) : 1e-9; /* :29-31 */
  for (int i = 0; i < dim; ++i) {
    if (S[i]) {
      if (x[i] < lo[i]) { S[i] = 0; C[i] = lo[i]; return 0; }
      else if (x[i] > hi[i]) { S[i] = 0; C[i] = hi[i]; return 0; }
    } else {
      if (C[i] == lo[i] && w[i] < 0) { S[i] = 1; return 0; }
      else if (C[i] == hi[i] && w[i] > 0) { S[i] = 1; return 0; }
    }
  }
  for (int i = 0; i < dim; ++i)
    if (x[i] < lo[i] || x[i] > hi[i]) return 0; /* :66 */
  for (int i = 0; i < dim; ++i) {
    if (x[i] == lo[i] && w[i] < 0) return 0; /* :72-76 */
    if (x[i] == hi[i] && w[i] > 0) return 0;
  }
  double nrm = 0;
  for (int i = 0; i < dim; ++i) {
    double s = 0;
    for (int j = 0; j < dim; ++j) s += A[(long)i * dim + j] * x[j];
    double d = s - (b[i] + w[i]);
    nrm += d * d;
  }
  if (sqrt(nrm) > tol) return 0; /* :83-85 */
  return 1;
}

/* lcp.cc:107-113 */
static double goodness(int dim, const double *x, const double *w) {
  double g = 0;
  for (int i = 0; i < dim; ++i) {
    if (!(x[i] > 0)) g += x[i];
    if (!(w[i] > 0)) g += w[i];
  }
  return g;
}

/* lcp.cc:157-274.  box_fix != 0 is NOT the reference: it adds the A(.,!S)x(!S)
 * terms the reference omits, so that bounded problems are solved correctly
 * (used only by orc_mixed_constraints(use_bounds=1)). */
static int murty_impl(int dim, const double *A, const double *b,
                      const double *lo, const double *hi, double *x, double *w,
                      int *pivots_out, int box_fix) {
  for (int i = 0; i < dim; ++i) /* :161-164 CHECKs */
    if (!(lo[i] < hi[i]) || !(lo[i] <= 0) || !(hi[i] > 0)) return 0;
  double p2 = pow(2.0, dim);
  const int max_iterations = p2 > 1000 ? 1000 : (int)p2; /* :168 */
  int iter = 0, pivots = 0;
  uint8_t *S = (uint8_t *)malloc(dim + 1);
  double *C = (double *)malloc(sizeof(double) * (dim + 1));
  double *bx = (double *)malloc(sizeof(double) * (dim + 1));
  double *bw = (double *)malloc(sizeof(double) * (dim + 1));
  double *sub = (double *)malloc(sizeof(double) * ((long)dim * dim + 1));
  double *rs = (double *)malloc(sizeof(double) * (dim + 1));
  int *idx = (int *)malloc(sizeof(int) * (dim + 1));
  for (int i = 0; i < dim; ++i) {
    S[i] = 1; x[i] = 0; w[i] = -b[i]; C[i] = lo[i]; /* :176-189 */
    bx[i] = x[i]; bw[i] = w[i];
  }
  /* box_fix only: the reference's start (x = 0, w = -b) passes its own check
   * whenever lo < 0 < hi ("WithBound solutions are trivial", lcp.cc:181), so
   * the corrected variant solves once for S = all before the first check. */
  int force = box_fix;
  while (iter < max_iterations) {
    if (force || !orc_check_murty(dim, A, b, x, w, S, C, lo, hi, 0)) {
      force = 0;
      int ns = 0;
      for (int i = 0; i < dim; ++i) if (S[i]) idx[ns++] = i;
      /* x(!S) first (values do not depend on the solve) :208-216 */
      for (int i = 0; i < dim; ++i)
        if (!S[i]) { if (C[i] == lo[i]) x[i] = lo[i]; if (C[i] == hi[i]) x[i] = hi[i]; }
      for (int r = 0; r < ns; ++r) {
        for (int c = 0; c < ns; ++c) sub[(long)r * ns + c] = A[(long)idx[r] * dim + idx[c]];
        double rhs = b[idx[r]];
        if (box_fix)
          for (int j = 0; j < dim; ++j) if (!S[j]) rhs -= A[(long)idx[r] * dim + j] * x[j];
        rs[r] = rhs;
      }
      ldlt_solve_inplace(ns, sub, rs, 1); /* :202-203 */
      for (int r = 0; r < ns; ++r) x[idx[r]] = rs[r];
      for (int i = 0; i < dim; ++i) { /* :219-223 */
        if (S[i]) { w[i] = 0; continue; }
        double s = 0;
        for (int r = 0; r < ns; ++r) s += A[(long)i * dim + idx[r]] * x[idx[r]];
        if (box_fix)
          for (int j = 0; j < dim; ++j) if (!S[j]) s += A[(long)i * dim + j] * x[j];
        w[i] = s - b[i];
      }
      ++pivots;
      /* :226, :125-137 */
      int same = 1;
      for (int i = 0; i < dim && same; ++i) if (x[i] != bx[i] || w[i] != bw[i]) same = 0;
      if (!same && goodness(dim, x, w) > goodness(dim, bx, bw)) {
        memcpy(bx, x, sizeof(double) * dim);
        memcpy(bw, w, sizeof(double) * dim);
      }
    } else {
      break;
    }
    ++iter;
  }
  if (!box_fix) { /* the "goodness" ranking assumes lo = 0; box_fix keeps the last iterate */
    memcpy(x, bx, sizeof(double) * dim); /* :241-242 */
    memcpy(w, bw, sizeof(double) * dim);
  }
  int ok = orc_check_murty(dim, A, b, x, w, S, C, lo, hi,
                           iter >= max_iterations ? 1e-8 : 0); /* :244-249 */
  if (pivots_out) *pivots_out = pivots;
  free(S); free(C); free(bx); free(bw); free(sub); free(rs); free(idx);
  return ok;
}

int orc_murty(int dim, const double *A, const double *b, const double *lo,
              const double *hi, double *x, double *w, int *pivots_out) {
  return murty_impl(dim, A, b, lo, hi, x, w, pivots_out, 0);
}

/* lcp.cc:276-336 */
int orc_mixed_constraints(int dim, const double *A, const double *b,
                          const uint8_t *C, const double *lo, const double *hi,
                          int use_bounds, double *x, double *w,
                          int *pivots_out) {
  int ne = 0, ni = 0;
  int *ie = (int *)malloc(sizeof(int) * (dim + 1));
  int *ii = (int *)malloc(sizeof(int) * (dim + 1));
  for (int i = 0; i < dim; ++i) { if (C[i]) ie[ne++] = i; else ii[ni++] = i; }
  double *Aee = (double *)malloc(sizeof(double) * ((long)ne * ne + 1));
  /* X = A_ee^-1 [A_ei | b_e]  (ne x (ni+1)) */
  double *X = (double *)malloc(sizeof(double) * ((long)ne * (ni + 1) + 1));
  for (int r = 0; r < ne; ++r) {
    for (int c = 0; c < ne; ++c) Aee[(long)r * ne + c] = A[(long)ie[r] * dim + ie[c]];
    for (int c = 0; c < ni; ++c) X[(long)r * (ni + 1) + c] = A[(long)ie[r] * dim + ii[c]];
    X[(long)r * (ni + 1) + ni] = b[ie[r]];
  }
  ldlt_solve_inplace(ne, Aee, X, ni + 1);
  /* lhs = A_ii - A_ie X[:, :ni];  rhs = b_i - A_ie X[:, ni]   (:293-294) */
  double *lhs = (double *)malloc(sizeof(double) * ((long)ni * ni + 1));
  double *rhs = (double *)malloc(sizeof(double) * (ni + 1));
  for (int r = 0; r < ni; ++r) {
    for (int c = 0; c <= ni; ++c) {
      double s = 0;
      for (int k = 0; k < ne; ++k) s += A[(long)ii[r] * dim + ie[k]] * X[(long)k * (ni + 1) + c];
      if (c < ni) lhs[(long)r * ni + c] = A[(long)ii[r] * dim + ii[c]] - s;
      else rhs[r] = b[ii[r]] - s;
    }
  }
  double *xi = (double *)malloc(sizeof(double) * (ni + 1));
  double *wi = (double *)malloc(sizeof(double) * (ni + 1));
  double *l2 = (double *)malloc(sizeof(double) * (ni + 1));
  double *h2 = (double *)malloc(sizeof(double) * (ni + 1));
  for (int r = 0; r < ni; ++r) {
    /* :298 calls the no-bounds overload: 0 <= x < inf (quirk Q3) */
    l2[r] = use_bounds ? lo[ii[r]] : 0.0;
    h2[r] = use_bounds ? hi[ii[r]] : INFINITY;
  }
  int ok = murty_impl(ni, lhs, rhs, l2, h2, xi, wi, pivots_out, use_bounds);
  if (ok) {
    /* x_e = A_ee.ldlt().solve(b_e - A_ei x_i)  (:317) */
    double *be = (double *)malloc(sizeof(double) * (ne + 1));
    for (int r = 0; r < ne; ++r) {
      for (int c = 0; c < ne; ++c) Aee[(long)r * ne + c] = A[(long)ie[r] * dim + ie[c]];
      double s = 0;
      for (int c = 0; c < ni; ++c) s += A[(long)ie[r] * dim + ii[c]] * xi[c];
      be[r] = b[ie[r]] - s;
    }
    ldlt_solve_inplace(ne, Aee, be, 1);
    for (int i = 0; i < dim; ++i) w[i] = 0;
    for (int r = 0; r < ne; ++r) x[ie[r]] = be[r];
    for (int r = 0; r < ni; ++r) { x[ii[r]] = xi[r]; w[ii[r]] = wi[r]; }
    free(be);
  }
  free(ie); free(ii); free(Aee); free(X); free(lhs); free(rhs);
  free(xi); free(wi); free(l2); free(h2);
  return ok;
}
