/* lcp_toolkit.c -- TEST INFRASTRUCTURE ONLY (see egs_oracle.h): a plain-C restatement of the
 * incremental-factor box LCP of the reference's toolkit/lcp.cc, the `lcp::SolveLCP` family SURVEY
 * row a13 / f4 names:
 *   Cholesky / LSolve / LTSolve / LLTSolve     toolkit/lcp.cc:46-72
 *   RankUpdate                                 toolkit/lcp.cc:76-83  (Eigen internal, see below)
 *   AddCholeskyRow                             toolkit/lcp.cc:91-102
 *   SwapCholeskyRows                           toolkit/lcp.cc:110-157
 *   MatrixPermutation::SwapRowsAndColumns      toolkit/lcp.cc:171-195 (lower triangle only)
 *   SolveLCP_BoxDantzig                        toolkit/lcp.cc:444-619
 * Matrices are row-major n x n with leading dimension n; like the reference, only the lower
 * triangle (row >= column) of A and L is ever read or written.
 *
 * Third-party piece: RankUpdate calls Eigen::internal::llt_rank_update_lower (Eigen 3.3.8 / 3.3.9,
 * pinned by the #error at toolkit/lcp.cc:41-43; Eigen is absent from /root/reference and from this
 * image).  Its published algorithm for a general sigma is the one of Gill, Golub, Murray and
 * Saunders, "Methods for modifying matrix factorizations" (1974), method C1, restated in
 * otk_rank_update; Eigen's faster Givens variant for sigma > 0 gives the same factor up to rounding,
 * which is all the reference's own tests ask of it (vs a refactorisation, 1e-9, toolkit/lcp.cc:
 * 1051-1078).  Operation order inside the triangular solves and products is NOT Eigen's (blocked,
 * vectorised, unknown); it is the plain column-oriented order written here, which the device code
 * repeats so that pivot decisions agree.  Parity of this path is therefore pinned by the
 * reference's property tests (restated in tests/test_oracle_lcp_toolkit.py), not by golden vectors. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "egs_oracle.h"

#define AT(M, r, c) (M)[(size_t)(r) * n + (c)]

/* in-place Cholesky of the lower triangle, column by column (toolkit/lcp.cc:46-48) */
int otk_cholesky(double *L, int n) {
  for (int j = 0; j < n; ++j) {
    double d = AT(L, j, j);
    for (int k = 0; k < j; ++k) d = d - AT(L, j, k) * AT(L, j, k);
    if (!(d > 0.0)) return j + 1;
    d = sqrt(d);
    AT(L, j, j) = d;
    for (int i = j + 1; i < n; ++i) {
      double s = AT(L, i, j);
      for (int k = 0; k < j; ++k) s = s - AT(L, i, k) * AT(L, j, k);
      AT(L, i, j) = s / d;
    }
  }
  return 0;
}

/* L y = b on the top-left m x m block, column-oriented (toolkit/lcp.cc:52-54) */
void otk_lsolve(const double *L, int n, int m, double *x) {
  for (int j = 0; j < m; ++j) {
    x[j] = x[j] / AT(L, j, j);
    for (int k = j + 1; k < m; ++k) x[k] = x[k] - AT(L, k, j) * x[j];
  }
}

/* L' x = y on the top-left m x m block (toolkit/lcp.cc:58-62) */
void otk_ltsolve(const double *L, int n, int m, double *x) {
  for (int j = m - 1; j >= 0; --j) {
    x[j] = x[j] / AT(L, j, j);
    for (int k = 0; k < j; ++k) x[k] = x[k] - AT(L, j, k) * x[j];
  }
}

void otk_lltsolve(const double *L, int n, int m, double *x) {   /* toolkit/lcp.cc:66-69 */
  otk_lsolve(L, n, m, x);
  otk_ltsolve(L, n, m, x);
}

/* rank-one modification of the p x p block of L at (i0, i0): L L' += sigma vec vec'
 * (toolkit/lcp.cc:76-83 -> Eigen llt_rank_update_lower, general-sigma variant; `temp` = p doubles).
 * Returns 0, or j + 1 if the modified matrix is not positive definite at column j. */
int otk_rank_update(double *L, int n, int i0, int p, const double *vec, double sigma, double *temp) {
  for (int k = 0; k < p; ++k) temp[k] = vec[k];
  double beta = 1.0;
  for (int j = 0; j < p; ++j) {
    const double Ljj = AT(L, i0 + j, i0 + j);
    const double dj = Ljj * Ljj;
    const double wj = temp[j];
    const double swj2 = sigma * (wj * wj);
    const double gamma = dj * beta + swj2;
    const double xx = dj + swj2 / beta;
    if (!(xx > 0.0)) return j + 1;
    const double nLjj = sqrt(xx);
    AT(L, i0 + j, i0 + j) = nLjj;
    beta = beta + swj2 / dj;
    const double f0 = wj / Ljj, f1 = nLjj / Ljj, f2 = (gamma != 0.0) ? nLjj * sigma * wj / gamma : 0.0;
    for (int k = j + 1; k < p; ++k) {
      const double lk = AT(L, i0 + k, i0 + j);
      temp[k] = temp[k] - f0 * lk;
      if (gamma != 0.0) AT(L, i0 + k, i0 + j) = f1 * lk + f2 * temp[k];
    }
  }
  return 0;
}

/* row m - 1 of L from the factor of the leading (m-1) x (m-1) block (toolkit/lcp.cc:91-102) */
int otk_add_cholesky_row(const double *A, int n, int m, double *L) {
  if (m == 1) {
    if (!(AT(A, 0, 0) > 0.0)) return 1;
    AT(L, 0, 0) = sqrt(AT(A, 0, 0));
    return 0;
  }
  double *ell = &AT(L, m - 1, 0);          /* solved in place in the new row */
  for (int k = 0; k < m - 1; ++k) ell[k] = AT(A, m - 1, k);
  otk_lsolve(L, n, m - 1, ell);
  double s = 0.0;
  for (int k = 0; k < m - 1; ++k) s = s + ell[k] * ell[k];
  const double d = AT(A, m - 1, m - 1) - s;
  if (!(d > 0.0)) return m;
  AT(L, m - 1, m - 1) = sqrt(d);
  return 0;
}

/* L (m x m factor of A's leading block) -> (m-1) x (m-1) factor of A with row/column m - 1 moved to i
 * (toolkit/lcp.cc:110-157); A itself is not changed.  work = 2 n doubles. */
int otk_swap_cholesky_rows(const double *A, int n, int i, int m, double *L, double *work) {
  if (m <= 1 || i == m - 1) return 0;
  double *wq = work, *temp = work + n;
  if (i == 0) {
    for (int k = 0; k < m - 1; ++k) wq[k] = AT(A, m - 1, k) - AT(A, k, 0);
    wq[0] = (AT(A, m - 1, m - 1) - AT(A, 0, 0)) * 0.5 + 1.0;
    int r = otk_rank_update(L, n, 0, m - 1, wq, 0.5, temp);
    if (r) return r;
    wq[0] = (AT(A, m - 1, m - 1) - AT(A, 0, 0)) * 0.5 - 1.0;
    return otk_rank_update(L, n, 0, m - 1, wq, -0.5, temp);
  }
  double *l1 = &AT(L, i, 0);               /* new row i of L, solved in place */
  for (int k = 0; k < i; ++k) l1[k] = AT(A, m - 1, k);
  otk_lsolve(L, n, i, l1);
  double s = 0.0;
  for (int k = 0; k < i; ++k) s = s + l1[k] * l1[k];
  const double d = AT(A, m - 1, m - 1) - s;
  if (!(d > 0.0)) return i + 1;
  const double e = sqrt(d);
  AT(L, i, i) = e;
  const int p = m - 2 - i;
  if (p > 0) {
    for (int k = 0; k < p; ++k) wq[k] = AT(L, i + 1 + k, i);       /* the original l2 */
    int r = otk_rank_update(L, n, i + 1, p, wq, 1.0, temp);
    if (r) return r;
    for (int k = 0; k < p; ++k) {
      double t = 0.0;
      for (int c = 0; c < i; ++c) t = t + AT(L, i + 1 + k, c) * l1[c];
      const double v = (AT(A, m - 1, i + 1 + k) - t) / e;
      AT(L, i + 1 + k, i) = v;
      wq[k] = v;
    }
    r = otk_rank_update(L, n, i + 1, p, wq, -1.0, temp);
    if (r) return r;
  }
  return 0;
}

/* A(i <-> j) touching the lower triangle only (toolkit/lcp.cc:171-195); perm maps current -> original */
void otk_swap_rows_and_columns(double *A, int n, int i, int j, int *perm) {
  if (i == j) return;
  if (i > j) { int t = i; i = j; j = t; }
  { int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
  for (int c = 0; c < i; ++c) { double t = AT(A, i, c); AT(A, i, c) = AT(A, j, c); AT(A, j, c) = t; }
  for (int r = j + 1; r < n; ++r) { double t = AT(A, r, i); AT(A, r, i) = AT(A, r, j); AT(A, r, j) = t; }
  for (int k = i + 1; k < j; ++k) { double t = AT(A, k, i); AT(A, k, i) = AT(A, j, k); AT(A, j, k) = t; }
  { double t = AT(A, i, i); AT(A, i, i) = AT(A, j, j); AT(A, j, j) = t; }
}

static void swapd(double *v, int a, int b) { double t = v[a]; v[a] = v[b]; v[b] = t; }

/* SolveLCP_BoxDantzig (toolkit/lcp.cc:444-619).  A (lower triangle) is permuted in place; perm_out[k] =
 * original index of the final row k.  Requires lo <= 0 <= hi, lo < hi.  Returns 1 (the reference always
 * returns true), 0 if a factor update meets a non-positive pivot, -1 on allocation failure.
 * *pivots = steps of the inner loop. */
int otk_box_dantzig(int n, double *A, const double *b, const double *lo_arg, const double *hi_arg,
                    double *x_out, double *w_out, int *perm_out, int *pivots) {
  double *buf = (double *)calloc((size_t)n * n + 10 * (size_t)n, sizeof(double));
  int *perm = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  if (!buf || !perm) { free(buf); free(perm); return -1; }
  double *L = buf, *x = L + (size_t)n * n, *w = x + n, *lo = w + n, *hi = lo + n, *dxS = hi + n, *dwNS = dxS + n,
         *limit = dwNS + n, *v = limit + n, *work = v + n;   /* work: 2 n */
  for (int i = 0; i < n; ++i) { lo[i] = lo_arg[i]; hi[i] = hi_arg[i]; perm[i] = i; }
  int index = 0, steps = 0, ok = 1;
  for (int i = 0; i < n && ok; ++i) {
    double s = 0.0;
    for (int k = 0; k < i; ++k) s = s + AT(A, i, k) * x[k];
    w[i] = s - b[i];
    x[i] = 0.0;
    if (w[i] == 0.0) continue;
    if (lo[i] == 0.0 && w[i] >= 0.0) continue;
    if (hi[i] == 0.0 && w[i] <= 0.0) continue;
    const double dir = (w[i] <= 0.0) ? 1.0 : -1.0;
    for (int k = 0; k < index; ++k) dxS[k] = -dir * AT(A, i, k);
    otk_lltsolve(L, n, index, dxS);
    const double delta_xi = dir;
    while (1) {
      ++steps;
      for (int r = index; r < i; ++r) {
        double t = 0.0;
        for (int k = 0; k < index; ++k) t = t + AT(A, r, k) * dxS[k];
        dwNS[r - index] = t + AT(A, i, r) * dir;
      }
      double delta_wi = 0.0;
      for (int k = 0; k < index; ++k) delta_wi = delta_wi + AT(A, i, k) * dxS[k];
      delta_wi = delta_wi + AT(A, i, i) * dir;
      double best_alpha = -w[i] / delta_wi;
      int best_index = i, index_i_into_set = 1;
      const double index_i_limit = (dir > 0.0) ? hi[i] : lo[i];
      {
        const double alpha = (index_i_limit - x[i]) / delta_xi;
        if (alpha > 0.0 && alpha < best_alpha) { best_alpha = alpha; best_index = i; index_i_into_set = 0; }
      }
      for (int j = 0; j < index; ++j) {
        limit[j] = (dxS[j] > 0.0) ? hi[j] : lo[j];
        const double alpha = (limit[j] - x[j]) / dxS[j];
        if (alpha > 0.0 && alpha < best_alpha) { best_alpha = alpha; best_index = j; }
      }
      for (int j = index; j < i; ++j) {
        const double alpha = -w[j] / dwNS[j - index];
        if (alpha > 0.0 && alpha < best_alpha) { best_alpha = alpha; best_index = j; }
      }
      for (int k = 0; k < index; ++k) x[k] = x[k] + best_alpha * dxS[k];
      x[i] = x[i] + best_alpha * delta_xi;
      for (int r = index; r < i; ++r) w[r] = w[r] + best_alpha * dwNS[r - index];
      w[i] = w[i] + best_alpha * delta_wi;
      index_i_into_set = (best_index == i && index_i_into_set);
      if (best_index < index) {
        x[best_index] = limit[best_index];
        if (otk_swap_cholesky_rows(A, n, best_index, index, L, work)) { ok = 0; break; }
        otk_swap_rows_and_columns(A, n, index - 1, best_index, perm);
        swapd(x, index - 1, best_index); swapd(lo, index - 1, best_index); swapd(hi, index - 1, best_index);
        --index;
        for (int k = 0; k < index; ++k) dxS[k] = -dir * AT(A, i, k);
        otk_lltsolve(L, n, index, dxS);
      } else if (best_index < i || index_i_into_set) {
        w[index_i_into_set ? i : best_index] = 0.0;
        otk_swap_rows_and_columns(A, n, index, best_index, perm);
        swapd(x, index, best_index); swapd(w, index, best_index); swapd(lo, index, best_index); swapd(hi, index, best_index);
        if (otk_add_cholesky_row(A, n, index + 1, L)) { ok = 0; break; }
        if (best_index != i) {
          double t = 0.0;
          for (int k = 0; k < index; ++k) t = t + AT(A, index, k) * dxS[k];
          const double value = (-dir * AT(A, i, index) - t) / (AT(L, index, index) * AT(L, index, index));
          dxS[index] = value;
          for (int k = 0; k < index; ++k) v[k] = AT(L, index, k);
          otk_ltsolve(L, n, index, v);
          for (int k = 0; k < index; ++k) dxS[k] = dxS[k] - value * v[k];
        }
        ++index;
      } else {
        x[i] = index_i_limit;
      }
      if (best_index == i) break;
    }
  }
  for (int k = 0; k < n; ++k) { x_out[perm[k]] = x[k]; w_out[perm[k]] = w[k]; if (perm_out) perm_out[k] = perm[k]; }
  if (pivots) *pivots = steps;
  free(buf); free(perm);
  return ok;
}

/* SolveLCP_BoxMurty on a LinearReducer (toolkit/lcp.cc:213-328, 380-442): principal pivoting that keeps the
 * Cholesky factor of the index set up to date by AddCholeskyRow / SwapCholeskyRows.  SolveLCP_Murty (:333-378)
 * is the same loop with lo = 0, hi = +inf (its extra tests `w > 0` inside and `x > 0` outside the set can never
 * fire: w is zeroed inside, x = c = 0 outside).  A (lower triangle) is permuted in place; perm_out[k] = original
 * index of the final row k.  Returns 1 solved, 0 iteration limit reached (the reference returns false) or a
 * non-positive pivot, -1 allocation failure.  *iters = iterations of the loop. */
int otk_box_murty(int n, double *A, const double *b, const double *lo, const double *hi, int max_iterations,
                  double *x, double *w, int *perm_out, int *iters) {
  double *buf = (double *)calloc((size_t)n * n + 8 * (size_t)n, sizeof(double));
  int *perm = (int *)malloc(sizeof(int) * 2 * (size_t)(n > 0 ? n : 1));
  if (!buf || !perm) { free(buf); free(perm); return -1; }
  int *iperm = perm + n;
  double *L = buf, *x_ = L + (size_t)n * n, *c = x_ + n, *c2 = c + n, *t = c2 + n, *x2 = t + n, *work = x2 + n;   /* work: 2 n */
  /* LinearReducer::LinearReducer (:213-224) */
  for (int r = 0; r < n; ++r)
    for (int q = 0; q <= r; ++q) AT(L, r, q) = AT(A, r, q);
  int ok = otk_cholesky(L, n) == 0, index = n, it = 0, solved = 0;
  for (int i = 0; i < n; ++i) { perm[i] = i; iperm[i] = i; x_[i] = b[i]; c[i] = 0.0; }
  if (ok) otk_lltsolve(L, n, n, x_);
  for (; ok && it < max_iterations; ++it) {
    /* SubSolve (:245-296) */
    if (index == 0) {
      for (int i = 0; i < n; ++i) x[i] = c[i];
    } else if (index >= n) {
      for (int i = 0; i < n; ++i) x[perm[i]] = x_[i];
    } else {
      for (int i = 0; i < n; ++i) c2[i] = c[perm[i]];
      for (int k = 0; k < index; ++k) {
        double s = 0.0;
        for (int r = index; r < n; ++r) s = s + AT(A, r, k) * (c2[r] - x_[r]);
        t[k] = s;
      }
      otk_lltsolve(L, n, index, t);
      for (int i = 0; i < index; ++i) x[perm[i]] = x_[i] - t[i];
      for (int i = index; i < n; ++i) x[perm[i]] = c[perm[i]];
    }
    /* MultiplyA (:298-322) on the rows outside the set, then w = A x - b there and 0 inside (:396-404) */
    for (int i = 0; i < n; ++i) x2[i] = x[perm[i]];
    for (int r = index; r < n; ++r) {
      double s = 0.0;
      for (int k = 0; k < index; ++k) s = s + AT(A, r, k) * x2[k];
      double u = 0.0;
      for (int k = index; k < n; ++k) u = u + ((k <= r) ? AT(A, r, k) : AT(A, k, r)) * x2[k];
      w[perm[r]] = (s + u) - b[perm[r]];
    }
    for (int i = 0; i < index; ++i) w[perm[i]] = 0.0;
    /* first violated index in the caller's order (:408-431) */
    int moved = 0;
    for (int i = 0; i < n && !moved; ++i) {
      const int p = iperm[i];
      if (p < index) {
        int out = 0;
        if (x[i] < lo[i]) { c[i] = lo[i]; out = 1; }
        else if (x[i] > hi[i]) { c[i] = hi[i]; out = 1; }
        if (out) {      /* RemoveIndex (:236-243) */
          if (otk_swap_cholesky_rows(A, n, p, index, L, work)) { ok = 0; break; }
          --index;
          if (index != p) {
            const int a = perm[index], bq = perm[p];
            otk_swap_rows_and_columns(A, n, index, p, perm);
            iperm[a] = p; iperm[bq] = index;
            swapd(x_, index, p);
          }
          moved = 1;
        }
      } else {
        int in = 0;
        if (c[i] == lo[i] && w[i] < 0.0) in = 1;
        else if (c[i] == hi[i] && w[i] > 0.0) in = 1;
        if (in) {       /* AddIndex (:226-234) */
          if (index != p) {
            const int a = perm[index], bq = perm[p];
            otk_swap_rows_and_columns(A, n, index, p, perm);
            iperm[a] = p; iperm[bq] = index;
            swapd(x_, index, p);
          }
          ++index;
          if (otk_add_cholesky_row(A, n, index, L)) { ok = 0; break; }
          c[i] = 0.0;
          moved = 1;
        }
      }
    }
    if (!ok) break;
    if (!moved) { solved = 1; break; }
  }
  for (int k = 0; k < n; ++k) if (perm_out) perm_out[k] = perm[k];
  if (iters) *iters = it;
  free(buf); free(perm);
  return ok && solved;
}

/* SolveLCP_BoxSchur (toolkit/lcp.cc:627-747) and the dispatch of lcp::SolveLCP it recurses into (:752-785).
 *   algorithm 0 = MURTY -> SolveLCP_BoxMurty on the bounded part, 1 = COTTLE_DANTZIG -> SolveLCP_BoxDantzig.
 *   nub_arg >= 0 is the reference's test hook (:623-626): the first nub_arg indexes are taken as unbounded
 *   without looking; -1 scans lo / hi with the two-pointer partition of :656-683.
 *   q6 != 0 keeps the reference's classification tests literally (`hi < -DBL_MAX`, `hi >= -DBL_MAX`: the
 *   lower bound alone decides); q6 == 0 tests hi against +DBL_MAX as was surely meant.
 * A (row-major, lower triangle only) is permuted in place: the partition's swaps always, and the inner
 * solver's pivoting order only when nub == 0 (otherwise the inner solver permutes the temporary R).
 * perm_out[k] = original index of row k after the PARTITION (the inner permutation is not reported, as in the
 * reference, whose `permutation` object only records the outer swaps).  *nub_out = test_nub_from_SolveLCP_BoxSchur.
 * Returns 1 solved, 0 failed (inner solver gave up / a non-positive pivot), -1 allocation failure.
 * Operation order of the dense steps (Cholesky of Z, Q = L^-1 B', R = C - Q'Q, the two solves) is the plain
 * column-oriented one of this file, not Eigen's blocked kernels (unknown here): results agree to rounding. */
int otk_box_schur(int n, double *A, const double *b_arg, const double *lo_arg, const double *hi_arg, int algorithm,
                  int max_iterations, int nub_arg, int q6, double *x, double *w, int *perm_out, int *nub_out, int *iters) {
  const double big = __DBL_MAX__;
  double *vec = (double *)malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
  int *perm = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  if (!vec || !perm) { free(vec); free(perm); return -1; }
  double *b = vec, *lo = b + n, *hi = lo + n;
  for (int i = 0; i < n; ++i) { b[i] = b_arg[i]; lo[i] = lo_arg[i]; hi[i] = hi_arg[i]; perm[i] = i; }
  int nub = nub_arg, did_swaps = 0, ret = 1, it = 0;
  if (nub < 0) {
    nub = 0;
    int nb = n - 1;
    while (1) {
      for (; nub <= nb; nub++) {          /* :660-663 */
        const int bounded = q6 ? (lo[nub] > -big || hi[nub] < -big) : (lo[nub] > -big || hi[nub] < big);
        if (bounded) break;
      }
      for (; nb >= nub; nb--) {           /* :665-669 */
        const int unbounded = q6 ? (lo[nb] <= -big && hi[nb] >= -big) : (lo[nb] <= -big && hi[nb] >= big);
        if (unbounded) break;
      }
      if (nub > nb) break;
      otk_swap_rows_and_columns(A, n, nub, nb, perm);
      swapd(b, nub, nb); swapd(lo, nub, nb); swapd(hi, nub, nb);
      did_swaps = 1;
    }
  }
  if (nub_out) *nub_out = nub;
  if (perm_out) for (int k = 0; k < n; ++k) perm_out[k] = perm[k];
  if (nub == n) {                          /* :687-692: x = A.llt().solve(b), w = 0 (not unpermuted: no swaps happened) */
    double *L = (double *)calloc((size_t)n * n, sizeof(double));
    if (!L) { free(vec); free(perm); return -1; }
    for (int r = 0; r < n; ++r) for (int c = 0; c <= r; ++c) AT(L, r, c) = AT(A, r, c);
    if (otk_cholesky(L, n)) ret = 0;
    for (int i = 0; i < n; ++i) { x[i] = b[i]; w[i] = 0.0; }
    if (ret) otk_lltsolve(L, n, n, x);
    free(L);
  } else if (nub == 0) {                   /* :695-700: entirely an LCP; did_swaps is false, A goes to the inner solver itself */
    (void)did_swaps;
    ret = algorithm == 1 ? otk_box_dantzig(n, A, b, lo, hi, x, w, NULL, &it)
                         : otk_box_murty(n, A, b, lo, hi, max_iterations, x, w, NULL, &it);
  } else {
    const int nb2 = n - nub;
    double *buf = (double *)calloc((size_t)nub * nub + (size_t)nub * nb2 + (size_t)nb2 * nb2 + 4 * (size_t)n, sizeof(double));
    if (!buf) { free(vec); free(perm); return -1; }
    double *L = buf, *Q = L + (size_t)nub * nub, *R = Q + (size_t)nub * nb2, *t = R + (size_t)nb2 * nb2, *rhs = t + n,
           *z = rhs + n, *w2 = z + n;
    /* Z = A[0:nub, 0:nub], B = A[nub:, 0:nub], C = A[nub:, nub:], lower triangles only (:704-712) */
    for (int r = 0; r < nub; ++r) for (int c = 0; c <= r; ++c) L[(size_t)r * nub + c] = AT(A, r, c);
    if (otk_cholesky(L, nub)) ret = 0;      /* L L' = Z */
    if (ret) {
      for (int j = 0; j < nb2; ++j) {       /* Q = L^-1 B', one column of B' (= row of B) at a time */
        for (int k = 0; k < nub; ++k) t[k] = AT(A, nub + j, k);
        otk_lsolve(L, nub, nub, t);
        for (int k = 0; k < nub; ++k) Q[(size_t)k * nb2 + j] = t[k];
      }
      for (int i = 0; i < nb2; ++i)         /* R = C - Q'Q, lower triangle (:717-719) */
        for (int j = 0; j <= i; ++j) {
          double s = 0.0;
          for (int k = 0; k < nub; ++k) s = s + Q[(size_t)k * nb2 + i] * Q[(size_t)k * nb2 + j];
          R[(size_t)i * nb2 + j] = AT(A, nub + i, nub + j) - s;
        }
      for (int k = 0; k < nub; ++k) t[k] = b[k];
      otk_lltsolve(L, nub, nub, t);         /* t = Z^-1 c */
      for (int i = 0; i < nb2; ++i) {       /* rhs = d - B t */
        double s = 0.0;
        for (int k = 0; k < nub; ++k) s = s + AT(A, nub + i, k) * t[k];
        rhs[i] = b[nub + i] - s;
      }
      ret = algorithm == 1 ? otk_box_dantzig(nb2, R, rhs, lo + nub, hi + nub, z, w2, NULL, &it)
                           : otk_box_murty(nb2, R, rhs, lo + nub, hi + nub, max_iterations, z, w2, NULL, &it);
      if (ret == 1) {
        for (int k = 0; k < nub; ++k) {     /* y = Z^-1 (c - B' z) */
          double s = 0.0;
          for (int i = 0; i < nb2; ++i) s = s + AT(A, nub + i, k) * z[i];
          t[k] = b[k] - s;
        }
        otk_lltsolve(L, nub, nub, t);
        for (int k = 0; k < nub; ++k) { x[perm[k]] = t[k]; w[perm[k]] = 0.0; }
        for (int i = 0; i < nb2; ++i) { x[perm[nub + i]] = z[i]; w[perm[nub + i]] = w2[i]; }
      }
    }
    free(buf);
  }
  if (iters) *iters = it;
  free(vec); free(perm);
  return ret;
}
