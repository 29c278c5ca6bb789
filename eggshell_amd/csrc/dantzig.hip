// dantzig.hip -- the incremental-factor box LCP of the reference's toolkit:
//   lcp::SolveLCP_BoxDantzig             toolkit/lcp.cc:444-619   (Cottle-Dantzig principal pivoting)
//   AddCholeskyRow / SwapCholeskyRows    toolkit/lcp.cc:91-157    (the factor grows / shrinks by one row,
//                                                                  O(n^2) per pivot instead of a fresh O(n^3))
//   RankUpdate                           toolkit/lcp.cc:76-83     (Eigen's llt_rank_update_lower: method C1 of
//                                                                  Gill, Golub, Murray, Saunders 1974)
//   MatrixPermutation::SwapRowsAndColumns toolkit/lcp.cc:171-195  (A permuted IN PLACE, lower triangle only)
// The algorithm is one long dependent chain of pivots, each a handful of O(n^2) sweeps, on the few dozen
// to ~100 rows the reference's ensembles hand it -- there is nothing in ONE problem to spread over a GPU, so
// one workgroup runs a whole solve and the GPU-natural unit is a BATCH: one workgroup per problem, thousands of
// problems per launch (egs_box_lcp_batch; what a batch of ensembles hands the solver).  Two instantiations:
//   n <= 96   one wavefront, A, L and every vector in LDS (two n x n fp64 matrices in 160 KB);
//   n <= 1024 four wavefronts, A permuted in place in global memory, L in a global work area, vectors in LDS.
// Row-parallel loops take a lane per row; every scalar chain (dot products, the ratio test's argmin) runs in the
// reference's element order on all lanes at once, so the pivot sequence -- and with it the permutation left in
// A -- is the sequential algorithm's.  Every loop has an exit: a step cap (the caller's, else 20 n + 1000) and
// an optional wall-clock limit (lcp::Settings::max_iterations / max_time, toolkit/lcp.h:161-167).
#include "dense_lcp.h"

#include <algorithm>
#include <stdexcept>
#include <vector>

namespace egs {

namespace {

struct LcpResult {
  int32_t ok, pivots, reason, pad;      // reason: 0 solved, 1 step limit, 2 non-positive pivot, 3 time limit
};

// `count` problems, one workgroup each.  Offsets are in doubles; A is permuted in place (lower triangle).
struct LcpSet {
  const int32_t *ids;                   // block -> problem number (NULL: the block number itself)
  const int32_t *n;
  const int64_t *a_off, *v_off;
  double *A;
  const double *b, *lo, *hi;
  double *x, *w;
  int32_t *perm;
  LcpResult *res;
  double *L;                            // global instantiation: work area for the factor, laid out as A
  int max_steps;                        // <= 0: 20 n + 1000
  long long max_ticks;                  // wall_clock64 ticks (100 MHz), 0 = no limit
};

// the barrier: a scheduling fence plus "my LDS / global writes have landed" for the workgroup
__device__ __forceinline__ void wsync() { __syncthreads(); }

// lane 0 looks at the clock, everyone gets the same answer
__device__ bool out_of_time(long long t0, long long max_ticks, int *flag, int lane) {
  if (max_ticks <= 0) return false;
  if (lane == 0) *flag = ((long long)wall_clock64() - t0 > max_ticks) ? 1 : 0;
  wsync();
  const bool r = *flag != 0;
  wsync();
  return r;
}

#define AT(M, r, c) (M)[(r) * n + (c)]

// L y = b on the top-left m x m block, column by column (toolkit/lcp.cc:52-54)
template <int NT>
__device__ void lsolve(const double *L, int n, int m, double *x, int lane) {
  for (int j = 0; j < m; ++j) {
    const double xj = x[j] / AT(L, j, j);
    wsync();
    if (lane == 0) x[j] = xj;
    for (int k = j + 1 + lane; k < m; k += NT) x[k] = x[k] - AT(L, k, j) * xj;
    wsync();
  }
}
// L' x = y (toolkit/lcp.cc:58-62)
template <int NT>
__device__ void ltsolve(const double *L, int n, int m, double *x, int lane) {
  for (int j = m - 1; j >= 0; --j) {
    const double xj = x[j] / AT(L, j, j);
    wsync();
    if (lane == 0) x[j] = xj;
    for (int k = lane; k < j; k += NT) x[k] = x[k] - AT(L, j, k) * xj;
    wsync();
  }
}
template <int NT>
__device__ void lltsolve(const double *L, int n, int m, double *x, int lane) {
  lsolve<NT>(L, n, m, x, lane);
  ltsolve<NT>(L, n, m, x, lane);
}

// L L' += sigma vec vec' on the p x p block at (i0, i0) (toolkit/lcp.cc:76-83); temp = p doubles, vec may be temp
template <int NT>
__device__ bool rank_update(double *L, int n, int i0, int p, const double *vec, double sigma, double *temp, int lane) {
  for (int k = lane; k < p; k += NT) temp[k] = vec[k];
  wsync();
  double beta = 1.0;
  for (int j = 0; j < p; ++j) {
    const double Ljj = AT(L, i0 + j, i0 + j);
    const double dj = Ljj * Ljj;
    const double wj = temp[j];
    const double swj2 = sigma * (wj * wj);
    const double gamma = dj * beta + swj2;
    const double xx = dj + swj2 / beta;
    if (!(xx > 0.0)) return false;
    const double nLjj = sqrt(xx);
    beta = beta + swj2 / dj;
    const double f0 = wj / Ljj, f1 = nLjj / Ljj, f2 = (gamma != 0.0) ? nLjj * sigma * wj / gamma : 0.0;
    wsync();
    if (lane == 0) AT(L, i0 + j, i0 + j) = nLjj;
    for (int k = j + 1 + lane; k < p; k += NT) {
      const double lk = AT(L, i0 + k, i0 + j);
      const double tk = temp[k] - f0 * lk;
      temp[k] = tk;
      if (gamma != 0.0) AT(L, i0 + k, i0 + j) = f1 * lk + f2 * tk;
    }
    wsync();
  }
  return true;
}

// toolkit/lcp.cc:91-102
template <int NT>
__device__ bool add_cholesky_row(const double *A, int n, int m, double *L, int lane) {
  if (m == 1) {
    const double d = AT(A, 0, 0);
    if (!(d > 0.0)) return false;
    wsync();
    if (lane == 0) AT(L, 0, 0) = sqrt(d);
    wsync();
    return true;
  }
  double *ell = &AT(L, m - 1, 0);
  for (int k = lane; k < m - 1; k += NT) ell[k] = AT(A, m - 1, k);
  wsync();
  lsolve<NT>(L, n, m - 1, ell, lane);
  double s = 0.0;
  for (int k = 0; k < m - 1; ++k) s = s + ell[k] * ell[k];
  const double d = AT(A, m - 1, m - 1) - s;
  if (!(d > 0.0)) return false;
  wsync();
  if (lane == 0) AT(L, m - 1, m - 1) = sqrt(d);
  wsync();
  return true;
}

// toolkit/lcp.cc:110-157; wq, temp = n doubles each
template <int NT>
__device__ bool swap_cholesky_rows(const double *A, int n, int i, int m, double *L, double *wq, double *temp, int lane) {
  if (m <= 1 || i == m - 1) return true;
  if (i == 0) {
    const double head = (AT(A, m - 1, m - 1) - AT(A, 0, 0)) * 0.5;
    for (int k = lane; k < m - 1; k += NT) wq[k] = (k == 0) ? head + 1.0 : AT(A, m - 1, k) - AT(A, k, 0);
    wsync();
    if (!rank_update<NT>(L, n, 0, m - 1, wq, 0.5, temp, lane)) return false;
    if (lane == 0) wq[0] = head - 1.0;
    wsync();
    return rank_update<NT>(L, n, 0, m - 1, wq, -0.5, temp, lane);
  }
  double *l1 = &AT(L, i, 0);
  for (int k = lane; k < i; k += NT) l1[k] = AT(A, m - 1, k);
  wsync();
  lsolve<NT>(L, n, i, l1, lane);
  double s = 0.0;
  for (int k = 0; k < i; ++k) s = s + l1[k] * l1[k];
  const double d = AT(A, m - 1, m - 1) - s;
  if (!(d > 0.0)) return false;
  const double e = sqrt(d);
  wsync();
  if (lane == 0) AT(L, i, i) = e;
  const int p = m - 2 - i;
  if (p > 0) {
    for (int k = lane; k < p; k += NT) wq[k] = AT(L, i + 1 + k, i);
    wsync();
    if (!rank_update<NT>(L, n, i + 1, p, wq, 1.0, temp, lane)) return false;
    for (int k = lane; k < p; k += NT) {
      double t = 0.0;
      for (int c = 0; c < i; ++c) t = t + AT(L, i + 1 + k, c) * l1[c];
      const double v = (AT(A, m - 1, i + 1 + k) - t) / e;
      AT(L, i + 1 + k, i) = v;
      wq[k] = v;
    }
    wsync();
    if (!rank_update<NT>(L, n, i + 1, p, wq, -1.0, temp, lane)) return false;
  }
  wsync();
  return true;
}

// toolkit/lcp.cc:171-195
template <int NT>
__device__ void swap_rows_and_columns(double *A, int n, int i, int j, int *perm, int lane) {
  if (i == j) return;
  if (i > j) { const int t = i; i = j; j = t; }
  wsync();
  for (int c = lane; c < i; c += NT) { const double t = AT(A, i, c); AT(A, i, c) = AT(A, j, c); AT(A, j, c) = t; }
  for (int r = j + 1 + lane; r < n; r += NT) { const double t = AT(A, r, i); AT(A, r, i) = AT(A, r, j); AT(A, r, j) = t; }
  for (int k = i + 1 + lane; k < j; k += NT) { const double t = AT(A, k, i); AT(A, k, i) = AT(A, j, k); AT(A, j, k) = t; }
  if (lane == 0) {
    const double t = AT(A, i, i); AT(A, i, i) = AT(A, j, j); AT(A, j, j) = t;
    const int q = perm[i]; perm[i] = perm[j]; perm[j] = q;
  }
  wsync();
}

__device__ __forceinline__ void swap_entry(double *v, int a, int b, int lane) {
  if (lane == 0) { const double t = v[a]; v[a] = v[b]; v[b] = t; }
}

// Per-problem pointers of a workgroup: GLOBAL = false copies A into LDS (written back at the end), true works in place.
template <int NT, bool GLOBAL>
struct LcpView {
  int n, prob;
  double *A, *L, *vec;      // vec: the first of the problem's LDS vectors
  double *gA;
  long long v_off;
  __device__ LcpView(const LcpSet &S, unsigned char *smem, int lane) {
    prob = S.ids ? S.ids[blockIdx.x] : (int)blockIdx.x;
    n = S.n[prob];
    gA = S.A + S.a_off[prob];
    v_off = S.v_off[prob];
    double *base = reinterpret_cast<double *>(smem);
    if (GLOBAL) { A = gA; L = S.L + S.a_off[prob]; vec = base; }
    else { A = base; L = A + (size_t)n * n; vec = L + (size_t)n * n; }
  }
};

// SolveLCP_BoxDantzig, toolkit/lcp.cc:444-619.
template <int NT, bool GLOBAL>
__global__ void __launch_bounds__(NT) box_dantzig_kernel(const LcpSet S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const LcpView<NT, GLOBAL> V(S, smem, lane);
  const int n = V.n;
  double *A = V.A, *L = V.L;
  double *x = V.vec, *w = x + n, *lo = w + n, *hi = lo + n, *b = hi + n, *dxS = b + n, *dwNS = dxS + n,
         *limit = dwNS + n, *v = limit + n, *wq = v + n, *temp = wq + n;
  int *perm = reinterpret_cast<int *>(temp + n), *flag = perm + 2 * n;
  const double *gb = S.b + V.v_off, *glo = S.lo + V.v_off, *ghi = S.hi + V.v_off;
  for (int k = lane; k < n * n; k += NT) { if (!GLOBAL) A[k] = V.gA[k]; L[k] = 0.0; }
  for (int k = lane; k < n; k += NT) { x[k] = 0.0; w[k] = 0.0; lo[k] = glo[k]; hi[k] = ghi[k]; b[k] = gb[k]; perm[k] = k; }
  wsync();
  const int max_steps = S.max_steps > 0 ? S.max_steps : 20 * n + 1000;
  const long long t0 = (long long)wall_clock64();
  int reason = 0;

  int index = 0, steps = 0;
  bool ok = true;
  for (int i = 0; i < n && ok; ++i) {
    double s = 0.0;
    for (int k = 0; k < i; ++k) s = s + AT(A, i, k) * x[k];
    const double wi0 = s - b[i];
    wsync();
    if (lane == 0) { w[i] = wi0; x[i] = 0.0; }
    wsync();
    if (wi0 == 0.0) continue;
    if (lo[i] == 0.0 && wi0 >= 0.0) continue;
    if (hi[i] == 0.0 && wi0 <= 0.0) continue;
    const double dir = (wi0 <= 0.0) ? 1.0 : -1.0;
    for (int k = lane; k < index; k += NT) dxS[k] = -dir * AT(A, i, k);
    wsync();
    lltsolve<NT>(L, n, index, dxS, lane);
    const double delta_xi = dir;
    while (true) {
      if (++steps > max_steps) { ok = false; reason = 1; break; }
      if (out_of_time(t0, S.max_ticks, flag, lane)) { ok = false; reason = 3; break; }
      // delta_w on the rows outside the set, one lane per row, the row's products in column order
      for (int r = index + lane; r < i; r += NT) {
        double t = 0.0;
        for (int k = 0; k < index; ++k) t = t + AT(A, r, k) * dxS[k];
        dwNS[r - index] = t + AT(A, i, r) * dir;
      }
      double delta_wi = 0.0;
      for (int k = 0; k < index; ++k) delta_wi = delta_wi + AT(A, i, k) * dxS[k];
      delta_wi = delta_wi + AT(A, i, i) * dir;
      for (int j = lane; j < index; j += NT) limit[j] = (dxS[j] > 0.0) ? hi[j] : lo[j];
      wsync();
      // the ratio test, in the reference's scan order on every lane
      double best_alpha = -w[i] / delta_wi;
      int best_index = i;
      bool index_i_into_set = true;
      const double index_i_limit = (dir > 0.0) ? hi[i] : lo[i];
      {
        const double alpha = (index_i_limit - x[i]) / delta_xi;
        if (alpha > 0.0 && alpha < best_alpha) { best_alpha = alpha; best_index = i; index_i_into_set = false; }
      }
      for (int j = 0; j < index; ++j) {
        const double alpha = (limit[j] - x[j]) / dxS[j];
        if (alpha > 0.0 && alpha < best_alpha) { best_alpha = alpha; best_index = j; }
      }
      for (int j = index; j < i; ++j) {
        const double alpha = -w[j] / dwNS[j - index];
        if (alpha > 0.0 && alpha < best_alpha) { best_alpha = alpha; best_index = j; }
      }
      wsync();
      for (int k = lane; k < index; k += NT) x[k] = x[k] + best_alpha * dxS[k];
      for (int r = index + lane; r < i; r += NT) w[r] = w[r] + best_alpha * dwNS[r - index];
      if (lane == 0) { x[i] = x[i] + best_alpha * delta_xi; w[i] = w[i] + best_alpha * delta_wi; }
      wsync();
      index_i_into_set = (best_index == i && index_i_into_set);
      if (best_index < index) {
        if (lane == 0) x[best_index] = limit[best_index];
        wsync();
        if (!swap_cholesky_rows<NT>(A, n, best_index, index, L, wq, temp, lane)) { ok = false; reason = 2; break; }
        swap_rows_and_columns<NT>(A, n, index - 1, best_index, perm, lane);
        swap_entry(x, index - 1, best_index, lane); swap_entry(lo, index - 1, best_index, lane); swap_entry(hi, index - 1, best_index, lane);
        --index;
        wsync();
        for (int k = lane; k < index; k += NT) dxS[k] = -dir * AT(A, i, k);
        wsync();
        lltsolve<NT>(L, n, index, dxS, lane);
      } else if (best_index < i || index_i_into_set) {
        if (lane == 0) w[index_i_into_set ? i : best_index] = 0.0;
        wsync();
        swap_rows_and_columns<NT>(A, n, index, best_index, perm, lane);
        swap_entry(x, index, best_index, lane); swap_entry(w, index, best_index, lane);
        swap_entry(lo, index, best_index, lane); swap_entry(hi, index, best_index, lane);
        wsync();
        if (!add_cholesky_row<NT>(A, n, index + 1, L, lane)) { ok = false; reason = 2; break; }
        if (best_index != i) {
          double t = 0.0;
          for (int k = 0; k < index; ++k) t = t + AT(A, index, k) * dxS[k];
          const double value = (-dir * AT(A, i, index) - t) / (AT(L, index, index) * AT(L, index, index));
          for (int k = lane; k < index; k += NT) v[k] = AT(L, index, k);
          wsync();
          if (lane == 0) dxS[index] = value;
          ltsolve<NT>(L, n, index, v, lane);
          for (int k = lane; k < index; k += NT) dxS[k] = dxS[k] - value * v[k];
          wsync();
        }
        ++index;
      } else {
        if (lane == 0) x[i] = index_i_limit;
        wsync();
      }
      if (best_index == i) break;
    }
  }
  wsync();
  for (int k = lane; k < n; k += NT) { S.x[V.v_off + perm[k]] = x[k]; S.w[V.v_off + perm[k]] = w[k]; if (S.perm) S.perm[V.v_off + k] = perm[k]; }
  if (!GLOBAL)
    for (int k = lane; k < n * n; k += NT) {
      const int r = k / n, c = k - r * n;
      if (c <= r) V.gA[k] = A[k];      // the lower triangle only, as the reference
    }
  if (lane == 0) { LcpResult r; r.ok = ok ? 1 : 0; r.pivots = steps; r.reason = ok ? 0 : reason; r.pad = 0; S.res[V.prob] = r; }
}

// in-place Cholesky of the lower triangle, column by column, a lane per row below the pivot (toolkit/lcp.cc:46-48)
template <int NT>
__device__ bool cholesky(double *L, int n, int lane) {
  for (int j = 0; j < n; ++j) {
    double d = AT(L, j, j);
    for (int k = 0; k < j; ++k) d = d - AT(L, j, k) * AT(L, j, k);
    if (!(d > 0.0)) return false;
    d = sqrt(d);
    wsync();
    if (lane == 0) AT(L, j, j) = d;
    for (int i = j + 1 + lane; i < n; i += NT) {
      double s = AT(L, i, j);
      for (int k = 0; k < j; ++k) s = s - AT(L, i, k) * AT(L, j, k);
      AT(L, i, j) = s / d;
    }
    wsync();
  }
  return true;
}

// SolveLCP_BoxMurty on a LinearReducer (toolkit/lcp.cc:213-328, 380-442); SolveLCP_Murty (:333-378) is the same loop
// with lo = 0, hi = +inf.
template <int NT, bool GLOBAL>
__global__ void __launch_bounds__(NT) box_murty_kernel(const LcpSet S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const LcpView<NT, GLOBAL> V(S, smem, lane);
  const int n = V.n;
  double *A = V.A, *L = V.L;
  double *x = V.vec, *w = x + n, *lo = w + n, *hi = lo + n, *b = hi + n, *xs = b + n, *c = xs + n, *c2 = c + n,
         *t = c2 + n, *wq = t + n, *temp = wq + n;
  int *perm = reinterpret_cast<int *>(temp + n), *iperm = perm + n, *flag = iperm + n;
  const double *gb = S.b + V.v_off, *glo = S.lo + V.v_off, *ghi = S.hi + V.v_off;
  for (int k = lane; k < n * n; k += NT) {
    const int r = k / n, q = k - r * n;
    const double a = V.gA[k];
    if (!GLOBAL) A[k] = a;
    L[k] = (q <= r) ? a : 0.0;
  }
  for (int k = lane; k < n; k += NT) { x[k] = 0.0; w[k] = 0.0; lo[k] = glo[k]; hi[k] = ghi[k]; b[k] = gb[k]; xs[k] = gb[k]; c[k] = 0.0; perm[k] = k; iperm[k] = k; }
  wsync();
  const int max_iterations = S.max_steps > 0 ? S.max_steps : 20 * n + 1000;
  const long long t0 = (long long)wall_clock64();
  int reason = 0;
  // LinearReducer::LinearReducer (:213-224): factor all of A, xs = A^-1 b
  bool ok = cholesky<NT>(L, n, lane);
  if (!ok) reason = 2;
  if (ok) lltsolve<NT>(L, n, n, xs, lane);
  int index = n, it = 0;
  bool solved = false;
  for (; ok && it < max_iterations; ++it) {
    if (out_of_time(t0, S.max_ticks, flag, lane)) { ok = false; reason = 3; break; }
    // SubSolve (:245-296)
    wsync();
    if (index == 0) {
      for (int i = lane; i < n; i += NT) x[i] = c[i];
    } else if (index >= n) {
      for (int i = lane; i < n; i += NT) x[perm[i]] = xs[i];
    } else {
      for (int i = lane; i < n; i += NT) c2[i] = c[perm[i]];
      wsync();
      for (int k = lane; k < index; k += NT) {
        double s = 0.0;
        for (int r = index; r < n; ++r) s = s + AT(A, r, k) * (c2[r] - xs[r]);
        t[k] = s;
      }
      wsync();
      lltsolve<NT>(L, n, index, t, lane);
      for (int i = lane; i < n; i += NT) x[perm[i]] = (i < index) ? xs[i] - t[i] : c[perm[i]];
    }
    wsync();
    // MultiplyA (:298-322) on the rows outside the set; w = A x - b there, 0 inside (:396-404).  c2 = x permuted
    for (int i = lane; i < n; i += NT) c2[i] = x[perm[i]];
    wsync();
    for (int r = lane; r < n; r += NT) {
      if (r < index) { w[perm[r]] = 0.0; continue; }
      double s = 0.0;
      for (int k = 0; k < index; ++k) s = s + AT(A, r, k) * c2[k];
      double u = 0.0;
      for (int k = index; k < n; ++k) u = u + ((k <= r) ? AT(A, r, k) : AT(A, k, r)) * c2[k];
      w[perm[r]] = (s + u) - b[perm[r]];
    }
    wsync();
    // first violated index in the caller's order (:408-431), found by every lane
    int who = -1, dirn = 0;      // dirn: -1 leaves the set clamped at lo, -2 at hi, +1 enters the set
    for (int i = 0; i < n; ++i) {
      if (iperm[i] < index) {
        if (x[i] < lo[i]) { who = i; dirn = -1; break; }
        if (x[i] > hi[i]) { who = i; dirn = -2; break; }
      } else {
        if ((c[i] == lo[i] && w[i] < 0.0) || (c[i] == hi[i] && w[i] > 0.0)) { who = i; dirn = 1; break; }
      }
    }
    if (who < 0) { solved = true; break; }
    const int p = iperm[who];
    wsync();
    if (dirn < 0) {       // RemoveIndex (:236-243)
      if (lane == 0) c[who] = (dirn == -1) ? lo[who] : hi[who];
      if (!swap_cholesky_rows<NT>(A, n, p, index, L, wq, temp, lane)) { ok = false; reason = 2; break; }
      --index;
    }
    if (index != p) {
      const int a = perm[index], bq = perm[p];
      swap_rows_and_columns<NT>(A, n, index, p, perm, lane);
      if (lane == 0) { iperm[a] = p; iperm[bq] = index; const double tt = xs[index]; xs[index] = xs[p]; xs[p] = tt; }
      wsync();
    }
    if (dirn > 0) {       // AddIndex (:226-234)
      ++index;
      if (!add_cholesky_row<NT>(A, n, index, L, lane)) { ok = false; reason = 2; break; }
      if (lane == 0) c[who] = 0.0;
    }
    wsync();
  }
  wsync();
  for (int k = lane; k < n; k += NT) { S.x[V.v_off + k] = x[k]; S.w[V.v_off + k] = w[k]; if (S.perm) S.perm[V.v_off + k] = perm[k]; }
  if (!GLOBAL)
    for (int k = lane; k < n * n; k += NT) {
      const int r = k / n, q = k - r * n;
      if (q <= r) V.gA[k] = A[k];
    }
  if (lane == 0) {
    LcpResult r;
    r.ok = (ok && solved) ? 1 : 0; r.pivots = it; r.reason = (ok && solved) ? 0 : (ok ? 1 : reason); r.pad = 0;
    S.res[V.prob] = r;
  }
}

#undef AT

struct HipErr : std::runtime_error {
  using std::runtime_error::runtime_error;
};
void chk(hipError_t e, const char *what) {
  if (e != hipSuccess) throw HipErr(std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(call) chk((call), #call)

template <typename T>
struct DBuf {      // plain hipMalloc / hipFree on purpose: with a stream-ordered pool allocation (hipMallocAsync /
  T *p = nullptr;  // hipFreeAsync) back-to-back calls that got the same block back saw stale lines of the previous
                   // call's matrix on this stack (ROCm 7.2; tests/test_gpu_dantzig.py::test_repeated_calls_are_independent)
  explicit DBuf(size_t n) { if (n) HIPCHK(hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T))); }
  ~DBuf() { if (p) (void)hipFree(p); }
  DBuf(const DBuf &) = delete;
  DBuf &operator=(const DBuf &) = delete;
};

size_t lds_bytes(int n, bool global) {
  return ((global ? 0 : 2 * (size_t)n * n) + 11 * (size_t)n) * sizeof(double) + (3 * (size_t)n + 4) * sizeof(int);
}

template <int NT, bool GLOBAL>
void launch_kind(hipStream_t stream, int algorithm, const LcpSet &S, int blocks, size_t lds) {
  if (blocks <= 0) return;
  auto kd = box_dantzig_kernel<NT, GLOBAL>;
  auto km = box_murty_kernel<NT, GLOBAL>;
  const void *fn = algorithm == 1 ? reinterpret_cast<const void *>(kd) : reinterpret_cast<const void *>(km);
  if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (algorithm == 1) hipLaunchKernelGGL(kd, dim3(blocks), dim3(NT), lds, stream, S);
  else hipLaunchKernelGGL(km, dim3(blocks), dim3(NT), lds, stream, S);
  HIPCHK(hipGetLastError());
}

// Everything on the device already: matrices at dA + a_off[k], vectors at v_off[k].  hn / ha / hv = the host copies
// of the size and offset tables.  One launch for the problems that fit LDS, one for the larger ones.
void solve_set(hipStream_t stream, int algorithm, int count, const int32_t *hn, const int64_t *ha, const int64_t *hv, double *dA,
               const double *db, const double *dlo, const double *dhi, int max_steps, double max_seconds, double *dx, double *dw,
               int32_t *dperm, LcpResult *dres) {
  std::vector<int32_t> small, large;
  int n_small = 0, n_large = 0;
  int64_t a_total = 0;
  for (int k = 0; k < count; ++k) {
    if (hn[k] <= kDantzigMaxRows) { small.push_back(k); n_small = std::max(n_small, (int)hn[k]); }
    else { large.push_back(k); n_large = std::max(n_large, (int)hn[k]); }
    a_total = std::max(a_total, ha[k] + (int64_t)hn[k] * hn[k]);
  }
  DBuf<int32_t> d_n(count), d_ids(count);
  DBuf<int64_t> d_off(2 * (size_t)count);
  DBuf<double> d_L(large.empty() ? 0 : (size_t)a_total);
  std::vector<int32_t> ids(small);
  ids.insert(ids.end(), large.begin(), large.end());
  HIPCHK(hipMemcpyAsync(d_n.p, hn, count * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(d_ids.p, ids.data(), count * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(d_off.p, ha, count * sizeof(int64_t), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(d_off.p + count, hv, count * sizeof(int64_t), hipMemcpyHostToDevice, stream));
  LcpSet S{};
  S.n = d_n.p; S.a_off = d_off.p; S.v_off = d_off.p + count;
  S.A = dA; S.b = db; S.lo = dlo; S.hi = dhi; S.x = dx; S.w = dw; S.perm = dperm; S.res = dres; S.L = d_L.p;
  S.max_steps = max_steps;
  S.max_ticks = max_seconds > 0 ? (long long)(max_seconds * 1e8) + 1 : 0;     // wall_clock64 counts at 100 MHz
  S.ids = d_ids.p;
  launch_kind<64, false>(stream, algorithm, S, (int)small.size(), lds_bytes(n_small, false));
  S.ids = d_ids.p + small.size();
  launch_kind<256, true>(stream, algorithm, S, (int)large.size(), lds_bytes(n_large, true));
  HIPCHK(hipStreamSynchronize(stream));      // the tables above are freed on return
}

void check_problem(int algorithm, int n, const double *lo, const double *hi) {
  if (n <= 0 || n > kIncrementalMaxRows) throw std::invalid_argument("incremental box LCP: 1 <= n <= 1024");
  for (int i = 0; i < n; ++i) {
    // lo <= 0 <= hi (toolkit/lcp.h:134); Dantzig also needs lo < hi (toolkit/lcp.cc:448-450)
    if (!(lo[i] <= 0.0) || !(hi[i] >= 0.0) || (algorithm == 1 && !(lo[i] < hi[i])))
      throw std::invalid_argument("incremental box LCP: needs lo <= 0 <= hi (and lo < hi for Cottle-Dantzig)");
  }
}

const char *reason_text(int reason) {
  switch (reason) {
    case 1: return "incremental box LCP: iteration limit reached";
    case 2: return "incremental box LCP: a factor update met a non-positive pivot (A not positive definite?)";
    case 3: return "incremental box LCP: time limit reached";
    default: return "";
  }
}

}  // namespace

void box_lcp_incremental_batch(hipStream_t stream, int algorithm, int count, const int32_t *n, double *A, const double *b,
                               const double *lo, const double *hi, int max_steps, double max_seconds, double *x, double *w,
                               int32_t *perm, int32_t *ok, int32_t *pivots, int32_t *reason) {
  if (algorithm != 0 && algorithm != 1) throw std::invalid_argument("incremental box LCP: algorithm 0 (Murty) or 1 (Cottle-Dantzig)");
  if (count <= 0) return;
  std::vector<int64_t> ha(count), hv(count);
  int64_t at = 0, vt = 0;
  for (int k = 0; k < count; ++k) {
    ha[k] = at; hv[k] = vt;
    check_problem(algorithm, n[k], lo + vt, hi + vt);
    at += (int64_t)n[k] * n[k]; vt += n[k];
  }
  DBuf<double> dA((size_t)at), dv(5 * (size_t)vt);
  DBuf<int32_t> dperm((size_t)vt);
  DBuf<LcpResult> dres(count);
  double *db = dv.p, *dlo = dv.p + vt, *dhi = dv.p + 2 * vt, *dx = dv.p + 3 * vt, *dw = dv.p + 4 * vt;
  HIPCHK(hipMemcpyAsync(dA.p, A, (size_t)at * sizeof(double), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(db, b, (size_t)vt * sizeof(double), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(dlo, lo, (size_t)vt * sizeof(double), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(dhi, hi, (size_t)vt * sizeof(double), hipMemcpyHostToDevice, stream));
  solve_set(stream, algorithm, count, n, ha.data(), hv.data(), dA.p, db, dlo, dhi, max_steps, max_seconds, dx, dw, dperm.p, dres.p);
  std::vector<LcpResult> r(count);
  std::vector<double> hA((size_t)at);
  HIPCHK(hipMemcpyAsync(hA.data(), dA.p, (size_t)at * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipMemcpyAsync(x, dx, (size_t)vt * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipMemcpyAsync(w, dw, (size_t)vt * sizeof(double), hipMemcpyDeviceToHost, stream));
  if (perm) HIPCHK(hipMemcpyAsync(perm, dperm.p, (size_t)vt * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipMemcpyAsync(r.data(), dres.p, count * sizeof(LcpResult), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  // only the lower triangle of the caller's A is ever written (toolkit/lcp.h:73)
  for (int k = 0; k < count; ++k) {
    const int nk = n[k];
    double *Ak = A + ha[k];
    const double *Hk = hA.data() + ha[k];
    for (int rr = 0; rr < nk; ++rr)
      for (int c = 0; c <= rr; ++c) Ak[(size_t)rr * nk + c] = Hk[(size_t)rr * nk + c];
    if (ok) ok[k] = r[k].ok;
    if (pivots) pivots[k] = r[k].pivots;
    if (reason) reason[k] = r[k].reason;
  }
}

bool box_lcp_incremental(hipStream_t stream, int algorithm, int n, double *A, const double *b, const double *lo, const double *hi,
                         double *x, double *w, int32_t *perm, int max_steps, double max_seconds, int *pivots, std::string *msg) {
  int32_t nn = n, ok = 0, piv = 0, reason = 0;
  box_lcp_incremental_batch(stream, algorithm, 1, &nn, A, b, lo, hi, max_steps, max_seconds, x, w, perm, &ok, &piv, &reason);
  if (pivots) *pivots = piv;
  if (!ok && msg) *msg = reason_text(reason);
  return ok != 0;
}

bool box_lcp_incremental_device(hipStream_t stream, int algorithm, int n, double *dA, const double *db, const double *dlo,
                                const double *dhi, const double *h_lo, const double *h_hi, int max_steps, double max_seconds,
                                double *dx, double *dw, int *pivots, std::string *msg) {
  if (algorithm != 0 && algorithm != 1) throw std::invalid_argument("incremental box LCP: algorithm 0 (Murty) or 1 (Cottle-Dantzig)");
  check_problem(algorithm, n, h_lo, h_hi);
  DBuf<LcpResult> dres(1);
  const int32_t hn = n;
  const int64_t zero = 0;
  solve_set(stream, algorithm, 1, &hn, &zero, &zero, dA, db, dlo, dhi, max_steps, max_seconds, dx, dw, nullptr, dres.p);
  LcpResult r{};
  HIPCHK(hipMemcpy(&r, dres.p, sizeof r, hipMemcpyDeviceToHost));
  if (pivots) *pivots = r.pivots;
  if (!r.ok && msg) *msg = reason_text(r.reason);
  return r.ok != 0;
}

}  // namespace egs
