// dantzig.hip -- the incremental-factor box LCP of the reference's toolkit:
//   lcp::SolveLCP_BoxDantzig             toolkit/lcp.cc:444-619   (Cottle-Dantzig principal pivoting)
//   AddCholeskyRow / SwapCholeskyRows    toolkit/lcp.cc:91-157    (the factor grows / shrinks by one row,
//                                                                  O(n^2) per pivot instead of a fresh O(n^3))
//   RankUpdate                           toolkit/lcp.cc:76-83     (Eigen's llt_rank_update_lower: method C1 of
//                                                                  Gill, Golub, Murray, Saunders 1974)
//   MatrixPermutation::SwapRowsAndColumns toolkit/lcp.cc:171-195  (A permuted IN PLACE, lower triangle only)
// The algorithm is one long dependent chain of pivots, each a handful of O(n^2) sweeps, on the few dozen
// to ~100 rows the reference's ensembles hand it -- there is nothing to spread over a GPU, so ONE wavefront
// runs the whole solve with A, L and every vector in LDS (two n x n fp64 matrices: n <= 96 in 160 KB) and
// the host sees one launch and one read-back.  Row-parallel loops take a lane per row; every scalar chain
// (dot products, the ratio test's argmin) runs in the reference's element order on all lanes at once, so
// the pivot sequence -- and with it the permutation left in A -- is the sequential algorithm's.
// Larger problems go through the blocked solver of dense_lcp.hip (fresh MFMA Cholesky per pivot; same
// unique solution for symmetric positive definite A).
#include "dense_lcp.h"

#include <stdexcept>
#include <vector>

namespace egs {

namespace {

struct DantzigResult {
  int32_t ok, pivots;
};

// one wavefront: the barrier is a scheduling fence plus "my LDS writes have landed"
__device__ __forceinline__ void wsync() { __syncthreads(); }

#define AT(M, r, c) (M)[(r) * n + (c)]

// L y = b on the top-left m x m block, column by column (toolkit/lcp.cc:52-54)
__device__ void lsolve(const double *L, int n, int m, double *x, int lane) {
  for (int j = 0; j < m; ++j) {
    const double xj = x[j] / AT(L, j, j);
    wsync();
    if (lane == 0) x[j] = xj;
    for (int k = j + 1 + lane; k < m; k += 64) x[k] = x[k] - AT(L, k, j) * xj;
    wsync();
  }
}
// L' x = y (toolkit/lcp.cc:58-62)
__device__ void ltsolve(const double *L, int n, int m, double *x, int lane) {
  for (int j = m - 1; j >= 0; --j) {
    const double xj = x[j] / AT(L, j, j);
    wsync();
    if (lane == 0) x[j] = xj;
    for (int k = lane; k < j; k += 64) x[k] = x[k] - AT(L, j, k) * xj;
    wsync();
  }
}
__device__ void lltsolve(const double *L, int n, int m, double *x, int lane) {
  lsolve(L, n, m, x, lane);
  ltsolve(L, n, m, x, lane);
}

// L L' += sigma vec vec' on the p x p block at (i0, i0) (toolkit/lcp.cc:76-83); temp = p doubles, vec may be temp
__device__ bool rank_update(double *L, int n, int i0, int p, const double *vec, double sigma, double *temp, int lane) {
  for (int k = lane; k < p; k += 64) temp[k] = vec[k];
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
    for (int k = j + 1 + lane; k < p; k += 64) {
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
  for (int k = lane; k < m - 1; k += 64) ell[k] = AT(A, m - 1, k);
  wsync();
  lsolve(L, n, m - 1, ell, lane);
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
__device__ bool swap_cholesky_rows(const double *A, int n, int i, int m, double *L, double *wq, double *temp, int lane) {
  if (m <= 1 || i == m - 1) return true;
  if (i == 0) {
    const double head = (AT(A, m - 1, m - 1) - AT(A, 0, 0)) * 0.5;
    for (int k = lane; k < m - 1; k += 64) wq[k] = (k == 0) ? head + 1.0 : AT(A, m - 1, k) - AT(A, k, 0);
    wsync();
    if (!rank_update(L, n, 0, m - 1, wq, 0.5, temp, lane)) return false;
    if (lane == 0) wq[0] = head - 1.0;
    wsync();
    return rank_update(L, n, 0, m - 1, wq, -0.5, temp, lane);
  }
  double *l1 = &AT(L, i, 0);
  for (int k = lane; k < i; k += 64) l1[k] = AT(A, m - 1, k);
  wsync();
  lsolve(L, n, i, l1, lane);
  double s = 0.0;
  for (int k = 0; k < i; ++k) s = s + l1[k] * l1[k];
  const double d = AT(A, m - 1, m - 1) - s;
  if (!(d > 0.0)) return false;
  const double e = sqrt(d);
  wsync();
  if (lane == 0) AT(L, i, i) = e;
  const int p = m - 2 - i;
  if (p > 0) {
    for (int k = lane; k < p; k += 64) wq[k] = AT(L, i + 1 + k, i);
    wsync();
    if (!rank_update(L, n, i + 1, p, wq, 1.0, temp, lane)) return false;
    for (int k = lane; k < p; k += 64) {
      double t = 0.0;
      for (int c = 0; c < i; ++c) t = t + AT(L, i + 1 + k, c) * l1[c];
      const double v = (AT(A, m - 1, i + 1 + k) - t) / e;
      AT(L, i + 1 + k, i) = v;
      wq[k] = v;
    }
    wsync();
    if (!rank_update(L, n, i + 1, p, wq, -1.0, temp, lane)) return false;
  }
  wsync();
  return true;
}

// toolkit/lcp.cc:171-195
__device__ void swap_rows_and_columns(double *A, int n, int i, int j, int *perm, int lane) {
  if (i == j) return;
  if (i > j) { const int t = i; i = j; j = t; }
  wsync();
  for (int c = lane; c < i; c += 64) { const double t = AT(A, i, c); AT(A, i, c) = AT(A, j, c); AT(A, j, c) = t; }
  for (int r = j + 1 + lane; r < n; r += 64) { const double t = AT(A, r, i); AT(A, r, i) = AT(A, r, j); AT(A, r, j) = t; }
  for (int k = i + 1 + lane; k < j; k += 64) { const double t = AT(A, k, i); AT(A, k, i) = AT(A, j, k); AT(A, j, k) = t; }
  if (lane == 0) {
    const double t = AT(A, i, i); AT(A, i, i) = AT(A, j, j); AT(A, j, j) = t;
    const int q = perm[i]; perm[i] = perm[j]; perm[j] = q;
  }
  wsync();
}

__device__ __forceinline__ void swap_entry(double *v, int a, int b, int lane) {
  if (lane == 0) { const double t = v[a]; v[a] = v[b]; v[b] = t; }
}

// SolveLCP_BoxDantzig, toolkit/lcp.cc:444-619.  gA in/out (row-major n x n, lower triangle), gx / gw / gperm out.
__global__ void __launch_bounds__(64) box_dantzig_kernel(int n, double *gA, const double *gb, const double *glo, const double *ghi,
                                                         double *gx, double *gw, int32_t *gperm, DantzigResult *res,
                                                         int max_steps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *A = reinterpret_cast<double *>(smem);
  double *L = A + (size_t)n * n;
  double *x = L + (size_t)n * n, *w = x + n, *lo = w + n, *hi = lo + n, *b = hi + n, *dxS = b + n, *dwNS = dxS + n,
         *limit = dwNS + n, *v = limit + n, *wq = v + n, *temp = wq + n;
  int *perm = reinterpret_cast<int *>(temp + n);
  const int lane = threadIdx.x;
  for (int k = lane; k < n * n; k += 64) { A[k] = gA[k]; L[k] = 0.0; }
  for (int k = lane; k < n; k += 64) { x[k] = 0.0; w[k] = 0.0; lo[k] = glo[k]; hi[k] = ghi[k]; b[k] = gb[k]; perm[k] = k; }
  wsync();

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
    for (int k = lane; k < index; k += 64) dxS[k] = -dir * AT(A, i, k);
    wsync();
    lltsolve(L, n, index, dxS, lane);
    const double delta_xi = dir;
    while (true) {
      if (++steps > max_steps) { ok = false; break; }
      // delta_w on the rows outside the set, one lane per row, the row's products in column order
      for (int r = index + lane; r < i; r += 64) {
        double t = 0.0;
        for (int k = 0; k < index; ++k) t = t + AT(A, r, k) * dxS[k];
        dwNS[r - index] = t + AT(A, i, r) * dir;
      }
      double delta_wi = 0.0;
      for (int k = 0; k < index; ++k) delta_wi = delta_wi + AT(A, i, k) * dxS[k];
      delta_wi = delta_wi + AT(A, i, i) * dir;
      for (int j = lane; j < index; j += 64) limit[j] = (dxS[j] > 0.0) ? hi[j] : lo[j];
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
      for (int k = lane; k < index; k += 64) x[k] = x[k] + best_alpha * dxS[k];
      for (int r = index + lane; r < i; r += 64) w[r] = w[r] + best_alpha * dwNS[r - index];
      if (lane == 0) { x[i] = x[i] + best_alpha * delta_xi; w[i] = w[i] + best_alpha * delta_wi; }
      wsync();
      index_i_into_set = (best_index == i && index_i_into_set);
      if (best_index < index) {
        if (lane == 0) x[best_index] = limit[best_index];
        wsync();
        if (!swap_cholesky_rows(A, n, best_index, index, L, wq, temp, lane)) { ok = false; break; }
        swap_rows_and_columns(A, n, index - 1, best_index, perm, lane);
        swap_entry(x, index - 1, best_index, lane); swap_entry(lo, index - 1, best_index, lane); swap_entry(hi, index - 1, best_index, lane);
        --index;
        wsync();
        for (int k = lane; k < index; k += 64) dxS[k] = -dir * AT(A, i, k);
        wsync();
        lltsolve(L, n, index, dxS, lane);
      } else if (best_index < i || index_i_into_set) {
        if (lane == 0) w[index_i_into_set ? i : best_index] = 0.0;
        wsync();
        swap_rows_and_columns(A, n, index, best_index, perm, lane);
        swap_entry(x, index, best_index, lane); swap_entry(w, index, best_index, lane);
        swap_entry(lo, index, best_index, lane); swap_entry(hi, index, best_index, lane);
        wsync();
        if (!add_cholesky_row(A, n, index + 1, L, lane)) { ok = false; break; }
        if (best_index != i) {
          double t = 0.0;
          for (int k = 0; k < index; ++k) t = t + AT(A, index, k) * dxS[k];
          const double value = (-dir * AT(A, i, index) - t) / (AT(L, index, index) * AT(L, index, index));
          for (int k = lane; k < index; k += 64) v[k] = AT(L, index, k);
          wsync();
          if (lane == 0) dxS[index] = value;
          ltsolve(L, n, index, v, lane);
          for (int k = lane; k < index; k += 64) dxS[k] = dxS[k] - value * v[k];
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
  for (int k = lane; k < n; k += 64) { gx[perm[k]] = x[k]; gw[perm[k]] = w[k]; gperm[k] = perm[k]; }
  for (int k = lane; k < n * n; k += 64) {
    const int r = k / n, c = k - r * n;
    if (c <= r) gA[k] = A[k];      // the lower triangle only, as the reference
  }
  if (lane == 0) { res->ok = ok ? 1 : 0; res->pivots = steps; }
}

// in-place Cholesky of the lower triangle, column by column, a lane per row below the pivot (toolkit/lcp.cc:46-48)
__device__ bool cholesky(double *L, int n, int lane) {
  for (int j = 0; j < n; ++j) {
    double d = AT(L, j, j);
    for (int k = 0; k < j; ++k) d = d - AT(L, j, k) * AT(L, j, k);
    if (!(d > 0.0)) return false;
    d = sqrt(d);
    wsync();
    if (lane == 0) AT(L, j, j) = d;
    for (int i = j + 1 + lane; i < n; i += 64) {
      double s = AT(L, i, j);
      for (int k = 0; k < j; ++k) s = s - AT(L, i, k) * AT(L, j, k);
      AT(L, i, j) = s / d;
    }
    wsync();
  }
  return true;
}

// SolveLCP_BoxMurty on a LinearReducer (toolkit/lcp.cc:213-328, 380-442); SolveLCP_Murty (:333-378) is the same loop
// with lo = 0, hi = +inf.  gA in/out (lower triangle), gx / gw / gperm out.
__global__ void __launch_bounds__(64) box_murty_kernel(int n, double *gA, const double *gb, const double *glo, const double *ghi,
                                                       double *gx, double *gw, int32_t *gperm, DantzigResult *res,
                                                       int max_iterations) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *A = reinterpret_cast<double *>(smem);
  double *L = A + (size_t)n * n;
  double *x = L + (size_t)n * n, *w = x + n, *lo = w + n, *hi = lo + n, *b = hi + n, *xs = b + n, *c = xs + n, *c2 = c + n,
         *t = c2 + n, *wq = t + n, *temp = wq + n;
  int *perm = reinterpret_cast<int *>(temp + n), *iperm = perm + n;
  const int lane = threadIdx.x;
  for (int k = lane; k < n * n; k += 64) {
    const int r = k / n, q = k - r * n;
    A[k] = gA[k];
    L[k] = (q <= r) ? gA[k] : 0.0;
  }
  for (int k = lane; k < n; k += 64) { x[k] = 0.0; w[k] = 0.0; lo[k] = glo[k]; hi[k] = ghi[k]; b[k] = gb[k]; xs[k] = gb[k]; c[k] = 0.0; perm[k] = k; iperm[k] = k; }
  wsync();
  // LinearReducer::LinearReducer (:213-224): factor all of A, xs = A^-1 b
  bool ok = cholesky(L, n, lane);
  if (ok) lltsolve(L, n, n, xs, lane);
  int index = n, it = 0;
  bool solved = false;
  for (; ok && it < max_iterations; ++it) {
    // SubSolve (:245-296)
    wsync();
    if (index == 0) {
      for (int i = lane; i < n; i += 64) x[i] = c[i];
    } else if (index >= n) {
      for (int i = lane; i < n; i += 64) x[perm[i]] = xs[i];
    } else {
      for (int i = lane; i < n; i += 64) c2[i] = c[perm[i]];
      wsync();
      for (int k = lane; k < index; k += 64) {
        double s = 0.0;
        for (int r = index; r < n; ++r) s = s + AT(A, r, k) * (c2[r] - xs[r]);
        t[k] = s;
      }
      wsync();
      lltsolve(L, n, index, t, lane);
      for (int i = lane; i < n; i += 64) x[perm[i]] = (i < index) ? xs[i] - t[i] : c[perm[i]];
    }
    wsync();
    // MultiplyA (:298-322) on the rows outside the set; w = A x - b there, 0 inside (:396-404).  c2 = x permuted
    for (int i = lane; i < n; i += 64) c2[i] = x[perm[i]];
    wsync();
    for (int r = lane; r < n; r += 64) {
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
      if (!swap_cholesky_rows(A, n, p, index, L, wq, temp, lane)) { ok = false; break; }
      --index;
    }
    if (index != p) {
      const int a = perm[index], bq = perm[p];
      swap_rows_and_columns(A, n, index, p, perm, lane);
      if (lane == 0) { iperm[a] = p; iperm[bq] = index; const double tt = xs[index]; xs[index] = xs[p]; xs[p] = tt; }
      wsync();
    }
    if (dirn > 0) {       // AddIndex (:226-234)
      ++index;
      if (!add_cholesky_row(A, n, index, L, lane)) { ok = false; break; }
      if (lane == 0) c[who] = 0.0;
    }
    wsync();
  }
  wsync();
  for (int k = lane; k < n; k += 64) { gx[k] = x[k]; gw[k] = w[k]; gperm[k] = perm[k]; }
  for (int k = lane; k < n * n; k += 64) {
    const int r = k / n, q = k - r * n;
    if (q <= r) gA[k] = A[k];
  }
  if (lane == 0) { res->ok = (ok && solved) ? 1 : 0; res->pivots = it; }
}

#undef AT

struct HipErr : std::runtime_error {
  using std::runtime_error::runtime_error;
};
void chk(hipError_t e, const char *what) {
  if (e != hipSuccess) throw HipErr(std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(call) chk((call), #call)

}  // namespace

bool box_lcp_incremental(hipStream_t stream, int algorithm, int n, double *A, const double *b, const double *lo, const double *hi,
                         double *x, double *w, int32_t *perm, int max_steps, int *pivots, std::string *msg) {
  if (n <= 0 || n > kDantzigMaxRows) throw std::invalid_argument("box_lcp_incremental: 1 <= n <= 96");
  if (algorithm != 0 && algorithm != 1) throw std::invalid_argument("box_lcp_incremental: algorithm 0 (Murty) or 1 (Cottle-Dantzig)");
  for (int i = 0; i < n; ++i) {
    // lo <= 0 <= hi (toolkit/lcp.h:134); Dantzig also needs lo < hi (toolkit/lcp.cc:448-450)
    if (!(lo[i] <= 0.0) || !(hi[i] >= 0.0) || (algorithm == 1 && !(lo[i] < hi[i])))
      throw std::invalid_argument("box_lcp_incremental: needs lo <= 0 <= hi (and lo < hi for Cottle-Dantzig)");
  }
  const size_t nn = (size_t)n * n;
  // plain hipMalloc / hipFree on purpose: with a stream-ordered pool allocation (hipMallocAsync / hipFreeAsync)
  // back-to-back calls that got the same block back saw stale lines of the previous call's matrix on this
  // stack (ROCm 7.2; tests/test_gpu_dantzig.py::test_repeated_calls_are_independent)
  double *dA = nullptr, *dv = nullptr;
  int32_t *dperm = nullptr;
  DantzigResult *dres = nullptr;
  HIPCHK(hipMalloc(&dA, nn * sizeof(double)));
  HIPCHK(hipMalloc(&dv, 5 * (size_t)n * sizeof(double)));
  HIPCHK(hipMalloc(&dperm, (size_t)n * sizeof(int32_t)));
  HIPCHK(hipMalloc(&dres, sizeof(DantzigResult)));
  double *db = dv, *dlo = dv + n, *dhi = dv + 2 * n, *dx = dv + 3 * n, *dw = dv + 4 * n;
  bool good = false;
  try {
    HIPCHK(hipMemcpyAsync(dA, A, nn * sizeof(double), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(db, b, (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(dlo, lo, (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(dhi, hi, (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream));
    const size_t lds = (2 * nn + 11 * (size_t)n) * sizeof(double) + 2 * (size_t)n * sizeof(int);
    const int limit = max_steps > 0 ? max_steps : 0x7fffffff;
    if (algorithm == 1) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(box_dantzig_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(box_dantzig_kernel, dim3(1), dim3(64), lds, stream, n, dA, db, dlo, dhi, dx, dw, dperm, dres, limit);
    } else {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(box_murty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(box_murty_kernel, dim3(1), dim3(64), lds, stream, n, dA, db, dlo, dhi, dx, dw, dperm, dres, limit);
    }
    HIPCHK(hipGetLastError());
    DantzigResult r{};
    std::vector<int32_t> hperm(n);
    HIPCHK(hipMemcpyAsync(A, dA, nn * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(x, dx, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(w, dw, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(hperm.data(), dperm, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(&r, dres, sizeof(r), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (perm) for (int i = 0; i < n; ++i) perm[i] = hperm[i];
    if (pivots) *pivots = r.pivots;
    good = r.ok != 0;
    if (!good && msg) *msg = (max_steps > 0 && r.pivots >= max_steps) ? "incremental box LCP: iteration limit reached" : "incremental box LCP: a factor update met a non-positive pivot (A not positive definite?)";
  } catch (...) {
    (void)hipFree(dA); (void)hipFree(dv); (void)hipFree(dperm); (void)hipFree(dres);
    throw;
  }
  (void)hipFree(dA); (void)hipFree(dv); (void)hipFree(dperm); (void)hipFree(dres);
  return good;
}

}  // namespace egs
