// dense_iter.hip -- the reference's projected iterations on an EXPLICIT dense matrix:
//   sparse::{Jacobi,GaussSeidel,SOR}Iteration(const MatrixXd& A, const VectorXd& b[, C, x_lo, x_hi])
//       sparse_iterations.h:13-24, sparse_iterations.cc:72-144 (BaseIteration), :35-49 (GetResidualError)
//   sparse::MatrixSolveDiagonal / MatrixSolveLowerTriangle / MatrixSolveUpperTriangle (the dense twins)
//       sparse_iterations_utils.cc:25-40, 110-128, 245-262; ApplyProjection :12-21
// These are the solvers the reference's own unit tests run on random 3..50-row matrices (sparse_iterations.cc:355-513);
// the ensemble path uses the matrix-free twins (step_solve.hip and friends).  One workgroup does the whole solve --
// x0 = b, then sweep / residual / stopping test until err <= tol or max_iters sweeps, as the reference's loop -- with
// every vector in LDS and one read-back for the caller.  A sweep is a chain of n dependent scalar updates; rows work in
// parallel wherever the reference's summation order allows it:
//   * N x + b: a thread per row, the row's products in increasing column order;
//   * forward solve (Gauss-Seidel): column by column -- thread j finishes x_j, every thread i > j adds L(i,j) x_j to its
//     running sum, which is the reference's `substitutions += L(i, j) * x(j)` order, j increasing;
//   * backward solve (SOR): the reference sums U(i, j) x(j) for j = i + 1 .. n - 1 in INCREASING j, i.e. starting with the
//     value that was finished last, so row i's sum cannot start before x_{i+1} is known: thread i runs it when its turn
//     comes (n^2 / 2 dependent multiply-adds per sweep; fine at the reference's sizes);
//   * residual: w = A x - b a thread per row; the four partial sums of squares in index order by one thread, so that the
//     stopping test sees the bits the sequential code sees (oracle/dense_iter.c) and stops at the same sweep.
// Not restated: the spectral-radius gate of :113-121 (EigenSolver; the reference Panics when rho(M^-1 N) >= 1): a
// splitting that does not converge runs to max_iters and reports its residual.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <stdexcept>

#include "dense_lcp.h"

namespace egs {

namespace {

constexpr int kDenseIterMax = 1024;

__device__ __forceinline__ double dproj(double x, bool is_eq, double lo, double hi) {   // sparse_iterations_utils.cc:12-21
  if (is_eq) return x;
  if (x < lo) return lo;
  if (x > hi) return hi;
  return x;
}

struct DenseIterOut { double residual; int32_t iterations, pad; };

__global__ void __launch_bounds__(1024) dense_iterate_kernel(int n, const double *A, const double *b, const uint8_t *C, const double *lo,
                                                             const double *hi, int method, double ksor, int max_iters, double tol,
                                                             double *x_out, DenseIterOut *out) {
  __shared__ double x[kDenseIterMax], xn[kDenseIterMax], rhs[kDenseIterMax], w[kDenseIterMax];
  __shared__ double s_err;
  const int tid = threadIdx.x;
  auto residual = [&]() {           // sparse_iterations.cc:35-49
    for (int i = tid; i < n; i += 1024) {
      double t = 0.0;
      const double *row = A + (size_t)i * n;
      for (int j = 0; j < n; ++j) t = t + row[j] * x[j];
      w[i] = t - b[i];
    }
    __syncthreads();
    if (tid == 0) {
      double s_eq = 0.0, s_lo = 0.0, s_hi = 0.0, s_in = 0.0;
      for (int i = 0; i < n; ++i) {
        const double wi = w[i], xi = x[i];
        if (C[i]) s_eq += wi * wi;
        else {
          if (xi == lo[i] && wi < 0) s_lo += wi * wi;
          if (xi == hi[i] && wi > 0) s_hi += wi * wi;
          if (xi > lo[i] && xi < hi[i]) s_in += wi * wi;
        }
      }
      s_err = sqrt(s_eq) + (sqrt(s_lo) + sqrt(s_hi) + sqrt(s_in));
    }
    __syncthreads();
    return s_err;
  };
  for (int i = tid; i < n; i += 1024) x[i] = b[i];          // x0 = b (:124)
  __syncthreads();
  double err = residual();
  int it = 0;
  while (err > tol && it < max_iters) {
    for (int i = tid; i < n; i += 1024) {                     // rhs = N x + b (:130)
      const double *row = A + (size_t)i * n;
      double t = 0.0;
      if (method == 0) { for (int j = 0; j < n; ++j) if (j != i) t = t + (-row[j]) * x[j]; }
      else if (method == 1) { for (int j = i + 1; j < n; ++j) t = t + (-row[j]) * x[j]; }
      else {
        for (int j = 0; j < i; ++j) t = t + (-row[j]) * x[j];
        t = t + ((ksor - 1.0) * row[i]) * x[i];
      }
      rhs[i] = t + b[i];
      w[i] = 0.0;                                             // running substitution sums of the forward solve
    }
    __syncthreads();
    if (method == 0) {                                        // MatrixSolveDiagonal (utils :25-40)
      for (int i = tid; i < n; i += 1024) xn[i] = dproj(1.0 / A[(size_t)i * n + i] * rhs[i], C[i] != 0, lo[i], hi[i]);
      __syncthreads();
    } else if (method == 1) {                                 // MatrixSolveLowerTriangle (utils :110-128), column by column
      for (int j = 0; j < n; ++j) {
        if (tid == 0) xn[j] = dproj((rhs[j] - w[j]) / A[(size_t)j * n + j], C[j] != 0, lo[j], hi[j]);
        __syncthreads();
        const double xj = xn[j];
        for (int i = j + 1 + tid; i < n; i += 1024) w[i] += A[(size_t)i * n + j] * xj;
        __syncthreads();
      }
    } else {                                                  // MatrixSolveUpperTriangle (utils :245-262)
      for (int i = n - 1; i >= 0; --i) {
        if (tid == 0) {
          double sub = 0.0;
          const double *row = A + (size_t)i * n;
          for (int j = i + 1; j < n; ++j) sub += row[j] * xn[j];
          xn[i] = dproj((rhs[i] - sub) / (ksor * row[i]), C[i] != 0, lo[i], hi[i]);
        }
        __syncthreads();
      }
    }
    for (int i = tid; i < n; i += 1024) x[i] = xn[i];
    __syncthreads();
    err = residual();
    ++it;
  }
  for (int i = tid; i < n; i += 1024) x_out[i] = x[i];
  if (tid == 0) { out->residual = err; out->iterations = it; out->pad = 0; }
}

struct HipErr2 : std::runtime_error {
  using std::runtime_error::runtime_error;
};
void chk2(hipError_t e, const char *what) {
  if (e != hipSuccess) throw HipErr2(std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK2(call) chk2((call), #call)

}  // namespace

void dense_iterate(hipStream_t s, int n, const double *A, const double *b, const uint8_t *C, const double *lo, const double *hi,
                   int method, double omega, int max_iters, double tol, double *x, int *iterations, double *residual) {
  if (n < 0 || n > kDenseIterMax) throw std::invalid_argument("dense iteration: 0 <= n <= 1024");
  if (method < 0 || method > 2) throw std::invalid_argument("dense iteration: method 0 (Jacobi), 1 (Gauss-Seidel) or 2 (SOR)");
  if (!(omega > 0.0 && omega < 2.0)) throw std::invalid_argument("dense iteration: 0 < omega < 2");
  if (iterations) *iterations = 0;
  if (residual) *residual = 0.0;
  if (n == 0) return;                      // sparse_iterations.cc:79-81
  for (int i = 0; i < n; ++i)
    if (A[(size_t)i * n + i] == 0.0) throw std::invalid_argument("dense iteration: zero on the diagonal (the reference CHECKs det != 0)");
  double *dA = nullptr, *dv = nullptr;
  uint8_t *dC = nullptr;
  DenseIterOut *dout = nullptr;
  const size_t nn = (size_t)n * n;
  HIPCHK2(hipMalloc(reinterpret_cast<void **>(&dA), nn * sizeof(double)));
  HIPCHK2(hipMalloc(reinterpret_cast<void **>(&dv), 4 * (size_t)n * sizeof(double)));
  HIPCHK2(hipMalloc(reinterpret_cast<void **>(&dC), (size_t)n));
  HIPCHK2(hipMalloc(reinterpret_cast<void **>(&dout), sizeof(DenseIterOut)));
  try {
    double *db = dv, *dlo = dv + n, *dhi = dv + 2 * n, *dx = dv + 3 * n;
    HIPCHK2(hipMemcpyAsync(dA, A, nn * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK2(hipMemcpyAsync(db, b, n * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK2(hipMemcpyAsync(dlo, lo, n * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK2(hipMemcpyAsync(dhi, hi, n * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK2(hipMemcpyAsync(dC, C, (size_t)n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(dense_iterate_kernel, dim3(1), dim3(1024), 0, s, n, dA, db, dC, dlo, dhi, method, 1.0 / omega, max_iters, tol, dx, dout);
    HIPCHK2(hipGetLastError());
    DenseIterOut o{};
    HIPCHK2(hipMemcpyAsync(x, dx, n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK2(hipMemcpyAsync(&o, dout, sizeof o, hipMemcpyDeviceToHost, s));
    HIPCHK2(hipStreamSynchronize(s));
    if (iterations) *iterations = o.iterations;
    if (residual) *residual = o.residual;
  } catch (...) {
    (void)hipFree(dA); (void)hipFree(dv); (void)hipFree(dC); (void)hipFree(dout);
    throw;
  }
  (void)hipFree(dA); (void)hipFree(dv); (void)hipFree(dC); (void)hipFree(dout);
}

}  // namespace egs
