// dense_lcp.h -- dense direct LCP on the GPU (entry 3 of the C ABI):
// Lcp::MixedConstraintsSolver + Lcp::MurtyPrincipalPivot, eggshell/lcp.cc:141-336.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

namespace egs {

// A [N][N] row-major symmetric, b, C, lo, hi [N] on the host; x, w [N] out.
// use_bounds = false reproduces the reference (Murty on [0, inf), quirk Q3).
// block_pivoting = true replaces the reference's single-index rule by block
// principal pivoting (same solution, far fewer factorisations, no 1000-pivot cap).
// Returns the reference's bool; *pivots = number of principal pivots (solves).
// Throws std::invalid_argument / the HIP error type of capi.cpp's hip_check.
// max_pivots > 0 / max_seconds > 0: give up (return false) after that many principal pivots / that
// much wall time (lcp::Settings::max_iterations, max_time; toolkit/lcp.h:161-167).
bool dense_mixed_constraints(hipStream_t stream, int N, const double *A, const double *b, const uint8_t *C,
                             const double *lo, const double *hi, bool use_bounds, bool block_pivoting, double *x,
                             double *w, int *pivots, std::string *msg, int max_pivots = 0, double max_seconds = 0.0);
// The same with A (row-major, symmetric: the lower triangle is read) and b already on the device.
// x / w (host, [N]) may be NULL; dx_out (device, [N]) receives the solution if not NULL.
bool dense_mixed_constraints_device(hipStream_t stream, int N, const double *dA, const double *db, const uint8_t *C,
                                    const double *lo, const double *hi, bool use_bounds, bool block_pivoting,
                                    int max_pivots, double max_seconds, double *x, double *w, double *dx_out,
                                    int *pivots, std::string *msg);
// 2-norm condition number of a symmetric positive definite device matrix, what the reference gets from a JacobiSVD
// (utils.cc:256-261): lambda_max by power iteration on A times 1 / lambda_min by inverse iteration with the blocked
// Cholesky factor (N <= 1024; 60 iterations each, accurate to a few per cent unless the extreme eigenvalues are
// clustered, and never above the true value).  *pivot_bound (may be NULL) = (max L_ii / min L_ii)^2, the cheap lower
// bound, which is also what is returned beyond 1024 rows.  *spd = false (and +inf) if the factorisation breaks down.
double dense_condition_estimate(hipStream_t stream, int N, const double *dA, bool *spd, double *pivot_bound = nullptr);

// lcp::SolveLCP_BoxDantzig / SolveLCP_BoxMurty with the incremental Cholesky factor of toolkit/lcp.cc (dantzig.hip):
// one workgroup per problem.  n <= kDantzigMaxRows: one wavefront, everything in LDS; up to kIncrementalMaxRows:
// four wavefronts, the matrices in global memory.  A (host, row-major n x n; only the lower triangle is read) is
// permuted in place as the reference leaves it (lower triangle written back); perm[k] = original index of final row
// k (may be NULL).  Needs lo <= 0 <= hi (and lo < hi for Dantzig, toolkit/lcp.cc:448-450).
// algorithm 1 = SolveLCP_BoxDantzig (:444-619), 0 = SolveLCP_BoxMurty on a LinearReducer (:213-328, 380-442; with
// lo = 0, hi = +inf it is SolveLCP_Murty, :333-378).  max_steps > 0 = Settings::max_iterations, else 20 n + 1000;
// max_seconds > 0 = Settings::max_time.
constexpr int kDantzigMaxRows = 96;
constexpr int kIncrementalMaxRows = 1024;
bool box_lcp_incremental(hipStream_t stream, int algorithm, int n, double *A, const double *b, const double *lo, const double *hi,
                         double *x, double *w, int32_t *perm, int max_steps, double max_seconds, int *pivots, std::string *msg);
// `count` problems in one launch: problem k's matrix at A + sum_{j<k} n_j^2, its vectors at sum_{j<k} n_j.
// ok / pivots / reason [count] (reason: 0 solved, 1 step limit, 2 non-positive pivot, 3 time limit; may be NULL).
void box_lcp_incremental_batch(hipStream_t stream, int algorithm, int count, const int32_t *n, double *A, const double *b,
                               const double *lo, const double *hi, int max_steps, double max_seconds, double *x, double *w,
                               int32_t *perm, int32_t *ok, int32_t *pivots, int32_t *reason);
// One problem that lives on the device (dA n x n contiguous, permuted in place; dx / dw out); h_lo / h_hi = host
// copies of the bounds for the precondition check.
bool box_lcp_incremental_device(hipStream_t stream, int algorithm, int n, double *dA, const double *db, const double *dlo,
                                const double *dhi, const double *h_lo, const double *h_hi, int max_steps, double max_seconds,
                                double *dx, double *dw, int *pivots, std::string *msg);

// lcp::SolveLCP_BoxSchur (toolkit/lcp.cc:627-747): the two-pointer partition that brings the unbounded rows to the
// front (A's lower triangle permuted in place, perm[k] = original index of row k, *nub_out = their number), Z = L L',
// the Schur complement R = C - B Z^-1 B' and the reduced right-hand side on the blocked MFMA factorisation, the box
// LCP on R by the incremental-factor solver above (algorithm 0 / 1; beyond kIncrementalMaxRows bounded rows by block
// principal pivoting), then y = Z^-1 (c - B' z).  nub_arg >= 0 is the reference's test hook (:623-626), -1 scans the
// bounds; q6 keeps the reference's literal classification test.  A: host, row-major, only the
// lower triangle is read or written.
bool box_lcp_schur(hipStream_t stream, int n, double *A, const double *b, const double *lo, const double *hi, int algorithm,
                   int nub_arg, bool q6, int max_steps, double max_seconds, double *x, double *w, int32_t *perm, int *nub_out,
                   int *pivots, std::string *msg);

// sparse::{Jacobi,GaussSeidel,SOR}Iteration on an explicit dense matrix (sparse_iterations.cc:72-144; dense_iter.hip):
// A row-major n x n (n <= 1024), C / lo / hi as the reference's 5-argument overloads (all-equality for the 2-argument
// ones), method 0 / 1 / 2, omega = 1.5 in the reference (:15), max_iters = 500 (:19), tol = 1e-9 (constants.h:5).
void dense_iterate(hipStream_t stream, int n, const double *A, const double *b, const uint8_t *C, const double *lo,
                   const double *hi, int method, double omega, int max_iters, double tol, double *x, int *iterations,
                   double *residual);

}  // namespace egs
