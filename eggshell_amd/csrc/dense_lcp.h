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
// Condition estimate of a symmetric positive definite device matrix from its Cholesky factor:
// (max_i L_ii / min_i L_ii)^2, a lower bound of the 2-norm condition number the reference gets from
// a JacobiSVD (utils.cc:256-261).  *spd = false (and +inf) if the factorisation breaks down.
double dense_condition_estimate(hipStream_t stream, int N, const double *dA, bool *spd);

// lcp::SolveLCP_BoxDantzig with the incremental Cholesky factor of toolkit/lcp.cc (dantzig.hip): one wavefront,
// everything in LDS, n <= kDantzigMaxRows.  A (host, row-major n x n; only the lower triangle is read) is permuted
// in place as the reference leaves it (lower triangle written back); perm[k] = original index of final row k (may
// be NULL).  Needs lo <= 0 <= hi, lo < hi (toolkit/lcp.cc:448-450).  max_steps > 0: give up after that many steps.
constexpr int kDantzigMaxRows = 96;
// algorithm 1 = SolveLCP_BoxDantzig, 0 = SolveLCP_BoxMurty on a LinearReducer (toolkit/lcp.cc:213-328, 380-442; with
// lo = 0, hi = +inf it is SolveLCP_Murty, :333-378); max_steps = Settings::max_iterations.
bool box_lcp_incremental(hipStream_t stream, int algorithm, int n, double *A, const double *b, const double *lo, const double *hi,
                         double *x, double *w, int32_t *perm, int max_steps, int *pivots, std::string *msg);

}  // namespace egs
