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
bool dense_mixed_constraints(hipStream_t stream, int N, const double *A, const double *b, const uint8_t *C,
                             const double *lo, const double *hi, bool use_bounds, bool block_pivoting, double *x,
                             double *w, int *pivots, std::string *msg);

}  // namespace egs
