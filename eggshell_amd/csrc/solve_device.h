// solve_device.h -- device helpers shared by the solve kernels (kernels.hip,
// patch_solve.hip): the oracle-ordered arithmetic (oracle/pgs_fast.inc), the
// per-constraint register block, and the hand-placed LDS / global hand-offs.
// Everything lives in an anonymous namespace: one private copy per translation unit.
#pragma once
#include "kernels.h"

namespace egs {
namespace {

template <typename T> __device__ __forceinline__ T tfma(T a, T b, T c);
template <> __device__ __forceinline__ double tfma<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <> __device__ __forceinline__ float tfma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T>
__device__ __forceinline__ T dot6(const T *j, const T *a) {
  T s = j[0] * a[0];
  s = tfma(j[1], a[1], s);
  s = tfma(j[2], a[2], s);
  s = tfma(j[3], a[3], s);
  s = tfma(j[4], a[4], s);
  s = tfma(j[5], a[5], s);
  return s;
}

// three-term chain over one half (linear or angular) of a 3x6 row
template <typename T>
__device__ __forceinline__ T dot3h(const T *j, const T *a) {
  T s = j[0] * a[0];
  s = tfma(j[1], a[1], s);
  s = tfma(j[2], a[2], s);
  return s;
}
// (A x)_row without cfm: (p0 + p1) + (p2 + p3), oracle row_dot
template <typename T>
__device__ __forceinline__ T row_dot(const T *j0, const T *a0, const T *j1, const T *a1) {
  const T p0 = dot3h(j0, a0), p1 = dot3h(j0 + 3, a0 + 3);
  const T p2 = dot3h(j1, a1), p3 = dot3h(j1 + 3, a1 + 3);
  return (p0 + p1) + (p2 + p3);
}

// sparse_iterations_utils.cc:12-21, branch-free: same result for every input
// (NaN compares false and passes through, as in the reference).  Equality rows carry
// lo = -inf, hi = +inf in registers (clamp_bounds below), for which the two selects
// return x itself -- one select pair less on the dependent chain of every row.
template <typename T>
__device__ __forceinline__ T project(T x, T lo, T hi) {
  T r = x;
  r = (x > hi) ? hi : r;
  r = (x < lo) ? lo : r;
  return r;
}
template <typename T>
__device__ __forceinline__ void clamp_bounds(bool eq, T &lo, T &hi) {
  if (eq) { lo = -__builtin_huge_val(); hi = __builtin_huge_val(); }
}

// a += B d, rows applied in order 0,1,2 (oracle acc_add)
template <typename T>
__device__ __forceinline__ void acc_add(T *a, const T *B, const T *d) {
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    T t = tfma(B[3 * c + 0], d[0], a[c]);
    t = tfma(B[3 * c + 1], d[1], t);
    t = tfma(B[3 * c + 2], d[2], t);
    a[c] = t;
  }
}

// The same with B[c][r] = w_c J[r][c] formed on the fly (isotropic bodies: W is
// diagonal with equal linear and equal angular entries).  One rounding per
// product, as in the stored B, so the bits are those of acc_add.
template <typename T>
__device__ __forceinline__ void acc_add_iso(T *a, const T *J, T wl, T wa, const T *d) {
  // Opaque to the optimiser: otherwise the loop-invariant products w * J are hoisted out of
  // the sweep loop, i.e. B is back in 72 registers (or in scratch).
  asm volatile("" : "+v"(wl), "+v"(wa));
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const T w = c < 3 ? wl : wa;
    T t = tfma(w * J[c], d[0], a[c]);
    t = tfma(w * J[6 + c], d[1], t);
    t = tfma(w * J[12 + c], d[2], t);
    a[c] = t;
  }
}
__device__ __forceinline__ unsigned lds_load_acquire(const unsigned *p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store_release(unsigned *p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Per-lane constants of one constraint, held in registers for all sweeps.
template <typename REAL>
struct Cons {
  REAL J0[18], J1[18];  // 3x6 row-major, zero for a world side
  REAL B0[18], B1[18];  // (W J^T) as 6x3 row-major
  REAL D[9];            // J0 B0 + J1 B1
  REAL wl0, wa0, wl1, wa1;  // ISO mode: W = diag(wl, wl, wl, wa, wa, wa) per body, B is not stored
  REAL inv[3];          // 1 / ((D_rr + cfm) * kscale)
  REAL rhs[3], lo[3], hi[3];
  bool eq[3];
};

template <bool ISO, typename REAL>
__device__ __forceinline__ void acc_add_side0(REAL *a, const Cons<REAL> &c, const REAL *d) {
  if (ISO) acc_add_iso(a, c.J0, c.wl0, c.wa0, d); else acc_add(a, c.B0, d);
}
template <bool ISO, typename REAL>
__device__ __forceinline__ void acc_add_side1(REAL *a, const Cons<REAL> &c, const REAL *d) {
  if (ISO) acc_add_iso(a, c.J1, c.wl1, c.wa1, d); else acc_add(a, c.B1, d);
}

template <typename REAL, bool ISO = false>
__device__ __forceinline__ void load_cons(const SolveArgs<REAL> &A, int cidx, bool has0, bool has1,
                                          int body0, int body1, Cons<REAL> &c) {
#pragma unroll
  for (int k = 0; k < 18; ++k) {
    c.J0[k] = has0 ? A.J0[(size_t)cidx * 18 + k] : REAL(0);
    c.J1[k] = has1 ? A.J1[(size_t)cidx * 18 + k] : REAL(0);
  }
  c.wl0 = c.wa0 = c.wl1 = c.wa1 = REAL(0);
  if (ISO) {
    // B[k][q] = w_k J[q][k] is formed where it is needed; D = J0 B0 + J1 B1 in the stored-B order
    if (has0) { c.wl0 = A.Minv[(size_t)body0 * 36]; c.wa0 = A.Minv[(size_t)body0 * 36 + 21]; }
    if (has1) { c.wl1 = A.Minv[(size_t)body1 * 36]; c.wa1 = A.Minv[(size_t)body1 * 36 + 21]; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        REAL d0 = c.J0[6 * r] * (c.wl0 * c.J0[6 * q]);
#pragma unroll
        for (int k = 1; k < 6; ++k) d0 = tfma(c.J0[6 * r + k], (k < 3 ? c.wl0 : c.wa0) * c.J0[6 * q + k], d0);
        REAL d1 = c.J1[6 * r] * (c.wl1 * c.J1[6 * q]);
#pragma unroll
        for (int k = 1; k < 6; ++k) d1 = tfma(c.J1[6 * r + k], (k < 3 ? c.wl1 : c.wa1) * c.J1[6 * q + k], d1);
        c.D[3 * r + q] = d0 + d1;
      }
  } else {
#pragma unroll
    for (int k = 0; k < 18; ++k) { c.B0[k] = REAL(0); c.B1[k] = REAL(0); }
    if (has0) {
      const REAL *W = A.Minv + (size_t)body0 * 36;
  #pragma unroll
      for (int cc = 0; cc < 6; ++cc) {
        REAL Wr[6];
  #pragma unroll
        for (int k = 0; k < 6; ++k) Wr[k] = W[6 * cc + k];
  #pragma unroll
        for (int r = 0; r < 3; ++r) c.B0[3 * cc + r] = dot6(Wr, c.J0 + 6 * r);
      }
    }
    if (has1) {
      const REAL *W = A.Minv + (size_t)body1 * 36;
  #pragma unroll
      for (int cc = 0; cc < 6; ++cc) {
        REAL Wr[6];
  #pragma unroll
        for (int k = 0; k < 6; ++k) Wr[k] = W[6 * cc + k];
  #pragma unroll
        for (int r = 0; r < 3; ++r) c.B1[3 * cc + r] = dot6(Wr, c.J1 + 6 * r);
      }
    }
  #pragma unroll
    for (int r = 0; r < 3; ++r)
  #pragma unroll
      for (int q = 0; q < 3; ++q) {
        REAL d0 = c.J0[6 * r] * c.B0[q];
  #pragma unroll
        for (int k = 1; k < 6; ++k) d0 = tfma(c.J0[6 * r + k], c.B0[3 * k + q], d0);
        REAL d1 = c.J1[6 * r] * c.B1[q];
  #pragma unroll
        for (int k = 1; k < 6; ++k) d1 = tfma(c.J1[6 * r + k], c.B1[3 * k + q], d1);
        c.D[3 * r + q] = d0 + d1;
      }
}
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    c.inv[r] = REAL(1) / ((c.D[4 * r] + A.cfm) * A.kscale);
    c.rhs[r] = A.rhs[(size_t)cidx * 3 + r];
    c.lo[r] = A.lo[(size_t)cidx * 3 + r];
    c.hi[r] = A.hi[(size_t)cidx * 3 + r];
    c.eq[r] = A.is_eq[(size_t)cidx * 3 + r] != 0;
    clamp_bounds(c.eq[r], c.lo[r], c.hi[r]);
  }
}

// res_r = rhs_r - (J_r . a + cfm x_r)
template <typename REAL>
__device__ __forceinline__ void row_residuals(const Cons<REAL> &c, const REAL *a0, const REAL *a1,
                                              const REAL *x, REAL cfm, REAL *res) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    REAL full = tfma(cfm, x[r], row_dot(c.J0 + 6 * r, a0, c.J1 + 6 * r, a1));
    res[r] = c.rhs[r] - full;
  }
}

// One projected update of the 3 rows of a constraint; returns dx.
template <typename REAL, int METHOD>
__device__ __forceinline__ void update_rows(const Cons<REAL> &c, const REAL *res, REAL *x, REAL *dx) {
  if (METHOD == 0) {  // Jacobi: no intra-block coupling
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      REAL xn = project(tfma(res[r], c.inv[r], x[r]), c.lo[r], c.hi[r]);
      dx[r] = xn - x[r];
      x[r] = xn;
    }
  } else if (METHOD == 1) {  // forward
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      REAL t = res[r];
#pragma unroll
      for (int l = 0; l < r; ++l) t = tfma(-c.D[3 * r + l], dx[l], t);
      REAL xn = project(tfma(t, c.inv[r], x[r]), c.lo[r], c.hi[r]);
      dx[r] = xn - x[r];
      x[r] = xn;
    }
  } else {  // backward
#pragma unroll
    for (int r = 2; r >= 0; --r) {
      REAL t = res[r];
#pragma unroll
      for (int l = 2; l > r; --l) t = tfma(-c.D[3 * r + l], dx[l], t);
      REAL xn = project(tfma(t, c.inv[r], x[r]), c.lo[r], c.hi[r]);
      dx[r] = xn - x[r];
      x[r] = xn;
    }
  }
}

template <typename REAL>
__device__ __forceinline__ void lds_load6(const REAL *p, REAL *a) {
#pragma unroll
  for (int k = 0; k < 6; ++k) a[k] = p[k];
}
template <typename REAL>
__device__ __forceinline__ void lds_store6(REAL *p, const REAL *a) {
#pragma unroll
  for (int k = 0; k < 6; ++k) p[k] = a[k];
}


// ---- ordered LDS hand-off, hand-placed (see quad_solve.hip) ------------------
// DS instructions of a wavefront execute in issue order: both ticket loads, then
// the accumulator loads, ONE wait; stores of the accumulators, then the tickets.
__device__ __forceinline__ unsigned lds_addr(const void *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) void *)p;
}
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
// two-stage poll: tickets only (8 bytes per lane instead of 104) ...
__device__ __forceinline__ void poll_ticks(unsigned tk0, unsigned tk1, unsigned &t0, unsigned &t1) {
  asm volatile(
      "ds_read_b32 %0, %2\n\t"
      "ds_read_b32 %1, %3\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(t0), "=&v"(t1) : "v"(tk0), "v"(tk1) : "memory");
}
// ... then the accumulators of the lanes whose turn it is
__device__ __forceinline__ void load12(unsigned ac0, unsigned ac1, double (&a0)[6], double (&a1)[6]) {
  d2_t u0, u1, u2, v0, v1, v2;
  asm volatile(
      "ds_read_b128 %0, %6\n\t"
      "ds_read_b128 %1, %6 offset:16\n\t"
      "ds_read_b128 %2, %6 offset:32\n\t"
      "ds_read_b128 %3, %7\n\t"
      "ds_read_b128 %4, %7 offset:16\n\t"
      "ds_read_b128 %5, %7 offset:32\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(ac0), "v"(ac1) : "memory");
  a0[0] = u0.x; a0[1] = u0.y; a0[2] = u1.x; a0[3] = u1.y; a0[4] = u2.x; a0[5] = u2.y;
  a1[0] = v0.x; a1[1] = v0.y; a1[2] = v1.x; a1[3] = v1.y; a1[4] = v2.x; a1[5] = v2.y;
}
__device__ __forceinline__ void load12(unsigned ac0, unsigned ac1, float (&a0)[6], float (&a1)[6]) {
  f2_t u0, u1, u2, v0, v1, v2;
  asm volatile(
      "ds_read_b64 %0, %6\n\t"
      "ds_read_b64 %1, %6 offset:8\n\t"
      "ds_read_b64 %2, %6 offset:16\n\t"
      "ds_read_b64 %3, %7\n\t"
      "ds_read_b64 %4, %7 offset:8\n\t"
      "ds_read_b64 %5, %7 offset:16\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(ac0), "v"(ac1) : "memory");
  a0[0] = u0.x; a0[1] = u0.y; a0[2] = u1.x; a0[3] = u1.y; a0[4] = u2.x; a0[5] = u2.y;
  a1[0] = v0.x; a1[1] = v0.y; a1[2] = v1.x; a1[3] = v1.y; a1[4] = v2.x; a1[5] = v2.y;
}
__device__ __forceinline__ void store6(unsigned ac, const double (&a)[6]) {
  d2_t u0 = {a[0], a[1]}, u1 = {a[2], a[3]}, u2 = {a[4], a[5]};
  asm volatile(
      "ds_write_b128 %0, %1\n\t"
      "ds_write_b128 %0, %2 offset:16\n\t"
      "ds_write_b128 %0, %3 offset:32"
      :: "v"(ac), "v"(u0), "v"(u1), "v"(u2) : "memory");
}
__device__ __forceinline__ void store6(unsigned ac, const float (&a)[6]) {
  f2_t u0 = {a[0], a[1]}, u1 = {a[2], a[3]}, u2 = {a[4], a[5]};
  asm volatile(
      "ds_write_b64 %0, %1\n\t"
      "ds_write_b64 %0, %2 offset:8\n\t"
      "ds_write_b64 %0, %3 offset:16"
      :: "v"(ac), "v"(u0), "v"(u1), "v"(u2) : "memory");
}
__device__ __forceinline__ void store_tick(unsigned tick_addr, unsigned v) {
  asm volatile("ds_write_b32 %0, %1" :: "v"(tick_addr), "v"(v) : "memory");
}


// DPP move: the value of another lane of the same row of 16 (quad_perm, row_shr:n = 0x110 + n,
// row_shl:n = 0x100 + n); lanes without a source read 0
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ float dpp(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
  const int lo = dpp_i<CTRL>(__double2loint(v)), hi = dpp_i<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

template <typename T>
__device__ __forceinline__ T gld(const T *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void gst(T *p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


}  // namespace
}  // namespace egs
