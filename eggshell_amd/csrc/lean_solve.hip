// lean_solve.hip -- the static-timetable sweep of step_solve.hip for the batched fp64 case, cut to 128 VGPRs so that
// FOUR 256-constraint tiles are resident per CU instead of three (16 instead of 12 wavefronts: every SIMD has a pass
// to issue at every time step).
//
// step_solve_kernel's isotropic variant holds per lane J0 and J1 (36 doubles), three entries of the 3x3 diagonal
// block, 1 / denominators, rhs, bounds and lambda: 164 VGPRs.  Two facts take 42 of them away without changing one
// rounding:
//   * a constraint that joins two bodies has J1_lin = -J0_lin (contact.cc:66-99: [-Rn, ..] / [Rn, ..];
//     joints.cc:17-31: [I, ..] / [-I, ..]), so ONE 3x3 block L = J1_lin serves both sides: every product with
//     J0_lin is the exact negation of the product with L (-(a b) and fma(-a, b, -c) = -fma(a, b, c) are exact), and
//     the accumulator update fma(w (-L), dx, a) is fma(w L, -dx, a).  The host checks the property when blocks are
//     uploaded (egs_problem_set_blocks); the device assembly produces it by construction;
//   * rhs, the reciprocal denominators and the bounds are read once per update: they live in LDS, lane-major
//     (16-byte units: conflict-free ds_read_b128), 96 B per lane; the body weights sit behind the body's accumulator
//     in its LDS slot (64 B per body).  An update waits ONCE (accumulators, rhs and the first row's denominator and
//     bounds); the later rows' constants and the weights are requested while the row before them is being updated.
// Same device functions for the projection, same operation order as oracle/pgs_fast.inc: the results are those of
// step_solve_kernel / tile_solve_kernel bit for bit (signed zeros excepted: where J0_lin holds +0 for an entry whose
// mirror is also +0, a product that was +0 is -0 here; no finite value changes).
#include "kernels.h"
#include "solve_device.h"

namespace egs {

namespace {


__device__ __forceinline__ double neg(double v) { return -v; }

// p = j[0] a[0] + j[1] a[1] + j[2] a[2] in dot3h's order
__device__ __forceinline__ double dot3(const double *j, double a0, double a1, double a2) {
  double s = j[0] * a0;
  s = tfma(j[1], a1, s);
  s = tfma(j[2], a2, s);
  return s;
}

// phase 1 of an update: both accumulators and rhs, one wait
__device__ __forceinline__ void load_acc_rhs(unsigned ac0, unsigned ac1, unsigned pk, unsigned stride, double (&a0)[6], double (&a1)[6], double (&rhs)[3],
                                             double &inv0, double &lo0, double &hi0) {
  d2_t u0, u1, u2, v0, v1, v2, q0, q1, q2;
  asm volatile(
      "ds_read_b128 %0, %9\n\t"
      "ds_read_b128 %1, %9 offset:16\n\t"
      "ds_read_b128 %2, %9 offset:32\n\t"
      "ds_read_b128 %3, %10\n\t"
      "ds_read_b128 %4, %10 offset:16\n\t"
      "ds_read_b128 %5, %10 offset:32\n\t"
      "ds_read_b128 %6, %11\n\t"
      "ds_read_b128 %7, %12\n\t"
      "ds_read_b128 %8, %13\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(q0), "=&v"(q1), "=&v"(q2)
      : "v"(ac0), "v"(ac1), "v"(pk), "v"(pk + stride), "v"(pk + 2 * stride)
      : "memory");
  a0[0] = u0.x; a0[1] = u0.y; a0[2] = u1.x; a0[3] = u1.y; a0[4] = u2.x; a0[5] = u2.y;
  a1[0] = v0.x; a1[1] = v0.y; a1[2] = v1.x; a1[3] = v1.y; a1[4] = v2.x; a1[5] = v2.y;
  rhs[0] = q0.x; rhs[1] = q0.y; rhs[2] = q1.x; inv0 = q1.y; lo0 = q2.x; hi0 = q2.y;
}
// issued while the row residuals / the first row's update run, waited for just before their first use
__device__ __forceinline__ void issue2(unsigned a, unsigned b, d2_t &q0, d2_t &q1) {
  asm volatile(
      "ds_read_b128 %0, %2\n\t"
      "ds_read_b128 %1, %3"
      : "=&v"(q0), "=&v"(q1) : "v"(a), "v"(b) : "memory");
}
__device__ __forceinline__ void issue3(unsigned a, unsigned b, unsigned c, d2_t &q0, d2_t &q1, d2_t &q2) {
  asm volatile(
      "ds_read_b128 %0, %3\n\t"
      "ds_read_b128 %1, %4 offset:48\n\t"
      "ds_read_b128 %2, %5 offset:48"
      : "=&v"(q0), "=&v"(q1), "=&v"(q2) : "v"(a), "v"(b), "v"(c) : "memory");
}
__device__ __forceinline__ void landed(d2_t &q0, d2_t &q1) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1) :: "memory");
}
__device__ __forceinline__ void landed(d2_t &q0, d2_t &q1, d2_t &q2) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q2) :: "memory");
}

template <int METHOD>
__device__ __forceinline__ int lean_end(int depth, int P, int sweeps, int resume) {
  if (METHOD == 2) return (resume ? 0 : depth) + (sweeps >= 1 ? depth + P * (sweeps - 1) : 0);
  const int n_phases = sweeps + (resume ? 0 : 1);
  return n_phases >= 1 ? depth + P * (n_phases - 1) : 0;
}

template <int METHOD, int LB>
__global__ void __launch_bounds__(LB, 4) lean_step_kernel(const SolveArgs<double> A) {
  constexpr unsigned kStride = LB * 16;     // bytes between a lane's parked units
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [0, 96 B x lanes): the parked per-lane constants, unit k of lane t at (k * LB + t) * 16; then the accumulators
  d2_t *s_park = reinterpret_cast<d2_t *>(smem);
  double *s_acc = reinterpret_cast<double *>(smem + 6 * LB * sizeof(d2_t));
  const int tile = blockIdx.x, tid = threadIdx.x;
  const int nslots = A.tile_nslots[tile];
  const int32_t *slot_body = A.slot_body + A.tile_slot_off[tile];
  // a slot = the body's accumulator and its two weights (W = diag(wl, wl, wl, wa, wa, wa)): 64 bytes
  for (int s = tid; s < nslots; s += LB) {
    const int body = slot_body[s];
#pragma unroll
    for (int k = 0; k < 6; ++k) s_acc[s * 8 + k] = (A.resume && body >= 0) ? A.acc[(size_t)body * 6 + k] : 0.0;
    s_acc[s * 8 + 6] = body >= 0 ? A.Minv[(size_t)body * 36] : 0.0;
    s_acc[s * 8 + 7] = body >= 0 ? A.Minv[(size_t)body * 36 + 21] : 0.0;
  }
  const LaneDesc d = A.lanes[(size_t)tile * LB + tid];
  const bool active = d.cidx >= 0;
  const bool has0 = active && d.slot0 != 0, has1 = active && d.slot1 != 0;
  const int slot0 = active ? d.slot0 : 0, slot1 = active ? d.slot1 : 0;
  const int level = A.lane_level[(size_t)tile * LB + tid];
  const int P = A.tile_period[tile], depth = A.tile_depth[tile];

  // per-lane constants: L (the shared linear block), the two angular blocks, the body weights, three entries of D
  double L[9], G0[9], G1[9];      // G = angular 3x3, row-major
  double Dk[3] = {0.0, 0.0, 0.0};  // forward: D10, D20, D21; backward: D01, D02, D12
  double x[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 9; ++k) { L[k] = 0.0; G0[k] = 0.0; G1[k] = 0.0; }
  {
    d2_t pk[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) pk[k] = (d2_t){0.0, 0.0};
    if (active) {
      const double *j0 = A.J0 + (size_t)d.cidx * 18, *j1 = A.J1 + (size_t)d.cidx * 18;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          L[3 * r + k] = has1 ? j1[6 * r + k] : (has0 ? neg(j0[6 * r + k]) : 0.0);
          G0[3 * r + k] = has0 ? j0[6 * r + 3 + k] : 0.0;
          G1[3 * r + k] = has1 ? j1[6 * r + 3 + k] : 0.0;
        }
      double wl0 = 0.0, wa0 = 0.0, wl1 = 0.0, wa1 = 0.0;
      if (has0) { const double *W = A.Minv + (size_t)slot_body[slot0] * 36; wl0 = W[0]; wa0 = W[21]; }
      if (has1) { const double *W = A.Minv + (size_t)slot_body[slot1] * 36; wl1 = W[0]; wa1 = W[21]; }
      // D = J0 (W0 J0^T) + J1 (W1 J1^T) in load_cons's order; the linear part of side 0 is (-L)(w (-L)) = L (w L) exactly
      double Dfull[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          double d0 = 0.0, d1 = 0.0;
          if (has0) {
            d0 = L[3 * r] * (wl0 * L[3 * q]);
            d0 = tfma(L[3 * r + 1], wl0 * L[3 * q + 1], d0);
            d0 = tfma(L[3 * r + 2], wl0 * L[3 * q + 2], d0);
#pragma unroll
            for (int k = 0; k < 3; ++k) d0 = tfma(G0[3 * r + k], wa0 * G0[3 * q + k], d0);
          }
          if (has1) {
            d1 = L[3 * r] * (wl1 * L[3 * q]);
            d1 = tfma(L[3 * r + 1], wl1 * L[3 * q + 1], d1);
            d1 = tfma(L[3 * r + 2], wl1 * L[3 * q + 2], d1);
#pragma unroll
            for (int k = 0; k < 3; ++k) d1 = tfma(G1[3 * r + k], wa1 * G1[3 * q + k], d1);
          }
          Dfull[3 * r + q] = d0 + d1;
        }
      if (METHOD == 2) { Dk[0] = Dfull[1]; Dk[1] = Dfull[2]; Dk[2] = Dfull[5]; }
      else { Dk[0] = Dfull[3]; Dk[1] = Dfull[6]; Dk[2] = Dfull[7]; }
      double rhs[3], inv[3], lo[3], hi[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        inv[r] = 1.0 / ((Dfull[4 * r] + A.cfm) * A.kscale);
        rhs[r] = A.rhs[(size_t)d.cidx * 3 + r];
        lo[r] = A.lo[(size_t)d.cidx * 3 + r];
        hi[r] = A.hi[(size_t)d.cidx * 3 + r];
        clamp_bounds(A.is_eq[(size_t)d.cidx * 3 + r] != 0, lo[r], hi[r]);
        x[r] = A.resume ? A.x[(size_t)d.cidx * 3 + r] : rhs[r];
      }
      // in the order an update wants them: k0, k1, k2 = the rows in processing order (0, 1, 2 forward; 2, 1, 0 backward)
      constexpr int k0 = METHOD == 2 ? 2 : 0, k1 = 1, k2 = METHOD == 2 ? 0 : 2;
      pk[0] = (d2_t){rhs[0], rhs[1]};  pk[1] = (d2_t){rhs[2], inv[k0]}; pk[2] = (d2_t){lo[k0], hi[k0]};
      pk[3] = (d2_t){inv[k1], lo[k1]}; pk[4] = (d2_t){hi[k1], inv[k2]}; pk[5] = (d2_t){lo[k2], hi[k2]};
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) s_park[k * LB + tid] = pk[k];
  }
  const unsigned ac0 = lds_addr(s_acc + slot0 * 8), ac1 = lds_addr(s_acc + slot1 * 8), pkaddr = lds_addr(s_park + tid);
  const int t_end = lean_end<METHOD>(depth, P, A.sweeps, A.resume);
  __syncthreads();

  int sweep = A.resume ? 1 : 0;
  const int t0 = (METHOD == 2 && !A.resume) ? depth : 0;
  int due = (METHOD == 2 && A.resume) ? depth - 1 - level : level;
  if (!active || sweep > A.sweeps) due = 0x7fffffff;
  for (int t = 0; t < t_end; ++t) {
    if (due == t) {
      constexpr int k0 = METHOD == 2 ? 2 : 0, k1 = 1, k2 = METHOD == 2 ? 0 : 2;
      double a0[6], a1[6], rhs[3], inv0, lo0, hi0;
      load_acc_rhs(ac0, ac1, pkaddr, kStride, a0, a1, rhs, inv0, lo0, hi0);
      double dx[3];
      d2_t wq0, wq1;            // (wl0, wa0), (wl1, wa1): the bodies' weights, behind their accumulators
      if (sweep == 0) {
        d2_t dummy;
        issue3(pkaddr, ac0, ac1, dummy, wq0, wq1);
        dx[0] = x[0]; dx[1] = x[1]; dx[2] = x[2];
        landed(dummy, wq0, wq1);
      } else {
        // res_r = rhs_r - (cfm x_r + ((p0 + p1) + (p2 + p3))), p0 = J0_lin . a0_lin = -(L . a0_lin)
        double res[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const double p0 = neg(dot3(L + 3 * r, a0[0], a0[1], a0[2]));
          const double p1 = dot3(G0 + 3 * r, a0[3], a0[4], a0[5]);
          const double p2 = dot3(L + 3 * r, a1[0], a1[1], a1[2]);
          const double p3 = dot3(G1 + 3 * r, a1[3], a1[4], a1[5]);
          res[r] = rhs[r] - tfma(A.cfm, x[r], (p0 + p1) + (p2 + p3));
        }
        asm volatile("" : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]));
        d2_t u3, u4, u5;
        issue2(pkaddr + 3 * kStride, pkaddr + 4 * kStride, u3, u4);       // row k1's constants fly while row k0 is updated
        double tt = res[k0];
        double xn = project(tfma(tt, inv0, x[k0]), lo0, hi0);
        dx[k0] = xn - x[k0]; x[k0] = xn;
        landed(u3, u4);
        issue3(pkaddr + 5 * kStride, ac0, ac1, u5, wq0, wq1);             // row k2's bounds and the weights, during row k1
        tt = tfma(-Dk[METHOD == 2 ? 2 : 0], dx[k0], res[k1]);
        xn = project(tfma(tt, u3.x, x[k1]), u3.y, u4.x);
        dx[k1] = xn - x[k1]; x[k1] = xn;
        landed(u5, wq0, wq1);
        tt = tfma(-Dk[1], dx[k0], res[k2]);
        tt = tfma(-Dk[METHOD == 2 ? 0 : 2], dx[k1], tt);
        xn = project(tfma(tt, u4.y, x[k2]), u5.x, u5.y);
        dx[k2] = xn - x[k2]; x[k2] = xn;
      }
      // a += (w J^T) dx, rows in order 0, 1, 2 (acc_add_iso); side 0's linear part with -dx instead of -L
      if (has0) {
        const double w0 = wq0.x, w1 = wq0.y;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          double tq = tfma(w0 * L[c], neg(dx[0]), a0[c]);
          tq = tfma(w0 * L[3 + c], neg(dx[1]), tq);
          a0[c] = tfma(w0 * L[6 + c], neg(dx[2]), tq);
          double ta = tfma(w1 * G0[c], dx[0], a0[3 + c]);
          ta = tfma(w1 * G0[3 + c], dx[1], ta);
          a0[3 + c] = tfma(w1 * G0[6 + c], dx[2], ta);
        }
        store6(ac0, a0);
      }
      if (has1) {
        const double w0 = wq1.x, w1 = wq1.y;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          double tq = tfma(w0 * L[c], dx[0], a1[c]);
          tq = tfma(w0 * L[3 + c], dx[1], tq);
          a1[c] = tfma(w0 * L[6 + c], dx[2], tq);
          double ta = tfma(w1 * G1[c], dx[0], a1[3 + c]);
          ta = tfma(w1 * G1[3 + c], dx[1], ta);
          a1[3 + c] = tfma(w1 * G1[6 + c], dx[2], ta);
        }
        store6(ac1, a1);
      }
      due = (METHOD == 2 && sweep == 0) ? t0 + (depth - 1 - level) : due + P;
      if (++sweep > A.sweeps) due = 0x7fffffff;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // epilogue: lambda, w = A x - rhs, accumulators
  if (active) {
    double a0[6], a1[6], rhs[3], inv0, lo0, hi0;
    load_acc_rhs(ac0, ac1, pkaddr, kStride, a0, a1, rhs, inv0, lo0, hi0);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double p0 = neg(dot3(L + 3 * r, a0[0], a0[1], a0[2]));
      const double p1 = dot3(G0 + 3 * r, a0[3], a0[4], a0[5]);
      const double p2 = dot3(L + 3 * r, a1[0], a1[1], a1[2]);
      const double p3 = dot3(G1 + 3 * r, a1[3], a1[4], a1[5]);
      A.x[(size_t)d.cidx * 3 + r] = x[r];
      A.wres[(size_t)d.cidx * 3 + r] = tfma(A.cfm, x[r], (p0 + p1) + (p2 + p3)) - rhs[r];
    }
  }
  for (int s = tid + 1; s < nslots; s += LB) {
    const int body = slot_body[s];
    if (body < 0) continue;
#pragma unroll
    for (int k = 0; k < 6; ++k) A.acc[(size_t)body * 6 + k] = s_acc[s * 8 + k];
  }
}

size_t lean_lds_bytes(int block, int max_slots) { return 6 * (size_t)block * sizeof(d2_t) + (size_t)max_slots * 8 * sizeof(double); }

template <int LB>
void launch_lean(const SolveArgs<double> &b, int method, int n_tiles, hipStream_t s) {
  const size_t lds = lean_lds_bytes(LB, b.max_slots);
  auto k1 = lean_step_kernel<1, LB>;
  auto k2 = lean_step_kernel<2, LB>;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(method == 1 ? k1 : k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (method == 1) hipLaunchKernelGGL(k1, dim3(n_tiles), dim3(LB), lds, s, b);
  else hipLaunchKernelGGL(k2, dim3(n_tiles), dim3(LB), lds, s, b);
}

}  // namespace

void launch_lean_solve(const SolveArgs<double> &a, int method, int n_tiles, int block, hipStream_t s) {
  if (n_tiles <= 0) return;
  SolveArgs<double> b = a;
  b.n_tiles = n_tiles;
  if (block == 512) launch_lean<512>(b, method, n_tiles, s);
  else launch_lean<256>(b, method, n_tiles, s);
}

int occupancy_lean_solve(int block, int max_slots) {
  int n1 = 0, n2 = 0;
  const size_t lds = lean_lds_bytes(block, max_slots);
  if (block == 512) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, lean_step_kernel<1, 512>, 512, lds) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, lean_step_kernel<2, 512>, 512, lds) != hipSuccess) return 0;
  } else {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, lean_step_kernel<1, 256>, 256, lds) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, lean_step_kernel<2, 256>, 256, lds) != hipSuccess) return 0;
  }
  return n1 < n2 ? n1 : n2;
}

}  // namespace egs
