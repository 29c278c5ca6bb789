// kernels.hip -- gfx950 (CDNA4, wave64) kernels of the constraint-solve path.
//
//   assemble_kernel        K1-K4: Jacobian blocks, error, bounds, ODE rhs
//                          (joints.cc:3-35, contact.cc:14-117, ensembles.cc:569)
//   tile_solve_kernel      K5-K8: projected Jacobi / Gauss-Seidel / backward
//                          SOR over J W J^T + cfm I, one workgroup per tile of
//                          whole islands, body accumulators in LDS
//                          (sparse_iterations.cc:148-226)
//   global_solve_kernel    the same sweep for islands larger than a workgroup
//   residual_partials      K8 reduction (sparse_iterations.cc:51-69)
//   velocity_kernel        K9 (ensembles.cc:535, 572)
//
// Built with -ffp-contract=off: the only fused operations are the explicit
// fma() calls, in the order fixed by the CPU oracle (oracle/pgs_fast.inc), so
// the results can be compared bit for bit.
//
// How list order is kept without global barriers.  The reference sweeps the
// constraint list strictly in order.  Two constraints only interact through a
// shared body, so it is enough that, per body, its constraints run in list
// order, sweep after sweep.  Every body carries a ticket counter; constraint c
// with rank `pos` among the `cnt` constraints of that body may run in sweep s
// when ticket == s*cnt + pos (cnt-1-pos for the backward sweep) on BOTH of
// its bodies, and bumps both tickets when done.  The dependency graph is
// acyclic ((sweep, index) is a topological order), so there is always a lane
// that can run; consecutive sweeps pipeline through an island as a wavefront.
#include "kernels.h"
#include "solve_device.h"

#ifndef EGS_POLL_SLEEP
#define EGS_POLL_SLEEP 32  // s_sleep units (64 cycles) of a wavefront none of whose lanes is ready; s_wakeup ends it early
#endif

namespace egs {

namespace {

// Ticket-ordered a += B dx for every lane of the workgroup (list order per
// body).  Used for the initial accumulators (dx = x0) and the Jacobi sweep.
template <bool ISO, typename REAL>
__device__ __forceinline__ bool ordered_accumulate(REAL *s_acc, unsigned *s_tick, const Cons<REAL> &c,
                                                   const REAL *dx, bool active, bool has0, bool has1,
                                                   int slot0, int slot1, unsigned want0, unsigned want1,
                                                   unsigned spin_limit) {
  bool pending = active;
  bool ok = true;
  unsigned spins = 0;
  while (pending) {
    const unsigned t0 = has0 ? lds_load_acquire(s_tick + slot0) : want0;
    const unsigned t1 = has1 ? lds_load_acquire(s_tick + slot1) : want1;
    if (t0 == want0 && t1 == want1) {
      REAL a0[6], a1[6];
      if (has0) {
        lds_load6(s_acc + slot0 * 6, a0);
        acc_add_side0<ISO>(a0, c, dx);
        lds_store6(s_acc + slot0 * 6, a0);
      }
      if (has1) {
        lds_load6(s_acc + slot1 * 6, a1);
        acc_add_side1<ISO>(a1, c, dx);
        lds_store6(s_acc + slot1 * 6, a1);
      }
      if (has0) lds_store_release(s_tick + slot0, want0 + 1);
      if (has1) lds_store_release(s_tick + slot1, want1 + 1);
      pending = false;
    } else if (++spins > spin_limit) {
      ok = false;
      pending = false;
    }
  }
  return ok;
}

// ISO: every body's M^-1 block is diag(a, a, a, b, b, b) (boxes with I = c*Identity at
// their rest orientation -- every BASELINE pile).  B = M^-1 J^T is then w * J entry by
// entry and is formed on the fly instead of living in 72 VGPRs: 164 instead of 232 VGPRs,
// three wavefronts per SIMD instead of two, i.e. 768 instead of 512 constraints resident
// per CU.  Same products, same roundings, same bits.
template <typename REAL, int BLOCK, int METHOD, bool ISO>
__global__ void __launch_bounds__(BLOCK, ISO ? (sizeof(REAL) == 4 ? 4 : 3) : 1) tile_solve_kernel(const SolveArgs<REAL> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  REAL *s_acc = reinterpret_cast<REAL *>(smem);
  unsigned *s_tick = reinterpret_cast<unsigned *>(smem + (size_t)A.max_slots * 6 * sizeof(REAL));

  const int tile = blockIdx.x, tid = threadIdx.x;
  const int nslots = A.tile_nslots[tile];
  const int32_t *slot_body = A.slot_body + A.tile_slot_off[tile];

  for (int s = tid; s < nslots; s += BLOCK) {
    const int body = slot_body[s];
#pragma unroll
    for (int k = 0; k < 6; ++k)
      s_acc[s * 6 + k] = (A.resume && body >= 0) ? A.acc[(size_t)body * 6 + k] : REAL(0);
    s_tick[s] = 0u;
  }

  const LaneDesc d = A.lanes[(size_t)tile * BLOCK + tid];
  const bool active = d.cidx >= 0;
  const bool has0 = active && d.slot0 != 0, has1 = active && d.slot1 != 0;
  const int slot0 = d.slot0, slot1 = d.slot1;
  const unsigned cnt0 = d.cnt0, cnt1 = d.cnt1;
  const unsigned pos0 = d.pos0, pos1 = d.pos1;

  Cons<REAL> c;
  REAL x[3] = {REAL(0), REAL(0), REAL(0)};
  if (active) {
    load_cons<REAL, ISO>(A, d.cidx, has0, has1, has0 ? slot_body[slot0] : 0, has1 ? slot_body[slot1] : 0, c);
#pragma unroll
    for (int r = 0; r < 3; ++r) x[r] = A.resume ? A.x[(size_t)d.cidx * 3 + r] : c.rhs[r];
  }
  __syncthreads();

  bool ok = true;
  // Tickets count completed updates per body.  Sweep s (1-based) waits for
  // base + (s-1)*cnt + ord; the initial accumulation (x0 = rhs,
  // sparse_iterations.cc:202) takes tickets 0..cnt-1 unless resuming.
  const unsigned base0 = A.resume ? 0u : cnt0, base1 = A.resume ? 0u : cnt1;
  if (!A.resume) {
    ok = ordered_accumulate<ISO>(s_acc, s_tick, c, x, active, has0, has1, slot0, slot1, pos0, pos1, A.spin_limit);
    __syncthreads();
  }

  if (METHOD == 0) {
    for (int s = 1; s <= A.sweeps; ++s) {
      REAL dx[3] = {REAL(0), REAL(0), REAL(0)};
      if (active) {
        REAL a0[6], a1[6], res[3];
        lds_load6(s_acc + slot0 * 6, a0);
        lds_load6(s_acc + slot1 * 6, a1);
        row_residuals(c, a0, a1, x, A.cfm, res);
        update_rows<REAL, 0>(c, res, x, dx);
      }
      __syncthreads();  // every lane has read the old accumulators
      ok &= ordered_accumulate<ISO>(s_acc, s_tick, c, dx, active, has0, has1, slot0, slot1,
                               base0 + (unsigned)(s - 1) * cnt0 + pos0,
                               base1 + (unsigned)(s - 1) * cnt1 + pos1, A.spin_limit);
      __syncthreads();
      if (!ISO && A.hist_x) {   // snapshots for the per-sweep stopping test: the sweep is complete here
        if (active) {
          REAL *hx = A.hist_x + ((size_t)(s - 1) * A.m + d.cidx) * 3;
          hx[0] = x[0]; hx[1] = x[1]; hx[2] = x[2];
        }
        for (int q = tid + 1; q < nslots; q += BLOCK) {
          if (slot_body[q] < 0) continue;   // slot numbers have gaps (plan.cpp, bank-aware numbering)
          REAL *ha = A.hist_acc + ((size_t)(s - 1) * A.n_bodies + slot_body[q]) * 6;
#pragma unroll
          for (int k = 0; k < 6; ++k) ha[k] = s_acc[q * 6 + k];
        }
      }
    }
  } else {
    const unsigned ord0 = (METHOD == 2) ? cnt0 - 1u - pos0 : pos0;
    const unsigned ord1 = (METHOD == 2) ? cnt1 - 1u - pos1 : pos1;
    unsigned want0 = base0 + ord0, want1 = base1 + ord1;
    int sweep = 1;
    unsigned spins = 0;
    bool alive = active && A.sweeps >= 1;
    const unsigned tk0 = lds_addr(s_tick + slot0), tk1 = lds_addr(s_tick + slot1);
    const unsigned ac0 = lds_addr(s_acc + slot0 * 6), ac1 = lds_addr(s_acc + slot1 * 6);
    while (alive) {
      unsigned t0, t1;
      REAL a0[6], a1[6];
      // Two stages: every wavefront of the (two) resident tiles polls, so the poll reads the
      // tickets only (8 B per lane); the 96 B of accumulators follow for the lanes whose turn
      // it is.  One extra LDS round trip on the hand-off, but 13x less polling traffic in front
      // of the working wavefront's LDS operations: +5 % on the batched C3 solve.  A wavefront
      // with no ready lane sleeps (up to 2048 cycles) and is woken by the s_wakeup that follows
      // every ticket store in its workgroup: another +7 %.
      poll_ticks(tk0, tk1, t0, t1);                 // world slot 0: ticket ignored, zeros
      const bool ready = (!has0 || t0 == want0) && (!has1 || t1 == want1);
      if (ready) load12(ac0, ac1, a0, a1);
      if (ready) {
        REAL res[3], dx[3] = {REAL(0), REAL(0), REAL(0)};
        row_residuals(c, a0, a1, x, A.cfm, res);
        update_rows<REAL, METHOD>(c, res, x, dx);
        if (has0) { acc_add_side0<ISO>(a0, c, dx); store6(ac0, a0); }
        if (has1) { acc_add_side1<ISO>(a1, c, dx); store6(ac1, a1); }
        if (has0) store_tick(tk0, want0 + 1);
        if (has1) store_tick(tk1, want1 + 1);
        // the workgroup's sleeping wavefronts poll now instead of when their s_sleep expires
        asm volatile("s_wakeup");
        if (!ISO && A.hist_x) {   // snapshots for the per-sweep stopping test (kernels.h)
          REAL *hx = A.hist_x + ((size_t)(sweep - 1) * A.m + d.cidx) * 3;
          hx[0] = x[0]; hx[1] = x[1]; hx[2] = x[2];
          if (has0 && ord0 == cnt0 - 1u) {      // this was body0's last update of the sweep
            REAL *ha = A.hist_acc + ((size_t)(sweep - 1) * A.n_bodies + slot_body[slot0]) * 6;
#pragma unroll
            for (int k = 0; k < 6; ++k) ha[k] = a0[k];
          }
          if (has1 && ord1 == cnt1 - 1u) {
            REAL *ha = A.hist_acc + ((size_t)(sweep - 1) * A.n_bodies + slot_body[slot1]) * 6;
#pragma unroll
            for (int k = 0; k < 6; ++k) ha[k] = a1[k];
          }
        }
        want0 += cnt0; want1 += cnt1;
        spins = 0;
        alive = ++sweep <= A.sweeps;
      } else if (++spins > A.spin_limit) {
        ok = false;
        alive = false;
      }
      if (!__any(ready)) __builtin_amdgcn_s_sleep(EGS_POLL_SLEEP);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }

  if (!ok) atomicOr(A.error_flag, 1);

  // epilogue: lambda, w = A x - rhs, accumulators
  if (active) {
    REAL a0[6], a1[6];
    lds_load6(s_acc + slot0 * 6, a0);
    lds_load6(s_acc + slot1 * 6, a1);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      REAL w = tfma(A.cfm, x[r], row_dot(c.J0 + 6 * r, a0, c.J1 + 6 * r, a1)) - c.rhs[r];
      A.x[(size_t)d.cidx * 3 + r] = x[r];
      A.wres[(size_t)d.cidx * 3 + r] = w;
    }
  }
  for (int s = tid + 1; s < nslots; s += BLOCK) {
    const int body = slot_body[s];
    if (body < 0) continue;   // unused slot number
#pragma unroll
    for (int k = 0; k < 6; ++k) A.acc[(size_t)body * 6 + k] = s_acc[s * 6 + k];
  }
}

// --------------------------------------------------------------------------
// residual partial sums (sparse_iterations.cc:51-69): per block 4 sums of w^2
// over {equality, at lo & w<0, at hi & w>0, strictly inside}.
template <typename REAL>
__global__ void __launch_bounds__(256) residual_partials_kernel(int rows, const REAL *wres, const REAL *x,
                                                                const REAL *lo, const REAL *hi,
                                                                const uint8_t *is_eq, double *out) {
  __shared__ double red[4][256];
  double e = 0, a = 0, b = 0, c = 0;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) {
    const double w = (double)wres[r];
    const REAL xv = x[r], l = lo[r], h = hi[r];
    if (is_eq[r]) e += w * w;
    else {
      if (xv == l && w < 0) a += w * w;
      if (xv == h && w > 0) b += w * w;
      if (xv > l && xv < h) c += w * w;
    }
  }
  red[0][threadIdx.x] = e; red[1][threadIdx.x] = a; red[2][threadIdx.x] = b; red[3][threadIdx.x] = c;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s)
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x < 4) out[blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

// The same four sums for every sweep of a recorded chunk (SolveArgs::hist_x / hist_acc):
// w = cfm x + J a - rhs from the snapshots, evaluated with the solve kernels' own epilogue
// expression, then the reduction of residual_partials_kernel -- so each sweep's value is the
// one a launch stopped after that sweep would have produced, bit for bit.
// One block walks its rows ONCE per group of G recorded sweeps: the row's J blocks, bounds and rhs are read
// once and serve G sweeps (per sweep only lambda and the two accumulators change), instead of re-streaming
// 300 B of J per row and sweep.  Per sweep every thread adds the same rows in the same order and the block
// reduces them the same way as before, so the sums have the same bits.
template <typename REAL, int G>
__global__ void __launch_bounds__(256) hist_residual_kernel(const SolveArgs<REAL> A, double *out, int write_sweep, int sweeps) {
  __shared__ double red[4][256];
  const int s0 = blockIdx.y * G;          // first sweep of the group, 0-based
  const int rows = 3 * A.m;
  double acc[G][4];
#pragma unroll
  for (int g = 0; g < G; ++g) { acc[g][0] = 0; acc[g][1] = 0; acc[g][2] = 0; acc[g][3] = 0; }
  for (int r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) {
    const int i = r / 3, rr = r - 3 * i;
    const int b0 = A.body0[i], b1 = A.body1[i];
    REAL j0[6], j1[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      j0[k] = b0 >= 0 ? A.J0[(size_t)i * 18 + 6 * rr + k] : REAL(0);
      j1[k] = b1 >= 0 ? A.J1[(size_t)i * 18 + 6 * rr + k] : REAL(0);
    }
    const REAL l = A.lo[r], h = A.hi[r], rhs = A.rhs[r];
    const bool eq = A.is_eq[r] != 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int sweep = s0 + g;
      if (sweep >= sweeps) break;
      const REAL *ha = A.hist_acc + (size_t)sweep * A.n_bodies * 6;
      REAL a0[6], a1[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        a0[k] = b0 >= 0 ? ha[(size_t)b0 * 6 + k] : REAL(0);
        a1[k] = b1 >= 0 ? ha[(size_t)b1 * 6 + k] : REAL(0);
      }
      const REAL xv = A.hist_x[(size_t)sweep * A.m * 3 + r];
      const REAL wr = tfma(A.cfm, xv, row_dot(j0, a0, j1, a1)) - rhs;
      if (write_sweep == sweep + 1) A.wres[r] = wr;
      const double w = (double)wr;
      if (eq) acc[g][0] += w * w;
      else {
        if (xv == l && w < 0) acc[g][1] += w * w;
        if (xv == h && w > 0) acc[g][2] += w * w;
        if (xv > l && xv < h) acc[g][3] += w * w;
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int sweep = s0 + g;
    if (sweep >= sweeps) break;
    __syncthreads();
    red[0][threadIdx.x] = acc[g][0]; red[1][threadIdx.x] = acc[g][1]; red[2][threadIdx.x] = acc[g][2]; red[3][threadIdx.x] = acc[g][3];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s)
        for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x < 4) out[((size_t)sweep * gridDim.x + blockIdx.x) * 4 + threadIdx.x] = red[threadIdx.x][0];
  }
}

// --------------------------------------------------------------------------
// K9: v_new = v + dt (W f_ext + a)        (ensembles.cc:535, 572)
template <typename REAL>
__global__ void __launch_bounds__(256) velocity_kernel(int n, const double *v, const double *w, const double *Wf,
                                                       const REAL *acc, double dt, double *v6) {
  const int e = blockIdx.x * 256 + threadIdx.x;   // one lane per (body, component): every access contiguous
  if (e >= 6 * n) return;
  const int b = e / 6, r = e - 6 * b;
  const double vel = r < 3 ? v[(size_t)b * 3 + r] : w[(size_t)b * 3 + r - 3];
  v6[e] = vel + dt * (Wf[e] + (double)acc[e]);
}

// Wf = M^-1 f_ext per body, the expression of ComputeVDot's M_inverse_ * external_force_torque_
// restricted to the diagonal block (ensembles.cc:535)
__global__ void __launch_bounds__(256) mass_times_force_kernel(int n, const double *Minv, const double *f_ext, double *Wf) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 6 * n) return;
  const int b = e / 6, r = e - 6 * b;
  const double *M = Minv + (size_t)b * 36 + 6 * r, *f = f_ext + (size_t)b * 6;
  Wf[e] = ((((M[0] * f[0] + M[1] * f[1]) + M[2] * f[2]) + M[3] * f[3]) + M[4] * f[4]) + M[5] * f[5];
}

// StepPositions_ODE (ensembles.cc:577-591): p += dt (v + v_new)/2,
// R = Q(dt (w + w_new)/2) R with Q = WtoQ (utils.cc:82-89); then the new
// velocities become the body state for the next step.
__global__ void __launch_bounds__(256) advance_kernel(int n, double *pos, double *R, double *v, double *w,
                                                      const double *v6, double dt) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n) return;
  double wm[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double vm = (v[(size_t)b * 3 + k] + v6[(size_t)b * 6 + k]) / 2.0;
    pos[(size_t)b * 3 + k] = pos[(size_t)b * 3 + k] + dt * vm;
    wm[k] = (w[(size_t)b * 3 + k] + v6[(size_t)b * 6 + 3 + k]) / 2.0;
  }
  const double n2 = (wm[0] * wm[0] + wm[1] * wm[1]) + wm[2] * wm[2];
  const double nrm = sqrt(n2);
  double ax[3] = {wm[0], wm[1], wm[2]};
  if (n2 > 0) { ax[0] = wm[0] / nrm; ax[1] = wm[1] / nrm; ax[2] = wm[2] / nrm; }
  const double half = 0.5 * (nrm * dt);
  const double sn = sin(half), cs = cos(half);
  const double qw = cs, qx = sn * ax[0], qy = sn * ax[1], qz = sn * ax[2];
  const double tx = 2.0 * qx, ty = 2.0 * qy, tz = 2.0 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  const double Q[9] = {1.0 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.0 - (txx + tzz), tyz - twx,
                       txz - twy, tyz + twx, 1.0 - (txx + tyy)};
  double Ro[9], Rn[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) Ro[k] = R[(size_t)b * 9 + k];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Rn[3 * i + j] = (Q[3 * i] * Ro[j] + Q[3 * i + 1] * Ro[3 + j]) + Q[3 * i + 2] * Ro[6 + j];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[(size_t)b * 9 + k] = Rn[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) { v[(size_t)b * 3 + k] = v6[(size_t)b * 6 + k]; w[(size_t)b * 3 + k] = v6[(size_t)b * 6 + 3 + k]; }
}

template <typename REAL>
__global__ void __launch_bounds__(256) convert_kernel(int count, const double *src, REAL *dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < count) dst[i] = (REAL)src[i];
}

// flag &= every 6x6 block is diag(a, a, a, b, b, b) exactly (the tile kernel's ISO variant)
template <typename REAL>
__global__ void __launch_bounds__(256) minv_iso_kernel(int n, const REAL *W, int *flag) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  bool iso = true;
  if (b < n) {
    const REAL *w = W + (size_t)b * 36;
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) {
        const REAL v = w[6 * r + c];
        if (r != c) iso &= (v == REAL(0));
        else iso &= (v == w[r < 3 ? 0 : 21]);
      }
  }
  if (!iso) atomicAnd(flag, 0);
}

// --------------------------------------------------------------------------
// K1-K4 assembly.  Operation order mirrors oracle/model.c.
__device__ __forceinline__ double d3(const double *a, const double *b) {
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
__device__ __forceinline__ void mv3(const double *A, const double *v, double *o) {
  o[0] = (A[0] * v[0] + A[1] * v[1]) + A[2] * v[2];
  o[1] = (A[3] * v[0] + A[4] * v[1]) + A[5] * v[2];
  o[2] = (A[6] * v[0] + A[7] * v[1]) + A[8] * v[2];
}
__device__ __forceinline__ void mm3(const double *A, const double *B, double *O) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      O[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void crossmat(const double *a, double *m) {  // utils.cc:16-24
  m[0] = 0;     m[1] = -a[2]; m[2] = a[1];
  m[3] = a[2];  m[4] = 0;     m[5] = -a[0];
  m[6] = -a[1]; m[7] = a[0];  m[8] = 0;
}
__device__ __forceinline__ void quat_to_R(double w, double x, double y, double z, double *R) {
  double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w;
  double txx = tx * x, txy = ty * x, txz = tz * x;
  double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}
// utils.cc:233-237 (Eigen FromTwoVectors(a, z).toRotationMatrix()); for the
// antiparallel case see DESIGN.md (deterministic axis instead of Eigen's SVD).
__device__ void align_to_z(const double *a, double *Rout) {
  double v0[3];
  const double na = d3(a, a);
  if (na > 0) { const double s = sqrt(na); v0[0] = a[0] / s; v0[1] = a[1] / s; v0[2] = a[2] / s; }
  else { v0[0] = a[0]; v0[1] = a[1]; v0[2] = a[2]; }
  const double v1[3] = {0.0 / 1.0, 0.0 / 1.0, 1.0 / 1.0};
  double c = d3(v1, v0);
  double qw, q[3];
  if (c < -1.0 + 1e-12) {
    if (c < -1.0) c = -1.0;
    const double ax = fabs(v0[0]), ay = fabs(v0[1]), az = fabs(v0[2]);
    double e[3] = {0, 0, 0};
    if (ax <= ay && ax <= az) e[0] = 1; else if (ay <= az) e[1] = 1; else e[2] = 1;
    double axis[3] = {v0[1] * e[2] - v0[2] * e[1], v0[2] * e[0] - v0[0] * e[2], v0[0] * e[1] - v0[1] * e[0]};
    const double n = sqrt(d3(axis, axis));
    axis[0] /= n; axis[1] /= n; axis[2] /= n;
    const double w2 = (1.0 + c) * 0.5;
    qw = sqrt(w2);
    const double sv = sqrt(1.0 - w2);
    q[0] = axis[0] * sv; q[1] = axis[1] * sv; q[2] = axis[2] * sv;
  } else {
    const double axis[3] = {v0[1] * v1[2] - v0[2] * v1[1], v0[2] * v1[0] - v0[0] * v1[2], v0[0] * v1[1] - v0[1] * v1[0]};
    const double s = sqrt((1.0 + c) * 2.0);
    const double invs = 1.0 / s;
    q[0] = axis[0] * invs; q[1] = axis[1] * invs; q[2] = axis[2] * invs;
    qw = s * 0.5;
  }
  quat_to_R(qw, q[0], q[1], q[2], Rout);
}

__device__ __forceinline__ double dot6p(const double *a, const double *b) {
  return ((((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3]) + a[4] * b[4]) + a[5] * b[5];
}

template <typename REAL>
__global__ void __launch_bounds__(256) assemble_kernel(const AssembleArgs A) {
  // J blocks leave through LDS: a lane's 18 values are 144 B apart from its neighbour's, so direct
  // stores hit 64 lines per instruction; staged, the workgroup writes its 256 x 18 block contiguously
  __shared__ REAL stage[256 * 19];   // row stride 19: conflict-free column walks
  const int i_raw = blockIdx.x * 256 + threadIdx.x;
  const bool live = i_raw < A.m;
  const int i = live ? i_raw : A.m - 1;   // the tail lanes recompute the last constraint and store nothing
  const int b0 = A.body0[i], b1 = A.body1[i];
  double d[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) d[k] = A.data[(size_t)i * 7 + k];
  double j0[18], j1[18], e[3], lo[3], hi[3];
  bool eq;
#pragma unroll
  for (int k = 0; k < 18; ++k) { j0[k] = 0.0; j1[k] = 0.0; }
  if (A.kind[i] == 0) {  // joints.cc:3-35
    double R0[9], rc0[3], cm[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R0[k] = A.R[(size_t)b0 * 9 + k];
    mv3(R0, d, rc0);
    crossmat(rc0, cm);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      j0[6 * r + r] = 1.0;
#pragma unroll
      for (int c = 0; c < 3; ++c) j0[6 * r + 3 + c] = -1.0 * cm[3 * r + c];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) e[k] = A.pos[(size_t)b0 * 3 + k] + rc0[k];
    if (b1 >= 0) {
      double R1[9], rc1[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) R1[k] = A.R[(size_t)b1 * 9 + k];
      mv3(R1, d + 3, rc1);
      crossmat(rc1, cm);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        j1[6 * r + r] = -1.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) j1[6 * r + 3 + c] = cm[3 * r + c];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) e[k] = (e[k] - A.pos[(size_t)b1 * 3 + k]) - rc1[k];
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) e[k] = e[k] - d[3 + k];
    }
    eq = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[k] = 0.0; hi[k] = 0.0; }
  } else {  // contact.cc:14-117, FrictionModel::BOX
    double Rn[9];
    align_to_z(d + 3, Rn);
    if (b0 >= 0) {
      double rel[3], cm[9], rw[9];
#pragma unroll
      for (int k = 0; k < 3; ++k) rel[k] = d[k] - A.pos[(size_t)b0 * 3 + k];
      crossmat(rel, cm);
      mm3(Rn, cm, rw);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) { j0[6 * r + c] = -Rn[3 * r + c]; j0[6 * r + 3 + c] = rw[3 * r + c]; }
    }
    if (b1 >= 0) {
      double rel[3], cm[9], rw[9];
#pragma unroll
      for (int k = 0; k < 3; ++k) rel[k] = d[k] - A.pos[(size_t)b1 * 3 + k];
      crossmat(rel, cm);
#pragma unroll
      for (int k = 0; k < 9; ++k) cm[k] = -1.0 * cm[k];
      mm3(Rn, cm, rw);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) { j1[6 * r + c] = Rn[3 * r + c]; j1[6 * r + 3 + c] = rw[3 * r + c]; }
    }
    e[0] = 0.0; e[1] = 0.0; e[2] = -d[6];
    eq = false;
    lo[0] = -1.0; lo[1] = -1.0; lo[2] = 0.0;
    hi[0] = 1.0; hi[1] = 1.0; hi[2] = INFINITY;
  }
  // rhs = -(erp/dt^2) err - J (v/dt + W f)      ensembles.cc:569-570
  double u0[6] = {0, 0, 0, 0, 0, 0}, u1[6] = {0, 0, 0, 0, 0, 0};
  if (b0 >= 0) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double vel = r < 3 ? A.v[(size_t)b0 * 3 + r] : A.w[(size_t)b0 * 3 + r - 3];
      u0[r] = vel / A.dt + A.Wf[(size_t)b0 * 6 + r];
    }
  }
  if (b1 >= 0) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double vel = r < 3 ? A.v[(size_t)b1 * 3 + r] : A.w[(size_t)b1 * 3 + r - 3];
      u1[r] = vel / A.dt + A.Wf[(size_t)b1 * 6 + r];
    }
  }
  const double kk = -A.erp / A.dt / A.dt;
  REAL *J0o = reinterpret_cast<REAL *>(A.J0), *J1o = reinterpret_cast<REAL *>(A.J1);
  REAL *loo = reinterpret_cast<REAL *>(A.lo), *hio = reinterpret_cast<REAL *>(A.hi);
  REAL *rhso = reinterpret_cast<REAL *>(A.rhs);
  {
    const int first = blockIdx.x * 256;
    const int count = (A.m - first) < 256 ? (A.m - first) : 256;
#pragma unroll
    for (int k = 0; k < 18; ++k) stage[threadIdx.x * 19 + k] = (REAL)j0[k];
    __syncthreads();
    for (int e = threadIdx.x; e < count * 18; e += 256) J0o[(size_t)first * 18 + e] = stage[(e / 18) * 19 + e % 18];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 18; ++k) stage[threadIdx.x * 19 + k] = (REAL)j1[k];
    __syncthreads();
    for (int e = threadIdx.x; e < count * 18; e += 256) J1o[(size_t)first * 18 + e] = stage[(e / 18) * 19 + e % 18];
  }
  if (!live) return;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double ju = dot6p(j0 + 6 * r, u0) + dot6p(j1 + 6 * r, u1);
    rhso[(size_t)i * 3 + r] = (REAL)(kk * e[r] - ju);
    A.err[(size_t)i * 3 + r] = e[r];
    loo[(size_t)i * 3 + r] = (REAL)lo[r];
    hio[(size_t)i * 3 + r] = (REAL)hi[r];
    A.is_eq[(size_t)i * 3 + r] = eq ? 1 : 0;
  }
}


// --------------------------------------------------------------------------
// Cross-workgroup path: islands larger than a tile.  Same ticket protocol, but
// body accumulators and tickets live in global memory and the lanes of a
// persistent grid (<= one 256-thread workgroup per CU, so every workgroup is
// resident) each own a strided slice of the constraint list.  Hand-off between
// workgroups: sc1 (write-through) payload stores -> s_waitcnt vmcnt(0) -> sc1
// ticket store; sc1 ticket poll -> sc1 payload loads, all by the same lane.
// Every wait is bounded.
// B = W J^T, D, den per global constraint; x0 = rhs; dx workspace = x0.
template <typename REAL>
__global__ void __launch_bounds__(256) global_prepare_kernel(const GlobalArgs<REAL> A) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= A.mg) return;
  const GlobalDesc d = A.cons[g];
  SolveArgs<REAL> S;
  S.Minv = A.Minv; S.J0 = A.J0; S.J1 = A.J1; S.is_eq = A.is_eq; S.lo = A.lo; S.hi = A.hi; S.rhs = A.rhs;
  S.cfm = A.cfm; S.kscale = A.kscale;
  Cons<REAL> c;
  load_cons(S, d.cidx, d.body0 >= 0, d.body1 >= 0, d.body0, d.body1, c);
#pragma unroll
  for (int k = 0; k < 18; ++k) { A.B0[(size_t)g * 18 + k] = c.B0[k]; A.B1[(size_t)g * 18 + k] = c.B1[k]; }
#pragma unroll
  for (int k = 0; k < 9; ++k) A.D[(size_t)g * 9 + k] = c.D[k];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    A.den[(size_t)g * 3 + r] = c.inv[r];
    if (!A.resume) A.x[(size_t)d.cidx * 3 + r] = c.rhs[r];
    A.dx[(size_t)g * 3 + r] = A.resume ? REAL(0) : c.rhs[r];
  }
}

// B = W J^T, D, 1/den for every constraint, indexed by list position (quad path).
template <typename REAL>
__global__ void __launch_bounds__(256) cons_prepare_kernel(const SolveArgs<REAL> A) {
  __shared__ REAL stage[256 * 19];   // the outputs leave through LDS, contiguously (see assemble_kernel)
  const int i_raw = blockIdx.x * 256 + threadIdx.x;
  const int i = i_raw < A.m ? i_raw : A.m - 1;
  const int first = blockIdx.x * 256;
  const int count = (A.m - first) < 256 ? (A.m - first) : 256;
  const int b0 = A.body0[i], b1 = A.body1[i];
  Cons<REAL> c;
  load_cons(A, i, b0 >= 0, b1 >= 0, b0, b1, c);
  auto flush = [&](const REAL *vals, int per, REAL *out) {
    for (int k = 0; k < per; ++k) stage[threadIdx.x * 19 + k] = vals[k];
    __syncthreads();
    for (int e = threadIdx.x; e < count * per; e += 256) out[(size_t)first * per + e] = stage[(e / per) * 19 + e % per];
    __syncthreads();
  };
  flush(c.B0, 18, A.wsB0);
  flush(c.B1, 18, A.wsB1);
  flush(c.D, 9, A.wsD);
  flush(c.inv, 3, A.wsInv);
}

template <typename REAL>
__device__ __forceinline__ void gload_cons(const GlobalArgs<REAL> &A, int g, const GlobalDesc &d, Cons<REAL> &c) {
  const bool has0 = d.body0 >= 0, has1 = d.body1 >= 0;
#pragma unroll
  for (int k = 0; k < 18; ++k) {
    c.J0[k] = has0 ? A.J0[(size_t)d.cidx * 18 + k] : REAL(0);
    c.J1[k] = has1 ? A.J1[(size_t)d.cidx * 18 + k] : REAL(0);
    c.B0[k] = A.B0[(size_t)g * 18 + k];
    c.B1[k] = A.B1[(size_t)g * 18 + k];
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) c.D[k] = A.D[(size_t)g * 9 + k];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    c.inv[r] = A.den[(size_t)g * 3 + r];
    c.rhs[r] = A.rhs[(size_t)d.cidx * 3 + r];
    c.lo[r] = A.lo[(size_t)d.cidx * 3 + r];
    c.hi[r] = A.hi[(size_t)d.cidx * 3 + r];
    c.eq[r] = A.is_eq[(size_t)d.cidx * 3 + r] != 0;
    clamp_bounds(c.eq[r], c.lo[r], c.hi[r]);
  }
}

template <typename REAL, int METHOD>
__global__ void __launch_bounds__(256) global_solve_kernel(const GlobalArgs<REAL> A) {
  const int L = gridDim.x * 256, lane = blockIdx.x * 256 + threadIdx.x;
  const int per_lane = A.per_lane;
  const bool rel = A.resume || A.mode == 1;  // tickets start at 0 for this launch's first phase
  int p = (A.resume && A.mode == 0) ? 1 : 0; // phase 0 = ordered accumulate of dx
  const int p_last = A.mode == 1 ? 0 : A.sweeps;
  int k = 0;
  bool ok = true;
  unsigned spins = 0;
  bool alive = p <= p_last;
  while (alive) {
    const bool backward = (METHOD == 2) && p >= 1;
    const int kk = backward ? per_lane - 1 - k : k;
    const int g = kk * L + lane;
    bool advance = false, ready = false;
    if (g >= A.mg) {
      advance = true;
    } else {
      const GlobalDesc d = A.cons[g];
      const bool has0 = d.body0 >= 0, has1 = d.body1 >= 0;
      const unsigned cnt0 = d.cnt0, cnt1 = d.cnt1, pos0 = d.pos0, pos1 = d.pos1;
      unsigned want0, want1;
      const unsigned o0 = backward ? cnt0 - 1u - pos0 : pos0, o1 = backward ? cnt1 - 1u - pos1 : pos1;
      if (p == 0) { want0 = pos0; want1 = pos1; }
      else {
        want0 = (rel ? 0u : cnt0) + (unsigned)(p - 1) * cnt0 + o0;
        want1 = (rel ? 0u : cnt1) + (unsigned)(p - 1) * cnt1 + o1;
      }
      const unsigned t0 = has0 ? gld(A.tickets + d.body0) : want0;
      const unsigned t1 = has1 ? gld(A.tickets + d.body1) : want1;
      ready = (t0 == want0) && (t1 == want1);
      if (ready) {
        // Every hand-off byte is stored sc1 (write-through, gst) and loaded sc1
        // (L1-bypassing, gld), and the storing lane drains vmcnt before it bumps
        // the ticket, so no L1 invalidate / L2 write-back is needed (guide G16,
        // "sc1 both sides"); the wavefront-scope fence only pins compiler order.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        REAL a0[6], a1[6], dx[3];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          a0[q] = has0 ? gld(A.acc + (size_t)d.body0 * 6 + q) : REAL(0);
          a1[q] = has1 ? gld(A.acc + (size_t)d.body1 * 6 + q) : REAL(0);
        }
        Cons<REAL> c;
        gload_cons(A, g, d, c);
        if (p == 0) {
#pragma unroll
          for (int r = 0; r < 3; ++r) dx[r] = A.dx[(size_t)g * 3 + r];
        } else {
          REAL x[3], res[3];
#pragma unroll
          for (int r = 0; r < 3; ++r) { x[r] = A.x[(size_t)d.cidx * 3 + r]; dx[r] = REAL(0); }
          row_residuals(c, a0, a1, x, A.cfm, res);
          update_rows<REAL, METHOD>(c, res, x, dx);
#pragma unroll
          for (int r = 0; r < 3; ++r) A.x[(size_t)d.cidx * 3 + r] = x[r];
        }
        if (has0) {
          acc_add(a0, c.B0, dx);
#pragma unroll
          for (int q = 0; q < 6; ++q) gst(A.acc + (size_t)d.body0 * 6 + q, a0[q]);
        }
        if (has1) {
          acc_add(a1, c.B1, dx);
#pragma unroll
          for (int q = 0; q < 6; ++q) gst(A.acc + (size_t)d.body1 * 6 + q, a1[q]);
        }
        if (A.hist_x && p >= 1) {   // snapshots for the per-sweep stopping test (kernels.h)
          const size_t sw = (size_t)(p - 1);
#pragma unroll
          for (int r = 0; r < 3; ++r) A.hist_x[(sw * A.m + d.cidx) * 3 + r] = A.x[(size_t)d.cidx * 3 + r];
          if (has0 && o0 == cnt0 - 1u) {   // this was the body's last update of the sweep
#pragma unroll
            for (int q = 0; q < 6; ++q) A.hist_acc[(sw * A.n_bodies + d.body0) * 6 + q] = a0[q];
          }
          if (has1 && o1 == cnt1 - 1u) {
#pragma unroll
            for (int q = 0; q < 6; ++q) A.hist_acc[(sw * A.n_bodies + d.body1) * 6 + q] = a1[q];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (has0) gst(A.tickets + d.body0, want0 + 1u);
        if (has1) gst(A.tickets + d.body1, want1 + 1u);
        advance = true;
        spins = 0;
      } else if (++spins > A.spin_limit) {
        ok = false;
        alive = false;
      }
    }
    if (advance) {
      if (++k == per_lane) { k = 0; ++p; }
      alive = alive && p <= p_last;
    }
    if (!__any(ready)) __builtin_amdgcn_s_sleep(2);
  }
  if (!ok) atomicOr(A.error_flag, 1);
}

// Jacobi compute phase for the cross-workgroup path: x_new and dx from the
// accumulators of the previous sweep (no ordering needed: read-only on acc).
template <typename REAL>
__global__ void __launch_bounds__(256) global_jacobi_kernel(const GlobalArgs<REAL> A) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= A.mg) return;
  const GlobalDesc d = A.cons[g];
  Cons<REAL> c;
  gload_cons(A, g, d, c);
  REAL a0[6], a1[6], x[3], res[3], dx[3];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    a0[q] = d.body0 >= 0 ? A.acc[(size_t)d.body0 * 6 + q] : REAL(0);
    a1[q] = d.body1 >= 0 ? A.acc[(size_t)d.body1 * 6 + q] : REAL(0);
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) x[r] = A.x[(size_t)d.cidx * 3 + r];
  row_residuals(c, a0, a1, x, A.cfm, res);
  update_rows<REAL, 0>(c, res, x, dx);
#pragma unroll
  for (int r = 0; r < 3; ++r) { A.x[(size_t)d.cidx * 3 + r] = x[r]; A.dx[(size_t)g * 3 + r] = dx[r]; }
}

// w = A x - rhs for the cross-workgroup constraints, after the last sweep.
template <typename REAL>
__global__ void __launch_bounds__(256) global_wres_kernel(const GlobalArgs<REAL> A) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= A.mg) return;
  const GlobalDesc d = A.cons[g];
  REAL a0[6], a1[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    a0[q] = d.body0 >= 0 ? A.acc[(size_t)d.body0 * 6 + q] : REAL(0);
    a1[q] = d.body1 >= 0 ? A.acc[(size_t)d.body1 * 6 + q] : REAL(0);
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    REAL j0[6], j1[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      j0[q] = d.body0 >= 0 ? A.J0[(size_t)d.cidx * 18 + 6 * r + q] : REAL(0);
      j1[q] = d.body1 >= 0 ? A.J1[(size_t)d.cidx * 18 + 6 * r + q] : REAL(0);
    }
    A.wres[(size_t)d.cidx * 3 + r] = tfma(A.cfm, A.x[(size_t)d.cidx * 3 + r], row_dot(j0, a0, j1, a1)) - A.rhs[(size_t)d.cidx * 3 + r];
  }
}

}  // namespace

// --------------------------------------------------------------------------
// Dense J M^-1 J^T + cfm I (ensembles.cc:510, 513-521) from the block-sparse form: one lane per
// pair of constraints writes the 3x3 block (zero when the two share no body).  O(m^2) like the
// reference's dense product, for the sizes its dense solver is meant for (Chain / Cairn).
__global__ void __launch_bounds__(256) dense_system_kernel(int m, const int32_t *body0, const int32_t *body1,
                                                           const double *J0, const double *J1, const double *Minv,
                                                           double cfm, double *A) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)m * m) return;
  const int i = (int)(idx / m), j = (int)(idx % m);
  double blk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int si = 0; si < 2; ++si) {
    const int bi = si ? body1[i] : body0[i];
    if (bi < 0) continue;
    for (int sj = 0; sj < 2; ++sj) {
      const int bj = sj ? body1[j] : body0[j];
      if (bj != bi) continue;
      const double *Ji = (si ? J1 : J0) + (size_t)i * 18, *Jj = (sj ? J1 : J0) + (size_t)j * 18, *W = Minv + (size_t)bi * 36;
      double t[18];   // W Jj^T, 6x3
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          double v = W[6 * k] * Jj[6 * q];
#pragma unroll
          for (int l = 1; l < 6; ++l) v = __builtin_fma(W[6 * k + l], Jj[6 * q + l], v);
          t[3 * k + q] = v;
        }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          double v = blk[3 * r + q];
#pragma unroll
          for (int k = 0; k < 6; ++k) v = __builtin_fma(Ji[6 * r + k], t[3 * k + q], v);
          blk[3 * r + q] = v;
        }
    }
  }
  if (i == j) { blk[0] += cfm; blk[4] += cfm; blk[8] += cfm; }
  const size_t N = (size_t)3 * m;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 3; ++q) A[((size_t)3 * i + r) * N + 3 * j + q] = blk[3 * r + q];
}

// ---- launchers ------------------------------------------------------------
void launch_dense_system(int m, const int32_t *body0, const int32_t *body1, const double *J0, const double *J1,
                         const double *Minv, double cfm, double *A, hipStream_t s) {
  if (m <= 0) return;
  const size_t pairs = (size_t)m * m;
  hipLaunchKernelGGL(dense_system_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, m, body0, body1, J0, J1, Minv, cfm, A);
}

template <typename REAL>
void launch_tile_solve(const SolveArgs<REAL> &a, int method, int n_tiles, int block, hipStream_t s) {
  if (n_tiles <= 0) return;
  const size_t lds = (size_t)a.max_slots * (6 * sizeof(REAL) + sizeof(unsigned));
  const dim3 g(n_tiles), b(block);
#define EGS_LAUNCH_T(BLK, ISO)                                                                        \
  switch (method) {                                                                                   \
    case 0: hipLaunchKernelGGL((tile_solve_kernel<REAL, BLK, 0, ISO>), g, b, lds, s, a); break;       \
    case 1: hipLaunchKernelGGL((tile_solve_kernel<REAL, BLK, 1, ISO>), g, b, lds, s, a); break;       \
    default: hipLaunchKernelGGL((tile_solve_kernel<REAL, BLK, 2, ISO>), g, b, lds, s, a); break;      \
  }
#define EGS_LAUNCH(BLK) EGS_LAUNCH_T(BLK, false)
  if (block == 256 && a.iso) { EGS_LAUNCH_T(256, true) }
  else if (block == 256) { EGS_LAUNCH(256) }
  else if (block == 128) { EGS_LAUNCH(128) }
  else if (block == 64) { EGS_LAUNCH(64) }
  else { EGS_LAUNCH(512) }
#undef EGS_LAUNCH
#undef EGS_LAUNCH_T
}

template <typename REAL>
void launch_assemble(const AssembleArgs &a, hipStream_t s) {
  if (a.m <= 0) return;
  hipLaunchKernelGGL((assemble_kernel<REAL>), dim3((a.m + 255) / 256), dim3(256), 0, s, a);
}

template <typename REAL>
void launch_residual_partials(int rows, const REAL *wres, const REAL *x, const REAL *lo, const REAL *hi,
                              const uint8_t *is_eq, double *out, int blocks, hipStream_t s) {
  hipLaunchKernelGGL((residual_partials_kernel<REAL>), dim3(blocks), dim3(256), 0, s, rows, wres, x, lo, hi, is_eq, out);
}

template <typename REAL>
void launch_hist_residual(const SolveArgs<REAL> &a, int sweeps, int blocks, double *out, int write_sweep, hipStream_t s) {
  if (sweeps <= 0 || a.m <= 0) return;
  constexpr int G = 8;   // recorded sweeps that share one pass over the J blocks
  hipLaunchKernelGGL((hist_residual_kernel<REAL, G>), dim3(blocks, (sweeps + G - 1) / G), dim3(256), 0, s, a, out, write_sweep, sweeps);
}

template <typename REAL>
void launch_velocity(int n, const double *v, const double *w, const double *Wf, const REAL *acc, double dt, double *v6,
                     hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL((velocity_kernel<REAL>), dim3((6 * n + 255) / 256), dim3(256), 0, s, n, v, w, Wf, acc, dt, v6);
}

void launch_mass_times_force(int n, const double *Minv, const double *f_ext, double *Wf, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(mass_times_force_kernel, dim3((6 * n + 255) / 256), dim3(256), 0, s, n, Minv, f_ext, Wf);
}

void launch_advance(int n, double *pos, double *R, double *v, double *w, const double *v6, double dt, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(advance_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, pos, R, v, w, v6, dt);
}

template <typename REAL>
void launch_convert_minv(int count, const double *src, REAL *dst, hipStream_t s) {
  if (count <= 0) return;
  hipLaunchKernelGGL((convert_kernel<REAL>), dim3((count + 255) / 256), dim3(256), 0, s, count, src, dst);
}

template <typename REAL>
void launch_minv_iso(int n, const REAL *W, int *flag, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL((minv_iso_kernel<REAL>), dim3((n + 255) / 256), dim3(256), 0, s, n, W, flag);
}


template <typename REAL>
int occupancy_global_solve() {
  int a = 0, b = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, global_solve_kernel<REAL, 1>, 256, 0) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, global_solve_kernel<REAL, 2>, 256, 0) != hipSuccess) return 0;
  return a < b ? a : b;
}
template int occupancy_global_solve<double>();
template int occupancy_global_solve<float>();

template <typename REAL>
void launch_global_solve(const GlobalArgs<REAL> &a0, int max_blocks, hipStream_t s) {
  if (a0.mg <= 0) return;
  GlobalArgs<REAL> a = a0;
  const int flat_blocks = (a.mg + 255) / 256;
  // the persistent grid's workgroups wait on each other: never more than are resident together
  // (max_blocks = occupancy x CUs, capped at one per CU; the lanes take more constraints each instead)
  const int cap = max_blocks > 0 ? max_blocks : 1;
  const int grid = flat_blocks < cap ? flat_blocks : cap;
  a.per_lane = (a.mg + grid * 256 - 1) / (grid * 256);
  const size_t tick_bytes = sizeof(uint32_t) * (size_t)(a.n_bodies > 0 ? a.n_bodies : 1);
  hipLaunchKernelGGL((global_prepare_kernel<REAL>), dim3(flat_blocks), dim3(256), 0, s, a);
  if (a.method == 0) {
    if (!a.resume) {  // accumulators from x0
      a.mode = 1;
      (void)hipMemsetAsync(a.tickets, 0, tick_bytes, s);
      hipLaunchKernelGGL((global_solve_kernel<REAL, 1>), dim3(grid), dim3(256), 0, s, a);
    }
    for (int it = 0; it < a.sweeps; ++it) {
      hipLaunchKernelGGL((global_jacobi_kernel<REAL>), dim3(flat_blocks), dim3(256), 0, s, a);
      a.mode = 1;
      (void)hipMemsetAsync(a.tickets, 0, tick_bytes, s);
      hipLaunchKernelGGL((global_solve_kernel<REAL, 1>), dim3(grid), dim3(256), 0, s, a);
    }
  } else {
    a.mode = 0;
    (void)hipMemsetAsync(a.tickets, 0, tick_bytes, s);
    if (a.method == 1) hipLaunchKernelGGL((global_solve_kernel<REAL, 1>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((global_solve_kernel<REAL, 2>), dim3(grid), dim3(256), 0, s, a);
  }
  hipLaunchKernelGGL((global_wres_kernel<REAL>), dim3(flat_blocks), dim3(256), 0, s, a);
}

template <typename REAL>
void launch_global_wres(const GlobalArgs<REAL> &a, hipStream_t s) {
  if (a.mg <= 0) return;
  hipLaunchKernelGGL((global_wres_kernel<REAL>), dim3((a.mg + 255) / 256), dim3(256), 0, s, a);
}

template <typename REAL>
void launch_cons_prepare(const SolveArgs<REAL> &a, hipStream_t s) {
  if (a.m <= 0) return;
  hipLaunchKernelGGL((cons_prepare_kernel<REAL>), dim3((a.m + 255) / 256), dim3(256), 0, s, a);
}

#define EGS_INSTANTIATE(REAL)                                                                                \
  template void launch_tile_solve<REAL>(const SolveArgs<REAL> &, int, int, int, hipStream_t);                \
  template void launch_global_solve<REAL>(const GlobalArgs<REAL> &, int, hipStream_t);                            \
  template void launch_cons_prepare<REAL>(const SolveArgs<REAL> &, hipStream_t);                             \
  template void launch_global_wres<REAL>(const GlobalArgs<REAL> &, hipStream_t);                            \
  template void launch_assemble<REAL>(const AssembleArgs &, hipStream_t);                                    \
  template void launch_residual_partials<REAL>(int, const REAL *, const REAL *, const REAL *, const REAL *,  \
                                               const uint8_t *, double *, int, hipStream_t);                 \
  template void launch_velocity<REAL>(int, const double *, const double *, const double *, const REAL *, double,   \
                                      double *, hipStream_t);                          \
  template void launch_convert_minv<REAL>(int, const double *, REAL *, hipStream_t);                         \
  template void launch_minv_iso<REAL>(int, const REAL *, int *, hipStream_t);                               \
  template void launch_hist_residual<REAL>(const SolveArgs<REAL> &, int, int, double *, int, hipStream_t);
EGS_INSTANTIATE(double)
EGS_INSTANTIATE(float)

}  // namespace egs
