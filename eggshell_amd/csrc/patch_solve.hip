// patch_solve.hip -- projected Gauss-Seidel / backward SOR for islands larger
// than a workgroup (connected piles), cut into body patches by plan.cpp.
//
// One workgroup per patch tile, one constraint per lane, constraint blocks in
// VGPRs exactly as in tile_solve_kernel.  Every body the patch touches has an LDS
// slot (hand-off ~ an LDS round trip).  A body shared with other patches travels
// with the sweep: its accumulator and ticket stay in the LDS of the patch that
// updated it last and cross global memory only where the list-order neighbour on
// that body sits in another patch (kPrevRemote / kNextRemote, plan.h): sc1
// write-through stores -> s_waitcnt vmcnt(0) -> sc1 ticket store; sc1 ticket poll
// -> sc1 loads (same lane on both ends, guide G16 "sc1 both sides").  Per-body list order is enforced by the same ticket protocol, so the
// result is the sequential list-order sweep, bit for bit.  All patch tiles of a
// launch must be co-resident (the host caps the grid); every wait is bounded.
#include "kernels.h"
#include "solve_device.h"

namespace egs {

namespace {

// HIST = true: records the per-sweep snapshots of SolveArgs::hist_x / hist_acc (tolerance-
// terminated solves, kernels.h) -- a variant of its own so that the plain kernel keeps its registers.
template <typename REAL, int METHOD, bool HIST>
__global__ void __launch_bounds__(256) patch_solve_kernel(const SolveArgs<REAL> A, uint32_t *g_tick) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  REAL *s_acc = reinterpret_cast<REAL *>(smem);
  unsigned *s_tick = reinterpret_cast<unsigned *>(smem + (size_t)A.max_slots * 6 * sizeof(REAL));
  const int tile = blockIdx.x, tid = threadIdx.x;
  const int nslots = A.tile_nslots[tile];
  const int32_t *slot_body = A.slot_body + A.tile_slot_off[tile];
  for (int s = tid; s < nslots; s += 256) {
    const int body = slot_body[s];
#pragma unroll
    for (int k = 0; k < 6; ++k) s_acc[s * 6 + k] = (A.resume && body >= 0) ? A.acc[(size_t)body * 6 + k] : REAL(0);
    s_tick[s] = 0u;
  }
  const LaneDesc d = A.lanes[(size_t)tile * 256 + tid];
  const bool active = d.cidx >= 0;
  const bool has0 = active && d.slot0 != 0, has1 = active && d.slot1 != 0;
  const int slot0 = d.slot0 & kSlotMask, slot1 = d.slot1 & kSlotMask;
  const bool sh0 = has0 && slot_body[slot0] < -1, sh1 = has1 && slot_body[slot1] < -1;   // bodies other patches touch too
  const bool prev0 = has0 && (d.slot0 & kPrevRemote) != 0, next0 = has0 && (d.slot0 & kNextRemote) != 0;
  const bool prev1 = has1 && (d.slot1 & kPrevRemote) != 0, next1 = has1 && (d.slot1 & kNextRemote) != 0;
  const int gb0 = active ? A.body0[d.cidx] : -1, gb1 = active ? A.body1[d.cidx] : -1;
  const unsigned cnt0 = d.cnt0, cnt1 = d.cnt1, pos0 = d.pos0, pos1 = d.pos1;
  REAL *ga0 = A.acc + (size_t)(gb0 >= 0 ? gb0 : 0) * 6, *ga1 = A.acc + (size_t)(gb1 >= 0 ? gb1 : 0) * 6;
  uint32_t *gt0 = g_tick + (gb0 >= 0 ? gb0 : 0), *gt1 = g_tick + (gb1 >= 0 ? gb1 : 0);

  Cons<REAL> c;
  REAL x[3] = {REAL(0), REAL(0), REAL(0)};
  if (active) {
    load_cons(A, d.cidx, has0, has1, gb0, gb1, c);
#pragma unroll
    for (int r = 0; r < 3; ++r) x[r] = A.resume ? A.x[(size_t)d.cidx * 3 + r] : c.rhs[r];
  }
  __syncthreads();

  const unsigned tk0 = lds_addr(s_tick + slot0), tk1 = lds_addr(s_tick + slot1);
  const unsigned ac0 = lds_addr(s_acc + slot0 * 6), ac1 = lds_addr(s_acc + slot1 * 6);
  const unsigned base0 = A.resume ? 0u : cnt0, base1 = A.resume ? 0u : cnt1;
  const unsigned ord0 = (METHOD == 2) ? cnt0 - 1u - pos0 : pos0, ord1 = (METHOD == 2) ? cnt1 - 1u - pos1 : pos1;
  // phase 0: accumulators from x0 = rhs in list order (skipped when resuming);
  // phases 1..sweeps: the projected sweeps.
  int phase = A.resume ? 1 : 0;
  unsigned want0 = A.resume ? ord0 : pos0, want1 = A.resume ? ord1 : pos1;
  bool ok = true;
  unsigned spins = 0;
  bool alive = active && phase <= A.sweeps;
  while (alive) {
    unsigned t0, t1;
    REAL a0[6], a1[6];
    // The accumulator comes from global memory when the predecessor on the body (in the order of
    // this phase: phase 0 and the forward sweep run the list, the backward sweep runs it from its
    // end) sits in another patch, and goes there when the successor does.  A shared body's first
    // update of a resumed launch reads global memory, its last update of the launch writes it.
    const bool fwd = METHOD == 1 || phase == 0;
    const unsigned o0 = phase == 0 ? pos0 : ord0, o1 = phase == 0 ? pos1 : ord1;
    const bool acq0 = (fwd ? prev0 : next0) || (sh0 && A.resume && phase == 1 && o0 == 0u);
    const bool acq1 = (fwd ? prev1 : next1) || (sh1 && A.resume && phase == 1 && o1 == 0u);
    const bool rel0 = (fwd ? next0 : prev0) || (sh0 && phase == A.sweeps && o0 == cnt0 - 1u);
    const bool rel1 = (fwd ? next1 : prev1) || (sh1 && phase == A.sweeps && o1 == cnt1 - 1u);
    unsigned g0 = want0, g1 = want1;
    if (acq0) g0 = gld(gt0);
    if (acq1) g1 = gld(gt1);
    poll_ticks(tk0, tk1, t0, t1);   // tickets first, accumulators only when it is this lane's turn (see kernels.hip)
    if (acq0) t0 = g0;
    if (acq1) t1 = g1;
    const bool ready = (!has0 || t0 == want0) && (!has1 || t1 == want1);
    if (ready) {
      load12(ac0, ac1, a0, a1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (acq0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) a0[q] = gld(ga0 + q);
      }
      if (acq1) {
#pragma unroll
        for (int q = 0; q < 6; ++q) a1[q] = gld(ga1 + q);
      }
      REAL dx[3] = {REAL(0), REAL(0), REAL(0)};
      if (phase == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) dx[r] = x[r];
      } else {
        REAL res[3];
        row_residuals(c, a0, a1, x, A.cfm, res);
        update_rows<REAL, METHOD>(c, res, x, dx);
      }
      if (has0) {
        acc_add(a0, c.B0, dx);
        if (rel0) {
#pragma unroll
          for (int q = 0; q < 6; ++q) gst(ga0 + q, a0[q]);
        } else {
          store6(ac0, a0);
        }
      }
      if (has1) {
        acc_add(a1, c.B1, dx);
        if (rel1) {
#pragma unroll
          for (int q = 0; q < 6; ++q) gst(ga1 + q, a1[q]);
        } else {
          store6(ac1, a1);
        }
      }
      if (HIST && phase >= 1) {   // snapshots for the per-sweep stopping test (kernels.h)
        const size_t sw = (size_t)(phase - 1);
        REAL *hx = A.hist_x + (sw * A.m + d.cidx) * 3;
        hx[0] = x[0]; hx[1] = x[1]; hx[2] = x[2];
        if (has0 && ord0 == cnt0 - 1u) {   // this was the body's last update of the sweep
          REAL *ha = A.hist_acc + (sw * A.n_bodies + gb0) * 6;
#pragma unroll
          for (int q = 0; q < 6; ++q) ha[q] = a0[q];
        }
        if (has1 && ord1 == cnt1 - 1u) {
          REAL *ha = A.hist_acc + (sw * A.n_bodies + gb1) * 6;
#pragma unroll
          for (int q = 0; q < 6; ++q) ha[q] = a1[q];
        }
      }
      if (rel0 || rel1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (has0) { if (rel0) gst(gt0, want0 + 1u); else store_tick(tk0, want0 + 1u); }
      if (has1) { if (rel1) gst(gt1, want1 + 1u); else store_tick(tk1, want1 + 1u); }
      asm volatile("s_wakeup");   // wavefronts of this workgroup that sleep on an LDS ticket
      ++phase;
      want0 = base0 + (unsigned)(phase - 1) * cnt0 + ord0;
      want1 = base1 + (unsigned)(phase - 1) * cnt1 + ord1;
      spins = 0;
      alive = phase <= A.sweeps;
    } else if (++spins > A.spin_limit) {
      ok = false;
      alive = false;
    }
    if (!__any(ready)) {
      // a lane that waits on a body shared with another workgroup has to keep looking at global
      // memory; a wavefront whose lanes all wait on LDS tickets is woken by s_wakeup
      if (__any(alive && (acq0 || acq1))) __builtin_amdgcn_s_sleep(1);
      else __builtin_amdgcn_s_sleep(32);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (!ok) atomicOr(A.error_flag, 1);
  if (active) {
#pragma unroll
    for (int r = 0; r < 3; ++r) A.x[(size_t)d.cidx * 3 + r] = x[r];
  }
  // private bodies: final accumulators to global (w = A x - rhs is computed by a
  // follow-up kernel, once every patch has finished)
  for (int s = tid + 1; s < nslots; s += 256) {
    const int body = slot_body[s];
    if (body < 0) continue;   // a shared body: its last update of the launch went to global memory
#pragma unroll
    for (int k = 0; k < 6; ++k) A.acc[(size_t)body * 6 + k] = s_acc[s * 6 + k];
  }
}

}  // namespace

template <typename REAL>
void launch_patch_solve(const SolveArgs<REAL> &a, int method, int n_tiles, uint32_t *tickets, hipStream_t s) {
  if (n_tiles <= 0) return;
  const size_t lds = (size_t)a.max_slots * (6 * sizeof(REAL) + sizeof(unsigned));
  const bool hist = a.hist_x != nullptr;
  if (method == 1 && hist) hipLaunchKernelGGL((patch_solve_kernel<REAL, 1, true>), dim3(n_tiles), dim3(256), lds, s, a, tickets);
  else if (method == 1) hipLaunchKernelGGL((patch_solve_kernel<REAL, 1, false>), dim3(n_tiles), dim3(256), lds, s, a, tickets);
  else if (hist) hipLaunchKernelGGL((patch_solve_kernel<REAL, 2, true>), dim3(n_tiles), dim3(256), lds, s, a, tickets);
  else hipLaunchKernelGGL((patch_solve_kernel<REAL, 2, false>), dim3(n_tiles), dim3(256), lds, s, a, tickets);
}

// Workgroups of this kernel one CU keeps resident (the smallest over its instantiations):
// patches wait on each other, so a launch must not have more patches than that times the CUs.
template <typename REAL>
int occupancy_patch_solve(size_t lds) {
  int best = 1 << 30, nb = 0;
#define EGS_OCC(M, H)                                                                                              \
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, patch_solve_kernel<REAL, M, H>, 256, lds) != hipSuccess) return 0; \
  best = nb < best ? nb : best;
  EGS_OCC(1, true) EGS_OCC(1, false) EGS_OCC(2, true) EGS_OCC(2, false)
#undef EGS_OCC
  return best;
}
template int occupancy_patch_solve<double>(size_t);
template int occupancy_patch_solve<float>(size_t);

template void launch_patch_solve<double>(const SolveArgs<double> &, int, int, uint32_t *, hipStream_t);
template void launch_patch_solve<float>(const SolveArgs<float> &, int, int, uint32_t *, hipStream_t);

}  // namespace egs
