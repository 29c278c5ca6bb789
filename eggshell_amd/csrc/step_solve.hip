// step_solve.hip -- projected Gauss-Seidel / backward SOR on a STATIC, time-stepped schedule.
//
// tile_solve_kernel finds the list order at run time: every lane polls its two bodies' tickets
// until it is its turn.  For a tile whose islands are regular (a pile of columns) that order is a
// fixed timetable: with level(c) = depth of constraint c in the list-order dependency DAG of one
// sweep and P = the largest level span of a body in the tile (plan.h), constraint c may run its
// update of sweep s at time
//     T(c, s) = level(c) + P * s                      (forward sweep; s = 0 is the x0 = rhs accumulation)
// because per body those times increase in list order and the first update of sweep s + 1 comes
// after the last one of sweep s.  The kernel walks the time steps with ONE workgroup barrier per
// step: the lanes due at that step run their update (accumulators in LDS, constants in VGPRs, the
// device functions of tile_solve_kernel: same arithmetic, same bits), everyone else goes straight to
// the barrier.  No tickets, no polling, no sleeping, no bounded spins -- and all lanes that are due
// together run in ONE pass (phase-major lane order puts them in the same wavefront), which the
// ticket kernel only approximates (23 of 32 lanes per pass).
// Backward SOR: the accumulation phase runs the list forward (steps 0 .. depth - 1), then the
// sweeps run it backward at depth + (depth - 1 - level) + P * (s - 1).
//
// GROUP > 1: one workgroup walks GROUP tiles on the same clock (the CU holds that many tiles anyway:
// registers).  Tiles that step independently collide on the SIMDs -- a pass is ~75 % instruction
// issue, two passes on one SIMD take 1.7 x as long (profiles/r02/microbench.json: update_iso_w8) --
// and in a regular tile the due wavefront is the same in every tile.  Sharing the barrier and
// rotating the lane -> wavefront assignment by the tile's number inside the group puts the GROUP
// passes of a time step on GROUP different SIMDs.
#include "kernels.h"
#include "solve_device.h"

namespace egs {

namespace {

template <int METHOD>
__device__ __forceinline__ int timetable_end(int depth, int P, int sweeps, int resume) {
  if (METHOD == 2) return (resume ? 0 : depth) + (sweeps >= 1 ? depth + P * (sweeps - 1) : 0);
  const int n_phases = sweeps + (resume ? 0 : 1);   // updates per lane: the accumulation shares the forward timetable
  return n_phases >= 1 ? depth + P * (n_phases - 1) : 0;
}

template <typename REAL, int BLOCK, int METHOD, bool ISO, int GROUP, bool HIST>
__global__ void __launch_bounds__(BLOCK * GROUP, (ISO && GROUP == 1) ? (sizeof(REAL) == 4 ? 4 : 3) : 1) step_solve_kernel(const SolveArgs<REAL> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WAVES = BLOCK / 64;
  const int sub = GROUP > 1 ? (int)threadIdx.x / BLOCK : 0;                 // wavefront-uniform
  const int phys = GROUP > 1 ? (int)threadIdx.x % BLOCK : (int)threadIdx.x;
  // lane of the tile this thread plays: wavefront j of sub-tile k plays the tile's wavefront (j + k) mod WAVES
  const int tid = (GROUP > 1 && WAVES > 1) ? (((phys >> 6) + sub) % WAVES) * 64 + (phys & 63) : phys;
  const int tile = blockIdx.x * GROUP + sub;
  const bool valid = GROUP == 1 || tile < A.n_tiles;
  REAL *s_acc = reinterpret_cast<REAL *>(smem) + (size_t)sub * A.max_slots * 6;

  const int nslots = valid ? A.tile_nslots[tile] : 0;
  const int32_t *slot_body = A.slot_body + (valid ? A.tile_slot_off[tile] : 0);
  for (int s = tid; s < nslots; s += BLOCK) {
    const int body = slot_body[s];
#pragma unroll
    for (int k = 0; k < 6; ++k) s_acc[s * 6 + k] = (A.resume && body >= 0) ? A.acc[(size_t)body * 6 + k] : REAL(0);
  }
  LaneDesc d;
  d.cidx = -1;
  if (valid) d = A.lanes[(size_t)tile * BLOCK + tid];
  const bool active = d.cidx >= 0;
  const bool has0 = active && d.slot0 != 0, has1 = active && d.slot1 != 0;
  const int slot0 = active ? d.slot0 : 0, slot1 = active ? d.slot1 : 0;
  const int level = valid ? A.lane_level[(size_t)tile * BLOCK + tid] : 0;
  const int P = valid ? A.tile_period[tile] : 1, depth = valid ? A.tile_depth[tile] : 1;

  Cons<REAL> c;
  REAL x[3] = {REAL(0), REAL(0), REAL(0)};
  if (active) {
    load_cons<REAL, ISO>(A, d.cidx, has0, has1, has0 ? slot_body[slot0] : 0, has1 ? slot_body[slot1] : 0, c);
#pragma unroll
    for (int r = 0; r < 3; ++r) x[r] = A.resume ? A.x[(size_t)d.cidx * 3 + r] : c.rhs[r];
  }
  const unsigned ac0 = lds_addr(s_acc + slot0 * 6), ac1 = lds_addr(s_acc + slot1 * 6);
  // snapshots for the per-sweep stopping test (kernels.h): is this lane the last update of its body in a sweep?
  const bool hist = HIST && A.hist_x != nullptr;   // isotropic variant: a separate instantiation, the plain one keeps its registers
  const bool last0 = (METHOD == 2) ? d.pos0 == 0 : d.pos0 + 1 == d.cnt0, last1 = (METHOD == 2) ? d.pos1 == 0 : d.pos1 + 1 == d.cnt1;

  // the clock runs until the longest timetable of the group has ended
  int t_end = 0;
#pragma unroll
  for (int k = 0; k < GROUP; ++k) {
    const int tk = blockIdx.x * GROUP + k;
    if (GROUP == 1 || tk < A.n_tiles) t_end = max(t_end, timetable_end<METHOD>(A.tile_depth[tk], A.tile_period[tk], A.sweeps, A.resume));
  }
  __syncthreads();

  // the timetable: `due` = the step of this lane's next update, `sweep` = which sweep that is (0 = accumulation)
  int sweep = A.resume ? 1 : 0;
  const int t0 = (METHOD == 2 && !A.resume) ? depth : 0;       // backward sweeps start after the forward accumulation
  int due = (METHOD == 2 && A.resume) ? depth - 1 - level : level;
  if (!active || sweep > A.sweeps) due = 0x7fffffff;
  for (int t = 0; t < t_end; ++t) {
    if (due == t) {
      REAL a0[6], a1[6];
      load12(ac0, ac1, a0, a1);
      REAL dx[3] = {REAL(0), REAL(0), REAL(0)};
      if (sweep == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) dx[r] = x[r];
      } else {
        REAL res[3];
        row_residuals(c, a0, a1, x, A.cfm, res);
        update_rows<REAL, METHOD>(c, res, x, dx);
      }
      if (has0) { acc_add_side0<ISO>(a0, c, dx); store6(ac0, a0); }
      if (has1) { acc_add_side1<ISO>(a1, c, dx); store6(ac1, a1); }
      if (hist && sweep >= 1) {
        REAL *hx = A.hist_x + ((size_t)(sweep - 1) * A.m + d.cidx) * 3;
        hx[0] = x[0]; hx[1] = x[1]; hx[2] = x[2];
        if (has0 && last0) {
          REAL *ha = A.hist_acc + ((size_t)(sweep - 1) * A.n_bodies + slot_body[slot0]) * 6;
#pragma unroll
          for (int k = 0; k < 6; ++k) ha[k] = a0[k];
        }
        if (has1 && last1) {
          REAL *ha = A.hist_acc + ((size_t)(sweep - 1) * A.n_bodies + slot_body[slot1]) * 6;
#pragma unroll
          for (int k = 0; k < 6; ++k) ha[k] = a1[k];
        }
      }
      // next: the first backward sweep starts at t0 and runs the list from its end
      due = (METHOD == 2 && sweep == 0) ? t0 + (depth - 1 - level) : due + P;
      if (++sweep > A.sweeps) due = 0x7fffffff;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wavefront's accumulator stores have landed
    __builtin_amdgcn_s_barrier();
  }

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

int step_group_env(int dflt) {
  const char *e = std::getenv("EGS_STEP_GROUP");
  if (!e) return dflt;
  const int g = std::atoi(e);
  return g >= 1 ? g : dflt;
}

}  // namespace

template <typename REAL>
void launch_step_solve(const SolveArgs<REAL> &a, int method, int n_tiles, int block, hipStream_t s) {
  if (n_tiles <= 0) return;
  SolveArgs<REAL> b = a;
  b.n_tiles = n_tiles;
#define EGS_LAUNCH_S(BLK, ISO, GRP) EGS_LAUNCH_SH(BLK, ISO, GRP, !ISO)
#define EGS_LAUNCH_SH(BLK, ISO, GRP, HIST)                                                                     \
  {                                                                                                            \
    const size_t lds = (size_t)b.max_slots * 6 * sizeof(REAL) * GRP;                                           \
    const dim3 g((n_tiles + GRP - 1) / GRP), t(BLK * GRP);                                                     \
    auto k1 = step_solve_kernel<REAL, BLK, 1, ISO, GRP, HIST>;                                                 \
    auto k2 = step_solve_kernel<REAL, BLK, 2, ISO, GRP, HIST>;                                                 \
    if (lds > 48 * 1024) {                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(method == 1 ? k1 : k2),                            \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
    }                                                                                                          \
    if (method == 1) hipLaunchKernelGGL(k1, g, t, lds, s, b);                                                  \
    else hipLaunchKernelGGL(k2, g, t, lds, s, b);                                                              \
  }
  if (block == 256 && a.iso) {
    // the CU holds three fp64 (168 VGPRs) resp. four fp32 (128) isotropic tiles.  Walking them on one clock
    // pays for the four fp32 tiles (C4: 0.281 ms against 0.368); with three fp64 tiles the 12-wavefront
    // barrier costs more than the collisions it avoids (C3 x 24: 1.06 ms against 0.95), so fp64 keeps GROUP = 1
    const int grp = step_group_env(sizeof(REAL) == 4 ? 4 : 1);
    if (b.hist_x != nullptr) EGS_LAUNCH_SH(256, true, 1, true)      // per-sweep snapshots of the stopping loop (kernels.h)
    else if constexpr (sizeof(REAL) == 4) {
      if (grp >= 4) EGS_LAUNCH_S(256, true, 4)
      else if (grp >= 2) EGS_LAUNCH_S(256, true, 2)
      else EGS_LAUNCH_S(256, true, 1)
    } else {
      if (grp >= 3) EGS_LAUNCH_S(256, true, 3)
      else EGS_LAUNCH_S(256, true, 1)
    }
  }
  else if (block == 256) EGS_LAUNCH_S(256, false, 1)
  else if (block == 128) EGS_LAUNCH_S(128, false, 1)
  else if (block == 64) EGS_LAUNCH_S(64, false, 1)
  else EGS_LAUNCH_S(512, false, 1)
#undef EGS_LAUNCH_S
#undef EGS_LAUNCH_SH
}

template void launch_step_solve<double>(const SolveArgs<double> &, int, int, int, hipStream_t);
template void launch_step_solve<float>(const SolveArgs<float> &, int, int, int, hipStream_t);

}  // namespace egs
