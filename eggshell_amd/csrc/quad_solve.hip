// quad_solve.hip -- latency-optimised projected Gauss-Seidel / backward SOR:
// FOUR lanes per constraint.  Used when the problem under-fills the GPU (one
// ensemble, a few piles): the dependency chain of an island, not throughput,
// sets the step time, so each update is made as short as possible.
//
// Lane q of a quad owns one quarter of the 3x12 Jacobian row block:
//   q = 0: body0 linear   q = 1: body0 angular   q = 2: body1 linear   q = 3: body1 angular
// i.e. 3 accumulator components, 9 J entries and 9 entries of B = M^-1 J^T.
// A row product J_r . a is four 3-term fma chains combined by a 2-step quad
// butterfly (DPP quad_perm) as (p0 + p1) + (p2 + p3) -- exactly the order of
// the oracle's row_dot, so results stay bit-identical to the 1-lane kernel and
// to the CPU oracle.  The 3-row projected solve is done redundantly by the four
// lanes (identical inputs, identical bits); each lane then updates its own 3
// accumulator components in LDS.  Ordering between constraints is the same
// per-body ticket protocol as tile_solve_kernel (see kernels.hip).
// Tile = 64 constraints = one 256-thread workgroup, or 256 constraints = one
// 1024-thread workgroup when islands are larger than 64 constraints.
#include <stdexcept>

#include <type_traits>

#include "kernels.h"
#include "solve_device.h"

#ifndef EGS_QUAD_SLEEP
#define EGS_QUAD_SLEEP 32
#endif

namespace egs {

namespace {

// quad_perm DPP (solve_device.h: dpp<CTRL>): value of lane (l ^ 1) / (l ^ 2) inside each group of 4 lanes
constexpr int kXor1 = 0xB1;  // quad_perm [1,0,3,2]
constexpr int kXor2 = 0x4E;  // quad_perm [2,3,0,1]

template <typename T>
__device__ __forceinline__ T quad_sum(T p) {  // (p0 + p1) + (p2 + p3) on every lane of the quad
  p = p + dpp<kXor1>(p);
  p = p + dpp<kXor2>(p);
  return p;
}

// ---- ordered LDS hand-off, hand-placed ---------------------------------------
// A wavefront's DS instructions execute in issue order.  Consumer: ticket load,
// then the 3 accumulator loads, ONE wait for all four -- if the ticket matches,
// the accumulators read after it are current (nobody else may write this body
// before we bump its ticket).  Producer: accumulator stores, then the ticket
// store; no wait in between.  (volatile C++ accesses were tried first: hipcc
// lowers them to flat sc0 sc1 loads with a vmcnt(0) after each.)
__device__ __forceinline__ void poll3(unsigned tick_addr, unsigned acc_addr, unsigned &t, double (&a)[3]) {
  asm volatile(
      "ds_read_b32 %0, %4\n\t"
      "ds_read_b64 %1, %5\n\t"
      "ds_read_b64 %2, %5 offset:8\n\t"
      "ds_read_b64 %3, %5 offset:16\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(t), "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2])
      : "v"(tick_addr), "v"(acc_addr)
      : "memory");
}
__device__ __forceinline__ void poll3(unsigned tick_addr, unsigned acc_addr, unsigned &t, float (&a)[3]) {
  asm volatile(
      "ds_read_b32 %0, %4\n\t"
      "ds_read_b32 %1, %5\n\t"
      "ds_read_b32 %2, %5 offset:4\n\t"
      "ds_read_b32 %3, %5 offset:8\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(t), "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2])
      : "v"(tick_addr), "v"(acc_addr)
      : "memory");
}
__device__ __forceinline__ void store3(unsigned acc_addr, const double (&a)[3]) {
  asm volatile(
      "ds_write_b64 %0, %1\n\t"
      "ds_write_b64 %0, %2 offset:8\n\t"
      "ds_write_b64 %0, %3 offset:16"
      :: "v"(acc_addr), "v"(a[0]), "v"(a[1]), "v"(a[2])
      : "memory");
}
__device__ __forceinline__ void store3(unsigned acc_addr, const float (&a)[3]) {
  asm volatile(
      "ds_write_b32 %0, %1\n\t"
      "ds_write_b32 %0, %2 offset:4\n\t"
      "ds_write_b32 %0, %3 offset:8"
      :: "v"(acc_addr), "v"(a[0]), "v"(a[1]), "v"(a[2])
      : "memory");
}

// ---- data-tagged granules: the hand-off between body patches ----------------------------------
// A shared body's accumulator used to cross patches as payload + flag: sc1 stores -> s_waitcnt vmcnt(0) -> sc1 ticket
// store on one side, ticket poll -> payload loads on the other: two dependent round trips through the memory side on
// each side (MI355X_MICROARCH.md price list, handoff-flag: 1.3 us idle, 3.8-4.9 under load; ~4 us measured here).
// Now every component travels as ONE naturally aligned 16-byte granule {value, tag}, tag = launch epoch << 32 | the
// body's ticket AFTER the update, written by one global_store_dwordx4 sc1 and read by one global_load_dwordx4 sc1
// (observed untorn on gfx950 / ROCm 7.2): the producer neither waits for its stores nor writes a flag, the consumer's
// poll IS the payload load (handoff-1to1: 0.8-1.0 us idle).  A body's six granules share one 96-byte stretch laid out
// [component][half], so that the two lanes of a side read / write 32 contiguous bytes per instruction.  Tickets only
// grow within a launch and the epoch separates launches, so a matching tag can only be the predecessor's store.
typedef unsigned gran_u4 __attribute__((ext_vector_type(4)));
struct alignas(16) Gran { unsigned long long bits, tag; };

template <typename REAL> __device__ __forceinline__ unsigned long long gran_bits(REAL v);
template <> __device__ __forceinline__ unsigned long long gran_bits<double>(double v) { return (unsigned long long)__double_as_longlong(v); }
template <> __device__ __forceinline__ unsigned long long gran_bits<float>(float v) { return (unsigned long long)__float_as_uint(v); }
template <typename REAL> __device__ __forceinline__ REAL gran_value(unsigned lo, unsigned hi);
template <> __device__ __forceinline__ double gran_value<double>(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }
template <> __device__ __forceinline__ float gran_value<float>(unsigned lo, unsigned) { return __uint_as_float(lo); }

// this lane's three granules (stride 2: the other half's sit in between), no wait, no flag
template <typename REAL>
__device__ __forceinline__ void gran_store3(Gran *p, const REAL (&a)[3], unsigned long long tag) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const unsigned long long b = gran_bits<REAL>(a[k]);
    const gran_u4 v = {(unsigned)b, (unsigned)(b >> 32), (unsigned)tag, (unsigned)(tag >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p + 2 * k), "v"(v) : "memory");
  }
}
// one look: three loads in flight, one wait; true (and the values) iff all three carry `tag`
template <typename REAL>
__device__ __forceinline__ bool gran_poll3(const Gran *p, unsigned long long tag, REAL (&a)[3]) {
  gran_u4 g0, g1, g2;
  asm volatile(
      "global_load_dwordx4 %0, %3, off sc1\n\t"
      "global_load_dwordx4 %1, %3, off offset:32 sc1\n\t"
      "global_load_dwordx4 %2, %3, off offset:64 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(g0), "=&v"(g1), "=&v"(g2) : "v"(p) : "memory");
  const unsigned tl = (unsigned)tag, th = (unsigned)(tag >> 32);
  a[0] = gran_value<REAL>(g0.x, g0.y); a[1] = gran_value<REAL>(g1.x, g1.y); a[2] = gran_value<REAL>(g2.x, g2.y);
  return g0.z == tl && g0.w == th && g1.z == tl && g1.w == th && g2.z == tl && g2.w == th;
}

// QT = constraints per tile; the workgroup has 4 * QT threads (64 -> 256, 256 -> 1024).
// PATCH = true: the tile is a body patch of an island larger than a workgroup
// (plan.cpp::build_patches).  A body that other workgroups touch too has an LDS slot
// here like a private one, and its accumulator travels with the sweep: it stays in
// the LDS of the patch that updated it last and crosses global memory (A.acc,
// g_tick) only where the list-order neighbour on that body sits in another patch
// (kPrevRemote / kNextRemote on the slot): sc1 stores -> s_waitcnt vmcnt(0) -> sc1
// ticket store / sc1 ticket poll -> sc1 loads, as in patch_solve_kernel.  The last
// update of the launch on such a body always goes to global memory, and the first
// one of a resumed launch always comes from there.  w = A x - rhs is left to a
// follow-up kernel (the shared accumulators are final only when every patch has
// finished).
// HIST = true: records the per-sweep snapshots of SolveArgs::hist_x / hist_acc (tolerance-
// terminated solves); a separate instantiation, so the plain kernel keeps its 96 VGPRs.
// RUNS = true (patches only, Plan::patch_runs): the lanes come in chunks of four quads = one DPP row: up to four consecutive
// constraints on the same two bodies (members, then placeholders with cidx = -2 that carry the chunk's slots).  A chunk is
// ONE node of the ticket protocol: its quads wait for the first member's ticket (pos and want are the FIRST member's on
// every slot, see below), the accumulators go from member to member through the row (row_shr:4, backward row_shl:4)
// and the slot at the end publishes the last member's ticket + 1.  Same updates, same order, same bits -- three of
// four hand-offs on such a body no longer go through LDS.
template <typename REAL, int METHOD, int QT, bool PATCH, bool HIST, bool RUNS = false>
__global__ void __launch_bounds__(4 * QT) quad_solve_kernel(const SolveArgs<REAL> A, uint32_t *g_tick) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  REAL *s_acc = reinterpret_cast<REAL *>(smem);
  unsigned *s_tick = reinterpret_cast<unsigned *>(smem + (size_t)A.max_slots * 6 * sizeof(REAL));

  const int tile = blockIdx.x, tid = threadIdx.x;
  const int q = tid & 3, side = q >> 1, half = q & 1;
  const int nslots = A.tile_nslots[tile];
  const int32_t *slot_body = A.slot_body + A.tile_slot_off[tile];
  for (int s = tid; s < nslots; s += 4 * QT) {
    const int body = slot_body[s];
#pragma unroll
    for (int k = 0; k < 6; ++k) s_acc[s * 6 + k] = (A.resume && body >= 0) ? A.acc[(size_t)body * 6 + k] : REAL(0);
    s_tick[s] = 0u;
  }

  const LaneDesc d = A.lanes[(size_t)tile * QT + (tid >> 2)];
  const bool active = d.cidx >= 0;
  const bool chain = RUNS ? d.cidx != -1 : active;      // RUNS: placeholders take part in the hand-off
  const int run_pos = (tid >> 2) & 3;                   // RUNS: this quad's place in its chunk
  int run_len = 1;                                      // RUNS: members of the chunk
  if (RUNS) {
    const unsigned long long act = __ballot(active);
    run_len = __popcll((act >> (tid & 48)) & 0xffffull) >> 2;
  }
  const int raw_slot = side ? d.slot1 : d.slot0;
  const bool has = chain && raw_slot != 0;         // this lane's body is a real body
  const int slot = PATCH ? (raw_slot & kSlotMask) : raw_slot;
  // ... shared with other workgroups: where its list-order neighbours on the body live
  const bool prev_remote = PATCH && has && (raw_slot & kPrevRemote) != 0, next_remote = PATCH && has && (raw_slot & kNextRemote) != 0;
  const bool sh = PATCH && has && slot_body[slot] < -1;
  // RUNS: the chunk's FIRST member's position on every slot (the plan stores first + place)
  const unsigned cnt = side ? d.cnt1 : d.cnt0, pos = (side ? d.pos1 : d.pos0) - (RUNS ? (unsigned)run_pos : 0u);
  const unsigned nrun = RUNS ? (unsigned)run_len : 1u;
  REAL *my_acc = s_acc + slot * 6 + 3 * half;      // slot 0 (world) stays zero
  unsigned *my_tick = s_tick + slot;
  REAL *g_acc = A.acc;
  uint32_t *g_t = g_tick;
  Gran *my_gran = nullptr;
  const bool gran = PATCH && A.gran != nullptr;      // uniform
  const unsigned long long epoch_hi = (unsigned long long)A.gran_epoch << 32;
  if (sh) {   // where the accumulator crosses between patches
    const int body = -slot_body[slot] - 2;      // (a shared body's slot holds -(body + 2))
    g_acc = A.acc + (size_t)body * 6 + 3 * half;
    g_t = g_tick + body;
    my_gran = reinterpret_cast<Gran *>(A.gran) + (size_t)body * 6 + half;     // [component][half]
  }

  REAL Jh[9], Bh[9], Dl[3], inv[3], rhs[3], lo[3], hi[3], x[3];
  bool eq[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) { Jh[k] = REAL(0); Bh[k] = REAL(0); }
#pragma unroll
  for (int r = 0; r < 3; ++r) { Dl[r] = inv[r] = rhs[r] = lo[r] = hi[r] = x[r] = REAL(0); eq[r] = true; }
  if (active) {
    const size_t c = (size_t)d.cidx;
    if (has) {
      const REAL *J = (side ? A.J1 : A.J0) + c * 18 + 3 * half;
      const REAL *B = (side ? A.wsB1 : A.wsB0) + c * 18 + 9 * half;  // rows 3*half .. 3*half+2 of the 6x3
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) Jh[3 * r + k] = J[6 * r + k];
#pragma unroll
      for (int k = 0; k < 9; ++k) Bh[k] = B[k];
    }
    const REAL *D = A.wsD + c * 9;
    if (METHOD == 1) { Dl[0] = D[3]; Dl[1] = D[6]; Dl[2] = D[7]; }   // D10, D20, D21
    else { Dl[0] = D[1]; Dl[1] = D[2]; Dl[2] = D[5]; }               // D01, D02, D12
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      inv[r] = A.wsInv[c * 3 + r];
      rhs[r] = A.rhs[c * 3 + r];
      lo[r] = A.lo[c * 3 + r];
      hi[r] = A.hi[c * 3 + r];
      eq[r] = A.is_eq[c * 3 + r] != 0;
      clamp_bounds(eq[r], lo[r], hi[r]);
      x[r] = A.resume ? A.x[c * 3 + r] : rhs[r];
    }
  }
  __syncthreads();

  bool ok = true;
  const unsigned base = A.resume ? 0u : cnt;
  if (!A.resume) {
    // accumulators from x0 = rhs (sparse_iterations.cc:202), list order per body
    bool pending = has;
    unsigned spins = 0;
    // list order: the value comes from global memory if the predecessor on the body is remote
    // (pos 0: both places hold zeros) and goes there if the successor is
    // (a launch without sweeps ends here: then the body's last update goes to global memory)
    const bool acq = prev_remote, rel = next_remote || (sh && A.sweeps == 0 && pos + nrun == cnt);
    // granules: the body's first update starts from zero (nobody wrote before it); a launch without sweeps leaves the
    // last value in A.acc, where the next launch and the follow-up kernels look for it
    const bool acq_g = gran && acq && pos != 0u, end_g = gran && sh && A.sweeps == 0 && pos + nrun == cnt;
    const bool end_slot = !RUNS || run_pos == 3;      // RUNS: the slot the chunk's last value ends up in
    while (pending) {
      REAL ga[3] = {REAL(0), REAL(0), REAL(0)};
      unsigned t;
      if (acq_g) t = gran_poll3<REAL>(my_gran, epoch_hi | pos, ga) ? pos : pos + 1u;
      else if (gran && acq) t = pos;
      else t = acq ? gld(g_t) : lds_load_acquire(my_tick);
      if (t == pos) {
        REAL a[3];
        if (gran && acq) {
#pragma unroll
          for (int k = 0; k < 3; ++k) a[k] = ga[k];
        } else if (acq) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int k = 0; k < 3; ++k) a[k] = gld(g_acc + k);
        } else {
#pragma unroll
          for (int k = 0; k < 3; ++k) a[k] = my_acc[k];
        }
        // (RUNS: the lanes of a chunk's row with this q see the same ticket at the same look, so all four are here)
#pragma unroll
        for (int sub = 0; sub < (RUNS ? 4 : 1); ++sub) {
          if (RUNS && sub > 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) a[k] = dpp<0x114>(a[k]);      // row_shr:4 = the same quarter of the previous member
          }
          if (!RUNS || (run_pos == sub && active)) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              REAL u = tfma(Bh[3 * k + 0], x[0], a[k]);
              u = tfma(Bh[3 * k + 1], x[1], u);
              a[k] = tfma(Bh[3 * k + 2], x[2], u);
            }
          }
        }
        if (!end_slot) {
        } else if (gran && rel) {
          if (end_g) {
#pragma unroll
            for (int k = 0; k < 3; ++k) gst(g_acc + k, a[k]);
          } else {
            gran_store3<REAL>(my_gran, a, epoch_hi | (pos + nrun));
          }
        } else if (rel) {
#pragma unroll
          for (int k = 0; k < 3; ++k) gst(g_acc + k, a[k]);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // both halves' stores (one wavefront) have landed
          if (half == 0) gst(g_t, pos + nrun);
        } else {
#pragma unroll
          for (int k = 0; k < 3; ++k) my_acc[k] = a[k];
          if (half == 0) lds_store_release(my_tick, pos + nrun);
        }
        pending = false;
      } else if (++spins > A.spin_limit) {
        ok = false;
        pending = false;
      }
    }
    __syncthreads();
  }

  {
    const unsigned ord = (METHOD == 2) ? cnt - pos - nrun : pos;      // the node's first update in sweep order
    unsigned want = base + ord;
    int sweep = 1;
    unsigned spins = 0;
    bool alive = chain && A.sweeps >= 1;
    const bool end_slot = !RUNS || ((METHOD == 1) ? run_pos == 3 : run_pos == 0);      // where the chunk's last value ends up
    // this member's own place in the sweep (HIST: the body's last update of the sweep is snapshot)
    const unsigned ord_m = RUNS ? ((METHOD == 2) ? cnt - 1u - (pos + (unsigned)run_pos) : pos + (unsigned)run_pos) : ord;
    const unsigned tick_addr = lds_addr(my_tick), acc_addr = lds_addr(my_acc);
    // sweep order: forward = list order, backward = reversed, so the neighbours swap roles
    // A wavefront none of whose lanes touches a body that other patches share runs the loop WITHOUT the cross-patch
    // code: the general loop's sixteen exec-masked branches (acquire / release / launch-boundary cases) and the
    // vmcnt(0) the compiler puts at their merge point cost ~0.35 us per look even when no lane takes them
    // (tools/trace_patches.py: 0.7 us per LDS-local hand-off on the critical chain against 0.36 in the tile kernel).
    // The plan puts a patch's boundary constraints first (plan.cpp), so the interior ones fill such wavefronts.
    // (a lane needs that code if a list-order neighbour on its body is remote, or if it is the first / last constraint of a
    //  shared body's list: the launch-boundary cases)
    const bool acq_flag = (METHOD == 1) ? prev_remote : next_remote;
    const bool lane_remote = has && (prev_remote || next_remote || (sh && (pos == 0u || pos + nrun == cnt)));
    const bool lane_acquires = has && (acq_flag || (sh && A.resume && ord == 0u));
    // three loops: 0 = no cross-patch code at all, 1 = releases only (stores; no global load, hence no vmcnt wait that
    // would hold the wavefront until its own write-through stores are acknowledged), 2 = everything
    const int wave_level = !PATCH ? 0 : (__any(lane_acquires ? 1 : 0) ? 2 : (__any(lane_remote ? 1 : 0) ? 1 : 0));
    auto sweeps_loop = [&](auto level_tag) {
    constexpr int LEVEL = decltype(level_tag)::value;
    constexpr bool REMOTE = LEVEL >= 1, ACQ = LEVEL == 2;
    const bool acq_side = ACQ && acq_flag, rel_side = REMOTE && ((METHOD == 1) ? next_remote : prev_remote);
    while (alive) {
      unsigned t;
      REAL a[3];
      // a shared body's first update of a resumed launch reads global memory, its last update of
      // the launch writes it (the accumulator must not stay behind in some patch's LDS)
      const bool first_of_launch = ACQ && sh && A.resume && sweep == 1 && ord == 0u;
      const bool last_of_launch = REMOTE && sh && sweep == A.sweeps && ord + nrun == cnt;
      const bool acq = acq_side || first_of_launch;
      const bool rel = rel_side || last_of_launch;
      // granules: a predecessor in another patch of THIS launch is polled for; at the launch boundary the value is in A.acc
      const bool acq_g = ACQ && gran && acq_side && !first_of_launch;
      REAL ga[3] = {REAL(0), REAL(0), REAL(0)};
      unsigned gt = want;
      if (acq_g) gt = gran_poll3<REAL>(my_gran, epoch_hi | want, ga) ? want : want + 1u;
      else if (ACQ && acq && !gran) gt = gld(g_t);
      poll3(tick_addr, acc_addr, t, a);
      if (ACQ && acq) t = gt;
      int rdy = (!has || t == want) ? 1 : 0;
      rdy &= dpp_i<kXor1>(rdy);
      rdy &= dpp_i<kXor2>(rdy);
      if (rdy) {
        if (acq_g) {
#pragma unroll
          for (int k = 0; k < 3; ++k) a[k] = ga[k];
        } else if (ACQ && acq) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int k = 0; k < 3; ++k) a[k] = gld(g_acc + k);
        }
        // one update of this lane's constraint: `v` = its quarter of the body's accumulator, in and out
        auto update = [&](REAL (&v)[3]) {
          REAL res[3], dx[3] = {REAL(0), REAL(0), REAL(0)};
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            REAL p = Jh[3 * r] * v[0];
            p = tfma(Jh[3 * r + 1], v[1], p);
            p = tfma(Jh[3 * r + 2], v[2], p);
            const REAL full = tfma(A.cfm, x[r], quad_sum(p));
            res[r] = rhs[r] - full;
          }
          if (METHOD == 1) {
            REAL t0 = res[0];
            REAL xn = project(tfma(t0, inv[0], x[0]), lo[0], hi[0]);
            dx[0] = xn - x[0]; x[0] = xn;
            REAL t1 = tfma(-Dl[0], dx[0], res[1]);
            xn = project(tfma(t1, inv[1], x[1]), lo[1], hi[1]);
            dx[1] = xn - x[1]; x[1] = xn;
            REAL t2 = tfma(-Dl[1], dx[0], res[2]);
            t2 = tfma(-Dl[2], dx[1], t2);
            xn = project(tfma(t2, inv[2], x[2]), lo[2], hi[2]);
            dx[2] = xn - x[2]; x[2] = xn;
          } else {
            REAL t2 = res[2];
            REAL xn = project(tfma(t2, inv[2], x[2]), lo[2], hi[2]);
            dx[2] = xn - x[2]; x[2] = xn;
            REAL t1 = tfma(-Dl[2], dx[2], res[1]);                 // D12
            xn = project(tfma(t1, inv[1], x[1]), lo[1], hi[1]);
            dx[1] = xn - x[1]; x[1] = xn;
            REAL t0 = tfma(-Dl[1], dx[2], res[0]);                 // D02
            t0 = tfma(-Dl[0], dx[1], t0);                          // D01
            xn = project(tfma(t0, inv[0], x[0]), lo[0], hi[0]);
            dx[0] = xn - x[0]; x[0] = xn;
          }
          if (has) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              REAL u = tfma(Bh[3 * k + 0], dx[0], v[k]);
              u = tfma(Bh[3 * k + 1], dx[1], u);
              v[k] = tfma(Bh[3 * k + 2], dx[2], u);
            }
          }
        };
        REAL an[3] = {a[0], a[1], a[2]};
        REAL an_hist[3] = {REAL(0), REAL(0), REAL(0)};
        if (!RUNS) {
          update(an);
#pragma unroll
          for (int k = 0; k < 3; ++k) an_hist[k] = an[k];
        } else {
          // the chunk's updates in sweep order; the accumulators go from quad to quad, LDS (or the other patch) sees
          // the first read and the last write only
#pragma unroll
          for (int sub = 0; sub < 4; ++sub) {
            if (sub > 0) {
#pragma unroll
              for (int k = 0; k < 3; ++k) an[k] = (METHOD == 1) ? dpp<0x114>(an[k]) : dpp<0x104>(an[k]);      // row_shr:4 / row_shl:4
            }
            if (run_pos == ((METHOD == 1) ? sub : 3 - sub) && active) {
              update(an);
#pragma unroll
              for (int k = 0; k < 3; ++k) an_hist[k] = an[k];
            }
          }
        }
        if (has && end_slot) {
          if (REMOTE && gran && rel) {
            if (last_of_launch) {
#pragma unroll
              for (int k = 0; k < 3; ++k) gst(g_acc + k, an[k]);
            } else {
              gran_store3<REAL>(my_gran, an, epoch_hi | (want + nrun));
            }
          } else if (REMOTE && rel) {
#pragma unroll
            for (int k = 0; k < 3; ++k) gst(g_acc + k, an[k]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (half == 0) gst(g_t, want + nrun);
          } else {
            store3(acc_addr, an);
            if (half == 0) store_tick(tick_addr, want + nrun);
          }
        }
        if (PATCH && A.trace != nullptr && q == 0 && active) A.trace[(size_t)(sweep - 1) * A.m + d.cidx] = wall_clock64();
        if (HIST) {   // snapshots for the per-sweep stopping test (kernels.h)
          if (q == 0 && active) {
            REAL *hx = A.hist_x + ((size_t)(sweep - 1) * A.m + d.cidx) * 3;
            hx[0] = x[0]; hx[1] = x[1]; hx[2] = x[2];
          }
          if (has && active && ord_m == cnt - 1u) {   // this was the body's last update of the sweep
            const int body = side ? A.body1[d.cidx] : A.body0[d.cidx];
            REAL *ha = A.hist_acc + ((size_t)(sweep - 1) * A.n_bodies + body) * 6 + 3 * half;
            ha[0] = an_hist[0]; ha[1] = an_hist[1]; ha[2] = an_hist[2];
          }
        }
        // Waiting wavefronts sleep and are woken by the next ticket store of their workgroup
        // (see kernels.hip): -5 % on a single C3 pile, -15 % with four piles per launch; slower
        // on body patches (waits on global tickets cannot be woken), so not there.
        asm volatile("s_wakeup");
        want += cnt;
        spins = 0;
        alive = ++sweep <= A.sweeps;
      } else if (++spins > A.spin_limit) {
        ok = false;
        alive = false;
      }
      // a wavefront with a lane that waits on a global ticket has to keep looking (nothing wakes
      // it); one whose lanes all wait on LDS tickets sleeps until a ticket store of its workgroup
      // (in a patch few wavefronts are awake at a time and the next constraint in line often sits in another one: a
      //  short nap -- 128 cycles instead of 2048 -- finds its ticket sooner: walls another 4 %)
      if (!__any(rdy) && !(ACQ && __any(alive && acq))) {
        if (PATCH) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(EGS_QUAD_SLEEP);
      }
    }
    };
    if (wave_level == 2) sweeps_loop(std::integral_constant<int, 2>{});
    else if (wave_level == 1) sweeps_loop(std::integral_constant<int, 1>{});
    else sweeps_loop(std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }

  if (!ok) atomicOr(A.error_flag, 1);

  if (PATCH) {
    if (active && q == 0) {
#pragma unroll
      for (int r = 0; r < 3; ++r) A.x[(size_t)d.cidx * 3 + r] = x[r];
    }
  } else if (active) {  // lambda and w = A x - rhs with the final accumulators
    REAL a[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) a[k] = my_acc[k];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      REAL p = Jh[3 * r] * a[0];
      p = tfma(Jh[3 * r + 1], a[1], p);
      p = tfma(Jh[3 * r + 2], a[2], p);
      const REAL w = tfma(A.cfm, x[r], quad_sum(p)) - rhs[r];
      if (q == 0) {
        A.x[(size_t)d.cidx * 3 + r] = x[r];
        A.wres[(size_t)d.cidx * 3 + r] = w;
      }
    }
  }
  for (int s = tid + 1; s < nslots; s += 4 * QT) {
    const int body = slot_body[s];
    if (body < 0) continue;   // a shared body: its last update of the launch went to global memory
#pragma unroll
    for (int k = 0; k < 6; ++k) A.acc[(size_t)body * 6 + k] = s_acc[s * 6 + k];
  }
}


__device__ __forceinline__ void load3(unsigned acc_addr, double (&a)[3]) {
  asm volatile(
      "ds_read_b64 %0, %3\n\t"
      "ds_read_b64 %1, %3 offset:8\n\t"
      "ds_read_b64 %2, %3 offset:16\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]) : "v"(acc_addr) : "memory");
}
__device__ __forceinline__ void load3(unsigned acc_addr, float (&a)[3]) {
  asm volatile(
      "ds_read_b32 %0, %3\n\t"
      "ds_read_b32 %1, %3 offset:4\n\t"
      "ds_read_b32 %2, %3 offset:8\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]) : "v"(acc_addr) : "memory");
}

// The 4-lanes-per-constraint sweep on the STATIC TIMETABLE of step_solve.hip (plan.h: level, period,
// depth): constraint c runs its update of sweep s at time step level(c) + P * s, one workgroup barrier
// per step, no tickets and no polling.  Same lane layout, same arithmetic and same bits as
// quad_solve_kernel; what goes away is the ticket round trip through LDS on the critical path of a
// dependent update (0.36 us -> see profiles/r02/microbench.json: chain_update_us_stepq_*).
template <typename REAL, int METHOD, int QT, bool HIST, bool RUNS>
__global__ void __launch_bounds__(4 * QT) step_quad_kernel(const SolveArgs<REAL> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  REAL *s_acc = reinterpret_cast<REAL *>(smem);

  const int tile = blockIdx.x, tid = threadIdx.x;
  const int q = tid & 3, side = q >> 1, half = q & 1;
  const int nslots = A.tile_nslots[tile];
  const int32_t *slot_body = A.slot_body + A.tile_slot_off[tile];
  for (int s = tid; s < nslots; s += 4 * QT) {
    const int body = slot_body[s];
#pragma unroll
    for (int k = 0; k < 6; ++k) s_acc[s * 6 + k] = (A.resume && body >= 0) ? A.acc[(size_t)body * 6 + k] : REAL(0);
  }

  const LaneDesc d = A.lanes[(size_t)tile * QT + (tid >> 2)];
  const bool active = d.cidx >= 0;
  // RUNS: a chunk of fewer than four constraints is padded with placeholders (cidx = -2, plan.h): they have the
  // chunk's slots and level, take part in the hand-off of the accumulator and update nothing
  const bool chain = RUNS ? d.cidx != -1 : active;
  const int slot = side ? d.slot1 : d.slot0;
  const bool has = chain && slot != 0;             // this lane's body is a real body
  const unsigned cnt = side ? d.cnt1 : d.cnt0, pos = side ? d.pos1 : d.pos0;
  REAL *my_acc = s_acc + slot * 6 + 3 * half;      // slot 0 (world) stays zero
  const int level = A.lane_level[(size_t)tile * QT + (tid >> 2)];
  const int P = A.tile_period[tile], depth = A.tile_depth[tile];

  REAL Jh[9], Bh[9], Dl[3], inv[3], rhs[3], lo[3], hi[3], x[3];
  bool eq[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) { Jh[k] = REAL(0); Bh[k] = REAL(0); }
#pragma unroll
  for (int r = 0; r < 3; ++r) { Dl[r] = inv[r] = rhs[r] = lo[r] = hi[r] = x[r] = REAL(0); eq[r] = true; }
  if (active) {
    const size_t c = (size_t)d.cidx;
    if (has) {
      const REAL *J = (side ? A.J1 : A.J0) + c * 18 + 3 * half;
      const REAL *B = (side ? A.wsB1 : A.wsB0) + c * 18 + 9 * half;  // rows 3*half .. 3*half+2 of the 6x3
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) Jh[3 * r + k] = J[6 * r + k];
#pragma unroll
      for (int k = 0; k < 9; ++k) Bh[k] = B[k];
    }
    const REAL *D = A.wsD + c * 9;
    if (METHOD == 1) { Dl[0] = D[3]; Dl[1] = D[6]; Dl[2] = D[7]; }   // D10, D20, D21
    else { Dl[0] = D[1]; Dl[1] = D[2]; Dl[2] = D[5]; }               // D01, D02, D12
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      inv[r] = A.wsInv[c * 3 + r];
      rhs[r] = A.rhs[c * 3 + r];
      lo[r] = A.lo[c * 3 + r];
      hi[r] = A.hi[c * 3 + r];
      eq[r] = A.is_eq[c * 3 + r] != 0;
      clamp_bounds(eq[r], lo[r], hi[r]);
      x[r] = A.resume ? A.x[c * 3 + r] : rhs[r];
    }
  }
  // length of the tile's timetable (step_solve.hip: timetable_end)
  int t_end;
  if (METHOD == 2) t_end = (A.resume ? 0 : depth) + (A.sweeps >= 1 ? depth + P * (A.sweeps - 1) : 0);
  else { const int n_phases = A.sweeps + (A.resume ? 0 : 1); t_end = n_phases >= 1 ? depth + P * (n_phases - 1) : 0; }
  __syncthreads();

  const unsigned acc_addr = lds_addr(my_acc);
  const bool last_of_body = ((METHOD == 2) ? cnt - 1u - pos : pos) == cnt - 1u;
  const int run_pos = (tid >> 2) & 3;      // RUNS: this constraint's place in its group of four (plan.h)
  int sweep = A.resume ? 1 : 0;
  const int t0 = (METHOD == 2 && !A.resume) ? depth : 0;       // backward sweeps start after the forward accumulation
  int due = (METHOD == 2 && A.resume) ? depth - 1 - level : level;
  if (!chain || sweep > A.sweeps) due = 0x7fffffff;

  REAL snap[3] = {REAL(0), REAL(0), REAL(0)};
  // one update of this lane's constraint: `an` = its quarter of the body's accumulator, in and out
  auto update = [&](REAL (&an)[3]) {
    REAL dx[3] = {REAL(0), REAL(0), REAL(0)};
    if (sweep == 0) {   // accumulators from x0 = rhs (sparse_iterations.cc:202)
#pragma unroll
      for (int r = 0; r < 3; ++r) dx[r] = x[r];
    } else {
      REAL res[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        REAL p = Jh[3 * r] * an[0];
        p = tfma(Jh[3 * r + 1], an[1], p);
        p = tfma(Jh[3 * r + 2], an[2], p);
        const REAL full = tfma(A.cfm, x[r], quad_sum(p));
        res[r] = rhs[r] - full;
      }
      if (METHOD == 1) {
        REAL t0r = res[0];
        REAL xn = project(tfma(t0r, inv[0], x[0]), lo[0], hi[0]);
        dx[0] = xn - x[0]; x[0] = xn;
        REAL t1 = tfma(-Dl[0], dx[0], res[1]);
        xn = project(tfma(t1, inv[1], x[1]), lo[1], hi[1]);
        dx[1] = xn - x[1]; x[1] = xn;
        REAL t2 = tfma(-Dl[1], dx[0], res[2]);
        t2 = tfma(-Dl[2], dx[1], t2);
        xn = project(tfma(t2, inv[2], x[2]), lo[2], hi[2]);
        dx[2] = xn - x[2]; x[2] = xn;
      } else {
        REAL t2 = res[2];
        REAL xn = project(tfma(t2, inv[2], x[2]), lo[2], hi[2]);
        dx[2] = xn - x[2]; x[2] = xn;
        REAL t1 = tfma(-Dl[2], dx[2], res[1]);                 // D12
        xn = project(tfma(t1, inv[1], x[1]), lo[1], hi[1]);
        dx[1] = xn - x[1]; x[1] = xn;
        REAL t0r = tfma(-Dl[1], dx[2], res[0]);                // D02
        t0r = tfma(-Dl[0], dx[1], t0r);                        // D01
        xn = project(tfma(t0r, inv[0], x[0]), lo[0], hi[0]);
        dx[0] = xn - x[0]; x[0] = xn;
      }
    }
    if (has) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        REAL u = tfma(Bh[3 * k + 0], dx[0], an[k]);
        u = tfma(Bh[3 * k + 1], dx[1], u);
        an[k] = tfma(Bh[3 * k + 2], dx[2], u);
      }
    }
    if (HIST) { snap[0] = an[0]; snap[1] = an[1]; snap[2] = an[2]; }   // the accumulator right after THIS update
  };
  // snapshots for the per-sweep stopping test (kernels.h), once per time step: every constraint of a group has
  // had its one update by then, so lambda and the accumulator copy taken in update() are those of this sweep
  auto record = [&]() {
    if (HIST && sweep >= 1 && active) {
      if (q == 0) {
        REAL *hx = A.hist_x + ((size_t)(sweep - 1) * A.m + d.cidx) * 3;
        hx[0] = x[0]; hx[1] = x[1]; hx[2] = x[2];
      }
      if (has && last_of_body) {   // this was the body's last update of the sweep
        const int body = slot_body[slot];
        REAL *ha = A.hist_acc + ((size_t)(sweep - 1) * A.n_bodies + body) * 6 + 3 * half;
        ha[0] = snap[0]; ha[1] = snap[1]; ha[2] = snap[2];
      }
    }
  };

  for (int t = 0; t < t_end; ++t) {
    if (due == t) {
      REAL an[3];
      load3(acc_addr, an);
      if (!RUNS) {
        update(an);
        if (has) store3(acc_addr, an);
      } else if (METHOD == 1 || t < t0) {
        // the group's four updates in list order; the accumulator goes from lane to lane (row_shr:4 =
        // the same quarter of the previous constraint), LDS sees the first read and the last write only
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          if (sub > 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) an[k] = dpp<0x114>(an[k]);
          }
          if (run_pos == sub && active) update(an);
        }
        if (has && run_pos == 3) store3(acc_addr, an);
      } else {
        // backward sweep: the list, and the group, from its end (row_shl:4)
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          if (sub > 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) an[k] = dpp<0x104>(an[k]);
          }
          if (run_pos == 3 - sub && active) update(an);
        }
        if (has && run_pos == 0) store3(acc_addr, an);
      }
      record();
      due = (METHOD == 2 && sweep == 0) ? t0 + (depth - 1 - level) : due + P;
      if (++sweep > A.sweeps) due = 0x7fffffff;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wavefront's accumulator stores have landed
    __builtin_amdgcn_s_barrier();
  }

  if (active) {  // lambda and w = A x - rhs with the final accumulators
    REAL a[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) a[k] = my_acc[k];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      REAL p = Jh[3 * r] * a[0];
      p = tfma(Jh[3 * r + 1], a[1], p);
      p = tfma(Jh[3 * r + 2], a[2], p);
      const REAL w = tfma(A.cfm, x[r], quad_sum(p)) - rhs[r];
      if (q == 0) {
        A.x[(size_t)d.cidx * 3 + r] = x[r];
        A.wres[(size_t)d.cidx * 3 + r] = w;
      }
    }
  }
  for (int s = tid + 1; s < nslots; s += 4 * QT) {
    const int body = slot_body[s];
    if (body < 0) continue;   // unused slot number
#pragma unroll
    for (int k = 0; k < 6; ++k) A.acc[(size_t)body * 6 + k] = s_acc[s * 6 + k];
  }
}

}  // namespace

template <typename REAL>
void launch_quad_solve(const SolveArgs<REAL> &a, int method, int n_tiles, int tile_size, hipStream_t s) {
  if (n_tiles <= 0) return;
  const size_t lds = (size_t)a.max_slots * (6 * sizeof(REAL) + sizeof(unsigned));
  uint32_t *none = nullptr;
  const bool hist = a.hist_x != nullptr;
#define EGS_QLAUNCH(METHOD, QT, HIST) \
  hipLaunchKernelGGL((quad_solve_kernel<REAL, METHOD, QT, false, HIST>), dim3(n_tiles), dim3(4 * QT), lds, s, a, none)
  if (tile_size == 64) {
    if (method == 1) { if (hist) EGS_QLAUNCH(1, 64, true); else EGS_QLAUNCH(1, 64, false); }
    else { if (hist) EGS_QLAUNCH(2, 64, true); else EGS_QLAUNCH(2, 64, false); }
  } else if (tile_size == 128) {
    if (method == 1) { if (hist) EGS_QLAUNCH(1, 128, true); else EGS_QLAUNCH(1, 128, false); }
    else { if (hist) EGS_QLAUNCH(2, 128, true); else EGS_QLAUNCH(2, 128, false); }
  } else if (tile_size == 256) {
    if (method == 1) { if (hist) EGS_QLAUNCH(1, 256, true); else EGS_QLAUNCH(1, 256, false); }
    else { if (hist) EGS_QLAUNCH(2, 256, true); else EGS_QLAUNCH(2, 256, false); }
#undef EGS_QLAUNCH
  } else {
    throw std::invalid_argument("launch_quad_solve: tile size must be 64, 128 or 256");
  }
}

// Body patches of oversize islands, 256 constraints = 1024 threads per patch
// (one workgroup per CU at 96 VGPRs: the host keeps n_tiles <= 256 so that all
// patches are co-resident).
template <typename REAL>
void launch_quad_patch_solve(const SolveArgs<REAL> &a, int method, int n_tiles, uint32_t *tickets, hipStream_t s) {
  if (n_tiles <= 0) return;
  const size_t lds = (size_t)a.max_slots * (6 * sizeof(REAL) + sizeof(unsigned));
  const bool hist = a.hist_x != nullptr;
#define EGS_QPLAUNCH(M, H)                                                                                                      \
  do {                                                                                                                         \
    if (a.patch_runs) hipLaunchKernelGGL((quad_solve_kernel<REAL, M, 256, true, H, true>), dim3(n_tiles), dim3(1024), lds, s, a, tickets); \
    else hipLaunchKernelGGL((quad_solve_kernel<REAL, M, 256, true, H, false>), dim3(n_tiles), dim3(1024), lds, s, a, tickets);             \
  } while (0)
  if (method == 1) { if (hist) EGS_QPLAUNCH(1, true); else EGS_QPLAUNCH(1, false); }
  else { if (hist) EGS_QPLAUNCH(2, true); else EGS_QPLAUNCH(2, false); }
#undef EGS_QPLAUNCH
}

template <typename REAL>
int occupancy_quad_patch_solve(size_t lds) {
  int best = 1 << 30, nb = 0;
#define EGS_OCC(M, H)                                                                                     \
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, quad_solve_kernel<REAL, M, 256, true, H, false>, 1024, lds) != hipSuccess) return 0; \
  best = nb < best ? nb : best;                                                                           \
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, quad_solve_kernel<REAL, M, 256, true, H, true>, 1024, lds) != hipSuccess) return 0;  \
  best = nb < best ? nb : best;
  EGS_OCC(1, true) EGS_OCC(1, false) EGS_OCC(2, true) EGS_OCC(2, false)
#undef EGS_OCC
  return best;
}
// workgroups of the 4-lane timetable kernel one CU keeps resident (tile_size constraints = 4 x as many threads)
template <typename REAL>
int occupancy_step_quad(int tile_size, size_t lds) {
  int nb = 0;
  hipError_t e = hipErrorInvalidValue;
  if (tile_size == 64) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, step_quad_kernel<REAL, 1, 64, false, false>, 256, lds);
  else if (tile_size == 128) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, step_quad_kernel<REAL, 1, 128, false, false>, 512, lds);
  else if (tile_size == 256) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, step_quad_kernel<REAL, 1, 256, false, false>, 1024, lds);
  return e == hipSuccess ? nb : 0;
}
template int occupancy_step_quad<double>(int, size_t);
template int occupancy_step_quad<float>(int, size_t);
template int occupancy_quad_patch_solve<double>(size_t);
template int occupancy_quad_patch_solve<float>(size_t);

template void launch_quad_solve<double>(const SolveArgs<double> &, int, int, int, hipStream_t);
template void launch_quad_solve<float>(const SolveArgs<float> &, int, int, int, hipStream_t);
template void launch_quad_patch_solve<double>(const SolveArgs<double> &, int, int, uint32_t *, hipStream_t);
template void launch_quad_patch_solve<float>(const SolveArgs<float> &, int, int, uint32_t *, hipStream_t);

template <typename REAL>
void launch_step_quad(const SolveArgs<REAL> &a, int method, int n_tiles, int tile_size, hipStream_t s) {
  if (n_tiles <= 0) return;
  const size_t lds = (size_t)a.max_slots * 6 * sizeof(REAL);
  const bool hist = a.hist_x != nullptr;
#define EGS_SQLAUNCH(METHOD, QT, HIST)                                                                           \
  do {                                                                                                           \
    if (a.runs) hipLaunchKernelGGL((step_quad_kernel<REAL, METHOD, QT, HIST, true>), dim3(n_tiles), dim3(4 * QT), lds, s, a); \
    else hipLaunchKernelGGL((step_quad_kernel<REAL, METHOD, QT, HIST, false>), dim3(n_tiles), dim3(4 * QT), lds, s, a);       \
  } while (0)
  if (tile_size == 64) {
    if (method == 1) { if (hist) EGS_SQLAUNCH(1, 64, true); else EGS_SQLAUNCH(1, 64, false); }
    else { if (hist) EGS_SQLAUNCH(2, 64, true); else EGS_SQLAUNCH(2, 64, false); }
  } else if (tile_size == 128) {
    if (method == 1) { if (hist) EGS_SQLAUNCH(1, 128, true); else EGS_SQLAUNCH(1, 128, false); }
    else { if (hist) EGS_SQLAUNCH(2, 128, true); else EGS_SQLAUNCH(2, 128, false); }
  } else if (tile_size == 256) {
    if (method == 1) { if (hist) EGS_SQLAUNCH(1, 256, true); else EGS_SQLAUNCH(1, 256, false); }
    else { if (hist) EGS_SQLAUNCH(2, 256, true); else EGS_SQLAUNCH(2, 256, false); }
#undef EGS_SQLAUNCH
  } else {
    throw std::invalid_argument("launch_step_quad: tile size must be 64, 128 or 256");
  }
}
template void launch_step_quad<double>(const SolveArgs<double> &, int, int, int, hipStream_t);
template void launch_step_quad<float>(const SolveArgs<float> &, int, int, int, hipStream_t);

}  // namespace egs
