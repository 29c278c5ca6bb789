// kernels.h -- launch interface between the C-ABI layer and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "plan.h"

namespace egs {

// Flat system + tile plan, all device pointers.
template <typename REAL>
struct SolveArgs {
  const LaneDesc *lanes;
  const int32_t *tile_nslots;
  const int32_t *tile_slot_off;
  const int32_t *slot_body;
  const REAL *Minv;            // [n][36]
  const REAL *J0, *J1;         // [m][18]
  const uint8_t *is_eq;        // [3m]
  const REAL *lo, *hi, *rhs;   // [3m]
  REAL *x;                     // [3m]  in (resume) / out
  REAL *acc;                   // [n][6] in (resume) / out
  REAL *wres;                  // [3m]  out: A x - rhs
  // per-constraint derived blocks (quad kernel only), written by cons_prepare:
  REAL *wsB0, *wsB1;           // [m][18]  W J^T as 6x3 row-major
  REAL *wsD, *wsInv;           // [m][9], [m][3]
  const int32_t *body0, *body1; // [m] (cons_prepare only)
  int32_t m;
  int32_t *error_flag;
  REAL cfm, kscale;
  int32_t sweeps, resume, max_slots;
  uint32_t spin_limit;
  // static timetable of the same sweep (step_solve.hip, plan.h): per lane its level, per tile the
  // period P and the depth of the level DAG
  const uint16_t *lane_level = nullptr;
  const int32_t *tile_period = nullptr, *tile_depth = nullptr;
  int32_t n_tiles = 0;
  int32_t runs = 0;            // the timetable counts groups of four constraints (plan.h: Plan::runs)
  int32_t patch_runs = 0;      // body patches: the lanes come in chunks of four quads (Plan::patch_runs, quad_solve.hip)
  int iso = 0;              // 1: every M^-1 block is diag(a,a,a,b,b,b): B is formed on the fly (tile kernel)
  // Per-sweep history (tolerance-terminated solves): x after sweep s and each body's
  // accumulator once its last constraint of sweep s has run, s = 1..sweeps of this launch.
  // hist_residual then evaluates the reference's per-iteration stopping test for the
  // whole chunk from these snapshots: one launch and one read-back per chunk instead of
  // one per sweep.  NULL = off.
  REAL *hist_x = nullptr;      // [sweeps][m][3]
  REAL *hist_acc = nullptr;    // [sweeps][n_bodies][6]
  int32_t n_bodies = 0;
  // body patches on the 4-lane kernel: a shared body's accumulator crosses between patches as six data-tagged
  // 16-byte granules {value, launch epoch << 32 | ticket} (quad_solve.hip); NULL = the flag protocol (g_tick)
  void *gran = nullptr;        // [n_bodies][6] x 16 B
  uint32_t gran_epoch = 0;
  // diagnostics (EGS_TRACE_UPDATES=1): completion time (100 MHz wall clock) of every update of the 4-lane patch kernel
  unsigned long long *trace = nullptr;   // [sweeps][m]
};

template <typename REAL>
struct GlobalArgs {
  const GlobalDesc *cons;      // [mg]
  int32_t mg, per_lane;        // constraints, constraints per lane
  int32_t n_bodies, pad0;
  const REAL *Minv, *J0, *J1;
  const uint8_t *is_eq;
  const REAL *lo, *hi, *rhs;
  REAL *x, *acc, *wres;
  REAL *B0, *B1, *D, *den, *dx; // workspace [mg][18|18|9|3|3]
  uint32_t *tickets;           // [n] zeroed before every launch
  int32_t *error_flag;
  REAL cfm, kscale;
  int32_t sweeps, resume, method;
  int32_t mode;                // 0 = full sweeps, 1 = ordered accumulate of dx only
  uint32_t spin_limit;
  // per-sweep snapshots for tolerance-terminated solves, as in SolveArgs (NULL = off)
  int32_t m = 0, pad1 = 0;     // all constraints of the problem (hist_x stride)
  REAL *hist_x = nullptr, *hist_acc = nullptr;
};

struct AssembleArgs {
  int32_t n, m;
  const double *pos, *R, *v, *w;                 // body state, fp64
  const double *Wf;                              // [n][6] M^-1 f_ext (launch_mass_times_force)
  const int32_t *kind, *body0, *body1;
  const double *data;                            // [m][7]
  double dt, erp;
  void *J0, *J1, *lo, *hi, *rhs;                 // REAL outputs
  double *err;                                   // [3m] fp64
  uint8_t *is_eq;
};

// Wf[b] = M_b^-1 f_ext,b: both frozen at Init (Q5), so assembly and the velocity update read
// these 48 B per body instead of the 288 B block and the force (same expression, same bits).
void launch_mass_times_force(int n, const double *Minv, const double *f_ext, double *Wf, hipStream_t s);
template <typename REAL>
void launch_tile_solve(const SolveArgs<REAL> &a, int method, int n_tiles,
                       int block, hipStream_t s);
// the same GS / SOR sweep on the plan's static timetable: one workgroup barrier per time step, no tickets
template <typename REAL>
void launch_step_solve(const SolveArgs<REAL> &a, int method, int n_tiles, int block, hipStream_t s);
// the same timetable in 128 VGPRs (lean_solve.hip): fp64, isotropic bodies, 256-constraint tiles, J1_lin = -J0_lin
void launch_lean_solve(const SolveArgs<double> &a, int method, int n_tiles, int block, hipStream_t s);
int occupancy_lean_solve(int block, int max_slots);
// ... and the 4-lanes-per-constraint schedule on the same timetable (quad_solve.hip)
template <typename REAL>
void launch_step_quad(const SolveArgs<REAL> &a, int method, int n_tiles, int tile_size, hipStream_t s);
// max_blocks: how many workgroups of the persistent grid may be launched (all must be resident)
template <typename REAL>
void launch_global_solve(const GlobalArgs<REAL> &a, int max_blocks, hipStream_t s);
// workgroups per CU the hardware keeps resident for the cross-workgroup kernels' exact
// instantiations (hipOccupancyMaxActiveBlocksPerMultiprocessor, the smallest over method /
// history variants); 0 if the query fails
template <typename REAL> int occupancy_global_solve();
template <typename REAL> int occupancy_patch_solve(size_t lds_bytes);
template <typename REAL> int occupancy_quad_patch_solve(size_t lds_bytes);
template <typename REAL> int occupancy_step_quad(int tile_size, size_t lds_bytes);
// w = A x - rhs for the constraints of a GlobalDesc list (after the last sweep)
template <typename REAL>
void launch_global_wres(const GlobalArgs<REAL> &a, hipStream_t s);
// oversize islands cut into body patches (plan.h): LDS for private bodies,
// global sc1 hand-off for shared ones
template <typename REAL>
void launch_patch_solve(const SolveArgs<REAL> &a, int method, int n_tiles, uint32_t *tickets, hipStream_t s);
// latency-optimised variant: 4 lanes per constraint, 64 constraints per tile
template <typename REAL>
void launch_cons_prepare(const SolveArgs<REAL> &a, hipStream_t s);
template <typename REAL>
void launch_quad_solve(const SolveArgs<REAL> &a, int method, int n_tiles, int tile_size, hipStream_t s);
template <typename REAL>
void launch_quad_patch_solve(const SolveArgs<REAL> &a, int method, int n_tiles, uint32_t *tickets, hipStream_t s);
template <typename REAL>
void launch_assemble(const AssembleArgs &a, hipStream_t s);
// partial sums of squares by row category: out[4*blocks]
template <typename REAL>
void launch_residual_partials(int rows, const REAL *wres, const REAL *x,
                              const REAL *lo, const REAL *hi,
                              const uint8_t *is_eq, double *out, int blocks,
                              hipStream_t s);
template <typename REAL>
void launch_velocity(int n, const double *v, const double *w, const double *Wf, const REAL *acc,
                     double dt, double *v6, hipStream_t s);
// Residual partial sums (the 4 categories of sparse_iterations.cc:51-69, `blocks` partial
// sums each, same reduction order as launch_residual_partials) for every sweep of a recorded
// chunk: out [sweeps][blocks][4].  write_sweep >= 1 also stores that sweep's w into wres.
template <typename REAL>
void launch_hist_residual(const SolveArgs<REAL> &a, int sweeps, int blocks, double *out, int write_sweep, hipStream_t s);
template <typename REAL>
void launch_convert_minv(int count, const double *src, REAL *dst, hipStream_t s);
// *flag (preset to 1) is cleared unless every 6x6 block is exactly diag(a, a, a, b, b, b)
template <typename REAL>
void launch_minv_iso(int n, const REAL *W, int *flag, hipStream_t s);

// A = J M^-1 J^T + cfm I, [3m][3m] row-major fp64 on the device (ensembles.cc:510, 513-521)
void launch_dense_system(int m, const int32_t *body0, const int32_t *body1, const double *J0, const double *J1,
                         const double *Minv, double cfm, double *A, hipStream_t s);

void launch_advance(int n, double *pos, double *R, double *v, double *w, const double *v6, double dt,
                    hipStream_t s);

constexpr int kResidualBlocks = 64;

}  // namespace egs
