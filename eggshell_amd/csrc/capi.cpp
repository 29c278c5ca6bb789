// capi.cpp -- the extern "C" boundary (include/eggshell_amd.h): contexts,
// device-resident problems, solve driver.  No torch types, no CPU fallback:
// without a usable HIP device every entry fails with EGS_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/eggshell_amd.h"
#include "collide.h"
#include "dense_lcp.h"
#include "kernels.h"
#include "matvec.h"
#include "plan.h"

using namespace egs;

// Page-locked staging for host->device uploads of plan tables and device->host
// reads of the contact topology: pageable std::vector memory makes every
// hipMemcpyAsync a synchronous bounce through the runtime's own staging buffer.
struct PinnedArena {
  std::vector<std::pair<char *, size_t>> blocks;
  size_t used = 0;   // in blocks.back()
  void *take(size_t bytes) {
    bytes = (bytes + 63) & ~size_t(63);
    if (blocks.empty() || used + bytes > blocks.back().second) {
      const size_t want = std::max(bytes, blocks.empty() ? size_t(1) << 20 : 2 * blocks.back().second);
      char *p = nullptr;
      if (hipHostMalloc(reinterpret_cast<void **>(&p), want, hipHostMallocDefault) != hipSuccess)
        throw std::runtime_error("hipHostMalloc failed");
      blocks.emplace_back(p, want);
      used = 0;
    }
    void *r = blocks.back().first + used;
    used += bytes;
    return r;
  }
  // call only when no copy from the arena is in flight: keeps the largest block
  void reset() {
    while (blocks.size() > 1) { (void)hipHostFree(blocks.front().first); blocks.erase(blocks.begin()); }
    used = 0;
  }
  ~PinnedArena() { for (auto &b : blocks) (void)hipHostFree(b.first); }
};

struct egs_context {
  PinnedArena pinned;
  int device = 0;
  int cu_count = 256;   // co-residency caps of the cross-workgroup kernels scale with it
  hipStream_t stream = nullptr;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  std::string error;
  // hipEvent pairs around every solve-kernel launch
  std::vector<hipEvent_t> kev;
  size_t kev_used = 0;
  // egs_solve_blocks is stateless for its caller, but a simulation calls it every step with
  // the same constraint graph: the last problem (schedule + device buffers) is kept and reused
  // when n, m, precision and body0/body1 are unchanged (4.6 -> 0.7 ms per call at C3).
  egs_problem *oneshot = nullptr;
};

namespace {

constexpr size_t kEventPairs = 4096;
constexpr uint32_t kSpinLimitDefault = 1u << 22;
// EGS_DEBUG_SPIN_LIMIT=k: bound of the device-side ordering waits (tests force a stall with 1)
inline uint32_t spin_limit() {
  const char *e = std::getenv("EGS_DEBUG_SPIN_LIMIT");
  const long v = e ? std::atol(e) : 0;
  return v > 0 ? (uint32_t)v : kSpinLimitDefault;
}
// The 4-lane schedule is the faster one while its tiles are all resident at once (one round): 5 x 64
// constraints per CU in fp64 (94 VGPRs), 8 x 64 in fp32 -- hipOccupancyMaxActiveBlocksPerMultiprocessor
// of the instantiation decides (C3 fp64: 4 piles 0.33 ms against 0.41 on the 1-lane schedule, 6 piles
// 0.53 against 0.45).  No 4-lane plan is built at all beyond the register file's 5 (fp64) / 8 (fp32)
// tiles per CU.
inline int quad_tiles_per_cu_max(int precision) { return precision == EGS_F32 ? 8 : 5; }
constexpr int kBigTileMinConstraints = 196608;   // 768 tiles of 256: from here 512-constraint tiles
// Which kernel takes the oversize islands of a GS / SOR solve.  Patches wait on each other, so
// all of a launch's patches must be co-resident: the limits are occupancy (workgroups per CU of
// the exact kernel instantiation, queried from the runtime) x CUs, never above what was
// measured on MI355X -- one 1024-thread patch (4 lanes per constraint) or two 256-thread
// patches (232 VGPRs) per CU.  A kernel change that lowers occupancy lowers the limit and the
// island falls through to the next schedule instead of stalling.
// The tile plan's GS / SOR sweep on its static timetable (step_solve.hip) or on tickets
// (tile_solve_kernel)?  The timetable takes depth + P x sweeps barrier steps, P = the largest level
// span of a body in a tile; the ticket sweep follows the true dependency chains, about
// depth + (largest per-body count) x sweeps updates long.  Regular islands (piles of columns) have
// P = the per-body count and the timetable wins (every lane due at a step shares ONE pass); an
// irregular island can have spans far beyond its counts, then the tickets win.
// EGS_STEP=0 / 1 forces one or the other.
inline bool use_static_timetable(const Plan &pl, int sweeps) {
  if (!pl.levels_ok || pl.n_tiles <= 0) return false;
  const char *e = std::getenv("EGS_STEP");
  if (e) return std::atoi(e) != 0;
  const double grp = pl.runs ? 4.0 : 1.0;     // with runs the timetable counts groups of four updates (plan.h)
  const double fixed = grp * ((double)pl.max_depth + (double)pl.max_period * sweeps);
  const double ticket = grp * (double)pl.max_depth + (double)pl.max_cnt * sweeps;
  return fixed <= 1.15 * ticket;
}

enum OversizeSchedule { kQuadPatches = 0, kLanePatches = 1, kAllGlobal = 2 };
inline OversizeSchedule choose_oversize_schedule(int n_patch_tiles, int quad_per_cu, int patch_per_cu, int cu_count,
                                                 bool patches_enabled, bool quad_patches_enabled) {
  if (n_patch_tiles <= 0 || !patches_enabled) return kAllGlobal;
  const long quad_cap = (long)std::min(quad_per_cu, 1) * cu_count, lane_cap = (long)std::min(patch_per_cu, 2) * cu_count;
  if (quad_patches_enabled && n_patch_tiles <= quad_cap) return kQuadPatches;
  if (n_patch_tiles <= lane_cap) return kLanePatches;
  return kAllGlobal;
}

// Isotropic bodies, batched work: the register-light tile kernel (no stored B, three
// 256-constraint tiles per CU in fp64, four in fp32) against the regular one (fp64:
// 512-constraint tiles, one per CU; fp32: three 256-constraint tiles).
// Tiles are dispatched in rounds of 3C resp. C, so the better choice depends on how the
// tile count quantises; per-round times (ms, C3 columns, 100 sweeps) measured on MI355X.
inline bool iso_schedule_pays(long m, int cu, int precision) {
  const long t = (m + 255) / 256;                       // 256-constraint tiles
  if (precision == EGS_F32) return t > 3L * cu;         // fp32: 4 instead of 3 tiles per CU (C4: +7 %)
  if (t < 2L * cu) return false;                       // fewer than two tiles per CU: registers are not the limit
  const long full3 = t / (3L * cu), rem3 = t % (3L * cu);
  // (timetable kernels: one / two / three isotropic tiles per CU walk a launch in 0.435 / 0.462 / 0.50 ms, a round of
  //  512-constraint regular tiles in 0.415 ms; the ticket kernels' figures were 0.45 / 0.57 / 0.66 and 0.53 -- same choices)
  const double iso = 0.50 * full3 + (rem3 == 0 ? 0.0 : rem3 <= cu ? 0.435 : rem3 <= 2L * cu ? 0.462 : 0.50);
  const double regular = 0.415 * ((t / 2 + cu - 1) / cu);
  return iso < regular;
}

struct HipError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

void hip_check(hipError_t e, const char *what) {
  if (e != hipSuccess) {
    char buf[256];
    std::snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    throw HipError(buf);
  }
}
#define HIPCHK(call) hip_check((call), #call)

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t count = 0, cap = 0;
  // Grow-only: a buffer that is already large enough is kept (a world re-plans
  // on every contact-topology change; hipMalloc/hipFree would dominate that).
  void alloc(size_t n) {
    if (n > cap) {
      const size_t want = p ? n + n / 4 : n;   // head-room only once a buffer has had to grow
      release();
      HIPCHK(hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(T)));
      cap = want;
    }
    count = n;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    count = cap = 0;
  }
  size_t bytes() const { return cap * sizeof(T); }
  ~DevBuf() { release(); }
};

template <typename T>
void upload(DevBuf<T> &d, const T *src, size_t n, hipStream_t s) {
  if (n == 0) return;
  HIPCHK(hipMemcpyAsync(d.p, src, n * sizeof(T), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));  // src may be a temporary
}

// alloc + copy through the context's pinned arena, without a synchronise; the
// caller synchronises before the arena is reset
template <typename T>
void stage(egs_context *ctx, DevBuf<T> &d, const std::vector<T> &src) {
  d.alloc(src.size());
  if (src.empty()) return;
  const size_t bytes = src.size() * sizeof(T);
  void *h = ctx->pinned.take(bytes);
  std::memcpy(h, src.data(), bytes);
  HIPCHK(hipMemcpyAsync(d.p, h, bytes, hipMemcpyHostToDevice, ctx->stream));
}

}  // namespace

struct egs_problem {
  egs_context *ctx = nullptr;
  int n = 0, m = 0, precision = EGS_F64;
  Plan plan;                     // 1 lane per constraint (built lazily when the quad schedule applies)
  bool tile_plan_ready = false;
  std::vector<int32_t> h_body0, h_body1;
  // plan
  DevBuf<LaneDesc> lanes;
  DevBuf<uint16_t> lane_level;           // static timetable of the tile plan (plan.h)
  DevBuf<int32_t> tile_period, tile_depth;
  DevBuf<int32_t> tile_nslots, tile_slot_off, slot_body;
  // latency-optimised schedule (4 lanes per constraint, 64-constraint tiles);
  // used for GS/SOR when the problem is small and every island fits a tile
  Plan planq;
  bool use_quad = false;
  // which kernel takes the oversize islands of a GS / SOR solve (choose_oversize_schedule; EGS_PATCH=0
  // forces the all-global kernel, EGS_QUAD_PATCH=0 the 1-lane patches) and how many workgroups the
  // all-global kernel's persistent grid may have: both from the runtime's occupancy of the kernels
  int last_iso = 0;            // the last tile launch used the isotropic-body variant
  int last_static = 0;         // ... ran on the static timetable (step_solve.hip)
  int last_lean = 0;           // ... in its 128-VGPR form (lean_solve.hip)
  bool lin_antisym = false;    // J1_lin == -J0_lin on every two-body constraint (device assembly: by construction;
                               // egs_problem_set_blocks: checked on the host), what lean_step_kernel relies on
  int oversize = 2;            // OversizeSchedule
  int global_max_blocks = 1;
  DevBuf<LaneDesc> q_lanes;
  DevBuf<uint16_t> q_lane_level;         // static timetable of the 4-lane plan
  DevBuf<int32_t> q_tile_period, q_tile_depth;
  DevBuf<int32_t> q_tile_nslots, q_tile_slot_off, q_slot_body;
  DevBuf<unsigned char> wsB0, wsB1, wsD, wsInv;
  DevBuf<GlobalDesc> gcons;
  DevBuf<uint32_t> gtickets;
  DevBuf<unsigned long long> trace;   // EGS_TRACE_UPDATES=1: [sweeps][m] completion times of the last patch launch
  int trace_sweeps = 0;
  DevBuf<unsigned char> ggran;   // [n][6] x 16 B: data-tagged granules of the 4-lane body patches (quad_solve.hip)
  uint32_t gran_epoch = 0;
  // oversize islands as body patches (GS/SOR): LDS for private bodies, global for shared
  DevBuf<LaneDesc> p_lanes;
  DevBuf<int32_t> p_tile_nslots, p_tile_slot_off, p_slot_body;
  // topology + state (fp64)
  DevBuf<int32_t> body0, body1, kind;
  DevBuf<double> pos, R, v, w, Minv_d, f_ext, data, err, v6, res_partials;
  DevBuf<double> Wf;            // M^-1 f_ext per body, rebuilt when either is re-uploaded
  bool wf_valid = false;
  // solver arrays, REAL = double or float (byte buffers)
  DevBuf<unsigned char> Minv_r, J0, J1, lo, hi, rhs, x, acc, wres;
  DevBuf<unsigned char> gB0, gB1, gD, gden, gdx;  // cross-workgroup workspace
  DevBuf<uint8_t> is_eq;
  // [0]: device-side ordering wait timed out (EGS_ERR_STALL).  STICKY: the solve kernels OR into
  // it and only reporting it clears it, so a stall in any step of an asynchronous run is seen.
  // [1]: scratch word of the isotropy check.
  DevBuf<int32_t> error_flag;
  int32_t *h_flag = nullptr;   // page-locked copy of [0], refreshed by an async copy after every solve
  double *h_hist = nullptr;    // page-locked landing area of the stopping loop's per-sweep residual sums (+ the flag)
  size_t h_hist_cap = 0;       // in doubles
  // per-sweep history of a chunk of sweeps (tolerance-terminated solves, see kernels.h)
  DevBuf<unsigned char> hist_x, hist_acc;
  DevBuf<double> hist_out;
  int hist_sweeps = 0;        // 0: off for the next launch; k: record k sweeps
  bool residual_pending = false;   // wres/x hold a finished solve whose residual sums were not reduced yet
  bool have_blocks = false, have_state = false, have_constraints = false, minv_r_valid = false;
  // stand-alone mat-vec (matvec_plan.h): schedule built at the first product after a topology change
  MatvecPlan mvplan;
  bool mv_ready = false;
  DevBuf<MvLane> mv_lanes;
  DevBuf<MvTile> mv_tiles;
  DevBuf<MvSlot> mv_slots;
  DevBuf<uint16_t> mv_ents;
  DevBuf<MvBoundary> mv_boundary;
  DevBuf<unsigned char> mv_T, mv_x, mv_y;
  DevBuf<unsigned char> tmp_rows;   // [3m] REAL scratch
  DevBuf<double> dense_A;        // J M^-1 J^T + cfm I of the dense path (egs_problem_dense_system), [3m][3m]
  double dense_cfm = -1.0;       // the cfm dense_A was built with (< 0: not built)
  // row types and bounds of the assembled system on the host (the dense path partitions by them): they depend on the
  // constraint kinds only (joints.cc:13-35, contact.cc:103-113), so they are read back once per set of kinds
  std::vector<uint8_t> h_rows_eq;
  std::vector<double> h_rows_lo, h_rows_hi;
  bool h_rows_valid = false;
  bool minv_iso = false;       // every M^-1 block is diag(a,a,a,b,b,b): the tile kernel keeps no B (EGS_ISO=0 disables)
  int last_iterations = 0;
  size_t real_size() const { return precision == EGS_F32 ? sizeof(float) : sizeof(double); }
};

namespace {

egs_status fail(egs_context *ctx, egs_status st, const std::string &msg) {
  if (ctx) ctx->error = msg;
  return st;
}

template <typename F>
egs_status guarded(egs_context *ctx, F &&f) {
  try {
    return f();
  } catch (const HipError &e) {
    return fail(ctx, EGS_ERR_HIP, e.what());
  } catch (const std::invalid_argument &e) {
    return fail(ctx, EGS_ERR_INVALID, e.what());
  } catch (const std::bad_alloc &) {
    return fail(ctx, EGS_ERR_INTERNAL, "host allocation failed");
  } catch (const std::exception &e) {   // std::logic_error from the planner etc.: a library bug, not a HIP failure
    return fail(ctx, EGS_ERR_INTERNAL, e.what());
  }
}

void upload_real(egs_problem *p, DevBuf<unsigned char> &dst, const double *src, size_t count) {
  if (!src || count == 0) return;
  hipStream_t s = p->ctx->stream;
  if (p->precision == EGS_F32) {
    std::vector<float> tmp(count);
    for (size_t i = 0; i < count; ++i) tmp[i] = (float)src[i];
    HIPCHK(hipMemcpyAsync(dst.p, tmp.data(), count * sizeof(float), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
  } else {
    HIPCHK(hipMemcpyAsync(dst.p, src, count * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
  }
}

void download_real(egs_problem *p, const DevBuf<unsigned char> &src, double *dst, size_t count) {
  if (!dst || count == 0) return;
  hipStream_t s = p->ctx->stream;
  if (p->precision == EGS_F32) {
    std::vector<float> tmp(count);
    HIPCHK(hipMemcpyAsync(tmp.data(), src.p, count * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (size_t i = 0; i < count; ++i) dst[i] = (double)tmp[i];
  } else {
    HIPCHK(hipMemcpyAsync(dst, src.p, count * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
  }
}

void ensure_minv_real(egs_problem *p) {
  if (p->minv_r_valid) return;
  const int count = p->n * 36;
  if (p->precision == EGS_F32)
    launch_convert_minv<float>(count, p->Minv_d.p, reinterpret_cast<float *>(p->Minv_r.p), p->ctx->stream);
  else
    launch_convert_minv<double>(count, p->Minv_d.p, reinterpret_cast<double *>(p->Minv_r.p), p->ctx->stream);
  p->minv_r_valid = true;
  // isotropy of the blocks, checked once per upload (4 bytes back)
  const char *ie = std::getenv("EGS_ISO");
  if (p->n > 0 && !(ie && std::atoi(ie) == 0)) {
    hipStream_t s = p->ctx->stream;
    int one = 1, flag = 0;
    int32_t *scratch = p->error_flag.p + 1;
    HIPCHK(hipMemcpyAsync(scratch, &one, sizeof(int), hipMemcpyHostToDevice, s));
    if (p->precision == EGS_F32) launch_minv_iso<float>(p->n, reinterpret_cast<const float *>(p->Minv_r.p), scratch, s);
    else launch_minv_iso<double>(p->n, reinterpret_cast<const double *>(p->Minv_r.p), scratch, s);
    HIPCHK(hipMemcpyAsync(&flag, scratch, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const bool iso = flag != 0;
    if (iso != p->minv_iso) p->tile_plan_ready = false;   // the preferred tile size depends on it
    p->minv_iso = iso;
  }
}

// ---- the sticky stall flag ---------------------------------------------------
// asynchronous refresh of the page-locked copy (4 bytes), enqueued behind a solve
void post_flag_copy(egs_problem *p) {
  HIPCHK(hipMemcpyAsync(p->h_flag, p->error_flag.p, sizeof(int32_t), hipMemcpyDeviceToHost, p->ctx->stream));
}
// A stall was seen: clear it (it is being reported now) and fail.  Synchronises.
egs_status report_stall(egs_problem *p) {
  hipStream_t s = p->ctx->stream;
  HIPCHK(hipMemsetAsync(p->error_flag.p, 0, sizeof(int32_t), s));
  HIPCHK(hipStreamSynchronize(s));
  *p->h_flag = 0;
  p->ctx->error = "device ordering wait timed out";
  return EGS_ERR_STALL;
}
// Non-blocking look at the page-locked copy: true once the copy behind a stalled solve has
// landed.  Entry points that enqueue more work call it first; entry points that have just
// synchronised see every earlier solve.
inline bool stall_seen(const egs_problem *p) { return p->h_flag && *static_cast<volatile int32_t *>(p->h_flag) != 0; }

void record_kernel_event(egs_context *ctx, bool begin) {
  if (ctx->kev.empty()) return;
  const size_t pair = ctx->kev_used % kEventPairs;
  HIPCHK(hipEventRecord(ctx->kev[2 * pair + (begin ? 0 : 1)], ctx->stream));
  if (!begin) ++ctx->kev_used;
}

void ensure_tile_plan(egs_problem *p);

template <typename REAL>
void launch_solve_t(egs_problem *p, int method, REAL cfm, REAL kscale, int sweeps, int resume) {
  egs_context *ctx = p->ctx;
  const bool quad = p->use_quad && method != EGS_JACOBI;
  p->last_static = 0;
  if (!quad) ensure_tile_plan(p);
  const bool patch = !quad && method != EGS_JACOBI && p->plan.n_patch_tiles > 0 && p->oversize != kAllGlobal;
  record_kernel_event(ctx, true);
  // oversize islands accumulate in global memory (all bodies on the all-global kernel, shared
  // bodies of patches): from zero, unless this launch continues the previous one
  if (!quad && !p->plan.global.empty() && !resume)
    HIPCHK(hipMemsetAsync(p->acc.p, 0, (size_t)(p->n > 0 ? p->n : 1) * 6 * sizeof(REAL), ctx->stream));
  if (quad || p->plan.n_tiles > 0) {
    SolveArgs<REAL> a;
    a.lanes = quad ? p->q_lanes.p : p->lanes.p;
    a.tile_nslots = quad ? p->q_tile_nslots.p : p->tile_nslots.p;
    a.tile_slot_off = quad ? p->q_tile_slot_off.p : p->tile_slot_off.p;
    a.slot_body = quad ? p->q_slot_body.p : p->slot_body.p;
    a.wsB0 = reinterpret_cast<REAL *>(p->wsB0.p); a.wsB1 = reinterpret_cast<REAL *>(p->wsB1.p);
    a.wsD = reinterpret_cast<REAL *>(p->wsD.p); a.wsInv = reinterpret_cast<REAL *>(p->wsInv.p);
    a.body0 = p->body0.p; a.body1 = p->body1.p; a.m = p->m;
    a.Minv = reinterpret_cast<const REAL *>(p->Minv_r.p);
    a.J0 = reinterpret_cast<const REAL *>(p->J0.p);
    a.J1 = reinterpret_cast<const REAL *>(p->J1.p);
    a.is_eq = p->is_eq.p;
    a.lo = reinterpret_cast<const REAL *>(p->lo.p);
    a.hi = reinterpret_cast<const REAL *>(p->hi.p);
    a.rhs = reinterpret_cast<const REAL *>(p->rhs.p);
    a.x = reinterpret_cast<REAL *>(p->x.p);
    a.acc = reinterpret_cast<REAL *>(p->acc.p);
    a.wres = reinterpret_cast<REAL *>(p->wres.p);
    a.error_flag = p->error_flag.p;
    a.cfm = cfm;
    a.kscale = kscale;
    a.sweeps = sweeps;
    a.resume = resume;
    a.max_slots = quad ? p->planq.max_slots : p->plan.max_slots;
    a.spin_limit = spin_limit();
    a.iso = (p->minv_iso && !quad && p->plan.block == 256 && iso_schedule_pays(p->m, ctx->cu_count, p->precision)) ? 1 : 0;
    a.n_bodies = p->n;
    if (p->hist_sweeps > 0) {
      a.hist_x = reinterpret_cast<REAL *>(p->hist_x.p);
      a.hist_acc = reinterpret_cast<REAL *>(p->hist_acc.p);
      // the ticket kernel's isotropic variant has no registers to spare for the snapshots; the timetable
      // kernel has an instantiation of its own for them
      if (!(method != EGS_JACOBI && !quad && use_static_timetable(p->plan, sweeps))) a.iso = 0;
    }
    {
      const char *ie = std::getenv("EGS_ISO");   // 2: force the variant wherever the bodies allow it (experiments)
      if (ie && std::atoi(ie) == 2 && p->minv_iso && !quad && p->plan.block == 256 &&
          (p->hist_sweeps == 0 || (method != EGS_JACOBI && use_static_timetable(p->plan, sweeps)))) a.iso = 1;
    }
    p->last_iso = quad ? 0 : a.iso;
    p->last_lean = 0;
    if (quad) {
      launch_cons_prepare<REAL>(a, ctx->stream);
      if (use_static_timetable(p->planq, sweeps)) {
        a.lane_level = p->q_lane_level.p; a.tile_period = p->q_tile_period.p; a.tile_depth = p->q_tile_depth.p;
        a.runs = p->planq.runs ? 1 : 0;
        launch_step_quad<REAL>(a, method, p->planq.n_tiles, p->planq.block, ctx->stream);
        p->last_static = 1;
      } else {
        launch_quad_solve<REAL>(a, method, p->planq.n_tiles, p->planq.block, ctx->stream);
        p->last_static = 0;
      }
    } else if (method != EGS_JACOBI && use_static_timetable(p->plan, sweeps)) {
      a.lane_level = p->lane_level.p; a.tile_period = p->tile_period.p; a.tile_depth = p->tile_depth.p;
      a.runs = p->plan.runs ? 1 : 0;
      bool lean = false;
      if constexpr (sizeof(REAL) == 8) {
        // the 128-VGPR form (lean_solve.hip): an experiment switch, off by default -- measured on MI355X it buys residency
        // (1024 instead of 768 constraints per CU) with a longer update and ends level with the 164-VGPR kernel
        // (DESIGN.md section 5); EGS_LEAN=1 selects it wherever its preconditions hold.  Never for the
        // snapshot-recording launches of the stopping loop.
        const char *le = std::getenv("EGS_LEAN");
        lean = le && std::atoi(le) != 0 && p->minv_iso && (p->plan.block == 256 || p->plan.block == 512) && p->lin_antisym &&
               a.hist_x == nullptr && !a.runs;
        if (lean) { a.iso = 1; launch_lean_solve(a, method, p->plan.n_tiles, p->plan.block, ctx->stream); p->last_iso = 1; }
      }
      if (!lean) launch_step_solve<REAL>(a, method, p->plan.n_tiles, p->plan.block, ctx->stream);
      p->last_static = 1;
      p->last_lean = lean ? 1 : 0;
    } else {
      launch_tile_solve<REAL>(a, method, p->plan.n_tiles, p->plan.block, ctx->stream);
      p->last_static = 0;
    }
  }
  if (patch && p->plan.block != 256)   // patch lanes are laid out for 256-thread workgroups
    throw std::logic_error("patch schedule built with a tile size other than 256");
  if (patch) {
    SolveArgs<REAL> a;
    a.lanes = p->p_lanes.p; a.tile_nslots = p->p_tile_nslots.p; a.tile_slot_off = p->p_tile_slot_off.p;
    a.slot_body = p->p_slot_body.p;
    a.wsB0 = a.wsB1 = a.wsD = a.wsInv = nullptr;
    a.body0 = p->body0.p; a.body1 = p->body1.p; a.m = p->m;
    a.Minv = reinterpret_cast<const REAL *>(p->Minv_r.p);
    a.J0 = reinterpret_cast<const REAL *>(p->J0.p); a.J1 = reinterpret_cast<const REAL *>(p->J1.p);
    a.is_eq = p->is_eq.p;
    a.lo = reinterpret_cast<const REAL *>(p->lo.p); a.hi = reinterpret_cast<const REAL *>(p->hi.p);
    a.rhs = reinterpret_cast<const REAL *>(p->rhs.p);
    a.x = reinterpret_cast<REAL *>(p->x.p); a.acc = reinterpret_cast<REAL *>(p->acc.p);
    a.wres = reinterpret_cast<REAL *>(p->wres.p);
    a.error_flag = p->error_flag.p;
    a.cfm = cfm; a.kscale = kscale; a.sweeps = sweeps; a.resume = resume;
    a.max_slots = p->plan.patch_max_slots; a.spin_limit = spin_limit();
    HIPCHK(hipMemsetAsync(p->gtickets.p, 0, sizeof(uint32_t) * (size_t)(p->n > 0 ? p->n : 1), ctx->stream));
    a.n_bodies = p->n;
    if (p->hist_sweeps > 0) {
      a.hist_x = reinterpret_cast<REAL *>(p->hist_x.p);
      a.hist_acc = reinterpret_cast<REAL *>(p->hist_acc.p);
    }
    if (p->oversize == kQuadPatches) {
      {   // hand-offs between patches as data-tagged granules (EGS_GRANULES=0: payload + flag, the round-2 protocol)
        const char *ge = std::getenv("EGS_GRANULES");
        if (!(ge && std::atoi(ge) == 0)) {
          const size_t bytes = (size_t)(p->n > 0 ? p->n : 1) * 6 * 16;
          if (p->ggran.cap < bytes) {
            p->ggran.alloc(bytes);
            HIPCHK(hipMemsetAsync(p->ggran.p, 0, bytes, ctx->stream));
            p->gran_epoch = 0;
          }
          if (++p->gran_epoch == 0) {     // the epoch wrapped: old tags could match again
            HIPCHK(hipMemsetAsync(p->ggran.p, 0, bytes, ctx->stream));
            p->gran_epoch = 1;
          }
          a.gran = p->ggran.p;
          a.gran_epoch = p->gran_epoch;
        }
      }
      if (std::getenv("EGS_TRACE_UPDATES") && sweeps > 0) {
        p->trace.alloc((size_t)sweeps * p->m);
        HIPCHK(hipMemsetAsync(p->trace.p, 0, (size_t)sweeps * p->m * sizeof(unsigned long long), ctx->stream));
        p->trace_sweeps = sweeps;
        a.trace = p->trace.p;
      }
      // 4 lanes per constraint, 1024-thread patches: the LDS hop is about half as long
      a.wsB0 = reinterpret_cast<REAL *>(p->wsB0.p); a.wsB1 = reinterpret_cast<REAL *>(p->wsB1.p);
      a.wsD = reinterpret_cast<REAL *>(p->wsD.p); a.wsInv = reinterpret_cast<REAL *>(p->wsInv.p);
      launch_cons_prepare<REAL>(a, ctx->stream);
      a.patch_runs = p->plan.patch_runs ? 1 : 0;
      launch_quad_patch_solve<REAL>(a, method, p->plan.n_patch_tiles, p->gtickets.p, ctx->stream);
    } else {
      launch_patch_solve<REAL>(a, method, p->plan.n_patch_tiles, p->gtickets.p, ctx->stream);
    }
    GlobalArgs<REAL> g{};
    g.cons = p->gcons.p; g.mg = (int)p->plan.global.size();
    g.J0 = a.J0; g.J1 = a.J1; g.rhs = a.rhs; g.x = a.x; g.acc = a.acc; g.wres = a.wres; g.cfm = cfm;
    launch_global_wres<REAL>(g, ctx->stream);
  } else if (!quad && !p->plan.global.empty()) {
    GlobalArgs<REAL> g;
    g.cons = p->gcons.p;
    g.mg = (int)p->plan.global.size();
    g.n_bodies = p->n; g.pad0 = 0; g.per_lane = 1; g.mode = 0;
    g.B0 = reinterpret_cast<REAL *>(p->gB0.p); g.B1 = reinterpret_cast<REAL *>(p->gB1.p);
    g.D = reinterpret_cast<REAL *>(p->gD.p); g.den = reinterpret_cast<REAL *>(p->gden.p);
    g.dx = reinterpret_cast<REAL *>(p->gdx.p);
    g.Minv = reinterpret_cast<const REAL *>(p->Minv_r.p);
    g.J0 = reinterpret_cast<const REAL *>(p->J0.p);
    g.J1 = reinterpret_cast<const REAL *>(p->J1.p);
    g.is_eq = p->is_eq.p;
    g.lo = reinterpret_cast<const REAL *>(p->lo.p);
    g.hi = reinterpret_cast<const REAL *>(p->hi.p);
    g.rhs = reinterpret_cast<const REAL *>(p->rhs.p);
    g.x = reinterpret_cast<REAL *>(p->x.p);
    g.acc = reinterpret_cast<REAL *>(p->acc.p);
    g.wres = reinterpret_cast<REAL *>(p->wres.p);
    g.tickets = p->gtickets.p;
    g.error_flag = p->error_flag.p;
    g.cfm = cfm;
    g.kscale = kscale;
    g.sweeps = sweeps;
    g.resume = resume;
    g.method = method;
    g.spin_limit = spin_limit();
    g.m = p->m;
    if (p->hist_sweeps > 0 && method != EGS_JACOBI) {
      g.hist_x = reinterpret_cast<REAL *>(p->hist_x.p);
      g.hist_acc = reinterpret_cast<REAL *>(p->hist_acc.p);
    }
    launch_global_solve<REAL>(g, p->global_max_blocks, ctx->stream);
  }
  record_kernel_event(ctx, false);
  HIPCHK(hipGetLastError());
}

void launch_solve(egs_problem *p, const egs_solve_params &prm, int sweeps, int resume) {
  ensure_minv_real(p);
  if (p->precision == EGS_F32) {
    const float ks = prm.method == EGS_SOR ? 1.0f / (float)prm.omega : 1.0f;
    launch_solve_t<float>(p, prm.method, (float)prm.cfm, ks, sweeps, resume);
  } else {
    const double ks = prm.method == EGS_SOR ? 1.0 / prm.omega : 1.0;
    launch_solve_t<double>(p, prm.method, prm.cfm, ks, sweeps, resume);
  }
}

void launch_residual(egs_problem *p) {
  const int rows = 3 * p->m;
  hipStream_t s = p->ctx->stream;
  if (p->precision == EGS_F32)
    launch_residual_partials<float>(rows, reinterpret_cast<const float *>(p->wres.p), reinterpret_cast<const float *>(p->x.p),
                                    reinterpret_cast<const float *>(p->lo.p), reinterpret_cast<const float *>(p->hi.p),
                                    p->is_eq.p, p->res_partials.p, kResidualBlocks, s);
  else
    launch_residual_partials<double>(rows, reinterpret_cast<const double *>(p->wres.p), reinterpret_cast<const double *>(p->x.p),
                                     reinterpret_cast<const double *>(p->lo.p), reinterpret_cast<const double *>(p->hi.p),
                                     p->is_eq.p, p->res_partials.p, kResidualBlocks, s);
}

// synchronises; returns the reference's residual metric and the error flag
double read_residual(egs_problem *p, int *err_flag) {
  double part[4 * kResidualBlocks];
  int32_t flag = 0;
  hipStream_t s = p->ctx->stream;
  HIPCHK(hipMemcpyAsync(part, p->res_partials.p, sizeof part, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(&flag, p->error_flag.p, sizeof flag, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  double sum[4] = {0, 0, 0, 0};
  for (int b = 0; b < kResidualBlocks; ++b)
    for (int k = 0; k < 4; ++k) sum[k] += part[4 * b + k];
  if (flag) {   // reported to the caller now: clear the sticky word
    HIPCHK(hipMemsetAsync(p->error_flag.p, 0, sizeof(int32_t), s));
    HIPCHK(hipStreamSynchronize(s));
    *p->h_flag = 0;
  }
  if (err_flag) *err_flag = flag;
  return std::sqrt(sum[0]) + (std::sqrt(sum[1]) + std::sqrt(sum[2]) + std::sqrt(sum[3]));
}

egs_status validate_params(egs_context *ctx, const egs_solve_params *prm) {
  if (!prm) return fail(ctx, EGS_ERR_INVALID, "params is NULL");
  if (prm->method < 0 || prm->method > 2) return fail(ctx, EGS_ERR_INVALID, "unknown method");
  if (prm->max_iters < 0) return fail(ctx, EGS_ERR_INVALID, "max_iters < 0");
  if (prm->method == EGS_SOR && !(prm->omega > 0 && prm->omega < 2))
    return fail(ctx, EGS_ERR_INVALID, "SOR needs 0 < omega < 2 (sparse_iterations.cc:15)");
  return EGS_OK;
}

void fill_stats(egs_problem *p, egs_solve_stats *st) {
  if (!p->use_quad) ensure_tile_plan(p);
  const Plan &pl = p->use_quad ? p->planq : p->plan;   // islands and ticket periods agree between the two
  st->n_islands = pl.n_islands;
  st->n_tiles = pl.n_tiles;
  st->n_global = (int32_t)pl.global.size();
  st->reserved = p->use_quad ? 1 : 0;  // 1: 4-lanes-per-constraint schedule for GS/SOR
  st->schedule = (p->use_quad ? EGS_SCHED_QUAD : 0) | (p->last_iso ? EGS_SCHED_ISO : 0) | (p->last_static ? EGS_SCHED_STATIC : 0) | (p->last_lean ? EGS_SCHED_LEAN : 0);
  if (!p->use_quad && !pl.global.empty())
    st->schedule |= p->oversize == kQuadPatches ? EGS_SCHED_QUAD_PATCHES : p->oversize == kLanePatches ? EGS_SCHED_LANE_PATCHES : EGS_SCHED_ALL_GLOBAL;
  st->tile_constraints = pl.block;
}

template <typename REAL>
void fill_history_args(egs_problem *p, SolveArgs<REAL> &a, REAL cfm) {
  a.J0 = reinterpret_cast<const REAL *>(p->J0.p); a.J1 = reinterpret_cast<const REAL *>(p->J1.p);
  a.body0 = p->body0.p; a.body1 = p->body1.p;
  a.is_eq = p->is_eq.p;
  a.lo = reinterpret_cast<const REAL *>(p->lo.p); a.hi = reinterpret_cast<const REAL *>(p->hi.p);
  a.rhs = reinterpret_cast<const REAL *>(p->rhs.p);
  a.wres = reinterpret_cast<REAL *>(p->wres.p);
  a.hist_x = reinterpret_cast<REAL *>(p->hist_x.p); a.hist_acc = reinterpret_cast<REAL *>(p->hist_acc.p);
  a.m = p->m; a.n_bodies = p->n; a.cfm = cfm;
}

// The solve driver: sparse_iterations.cc:148-226.
egs_status do_solve(egs_problem *p, const egs_solve_params *prm, egs_solve_stats *stats) {
  egs_context *ctx = p->ctx;
  if (egs_status st = validate_params(ctx, prm)) return st;
  if (!p->have_blocks) return fail(ctx, EGS_ERR_INVALID, "no system uploaded (set_blocks or assemble first)");
  if (p->m == 0) {  // sparse_iterations.cc:152-154
    p->last_iterations = 0;
    if (stats) { std::memset(stats, 0, sizeof *stats); fill_stats(p, stats); }
    return EGS_OK;
  }
  ensure_minv_real(p);   // also decides the isotropic fast path, hence the tile size
  if (!p->use_quad || prm->method == EGS_JACOBI) ensure_tile_plan(p);
  if (stall_seen(p)) return report_stall(p);   // an earlier asynchronous solve timed out
  if (!(prm->tol > 0)) {
    // tickets are 32-bit counters that advance by cnt per sweep: very long runs
    // are cut into resumed launches so they cannot wrap
    // ... and the static timetable counts time steps in a 32-bit int (t_end = depth + period x sweeps,
    // step_solve.hip / quad_solve.hip): the chunk also keeps that below INT_MAX.  64-bit arithmetic, then the clamp.
    const Plan &pl_used = p->use_quad ? p->planq : p->plan;
    const int64_t by_ticket = (int64_t)0xF0000000u / (int64_t)std::max(1, pl_used.max_cnt) - 2;
    const int64_t by_clock = ((int64_t)0x7fffffff - 2 * (int64_t)std::max(1, pl_used.max_depth) - 2) / (int64_t)std::max(1, pl_used.max_period) - 2;
    const int chunk_max = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(by_ticket, by_clock), 0x7fffffff));
    int done = 0;
    do {
      const int chunk = std::min(chunk_max, prm->max_iters - done);
      launch_solve(p, *prm, chunk, done > 0 ? 1 : 0);
      done += chunk;
    } while (done < prm->max_iters);
    p->last_iterations = prm->max_iters;
    p->residual_pending = !stats;   // nobody is looking: the reduction runs when egs_problem_get_stats asks
    if (!stats) post_flag_copy(p);  // ... and a stall shows up at the next call or the next synchronising getter
    if (stats) {
      int flag = 0;
      launch_residual(p);
      stats->residual = read_residual(p, &flag);
      stats->iterations = prm->max_iters;
      stats->status = flag ? EGS_ERR_STALL : EGS_OK;
      fill_stats(p, stats);
      if (flag) return fail(ctx, EGS_ERR_STALL, "device ordering wait timed out");
    }
    return EGS_OK;
  }
  // tol > 0: x0 = rhs, residual before iterating, then the reference's loop: one sweep,
  // one residual, stop at the first err <= tol (sparse_iterations.cc:204-221).
  const int every = prm->check_every > 0 ? prm->check_every : 1;
  int it = 0, flag = 0;
  launch_solve(p, *prm, 0, 0);
  launch_residual(p);
  // With recorded chunks (below) the residual of x0 is not waited for: its partial sums ride on the first chunk's
  // read-back and the chunk is launched at once.  Should x0 already satisfy the test (it never does in practice: the
  // reference starts from x0 = rhs) its state is simply produced again.
  const bool defer_first = prm->max_iters > 1 && !(std::getenv("EGS_DEFER_RESIDUAL") && std::atoi(std::getenv("EGS_DEFER_RESIDUAL")) == 0);
  double err = 0.0;
  // Fast form, same result: sweeps run in chunks of up to 64 per launch while the kernels
  // record x and the per-body accumulators after every sweep; one more kernel evaluates the
  // stopping test of every recorded sweep and ONE read-back per chunk finds the first sweep
  // that satisfies it.
  const bool quad = p->use_quad && prm->method != EGS_JACOBI;
  // ... and oversize islands on the patch kernels or the all-global kernel.  Only Jacobi on
  // oversize islands, which is a launch per sweep anyway, takes the plain loop below.
  if (!quad) ensure_tile_plan(p);
  const bool history = (quad || p->plan.global.empty() || prm->method != EGS_JACOBI) && prm->max_iters > 1;
  const bool deferred = history && defer_first;
  if (!deferred) err = read_residual(p, &flag);
  if (history) {
    const size_t rs = p->real_size(), m = (size_t)p->m, n = (size_t)(p->n > 0 ? p->n : 1);
    const size_t per_sweep = (3 * m + 6 * n) * rs;
    // up to 256 recorded sweeps per launch within 2 GiB of snapshots (24 C3 piles: 14 MB per sweep).  The first
    // launch records at most 64 and every further one twice as many as the one before: a solve that converges
    // early wastes little, a long one pays the timetable's fill and the read-back once per 256 sweeps
    int K = (int)std::min<size_t>(256, std::max<size_t>(1, (size_t(2048) << 20) / per_sweep));
    K = std::min(K, prm->max_iters);
    int k_cur = std::min(K, 64);
    p->hist_x.alloc((size_t)K * 3 * m * rs);
    p->hist_acc.alloc((size_t)K * 6 * n * rs);
    p->hist_out.alloc((size_t)K * kResidualBlocks * 4);
    // bodies without constraints never get a snapshot written: theirs stays zero
    HIPCHK(hipMemsetAsync(p->hist_acc.p, 0, (size_t)K * 6 * n * rs, ctx->stream));
    // read-backs land in page-locked memory: a pageable destination makes each of the two copies per launch a
    // synchronous bounce through the runtime's staging buffer
    const size_t part_len = (size_t)K * kResidualBlocks * 4, first_len = (size_t)kResidualBlocks * 4;
    if (p->h_hist_cap < part_len + 1 + first_len) {
      if (p->h_hist) { HIPCHK(hipStreamSynchronize(ctx->stream)); (void)hipHostFree(p->h_hist); p->h_hist = nullptr; p->h_hist_cap = 0; }
      HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&p->h_hist), (part_len + 1 + first_len) * sizeof(double), hipHostMallocDefault));
      p->h_hist_cap = part_len + 1 + first_len;
    }
    double *part = p->h_hist;
    int32_t *h_f32 = reinterpret_cast<int32_t *>(p->h_hist + part_len);
    double *first_part = p->h_hist + part_len + 1;
    bool first_pending = deferred;
    if (deferred) {
      HIPCHK(hipMemcpyAsync(first_part, p->res_partials.p, first_len * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      err = std::numeric_limits<double>::infinity();
    }
    while (!flag && err > prm->tol && it < prm->max_iters) {
      const int chunk = std::min(k_cur, prm->max_iters - it);
      k_cur = std::min(2 * k_cur, K);
      p->hist_sweeps = chunk;
      launch_solve(p, *prm, chunk, 1);
      p->hist_sweeps = 0;
      auto residual_pass = [&](int write_sweep) {
        if (p->precision == EGS_F32) {
          SolveArgs<float> a{};
          fill_history_args(p, a, (float)prm->cfm);
          launch_hist_residual<float>(a, chunk, kResidualBlocks, p->hist_out.p, write_sweep, ctx->stream);
        } else {
          SolveArgs<double> a{};
          fill_history_args(p, a, prm->cfm);
          launch_hist_residual<double>(a, chunk, kResidualBlocks, p->hist_out.p, write_sweep, ctx->stream);
        }
      };
      residual_pass(0);
      HIPCHK(hipMemcpyAsync(part, p->hist_out.p, (size_t)chunk * kResidualBlocks * 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipMemcpyAsync(h_f32, p->error_flag.p, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      flag = *h_f32;
      if (flag) { HIPCHK(hipMemsetAsync(p->error_flag.p, 0, sizeof(int32_t), ctx->stream)); break; }
      if (first_pending) {      // the residual of x0, read with this chunk
        first_pending = false;
        double sum0[4] = {0, 0, 0, 0};
        for (int b = 0; b < kResidualBlocks; ++b)
          for (int k = 0; k < 4; ++k) sum0[k] += first_part[4 * b + k];
        const double e0 = std::sqrt(sum0[0]) + (std::sqrt(sum0[1]) + std::sqrt(sum0[2]) + std::sqrt(sum0[3]));
        if (!(e0 > prm->tol)) {      // x0 was the answer: produce its state again (lambda, accumulators, w, partial sums)
          launch_solve(p, *prm, 0, 0);
          launch_residual(p);
          err = e0;
          break;
        }
      }
      int stop = 0;   // first recorded sweep (1-based) at which the reference would stop
      double err_stop = 0, err_last = err;
      for (int sw = 1; sw <= chunk; ++sw) {
        const bool checked = ((it + sw) % every == 0) || (it + sw == prm->max_iters);
        if (!checked) continue;
        double sum[4] = {0, 0, 0, 0};
        const double *ps = part + (size_t)(sw - 1) * kResidualBlocks * 4;
        for (int b = 0; b < kResidualBlocks; ++b)
          for (int k = 0; k < 4; ++k) sum[k] += ps[4 * b + k];
        const double e = std::sqrt(sum[0]) + (std::sqrt(sum[1]) + std::sqrt(sum[2]) + std::sqrt(sum[3]));
        err_last = e;
        if (!(e > prm->tol)) { stop = sw; err_stop = e; break; }   // as the loop condition: a NaN residual stops it too
      }
      if (stop > 0 && stop < chunk) {
        // the answer is the snapshot of sweep `stop`: lambda, accumulators, w
        const size_t off_x = (size_t)(stop - 1) * 3 * m * rs, off_a = (size_t)(stop - 1) * 6 * (size_t)p->n * rs;
        HIPCHK(hipMemcpyAsync(p->x.p, p->hist_x.p + off_x, 3 * m * rs, hipMemcpyDeviceToDevice, ctx->stream));
        if (p->n > 0)
          HIPCHK(hipMemcpyAsync(p->acc.p, p->hist_acc.p + off_a, 6 * (size_t)p->n * rs, hipMemcpyDeviceToDevice, ctx->stream));
        residual_pass(stop);
        it += stop;
        err = err_stop;
        break;
      }
      it += chunk;
      err = err_last;
      if (stop == chunk) break;   // the launch's own epilogue state is the answer
    }
    p->residual_pending = it > 0;   // res_partials still hold the sums of x0; x / wres are final
    // the snapshots are scratch of THIS call: a problem object that once ran a tolerance-terminated solve on a large
    // system must not keep up to 2 GiB of HBM (and the page-locked mirror) for the rest of its life.  Small ones stay
    // (a converging Chain re-solves every step); the stream-ordered free waits for the kernels above.
    static const size_t keep_mb = [] { const char *e = std::getenv("EGS_HIST_KEEP_MB"); return e ? (size_t)std::atol(e) : (size_t)256; }();
    if (p->hist_x.bytes() + p->hist_acc.bytes() > (keep_mb << 20)) {
      HIPCHK(hipStreamSynchronize(ctx->stream));
      p->hist_x.release(); p->hist_acc.release(); p->hist_out.release();
    }
  } else {
    while (!flag && err > prm->tol && it < prm->max_iters) {
      const int chunk = std::min(every, prm->max_iters - it);
      launch_solve(p, *prm, chunk, 1);
      launch_residual(p);
      err = read_residual(p, &flag);
      it += chunk;
    }
  }
  p->last_iterations = it;
  if (stats) {
    stats->residual = err;
    stats->iterations = it;
    stats->status = flag ? EGS_ERR_STALL : EGS_OK;
    fill_stats(p, stats);
  }
  if (flag) return fail(ctx, EGS_ERR_STALL, "device ordering wait timed out");
  return EGS_OK;
}

void ensure_wf(egs_problem *p) {
  if (p->wf_valid) return;
  p->Wf.alloc((size_t)(p->n > 0 ? p->n : 1) * 6);
  launch_mass_times_force(p->n, p->Minv_d.p, p->f_ext.p, p->Wf.p, p->ctx->stream);
  p->wf_valid = true;
}

void do_assemble(egs_problem *p, double dt, double erp) {
  AssembleArgs a;
  a.n = p->n; a.m = p->m;
  a.pos = p->pos.p; a.R = p->R.p; a.v = p->v.p; a.w = p->w.p;
  ensure_wf(p);
  a.Wf = p->Wf.p;
  a.kind = p->kind.p; a.body0 = p->body0.p; a.body1 = p->body1.p;
  a.data = p->data.p;
  a.dt = dt; a.erp = erp;
  a.J0 = p->J0.p; a.J1 = p->J1.p; a.lo = p->lo.p; a.hi = p->hi.p; a.rhs = p->rhs.p;
  a.err = p->err.p; a.is_eq = p->is_eq.p;
  if (p->precision == EGS_F32) launch_assemble<float>(a, p->ctx->stream);
  else launch_assemble<double>(a, p->ctx->stream);
  HIPCHK(hipGetLastError());
  p->have_blocks = true;
  p->lin_antisym = true;     // joints.cc:17-31 and contact.cc:66-99 build [X, ..] / [-X, ..]
}

void do_velocity(egs_problem *p, double dt) {
  ensure_wf(p);
  if (p->precision == EGS_F32)
    launch_velocity<float>(p->n, p->v.p, p->w.p, p->Wf.p, reinterpret_cast<const float *>(p->acc.p), dt, p->v6.p, p->ctx->stream);
  else
    launch_velocity<double>(p->n, p->v.p, p->w.p, p->Wf.p, reinterpret_cast<const double *>(p->acc.p), dt, p->v6.p, p->ctx->stream);
  HIPCHK(hipGetLastError());
}


// The schedule of the stand-alone products (matvec_plan.h), built on first use.
void ensure_matvec_plan(egs_problem *p) {
  if (p->mv_ready) return;
  // 128 constraints per tile: 37 KB of LDS, four workgroups per CU keep loads in flight while others
  // compute (measured on 1 M contacts: 5.2-5.4 TB/s against 4.9-5.0 with 256).  EGS_MV_TILE=256 for experiments.
  const char *te = std::getenv("EGS_MV_TILE");
  const int forced = te ? std::atoi(te) : 0;
  p->mvplan = build_matvec_plan(p->n, p->m, p->h_body0.data(), p->h_body1.data(), forced == 256 ? 256 : 128);
  const MatvecPlan &pl = p->mvplan;
  stage(p->ctx, p->mv_lanes, pl.lanes);
  stage(p->ctx, p->mv_tiles, pl.tiles);
  stage(p->ctx, p->mv_slots, pl.slots);
  stage(p->ctx, p->mv_ents, pl.ents);
  stage(p->ctx, p->mv_boundary, pl.boundary);
  const size_t rs = p->real_size(), mm = (size_t)(p->m > 0 ? p->m : 1);
  p->mv_T.alloc((size_t)(pl.n_shared_entries > 0 ? pl.n_shared_entries : 1) * 6 * rs);
  p->mv_x.alloc(mm * 3 * rs);
  p->mv_y.alloc(mm * 3 * rs);
  HIPCHK(hipStreamSynchronize(p->ctx->stream));
  p->mv_ready = true;
}

template <typename REAL>
void launch_matvec_t(egs_problem *p, int parts, REAL eps, REAL scale, const REAL *x) {
  const MatvecPlan &pl = p->mvplan;
  MatvecArgs<REAL> a;
  a.lanes = p->mv_lanes.p; a.tiles = p->mv_tiles.p; a.slots = p->mv_slots.p; a.ents = p->mv_ents.p;
  a.boundary = p->mv_boundary.p; a.n_boundary = (int32_t)pl.boundary.size();
  a.max_slots = pl.max_slots;
  a.Minv = reinterpret_cast<const REAL *>(p->Minv_r.p);
  a.J0 = reinterpret_cast<const REAL *>(p->J0.p); a.J1 = reinterpret_cast<const REAL *>(p->J1.p);
  a.x = x; a.y = reinterpret_cast<REAL *>(p->mv_y.p); a.T = reinterpret_cast<REAL *>(p->mv_T.p);
  a.eps = eps; a.scale = scale; a.accumulate = 0;
  { const char *ne = std::getenv("EGS_MV_NT"); a.stream_nt = ne ? (std::atoi(ne) != 0) : 1; }   // +2-8 % measured
  record_kernel_event(p->ctx, true);
  if (parts == EGS_MV_FULL) {
    launch_matvec<REAL>(a, 8, pl.n_tiles, pl.block, p->ctx->stream);
  } else {   // Lx + Ux, Ux + Dx, Lx + Dx as the reference adds them (sparse_iterations_utils.cc:563-569, 606-622)
    for (int bit = 1; bit <= 4; bit <<= 1) {
      if (!(parts & bit)) continue;
      launch_matvec<REAL>(a, bit, pl.n_tiles, pl.block, p->ctx->stream);
      a.accumulate = 1;
    }
  }
  record_kernel_event(p->ctx, false);
  HIPCHK(hipGetLastError());
}

egs_status do_matvec(egs_problem *p, int32_t parts, double eps, double scale, const double *x, double *y) {
  egs_context *ctx = p->ctx;
  if (!(parts == EGS_MV_FULL || (parts >= 1 && parts <= 7)))
    return fail(ctx, EGS_ERR_INVALID, "parts must be EGS_MV_FULL or a combination of LOWER / UPPER / DIAG");
  if (!p->have_blocks) return fail(ctx, EGS_ERR_INVALID, "no system uploaded (set_blocks or assemble first)");
  if (p->m == 0) return EGS_OK;
  ensure_minv_real(p);
  ensure_matvec_plan(p);
  if (x) upload_real(p, p->mv_x, x, (size_t)p->m * 3);
  // x = NULL: the device-resident lambda of the last solve
  if (p->precision == EGS_F32)
    launch_matvec_t<float>(p, parts, (float)eps, (float)scale, reinterpret_cast<const float *>(x ? p->mv_x.p : p->x.p));
  else
    launch_matvec_t<double>(p, parts, eps, scale, reinterpret_cast<const double *>(x ? p->mv_x.p : p->x.p));
  if (y) download_real(p, p->mv_y, y, (size_t)p->m * 3);
  return EGS_OK;
}

// The dense system of Ensemble::ComputeVDot (ensembles.cc:510, 513-521) from the blocks the problem holds.
egs_status build_dense_system(egs_problem *p, double cfm) {
  egs_context *ctx = p->ctx;
  if (p->precision != EGS_F64) return fail(ctx, EGS_ERR_UNSUPPORTED, "the dense path is fp64 (the reference's is)");
  if (!p->have_blocks) return fail(ctx, EGS_ERR_INVALID, "no system uploaded (set_blocks or assemble first)");
  if ((size_t)p->m * 3 > 46340) return fail(ctx, EGS_ERR_INVALID, "dense system too large (more than 2^31 entries)");
  const size_t N = (size_t)p->m * 3;
  p->dense_A.alloc(N * N > 0 ? N * N : 1);
  launch_dense_system(p->m, p->body0.p, p->body1.p, reinterpret_cast<const double *>(p->J0.p),
                      reinterpret_cast<const double *>(p->J1.p), p->Minv_d.p, cfm, p->dense_A.p, ctx->stream);
  HIPCHK(hipGetLastError());
  p->dense_cfm = cfm;
  return EGS_OK;
}

// lambda -> the accumulators a_b = M_b^-1 sum_i J_ib^T lambda_i the velocity update reads: the solve
// kernels' own list-order accumulation (a launch without sweeps builds them from x0 = rhs, so lambda
// is lent to it as the rhs).
void accumulators_from_lambda(egs_problem *p) {
  egs_solve_params prm;
  egs_default_params(&prm);
  prm.method = EGS_GAUSS_SEIDEL; prm.tol = 0.0; prm.max_iters = 0;
  hipStream_t s = p->ctx->stream;
  const size_t bytes = (size_t)p->m * 3 * p->real_size();
  p->tmp_rows.alloc(bytes > 0 ? bytes : 1);
  HIPCHK(hipMemcpyAsync(p->tmp_rows.p, p->rhs.p, bytes, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(p->rhs.p, p->x.p, bytes, hipMemcpyDeviceToDevice, s));
  launch_solve(p, prm, 0, 0);     // x = "rhs" (= lambda), acc = sum B x in list order
  HIPCHK(hipMemcpyAsync(p->rhs.p, p->tmp_rows.p, bytes, hipMemcpyDeviceToDevice, s));
}

// egs_solve_blocks / egs_matvec_blocks are stateless for their caller; the context keeps the
// last problem (schedule + device buffers) and reuses it while the constraint graph is unchanged.
egs_status oneshot_problem(egs_context *ctx, int32_t n, int32_t m, const int32_t *body0, const int32_t *body1,
                           int32_t precision, egs_problem **out) {
  egs_problem *p = ctx->oneshot;
  const bool reuse = p && p->n == n && p->m == m && p->precision == precision &&
                     (m == 0 || (std::memcmp(p->h_body0.data(), body0, (size_t)m * sizeof(int32_t)) == 0 &&
                                 std::memcmp(p->h_body1.data(), body1, (size_t)m * sizeof(int32_t)) == 0));
  if (!reuse) {
    if (p) { egs_problem_destroy(p); ctx->oneshot = nullptr; }
    egs_status st = egs_problem_create(ctx, n, m, body0, body1, precision, &p);
    if (st != EGS_OK) return st;
    ctx->oneshot = p;
  }
  *out = p;
  return EGS_OK;
}
}  // namespace

extern "C" {

void egs_default_params(egs_solve_params *p) {
  if (!p) return;
  p->method = EGS_GAUSS_SEIDEL;
  p->max_iters = 500;   // sparse_iterations.cc:19
  p->check_every = 1;
  p->reserved = 0;
  p->omega = 1.5;       // sparse_iterations.cc:15
  p->cfm = 0.0;
  p->tol = 1e-9;        // constants.h:5
}

egs_status egs_context_create(int device_index, egs_context **out) {
  if (!out) return EGS_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return EGS_ERR_NO_DEVICE;
  if (device_index < 0 || device_index >= count) return EGS_ERR_INVALID;
  egs_context *ctx = new (std::nothrow) egs_context;
  if (!ctx) return EGS_ERR_HIP;
  ctx->device = device_index;
  egs_status st = guarded(ctx, [&]() -> egs_status {
    HIPCHK(hipSetDevice(device_index));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_index));
    if (prop.multiProcessorCount > 0) ctx->cu_count = prop.multiProcessorCount;
    set_patch_workgroups(ctx->cu_count);
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&ctx->t0));
    HIPCHK(hipEventCreate(&ctx->t1));
    ctx->kev.resize(2 * kEventPairs, nullptr);
    for (auto &e : ctx->kev) HIPCHK(hipEventCreate(&e));
    return EGS_OK;
  });
  if (st != EGS_OK) {
    egs_context_destroy(ctx);
    return st;
  }
  *out = ctx;
  return EGS_OK;
}

void egs_context_destroy(egs_context *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->oneshot) { egs_problem_destroy(ctx->oneshot); ctx->oneshot = nullptr; }
  for (auto e : ctx->kev) if (e) (void)hipEventDestroy(e);
  if (ctx->t0) (void)hipEventDestroy(ctx->t0);
  if (ctx->t1) (void)hipEventDestroy(ctx->t1);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *egs_last_error(const egs_context *ctx) { return ctx ? ctx->error.c_str() : "no context"; }

egs_status egs_context_synchronize(egs_context *ctx) {
  if (!ctx) return EGS_ERR_INVALID;
  return guarded(ctx, [&]() -> egs_status { HIPCHK(hipStreamSynchronize(ctx->stream)); return EGS_OK; });
}

egs_status egs_timer_start(egs_context *ctx) {
  if (!ctx) return EGS_ERR_INVALID;
  return guarded(ctx, [&]() -> egs_status { HIPCHK(hipEventRecord(ctx->t0, ctx->stream)); return EGS_OK; });
}

egs_status egs_timer_stop(egs_context *ctx, float *elapsed_ms) {
  if (!ctx || !elapsed_ms) return EGS_ERR_INVALID;
  return guarded(ctx, [&]() -> egs_status {
    HIPCHK(hipEventRecord(ctx->t1, ctx->stream));
    HIPCHK(hipEventSynchronize(ctx->t1));
    HIPCHK(hipEventElapsedTime(elapsed_ms, ctx->t0, ctx->t1));
    return EGS_OK;
  });
}

egs_status egs_kernel_time(egs_context *ctx, double *sum_ms, int64_t *launches, int reset) {
  if (!ctx) return EGS_ERR_INVALID;
  return guarded(ctx, [&]() -> egs_status {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const size_t cnt = std::min(ctx->kev_used, kEventPairs);
    double sum = 0;
    for (size_t i = 0; i < cnt; ++i) {
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, ctx->kev[2 * i], ctx->kev[2 * i + 1]));
      sum += ms;
    }
    if (sum_ms) *sum_ms = sum;
    if (launches) *launches = (int64_t)cnt;
    if (reset) ctx->kev_used = 0;
    return EGS_OK;
  });
}

namespace {

// The 1-lane-per-constraint schedule (tile / patch / global kernels).  Built on
// demand: a problem that runs on the quad schedule only needs it for Jacobi.
void ensure_tile_plan(egs_problem *p) {
  if (p->tile_plan_ready) return;
  hipStream_t s = p->ctx->stream;
  const int n = p->n, m = p->m;
  {
    // 256 constraints per tile; 512 once there are enough tiles to give every CU two
    // anyway (all 64 lanes of the working wavefront busy: +3-4 % on 16 batched C3 piles) --
    // but not for isotropic bodies, whose register-light kernel fits THREE 256-thread tiles per CU.
    // Oversize islands (patch / global kernels) always use 256 -- unless 512 makes every island fit.
    const char *te = std::getenv("EGS_TILE");   // experiment knob: 64/128/256/512 constraints per tile
    const int forced = te ? std::atoi(te) : 0;
    int tile = (forced == 64 || forced == 128 || forced == 256 || forced == 512) ? forced : (m >= kBigTileMinConstraints && p->precision == EGS_F64 && !(p->minv_iso && iso_schedule_pays(m, p->ctx->cu_count, p->precision)) ? 512 : 256);
    p->plan = build_plan(n, m, p->h_body0.data(), p->h_body1.data(), tile);
    if (!p->plan.global.empty()) {
      if (tile == 256 && !forced) {   // islands of 257..512 constraints: one 512-thread workgroup, all hand-offs in LDS
        Plan big = build_plan(n, m, p->h_body0.data(), p->h_body1.data(), 512);
        if (big.global.empty()) p->plan = std::move(big);
      } else if (tile != 256) {       // the patch kernels are 256-constraint workgroups, forced size or not
        p->plan = build_plan(n, m, p->h_body0.data(), p->h_body1.data(), 256);
      }
    }
  }
  const Plan &pl = p->plan;
  stage(p->ctx, p->lanes, pl.lanes);
  stage(p->ctx, p->lane_level, pl.lane_level);
  stage(p->ctx, p->tile_period, pl.tile_period);
  stage(p->ctx, p->tile_depth, pl.tile_depth);
  stage(p->ctx, p->tile_nslots, pl.tile_nslots);
  stage(p->ctx, p->tile_slot_off, pl.tile_slot_off);
  stage(p->ctx, p->slot_body, pl.slot_body);
  stage(p->ctx, p->gcons, pl.global);
  if (pl.n_patch_tiles > 0) {
    stage(p->ctx, p->p_lanes, pl.patch_lanes);
    stage(p->ctx, p->p_tile_nslots, pl.patch_tile_nslots);
    stage(p->ctx, p->p_tile_slot_off, pl.patch_tile_slot_off);
    stage(p->ctx, p->p_slot_body, pl.patch_slot_body);
  }
  {
    const char *pe = std::getenv("EGS_PATCH"), *qp = std::getenv("EGS_QUAD_PATCH");
    const bool f32 = p->precision == EGS_F32;
    const size_t lds = (size_t)pl.patch_max_slots * (6 * p->real_size() + sizeof(unsigned));
    const int occ_quad = pl.n_patch_tiles > 0 ? (f32 ? occupancy_quad_patch_solve<float>(lds) : occupancy_quad_patch_solve<double>(lds)) : 0;
    const int occ_lane = pl.n_patch_tiles > 0 ? (f32 ? occupancy_patch_solve<float>(lds) : occupancy_patch_solve<double>(lds)) : 0;
    const int occ_glob = pl.global.empty() ? 1 : (f32 ? occupancy_global_solve<float>() : occupancy_global_solve<double>());
    p->oversize = choose_oversize_schedule(pl.n_patch_tiles, occ_quad, occ_lane, p->ctx->cu_count, !(pe && std::atoi(pe) == 0),
                                           pl.block == 256 && !(qp && std::atoi(qp) == 0));
    // (a patch plan with runs is for the 4-lane kernel only: plan.cpp builds it when that kernel will take it; should it
    //  not -- no LDS left for a resident workgroup -- the island goes the all-global way rather than to a kernel that
    //  does not know the placeholders)
    if (pl.patch_runs && p->oversize == kLanePatches) p->oversize = kAllGlobal;
    p->global_max_blocks = std::max(1, std::min(occ_glob, 1) * p->ctx->cu_count);
    if (p->oversize == kQuadPatches) {
      const size_t rsz = p->real_size(), mm2 = (size_t)m;
      p->wsB0.alloc(mm2 * 18 * rsz); p->wsB1.alloc(mm2 * 18 * rsz); p->wsD.alloc(mm2 * 9 * rsz); p->wsInv.alloc(mm2 * 3 * rsz);
    }
  }
  const size_t mg = pl.global.size(), rsz = p->real_size();
  p->gB0.alloc(mg * 18 * rsz); p->gB1.alloc(mg * 18 * rsz); p->gD.alloc(mg * 9 * rsz);
  p->gden.alloc(mg * 3 * rsz); p->gdx.alloc(mg * 3 * rsz);
  HIPCHK(hipStreamSynchronize(s));
  p->tile_plan_ready = true;
}

// (Re)build everything that depends on the constraint topology.  Body state
// (pos, R, v, w, M^-1, f_ext) is kept when n is unchanged; buffers only grow.
void problem_set_topology(egs_problem *p, int32_t m, const int32_t *body0, const int32_t *body1, bool fresh = true) {
  egs_context *ctx = p->ctx;
  const int n = p->n;
  hipStream_t s = ctx->stream;
  const auto t_top0 = std::chrono::steady_clock::now();
  HIPCHK(hipStreamSynchronize(s));   // nothing may still read the pinned arena
  ctx->pinned.reset();
  p->m = m;
  p->h_body0.assign(body0, body0 + m);
  p->h_body1.assign(body1, body1 + m);
  p->tile_plan_ready = false;
  p->mv_ready = false;
  p->dense_cfm = -1.0;
  p->h_rows_valid = false;
  p->use_quad = false;
  p->have_blocks = false;
  p->have_constraints = false;
  p->last_iterations = 0;
  {  // quad schedule: small problems whose islands all fit 64-constraint tiles
    const char *env = std::getenv("EGS_QUAD");
    const int force = env ? std::atoi(env) : -1;
    if (m > 0 && force != 0 && (force == 1 || (long)m <= 64L * quad_tiles_per_cu_max(p->precision) * p->ctx->cu_count)) {
      const char *qe = std::getenv("EGS_QUAD_TILE");   // experiment knob: force 64 or 256
      const int qt = qe ? std::atoi(qe) : 0;
      // 64-constraint tiles when every island fits, else 1024-thread tiles of 256
      // runs (plan.h) while there is about one tile per CU
      const auto t_pl0 = std::chrono::steady_clock::now();
      p->planq = build_plan(n, m, body0, body1, (qt == 64 || qt == 128 || qt == 256) ? qt : kAutoQuadBlock, &p->planq,
                            p->ctx->cu_count);
      if (std::getenv("EGS_PLAN_TRACE"))
        std::fprintf(stderr, "plan trace: build_plan(quad) %.1f us for %d constraints\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_pl0).count(), m);
      bool one_round = true;
      if (force != 1 && p->planq.global.empty()) {
        const size_t qlds = (size_t)p->planq.max_slots * 6 * p->real_size();
        const int occ = p->precision == EGS_F32 ? occupancy_step_quad<float>(p->planq.block, qlds) : occupancy_step_quad<double>(p->planq.block, qlds);
        one_round = (long)p->planq.n_tiles <= (long)occ * p->ctx->cu_count;
      }
      if (p->planq.global.empty() && one_round) {
        const Plan &pq = p->planq;
        p->use_quad = true;
        stage(p->ctx, p->q_lanes, pq.lanes);
        stage(p->ctx, p->q_lane_level, pq.lane_level);
        stage(p->ctx, p->q_tile_period, pq.tile_period);
        stage(p->ctx, p->q_tile_depth, pq.tile_depth);
        stage(p->ctx, p->q_tile_nslots, pq.tile_nslots);
        stage(p->ctx, p->q_tile_slot_off, pq.tile_slot_off);
        stage(p->ctx, p->q_slot_body, pq.slot_body);
        const size_t rsz = p->real_size(), mm2 = (size_t)m;
        p->wsB0.alloc(mm2 * 18 * rsz); p->wsB1.alloc(mm2 * 18 * rsz); p->wsD.alloc(mm2 * 9 * rsz); p->wsInv.alloc(mm2 * 3 * rsz);
      }
    }
  }
  stage(p->ctx, p->body0, p->h_body0);
  stage(p->ctx, p->body1, p->h_body1);
  const size_t rs = p->real_size();
  const size_t nn = (size_t)(n > 0 ? n : 1), mm = (size_t)(m > 0 ? m : 1);
  p->kind.alloc(mm); p->data.alloc(mm * 7);
  p->err.alloc(mm * 3);
  p->J0.alloc(mm * 18 * rs); p->J1.alloc(mm * 18 * rs);
  p->lo.alloc(mm * 3 * rs); p->hi.alloc(mm * 3 * rs); p->rhs.alloc(mm * 3 * rs);
  p->x.alloc(mm * 3 * rs); p->wres.alloc(mm * 3 * rs);
  p->is_eq.alloc(mm * 3);
  HIPCHK(hipMemsetAsync(p->acc.p, 0, nn * 6 * rs, s));
  if (fresh) {   // a re-planned world overwrites both in its next solve; a new problem reads as zeros
    HIPCHK(hipMemsetAsync(p->x.p, 0, mm * 3 * rs, s));
    HIPCHK(hipMemsetAsync(p->wres.p, 0, mm * 3 * rs, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  if (std::getenv("EGS_PLAN_TRACE"))
    std::fprintf(stderr, "plan trace: problem_set_topology %.1f us in all\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_top0).count());
  // the 1-lane schedule is built at the first solve that needs it (ensure_tile_plan):
  // its tile size depends on the mass blocks, which arrive after the topology
}

egs_status check_topology(egs_context *ctx, int32_t n, int32_t m, const int32_t *body0, const int32_t *body1) {
  if (n < 0 || m < 0 || (m > 0 && (!body0 || !body1))) return fail(ctx, EGS_ERR_INVALID, "bad sizes / NULL topology");
  for (int i = 0; i < m; ++i) {
    // (the schedule builder checks the range too, but the 1-lane schedule is built lazily)
    if (body0[i] < -1 || body0[i] >= n || body1[i] < -1 || body1[i] >= n)
      return fail(ctx, EGS_ERR_INVALID, "body index out of range");
    if (body0[i] >= 0 && body0[i] == body1[i])
      return fail(ctx, EGS_ERR_INVALID, "constraint with the same body on both sides");
  }
  return EGS_OK;
}

}  // namespace

egs_status egs_problem_create(egs_context *ctx, int32_t n, int32_t m, const int32_t *body0,
                              const int32_t *body1, int32_t precision, egs_problem **out) {
  if (!ctx || !out) return EGS_ERR_INVALID;
  *out = nullptr;
  if (egs_status st = check_topology(ctx, n, m, body0, body1)) return st;
  if (precision != EGS_F64 && precision != EGS_F32) return fail(ctx, EGS_ERR_INVALID, "unknown precision");
  egs_problem *p = new (std::nothrow) egs_problem;
  if (!p) return fail(ctx, EGS_ERR_HIP, "host allocation failed");
  p->ctx = ctx; p->n = n; p->m = m; p->precision = precision;
  egs_status st = guarded(ctx, [&]() -> egs_status {
    HIPCHK(hipSetDevice(ctx->device));
    const size_t rs = p->real_size(), nn = (size_t)(n > 0 ? n : 1);
    p->gtickets.alloc(nn);
    p->pos.alloc(nn * 3); p->R.alloc(nn * 9); p->v.alloc(nn * 3); p->w.alloc(nn * 3);
    p->Minv_d.alloc(nn * 36); p->f_ext.alloc(nn * 6); p->v6.alloc(nn * 6);
    p->res_partials.alloc(4 * kResidualBlocks);
    p->Minv_r.alloc(nn * 36 * rs);
    p->acc.alloc(nn * 6 * rs);
    p->error_flag.alloc(2);
    HIPCHK(hipMemsetAsync(p->error_flag.p, 0, 2 * sizeof(int32_t), ctx->stream));
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&p->h_flag), 64, hipHostMallocDefault));
    *p->h_flag = 0;
    problem_set_topology(p, m, body0, body1);
    return EGS_OK;
  });
  if (st != EGS_OK) { if (p->h_flag) (void)hipHostFree(p->h_flag); delete p; return st; }
  *out = p;
  return EGS_OK;
}

egs_status egs_problem_create_batch(egs_context *ctx, int32_t n_ensembles, const int32_t *n_bodies,
                                    const int32_t *n_constraints, const int32_t *body0, const int32_t *body1,
                                    int32_t precision, egs_problem **out, int32_t *body_offset,
                                    int32_t *constraint_offset) {
  if (!ctx || !out) return EGS_ERR_INVALID;
  *out = nullptr;
  if (n_ensembles < 0 || (n_ensembles > 0 && (!n_bodies || !n_constraints)))
    return fail(ctx, EGS_ERR_INVALID, "bad ensemble count / NULL size tables");
  long nb = 0, nc = 0;
  for (int e = 0; e < n_ensembles; ++e) {
    if (n_bodies[e] < 0 || n_constraints[e] < 0) return fail(ctx, EGS_ERR_INVALID, "negative ensemble size");
    nb += n_bodies[e]; nc += n_constraints[e];
  }
  if (nb > INT32_MAX || nc > INT32_MAX) return fail(ctx, EGS_ERR_INVALID, "batch too large for 32-bit indices");
  if (nc > 0 && (!body0 || !body1)) return fail(ctx, EGS_ERR_INVALID, "NULL topology");
  std::vector<int32_t> g0, g1;
  try {
    g0.resize((size_t)nc); g1.resize((size_t)nc);
  } catch (const std::bad_alloc &) {
    return fail(ctx, EGS_ERR_HIP, "host allocation failed");
  }
  long bo = 0, co = 0;
  for (int e = 0; e < n_ensembles; ++e) {
    if (body_offset) body_offset[e] = (int32_t)bo;
    if (constraint_offset) constraint_offset[e] = (int32_t)co;
    for (int i = 0; i < n_constraints[e]; ++i) {
      const int32_t a = body0[co + i], b = body1[co + i];
      if (a < -1 || a >= n_bodies[e] || b < -1 || b >= n_bodies[e])
        return fail(ctx, EGS_ERR_INVALID, "ensemble-local body index out of range");
      g0[(size_t)(co + i)] = a < 0 ? -1 : (int32_t)(bo + a);
      g1[(size_t)(co + i)] = b < 0 ? -1 : (int32_t)(bo + b);
    }
    bo += n_bodies[e]; co += n_constraints[e];
  }
  if (body_offset) body_offset[n_ensembles] = (int32_t)bo;
  if (constraint_offset) constraint_offset[n_ensembles] = (int32_t)co;
  return egs_problem_create(ctx, (int32_t)nb, (int32_t)nc, g0.data(), g1.data(), precision, out);
}

void egs_problem_destroy(egs_problem *p) {
  if (!p) return;
  if (p->ctx && p->ctx->stream) (void)hipStreamSynchronize(p->ctx->stream);
  if (p->h_flag) (void)hipHostFree(p->h_flag);
  if (p->h_hist) (void)hipHostFree(p->h_hist);
  delete p;
}

egs_status egs_problem_set_blocks(egs_problem *p, const double *Minv, const double *J0, const double *J1,
                                  const uint8_t *is_eq, const double *lo, const double *hi, const double *rhs) {
  if (!p) return EGS_ERR_INVALID;
  return guarded(p->ctx, [&]() -> egs_status {
    const size_t n = p->n, m = p->m;
    if (Minv && n) { upload(p->Minv_d, Minv, n * 36, p->ctx->stream); p->minv_r_valid = false; p->wf_valid = false; }
    upload_real(p, p->J0, J0, m * 18);
    upload_real(p, p->J1, J1, m * 18);
    if (J0 || J1) {
      // lean_step_kernel keeps ONE linear block per constraint: allowed only if J1_lin = -J0_lin wherever both sides exist
      bool anti = J0 && J1 && p->precision == EGS_F64 && p->h_body0.size() == m && p->h_body1.size() == m;
      for (size_t i = 0; anti && i < m; ++i) {
        if (p->h_body0[i] < 0 || p->h_body1[i] < 0) continue;
        for (int r = 0; r < 3 && anti; ++r)
          for (int k = 0; k < 3; ++k)
            if (!(J0[i * 18 + 6 * r + k] == -J1[i * 18 + 6 * r + k])) { anti = false; break; }
      }
      p->lin_antisym = anti;
    }
    if (is_eq && m) upload(p->is_eq, is_eq, m * 3, p->ctx->stream);
    upload_real(p, p->lo, lo, m * 3);
    upload_real(p, p->hi, hi, m * 3);
    upload_real(p, p->rhs, rhs, m * 3);
    p->have_blocks = true;
    return EGS_OK;
  });
}

egs_status egs_problem_solve(egs_problem *p, const egs_solve_params *params, egs_solve_stats *stats) {
  if (!p) return EGS_ERR_INVALID;
  return guarded(p->ctx, [&]() -> egs_status { return do_solve(p, params, stats); });
}

egs_status egs_problem_get_lambda(egs_problem *p, double *x) {
  if (!p || !x) return EGS_ERR_INVALID;
  return guarded(p->ctx, [&]() -> egs_status {
    download_real(p, p->x, x, (size_t)p->m * 3);
    return stall_seen(p) ? report_stall(p) : EGS_OK;   // the download synchronised: every earlier solve is accounted for
  });
}

egs_status egs_problem_get_accumulators(egs_problem *p, double *a) {
  if (!p || !a) return EGS_ERR_INVALID;
  return guarded(p->ctx, [&]() -> egs_status {
    download_real(p, p->acc, a, (size_t)p->n * 6);
    return stall_seen(p) ? report_stall(p) : EGS_OK;
  });
}

egs_status egs_problem_set_state(egs_problem *p, const double *pos, const double *R, const double *v,
                                 const double *w, const double *Minv, const double *f_ext) {
  if (!p) return EGS_ERR_INVALID;
  return guarded(p->ctx, [&]() -> egs_status {
    const size_t n = p->n;
    hipStream_t s = p->ctx->stream;
    if (pos) upload(p->pos, pos, n * 3, s);
    if (R) upload(p->R, R, n * 9, s);
    if (v) upload(p->v, v, n * 3, s);
    if (w) upload(p->w, w, n * 3, s);
    if (Minv) { upload(p->Minv_d, Minv, n * 36, s); p->minv_r_valid = false; p->wf_valid = false; }
    if (f_ext) { upload(p->f_ext, f_ext, n * 6, s); p->wf_valid = false; }
    p->have_state = true;
    return EGS_OK;
  });
}

egs_status egs_problem_set_mass(egs_problem *p, const double *inv_mass, const double *inv_inertia) {
  if (!p) return EGS_ERR_INVALID;
  if (p->n > 0 && (!inv_mass || !inv_inertia)) return fail(p->ctx, EGS_ERR_INVALID, "NULL mass arrays");
  return guarded(p->ctx, [&]() -> egs_status {
    const size_t n = (size_t)p->n;
    std::vector<double> blocks(n * 36, 0.0);
    for (size_t b = 0; b < n; ++b) {
      double *W = blocks.data() + b * 36;
      for (int k = 0; k < 3; ++k) W[7 * k] = inv_mass[b];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) W[6 * (3 + r) + 3 + c] = inv_inertia[b * 9 + 3 * r + c];
    }
    if (n) { upload(p->Minv_d, blocks.data(), n * 36, p->ctx->stream); p->minv_r_valid = false; p->wf_valid = false; }
    return EGS_OK;
  });
}

egs_status egs_problem_set_constraints(egs_problem *p, const int32_t *kind, const double *data) {
  if (!p) return EGS_ERR_INVALID;
  if (p->m > 0 && (!kind || !data)) return fail(p->ctx, EGS_ERR_INVALID, "NULL constraint descriptors");
  for (int i = 0; i < p->m; ++i)
    if (kind[i] != EGS_JOINT_BALL && kind[i] != EGS_CONTACT_BOX)
      return fail(p->ctx, EGS_ERR_INVALID, "unknown constraint kind");
  return guarded(p->ctx, [&]() -> egs_status {
    upload(p->kind, kind, (size_t)p->m, p->ctx->stream);
    upload(p->data, data, (size_t)p->m * 7, p->ctx->stream);
    p->have_constraints = true;
    p->h_rows_valid = false;
    return EGS_OK;
  });
}

egs_status egs_problem_assemble(egs_problem *p, double dt, double erp) {
  if (!p) return EGS_ERR_INVALID;
  if (!p->have_state || !p->have_constraints) return fail(p->ctx, EGS_ERR_INVALID, "set_state and set_constraints first");
  if (!(dt > 0)) return fail(p->ctx, EGS_ERR_INVALID, "dt must be > 0");
  return guarded(p->ctx, [&]() -> egs_status { do_assemble(p, dt, erp); return EGS_OK; });
}

egs_status egs_problem_step(egs_problem *p, double dt, double erp, const egs_solve_params *params,
                            egs_solve_stats *stats) {
  if (!p) return EGS_ERR_INVALID;
  if (!p->have_state || !p->have_constraints) return fail(p->ctx, EGS_ERR_INVALID, "set_state and set_constraints first");
  if (!(dt > 0)) return fail(p->ctx, EGS_ERR_INVALID, "dt must be > 0");
  return guarded(p->ctx, [&]() -> egs_status {
    if (stall_seen(p)) return report_stall(p);   // an earlier (asynchronous) step timed out
    do_assemble(p, dt, erp);
    egs_status st = do_solve(p, params, stats);
    if (st != EGS_OK) return st;
    do_velocity(p, dt);
    return EGS_OK;
  });
}

egs_status egs_problem_get_blocks(egs_problem *p, double *J0, double *J1, uint8_t *is_eq, double *lo,
                                  double *hi, double *rhs, double *err) {
  if (!p) return EGS_ERR_INVALID;
  return guarded(p->ctx, [&]() -> egs_status {
    const size_t m = p->m;
    download_real(p, p->J0, J0, m * 18);
    download_real(p, p->J1, J1, m * 18);
    download_real(p, p->lo, lo, m * 3);
    download_real(p, p->hi, hi, m * 3);
    download_real(p, p->rhs, rhs, m * 3);
    hipStream_t s = p->ctx->stream;
    if (is_eq && m) HIPCHK(hipMemcpyAsync(is_eq, p->is_eq.p, m * 3, hipMemcpyDeviceToHost, s));
    if (err && m) HIPCHK(hipMemcpyAsync(err, p->err.p, m * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return EGS_OK;
  });
}

egs_status egs_problem_get_velocity(egs_problem *p, double *v6) {
  if (!p || !v6) return EGS_ERR_INVALID;
  return guarded(p->ctx, [&]() -> egs_status {
    if (p->n) HIPCHK(hipMemcpyAsync(v6, p->v6.p, (size_t)p->n * 6 * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    return stall_seen(p) ? report_stall(p) : EGS_OK;
  });
}

egs_status egs_problem_advance(egs_problem *p, double dt) {
  if (!p) return EGS_ERR_INVALID;
  if (!p->have_state) return fail(p->ctx, EGS_ERR_INVALID, "set_state and step first");
  return guarded(p->ctx, [&]() -> egs_status {
    if (stall_seen(p)) return report_stall(p);   // do not integrate a lambda that came out of a timed-out wait
    launch_advance(p->n, p->pos.p, p->R.p, p->v.p, p->w.p, p->v6.p, dt, p->ctx->stream);
    HIPCHK(hipGetLastError());
    return EGS_OK;
  });
}

egs_status egs_problem_get_state(egs_problem *p, double *pos, double *R, double *v, double *w) {
  if (!p) return EGS_ERR_INVALID;
  return guarded(p->ctx, [&]() -> egs_status {
    const size_t n = p->n;
    hipStream_t s = p->ctx->stream;
    if (pos && n) HIPCHK(hipMemcpyAsync(pos, p->pos.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (R && n) HIPCHK(hipMemcpyAsync(R, p->R.p, n * 9 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (v && n) HIPCHK(hipMemcpyAsync(v, p->v.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (w && n) HIPCHK(hipMemcpyAsync(w, p->w.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return stall_seen(p) ? report_stall(p) : EGS_OK;
  });
}

egs_status egs_problem_get_stats(egs_problem *p, egs_solve_stats *stats) {
  if (!p || !stats) return EGS_ERR_INVALID;
  return guarded(p->ctx, [&]() -> egs_status {
    std::memset(stats, 0, sizeof *stats);
    fill_stats(p, stats);
    stats->iterations = p->last_iterations;
    if (p->m == 0) return EGS_OK;
    int flag = 0;
    if (p->residual_pending) { launch_residual(p); p->residual_pending = false; }
    stats->residual = read_residual(p, &flag);
    stats->status = flag ? EGS_ERR_STALL : EGS_OK;
    return flag ? fail(p->ctx, EGS_ERR_STALL, "device ordering wait timed out") : EGS_OK;
  });
}

egs_status egs_solve_blocks(egs_context *ctx, int32_t n, const double *Minv, int32_t m, const int32_t *body0,
                            const int32_t *body1, const double *J0, const double *J1, const uint8_t *is_eq,
                            const double *lo, const double *hi, const double *rhs, const egs_solve_params *params,
                            int32_t precision, double *x, egs_solve_stats *stats) {
  if (!ctx) return EGS_ERR_INVALID;
  if (m > 0 && (!Minv || !body0 || !body1 || !J0 || !J1 || !is_eq || !lo || !hi || !rhs || !x))
    return fail(ctx, EGS_ERR_INVALID, "NULL array");
  egs_problem *p = nullptr;
  egs_status st = oneshot_problem(ctx, n, m, body0, body1, precision, &p);
  if (st != EGS_OK) return st;
  st = egs_problem_set_blocks(p, Minv, J0, J1, is_eq, lo, hi, rhs);
  egs_solve_stats local;
  if (st == EGS_OK) st = egs_problem_solve(p, params, stats ? stats : &local);
  if (st == EGS_OK && m > 0) st = egs_problem_get_lambda(p, x);
  return st;
}

egs_status egs_problem_matvec(egs_problem *p, int32_t parts, double eps, double scale, const double *x, double *y) {
  if (!p) return EGS_ERR_INVALID;
  return guarded(p->ctx, [&]() -> egs_status { return do_matvec(p, parts, eps, scale, x, y); });
}

egs_status egs_problem_get_matvec(egs_problem *p, double *y) {
  if (!p || !y) return EGS_ERR_INVALID;
  return guarded(p->ctx, [&]() -> egs_status {
    if (!p->mv_ready) return fail(p->ctx, EGS_ERR_INVALID, "egs_problem_matvec first");
    download_real(p, p->mv_y, y, (size_t)p->m * 3);
    return EGS_OK;
  });
}

egs_status egs_problem_get_wres(egs_problem *p, double *w) {
  if (!p || !w) return EGS_ERR_INVALID;
  return guarded(p->ctx, [&]() -> egs_status {
    download_real(p, p->wres, w, (size_t)p->m * 3);
    return stall_seen(p) ? report_stall(p) : EGS_OK;
  });
}

egs_status egs_matvec_blocks(egs_context *ctx, int32_t n, const double *Minv, int32_t m, const int32_t *body0,
                             const int32_t *body1, const double *J0, const double *J1, int32_t parts, double eps,
                             double scale, int32_t precision, const double *x, double *y) {
  if (!ctx) return EGS_ERR_INVALID;
  if (m > 0 && (!Minv || !body0 || !body1 || !J0 || !J1 || !x || !y)) return fail(ctx, EGS_ERR_INVALID, "NULL array");
  egs_problem *p = nullptr;
  egs_status st = oneshot_problem(ctx, n, m, body0, body1, precision, &p);
  if (st != EGS_OK) return st;
  st = egs_problem_set_blocks(p, Minv, J0, J1, nullptr, nullptr, nullptr, nullptr);
  if (st == EGS_OK) st = egs_problem_matvec(p, parts, eps, scale, x, y);
  return st;
}

int32_t egs_debug_choose_oversize_schedule(int32_t n_patch_tiles, int32_t quad_per_cu, int32_t patch_per_cu,
                                           int32_t cu_count, int32_t patches_enabled, int32_t quad_patches_enabled) {
  return (int32_t)choose_oversize_schedule(n_patch_tiles, quad_per_cu, patch_per_cu, cu_count, patches_enabled != 0,
                                           quad_patches_enabled != 0);
}

egs_status egs_debug_matvec_plan(int32_t n, int32_t m, const int32_t *body0, const int32_t *body1, int32_t tile_size,
                                 int32_t *n_tiles, int32_t *n_islands, int32_t *n_shared_bodies, int32_t *n_boundary,
                                 int32_t *cons_tile, int32_t *cons_lane) {
  if (n < 0 || m < 0 || (m > 0 && (!body0 || !body1))) return EGS_ERR_INVALID;
  try {
    const MatvecPlan pl = build_matvec_plan(n, m, body0, body1, tile_size);
    if (n_tiles) *n_tiles = pl.n_tiles;
    if (n_islands) *n_islands = pl.n_islands;
    if (n_shared_bodies) *n_shared_bodies = pl.n_shared_bodies;
    if (n_boundary) *n_boundary = (int32_t)pl.boundary.size();
    for (int t = 0; t < pl.n_tiles; ++t)
      for (int l = 0; l < pl.block; ++l) {
        const MvLane &d = pl.lanes[(size_t)t * pl.block + l];
        if (d.cidx < 0) continue;
        if (cons_tile) cons_tile[d.cidx] = t;
        if (cons_lane) cons_lane[d.cidx] = l;
      }
    return EGS_OK;
  } catch (const std::exception &) {
    return EGS_ERR_INVALID;
  }
}

egs_status egs_problem_dense_system(egs_problem *p, double cfm, double *A) {
  if (!p) return EGS_ERR_INVALID;
  return guarded(p->ctx, [&]() -> egs_status {
    if (egs_status st = build_dense_system(p, cfm)) return st;
    const size_t N = (size_t)p->m * 3;
    if (A && N) {
      HIPCHK(hipMemcpyAsync(A, p->dense_A.p, N * N * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
      HIPCHK(hipStreamSynchronize(p->ctx->stream));
    }
    return EGS_OK;
  });
}

egs_status egs_problem_dense_condition(egs_problem *p, double cfm, double *estimate) {
  if (!p || !estimate) return EGS_ERR_INVALID;
  return guarded(p->ctx, [&]() -> egs_status {
    if (egs_status st = build_dense_system(p, cfm)) return st;
    bool spd = true;
    *estimate = dense_condition_estimate(p->ctx->stream, 3 * p->m, p->dense_A.p, &spd);
    return EGS_OK;   // not positive definite: +inf, i.e. "ill-conditioned" to the caller (ensembles.cc:514)
  });
}

egs_status egs_dense_iterate(egs_context *ctx, int32_t N, const double *A, const double *b, const uint8_t *C, const double *lo,
                             const double *hi, const egs_solve_params *params, double *x, egs_solve_stats *stats) {
  if (!ctx) return EGS_ERR_INVALID;
  if (egs_status st = validate_params(ctx, params)) return st;
  if (N < 0 || (N > 0 && (!A || !b || !x))) return fail(ctx, EGS_ERR_INVALID, "NULL array");
  if (N > 0 && (C || lo || hi) && !(C && lo && hi)) return fail(ctx, EGS_ERR_INVALID, "C, lo and hi come together (or all NULL: every row an equality)");
  return guarded(ctx, [&]() -> egs_status {
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<uint8_t> all_eq;
    std::vector<double> zeros;
    if (!C) { all_eq.assign((size_t)std::max(N, 1), 1); zeros.assign((size_t)std::max(N, 1), 0.0); }   // sparse_iterations.cc:229-233
    int it = 0;
    double res = 0.0;
    dense_iterate(ctx->stream, N, A, b, C ? C : all_eq.data(), lo ? lo : zeros.data(), hi ? hi : zeros.data(), params->method,
                  params->omega, params->max_iters, params->tol, x, &it, &res);
    if (stats) {
      std::memset(stats, 0, sizeof *stats);
      stats->iterations = it;
      stats->residual = res;
      stats->status = EGS_OK;
    }
    return EGS_OK;
  });
}

egs_status egs_dense_condition(egs_context *ctx, int32_t N, const double *A, double *estimate, double *pivot_bound) {
  if (!ctx) return EGS_ERR_INVALID;
  if (N < 0 || !estimate || (N > 0 && !A)) return fail(ctx, EGS_ERR_INVALID, "NULL array");
  return guarded(ctx, [&]() -> egs_status {
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf<double> dA;
    dA.alloc((size_t)N * N);
    if (N) HIPCHK(hipMemcpyAsync(dA.p, A, (size_t)N * N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    bool spd = true;
    double pb = 1.0;
    *estimate = dense_condition_estimate(ctx->stream, N, dA.p, &spd, &pb);
    if (pivot_bound) *pivot_bound = spd ? pb : *estimate;
    return EGS_OK;
  });
}

egs_status egs_problem_step_dense(egs_problem *p, double dt, double erp, double cfm, int32_t use_bounds, int32_t *ok,
                                  int32_t *pivots) {
  if (!p) return EGS_ERR_INVALID;
  if (!p->have_state || !p->have_constraints) return fail(p->ctx, EGS_ERR_INVALID, "set_state and set_constraints first");
  if (!(dt > 0)) return fail(p->ctx, EGS_ERR_INVALID, "dt must be > 0");
  if (ok) *ok = 0;
  return guarded(p->ctx, [&]() -> egs_status {
    egs_context *ctx = p->ctx;
    hipStream_t s = ctx->stream;
    if (stall_seen(p)) return report_stall(p);
    do_assemble(p, dt, erp);                                  // J, err, bounds, rhs (ensembles.cc:565-570)
    if (p->m == 0) {                                           // v_dot = M^-1 f (ensembles.cc:504-505)
      HIPCHK(hipMemsetAsync(p->acc.p, 0, (size_t)(p->n > 0 ? p->n : 1) * 6 * p->real_size(), s));
      do_velocity(p, dt);
      if (ok) *ok = 1;
      return EGS_OK;
    }
    if (egs_status st = build_dense_system(p, cfm)) return st;   // ensembles.cc:510, 513-521
    const size_t rows = (size_t)p->m * 3;
    if (!p->h_rows_valid || p->h_rows_eq.size() != rows) {
      p->h_rows_eq.resize(rows); p->h_rows_lo.resize(rows); p->h_rows_hi.resize(rows);
      HIPCHK(hipMemcpyAsync(p->h_rows_eq.data(), p->is_eq.p, rows, hipMemcpyDeviceToHost, s));
      HIPCHK(hipMemcpyAsync(p->h_rows_lo.data(), p->lo.p, rows * sizeof(double), hipMemcpyDeviceToHost, s));
      HIPCHK(hipMemcpyAsync(p->h_rows_hi.data(), p->hi.p, rows * sizeof(double), hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      p->h_rows_valid = true;
    }
    const std::vector<uint8_t> &C = p->h_rows_eq;
    const std::vector<double> &lo = p->h_rows_lo, &hi = p->h_rows_hi;
    int piv = 0;
    std::string msg;
    // Lcp::MixedConstraintsSolver (ensembles.cc:531) on the device matrix; lambda lands in the problem's x
    const bool good = dense_mixed_constraints_device(s, (int)rows, p->dense_A.p, reinterpret_cast<const double *>(p->rhs.p), C.data(),
                                                     lo.data(), hi.data(), (use_bounds & 1) != 0, (use_bounds & 2) != 0, 0, 0.0,
                                                     nullptr, nullptr, reinterpret_cast<double *>(p->x.p), &piv, &msg);
    if (pivots) *pivots = piv;
    if (!good) return fail(ctx, EGS_ERR_LCP_FAILED, msg.empty() ? "MixedConstraintsSolver did not reach a solution" : msg);
    if (ok) *ok = 1;
    p->last_iterations = piv;
    accumulators_from_lambda(p);                              // a = M^-1 J^T lambda
    do_velocity(p, dt);                                       // ensembles.cc:535, 572
    return EGS_OK;
  });
}

egs_status egs_mixed_constraints_solve(egs_context *ctx, int32_t N, const double *A, const double *b,
                                       const uint8_t *C, const double *lo, const double *hi, int32_t use_bounds,
                                       double *x, double *w, int32_t *ok, int32_t *pivots) {
  return egs_mixed_constraints_solve_limits(ctx, N, A, b, C, lo, hi, use_bounds, 0, 0.0, x, w, ok, pivots);
}

egs_status egs_mixed_constraints_solve_limits(egs_context *ctx, int32_t N, const double *A, const double *b,
                                              const uint8_t *C, const double *lo, const double *hi, int32_t use_bounds,
                                              int32_t max_pivots, double max_seconds, double *x, double *w, int32_t *ok,
                                              int32_t *pivots) {
  if (!ctx) return EGS_ERR_INVALID;
  if (N < 0 || (N > 0 && (!A || !b || !C || !lo || !hi || !x || !w))) return fail(ctx, EGS_ERR_INVALID, "NULL array");
  if (ok) *ok = 0;
  return guarded(ctx, [&]() -> egs_status {
    HIPCHK(hipSetDevice(ctx->device));
    int piv = 0;
    std::string msg;
    const bool good = dense_mixed_constraints(ctx->stream, N, A, b, C, lo, hi, (use_bounds & 1) != 0,
                                              (use_bounds & 2) != 0, x, w, &piv, &msg, max_pivots, max_seconds);
    if (ok) *ok = good ? 1 : 0;
    if (pivots) *pivots = piv;
    if (!good) return fail(ctx, EGS_ERR_LCP_FAILED, msg.empty() ? "MixedConstraintsSolver did not reach a solution" : msg);
    return EGS_OK;
  });
}

static egs_status box_lcp_incremental_entry(egs_context *ctx, int algorithm, int32_t n, double *A, const double *b, const double *lo,
                                     const double *hi, int32_t max_steps, double *x, double *w, int32_t *perm, int32_t *ok,
                                     int32_t *pivots) {
  if (!ctx) return EGS_ERR_INVALID;
  if (ok) *ok = 0;
  if (n < 1 || n > kIncrementalMaxRows) return fail(ctx, EGS_ERR_INVALID, "incremental box LCP: 1 <= n <= 1024");
  if (!A || !b || !lo || !hi || !x || !w) return fail(ctx, EGS_ERR_INVALID, "NULL array");
  return guarded(ctx, [&]() -> egs_status {
    HIPCHK(hipSetDevice(ctx->device));
    int piv = 0;
    std::string msg;
    const bool good = box_lcp_incremental(ctx->stream, algorithm, n, A, b, lo, hi, x, w, perm, max_steps, 0.0, &piv, &msg);
    if (ok) *ok = good ? 1 : 0;
    if (pivots) *pivots = piv;
    if (!good) return fail(ctx, EGS_ERR_LCP_FAILED, msg.empty() ? "the box LCP solver did not reach a solution" : msg);
    return EGS_OK;
  });
}

egs_status egs_box_lcp_dantzig(egs_context *ctx, int32_t n, double *A, const double *b, const double *lo, const double *hi,
                               int32_t max_steps, double *x, double *w, int32_t *perm, int32_t *ok, int32_t *pivots) {
  return box_lcp_incremental_entry(ctx, 1, n, A, b, lo, hi, max_steps, x, w, perm, ok, pivots);
}
egs_status egs_box_lcp_murty(egs_context *ctx, int32_t n, double *A, const double *b, const double *lo, const double *hi,
                             int32_t max_iterations, double *x, double *w, int32_t *perm, int32_t *ok, int32_t *iterations) {
  return box_lcp_incremental_entry(ctx, 0, n, A, b, lo, hi, max_iterations, x, w, perm, ok, iterations);
}

egs_status egs_box_lcp_batch(egs_context *ctx, int32_t algorithm, int32_t count, const int32_t *n, double *A, const double *b,
                             const double *lo, const double *hi, int32_t max_steps, double max_seconds, double *x, double *w,
                             int32_t *perm, int32_t *ok, int32_t *pivots) {
  if (!ctx) return EGS_ERR_INVALID;
  if (count < 0 || (count > 0 && (!n || !A || !b || !lo || !hi || !x || !w || !ok))) return fail(ctx, EGS_ERR_INVALID, "NULL array");
  if (algorithm != 0 && algorithm != 1) return fail(ctx, EGS_ERR_INVALID, "algorithm: 0 (Murty) or 1 (Cottle-Dantzig)");
  for (int k = 0; k < count; ++k) {
    ok[k] = 0;
    if (n[k] < 1 || n[k] > kIncrementalMaxRows) return fail(ctx, EGS_ERR_INVALID, "incremental box LCP: 1 <= n <= 1024");
  }
  return guarded(ctx, [&]() -> egs_status {
    HIPCHK(hipSetDevice(ctx->device));
    box_lcp_incremental_batch(ctx->stream, algorithm, count, n, A, b, lo, hi, max_steps, max_seconds, x, w, perm, ok, pivots, nullptr);
    return EGS_OK;     // per-problem outcome in ok[]
  });
}

egs_status egs_box_lcp_schur(egs_context *ctx, int32_t n, double *A, const double *b, const double *lo, const double *hi,
                             int32_t algorithm, int32_t nub, int32_t reference_quirks, int32_t max_iterations, double max_seconds,
                             double *x, double *w, int32_t *perm, int32_t *ok, int32_t *nub_out, int32_t *pivots) {
  if (!ctx) return EGS_ERR_INVALID;
  if (ok) *ok = 0;
  if (n < 1 || nub > n) return fail(ctx, EGS_ERR_INVALID, "SolveLCP_BoxSchur: n >= 1, nub <= n");
  if (!A || !b || !lo || !hi || !x || !w) return fail(ctx, EGS_ERR_INVALID, "NULL array");
  if (algorithm != 0 && algorithm != 1) return fail(ctx, EGS_ERR_INVALID, "algorithm: 0 (Murty) or 1 (Cottle-Dantzig)");
  return guarded(ctx, [&]() -> egs_status {
    HIPCHK(hipSetDevice(ctx->device));
    int piv = 0, nub_found = 0;
    std::string msg;
    const bool good = box_lcp_schur(ctx->stream, n, A, b, lo, hi, algorithm, nub, reference_quirks != 0, max_iterations, max_seconds,
                                    x, w, perm, &nub_found, &piv, &msg);
    if (ok) *ok = good ? 1 : 0;
    if (pivots) *pivots = piv;
    if (nub_out) *nub_out = nub_found;
    if (!good) return fail(ctx, EGS_ERR_LCP_FAILED, msg.empty() ? "SolveLCP_BoxSchur did not reach a solution" : msg);
    return EGS_OK;
  });
}

egs_status egs_update_contacts_joints(egs_context *ctx, int32_t n, const double *pos, const double *R,
                                      const double *side, int32_t m_joints, const int32_t *jb0, const int32_t *jb1,
                                      const double *jdata, int32_t max_contacts, int32_t *m_out, int32_t *body0,
                                      int32_t *body1, double *data) {
  if (!ctx) return EGS_ERR_INVALID;
  if (n < 0 || m_joints < 0 || !m_out || (n > 0 && (!pos || !R || !side)) || (m_joints > 0 && (!jb0 || !jb1 || !jdata)) ||
      (max_contacts > 0 && (!body0 || !body1 || !data)))
    return fail(ctx, EGS_ERR_INVALID, "NULL array");
  for (int q = 0; q < m_joints; ++q)
    if (jb0[q] < -1 || jb0[q] >= n || jb1[q] < -1 || jb1[q] >= n) return fail(ctx, EGS_ERR_INVALID, "joint body index out of range");
  *m_out = 0;
  // only body-body joints can prune (the pair scan never visits the ground, quirk Q4)
  std::vector<int32_t> b0, b1; std::vector<double> jd;
  for (int q = 0; q < m_joints; ++q)
    if (jb0[q] >= 0 && jb1[q] >= 0) { b0.push_back(jb0[q]); b1.push_back(jb1[q]); jd.insert(jd.end(), jdata + 7 * (size_t)q, jdata + 7 * (size_t)q + 7); }
  return guarded(ctx, [&]() -> egs_status {
    HIPCHK(hipSetDevice(ctx->device));
    *m_out = update_contacts(ctx->stream, n, pos, R, side, max_contacts, body0, body1, data, nullptr, nullptr,
                             (int)b0.size(), b0.data(), b1.data(), jd.data());
    return EGS_OK;
  });
}

egs_status egs_update_contacts(egs_context *ctx, int32_t n, const double *pos, const double *R, const double *side,
                               int32_t max_contacts, int32_t *m_out, int32_t *body0, int32_t *body1, double *data) {
  if (!ctx) return EGS_ERR_INVALID;
  if (n < 0 || !m_out || (n > 0 && (!pos || !R || !side)) || (max_contacts > 0 && (!body0 || !body1 || !data)))
    return fail(ctx, EGS_ERR_INVALID, "NULL array");
  *m_out = 0;
  return guarded(ctx, [&]() -> egs_status {
    HIPCHK(hipSetDevice(ctx->device));
    *m_out = update_contacts(ctx->stream, n, pos, R, side, max_contacts, body0, body1, data, nullptr, nullptr);
    return EGS_OK;
  });
}

egs_status egs_debug_plan(int32_t n, int32_t m, const int32_t *body0, const int32_t *body1, int32_t tile_size,
                          int32_t *n_islands, int32_t *n_tiles, int32_t *n_global, int32_t *cons_tile,
                          int32_t *pos0, int32_t *cnt0, int32_t *pos1, int32_t *cnt1) {
  if (n < 0 || m < 0 || (m > 0 && (!body0 || !body1))) return EGS_ERR_INVALID;
  try {
    const Plan pl = build_plan(n, m, body0, body1, tile_size);
    if (n_islands) *n_islands = pl.n_islands;
    if (n_tiles) *n_tiles = pl.n_tiles;
    if (n_global) *n_global = (int32_t)pl.global.size();
    for (int i = 0; i < m; ++i) {
      if (cons_tile) cons_tile[i] = -1;
    }
    for (int t = 0; t < pl.n_tiles; ++t)
      for (int l = 0; l < pl.block; ++l) {
        const LaneDesc &d = pl.lanes[(size_t)t * pl.block + l];
        if (d.cidx < 0) continue;
        if (cons_tile) cons_tile[d.cidx] = t;
        if (pos0) pos0[d.cidx] = d.pos0;
        if (cnt0) cnt0[d.cidx] = d.cnt0;
        if (pos1) pos1[d.cidx] = d.pos1;
        if (cnt1) cnt1[d.cidx] = d.cnt1;
      }
    for (const GlobalDesc &g : pl.global) {
      if (pos0) pos0[g.cidx] = g.pos0;
      if (cnt0) cnt0[g.cidx] = g.cnt0;
      if (pos1) pos1[g.cidx] = g.pos1;
      if (cnt1) cnt1[g.cidx] = g.cnt1;
    }
    return EGS_OK;
  } catch (const std::exception &) {
    return EGS_ERR_INVALID;
  }
}

egs_status egs_problem_debug_trace(egs_problem *p, uint64_t *out, int64_t count, int32_t *sweeps) {
  if (!p || !out) return EGS_ERR_INVALID;
  return guarded(p->ctx, [&]() -> egs_status {
    if (sweeps) *sweeps = p->trace_sweeps;
    const int64_t have = (int64_t)p->trace_sweeps * p->m;
    if (have <= 0 || count < have) return fail(p->ctx, EGS_ERR_INVALID, "no trace recorded (EGS_TRACE_UPDATES=1, an island on 4-lane patches) or buffer too small");
    HIPCHK(hipMemcpyAsync(out, p->trace.p, (size_t)have * sizeof(uint64_t), hipMemcpyDeviceToHost, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    return EGS_OK;
  });
}

egs_status egs_debug_plan_patches(int32_t n, int32_t m, const int32_t *body0, const int32_t *body1, int32_t *n_patches,
                                  int32_t *cons_patch, int32_t *cons_lane, int32_t *remote0, int32_t *remote1) {
  if (n < 0 || m < 0 || (m > 0 && (!body0 || !body1))) return EGS_ERR_INVALID;
  try {
    const Plan pl = build_plan(n, m, body0, body1, 256);
    if (n_patches) *n_patches = pl.n_patch_tiles;
    for (int i = 0; i < m; ++i) {
      if (cons_patch) cons_patch[i] = -1;
      if (cons_lane) cons_lane[i] = -1;
      if (remote0) remote0[i] = 0;
      if (remote1) remote1[i] = 0;
    }
    for (int t = 0; t < pl.n_patch_tiles; ++t)
      for (int l = 0; l < pl.block; ++l) {
        const LaneDesc &d = pl.patch_lanes[(size_t)t * pl.block + l];
        if (d.cidx < 0) continue;
        if (cons_patch) cons_patch[d.cidx] = t;
        if (cons_lane) cons_lane[d.cidx] = l;
        if (remote0) remote0[d.cidx] = ((d.slot0 & kPrevRemote) ? 1 : 0) | ((d.slot0 & kNextRemote) ? 2 : 0);
        if (remote1) remote1[d.cidx] = ((d.slot1 & kPrevRemote) ? 1 : 0) | ((d.slot1 & kNextRemote) ? 2 : 0);
      }
    return EGS_OK;
  } catch (const std::exception &) {
    return EGS_ERR_INVALID;
  }
}

egs_status egs_debug_plan_slots(int32_t n, int32_t m, const int32_t *body0, const int32_t *body1, int32_t tile_size,
                                int32_t *lane, int32_t *slot0, int32_t *slot1, int32_t *tile_nslots) {
  if (n < 0 || m < 0 || (m > 0 && (!body0 || !body1))) return EGS_ERR_INVALID;
  try {
    const Plan pl = build_plan(n, m, body0, body1, tile_size);
    for (int i = 0; i < m; ++i) {
      if (lane) lane[i] = -1;
      if (slot0) slot0[i] = -1;
      if (slot1) slot1[i] = -1;
      if (tile_nslots) tile_nslots[i] = -1;
    }
    for (int t = 0; t < pl.n_tiles; ++t)
      for (int l = 0; l < pl.block; ++l) {
        const LaneDesc &d = pl.lanes[(size_t)t * pl.block + l];
        if (d.cidx < 0) continue;
        if (lane) lane[d.cidx] = l;
        if (slot0) slot0[d.cidx] = d.slot0;
        if (slot1) slot1[d.cidx] = d.slot1;
        if (tile_nslots) tile_nslots[d.cidx] = pl.tile_nslots[t];
      }
    return EGS_OK;
  } catch (const std::exception &) {
    return EGS_ERR_INVALID;
  }
}

egs_status egs_debug_plan_timetable(int32_t n, int32_t m, const int32_t *body0, const int32_t *body1, int32_t tile_size,
                                    int32_t *level, int32_t *period, int32_t *depth, int32_t *runs) {
  if (n < 0 || m < 0 || (m > 0 && (!body0 || !body1))) return EGS_ERR_INVALID;
  try {
    const Plan pl = build_plan(n, m, body0, body1, tile_size, nullptr, tile_size == 0 ? (1 << 30) : 0);
    if (!pl.levels_ok) return EGS_ERR_INVALID;
    if (runs) *runs = pl.runs ? 1 : 0;
    for (int i = 0; i < m; ++i) {
      if (level) level[i] = -1;
      if (period) period[i] = -1;
      if (depth) depth[i] = -1;
    }
    for (int t = 0; t < pl.n_tiles; ++t)
      for (int l = 0; l < pl.block; ++l) {
        const LaneDesc &d = pl.lanes[(size_t)t * pl.block + l];
        if (d.cidx < 0) continue;
        if (level) level[d.cidx] = pl.lane_level[(size_t)t * pl.block + l];
        if (period) period[d.cidx] = pl.tile_period[t];
        if (depth) depth[d.cidx] = pl.tile_depth[t];
      }
    return EGS_OK;
  } catch (const std::exception &) {
    return EGS_ERR_INVALID;
  }
}

}  // extern "C"

// ---------------------------------------------------------------------------
// egs_world: Ensemble::Step (ensembles.cc:390-427) resident on the device.
// Body state lives in the current egs_problem's buffers; each step runs
// UpdateContacts (Collider) -> [re-plan only if the constraint topology
// changed] -> assemble -> solve -> velocity -> StepPositions_ODE, and only the
// contact topology (8 bytes per contact) crosses PCIe, to feed the host plan.
struct egs_world {
  egs_context *ctx = nullptr;
  int n = 0, precision = EGS_F64;
  egs_problem *prob = nullptr;
  DevBuf<double> dside;
  Collider col;
  std::vector<int32_t> jb0, jb1;       // permanent constraints (joints), listed first (ensembles.cc:234-239)
  std::vector<double> jdata;
  DevBuf<int32_t> djb0, djb1;          // the same on the device, for joint-vs-contact pruning
  DevBuf<double> djdata;
  std::vector<int32_t> topo_b0, topo_b1;
  PinnedArena topo_pinned;             // page-locked landing area for the contact topology
  int32_t *h_b0 = nullptr, *h_b1 = nullptr;
  size_t h_cap = 0;
  int m_contacts = 0;
  int replans = 0;
  bool have_bodies = false;
  // EGS_WORLD_TRACE=1: host wall time per phase of egs_world_step, printed by egs_world_destroy
  bool trace = false;
  double t_phase[5] = {0, 0, 0, 0, 0};   // collide, topology D2H + compare, re-plan, solve + integrate (enqueue), steps
};

namespace {

void world_make_problem(egs_world *w, const int32_t *b0, const int32_t *b1, int m) {
  hipStream_t s = w->ctx->stream;
  if (!w->prob) {
    egs_problem *created = nullptr;
    egs_status st = egs_problem_create(w->ctx, w->n, m, b0, b1, w->precision, &created);
    if (st != EGS_OK) throw HipError(std::string("world: egs_problem_create: ") + egs_last_error(w->ctx));
    w->prob = created;
  } else {  // same bodies, new constraint list: the body state stays where it is
    if (check_topology(w->ctx, w->n, m, b0, b1) != EGS_OK)
      throw std::invalid_argument(egs_last_error(w->ctx));
    problem_set_topology(w->prob, m, b0, b1, /*fresh=*/false);
  }
  egs_problem *np = w->prob;
  // constraint kinds: joints first, then contacts; joint descriptors are static
  std::vector<int32_t> kind((size_t)(m > 0 ? m : 1), EGS_CONTACT_BOX);
  const int mj = (int)w->jb0.size();
  for (int i = 0; i < mj; ++i) kind[i] = EGS_JOINT_BALL;
  if (m > 0) {
    kind.resize((size_t)m);
    stage(w->ctx, np->kind, kind);   // through the pinned arena (reset only after a synchronise): no wait here
    if (mj > 0) upload(np->data, w->jdata.data(), (size_t)mj * 7, s);
  }
  np->have_constraints = true;
  np->h_rows_valid = false;
  w->topo_b0.assign(b0, b0 + m); w->topo_b1.assign(b1, b1 + m);
  ++w->replans;
}

}  // namespace

extern "C" {

egs_status egs_world_create(egs_context *ctx, int32_t n_bodies, int32_t precision, egs_world **out) {
  if (!ctx || !out || n_bodies < 0) return EGS_ERR_INVALID;
  *out = nullptr;
  if (precision != EGS_F64 && precision != EGS_F32) return fail(ctx, EGS_ERR_INVALID, "unknown precision");
  egs_world *w = new (std::nothrow) egs_world;
  if (!w) return fail(ctx, EGS_ERR_HIP, "host allocation failed");
  w->ctx = ctx; w->n = n_bodies; w->precision = precision;
  { const char *te = std::getenv("EGS_WORLD_TRACE"); w->trace = te && std::atoi(te) != 0; }
  egs_status st = guarded(ctx, [&]() -> egs_status {
    HIPCHK(hipSetDevice(ctx->device));
    w->dside.alloc((size_t)(n_bodies > 0 ? n_bodies : 1) * 3);
    return EGS_OK;
  });
  if (st != EGS_OK) { delete w; return st; }
  *out = w;
  return EGS_OK;
}

void egs_world_destroy(egs_world *w) {
  if (!w) return;
  if (w->trace && w->t_phase[4] > 0) {
    const double k = 1.0 / w->t_phase[4];
    std::fprintf(stderr, "egs_world trace (%d steps, %d re-plans), us/step: collide %.1f  topology %.1f  re-plan %.1f  solve+integrate %.1f\n",
                 (int)w->t_phase[4], w->replans, w->t_phase[0] * k, w->t_phase[1] * k, w->t_phase[2] * k, w->t_phase[3] * k);
  }
  if (w->prob) egs_problem_destroy(w->prob);
  delete w;
}

egs_status egs_world_set_bodies(egs_world *w, const double *pos, const double *R, const double *v, const double *wv,
                                const double *Minv, const double *f_ext, const double *side_lengths) {
  if (!w) return EGS_ERR_INVALID;
  // the first call needs everything; afterwards NULL = keep (M^-1, f_ext and the side lengths
  // are frozen at Init in the reference, Q5)
  if (w->n > 0 && !w->have_bodies && (!pos || !R || !v || !wv || !Minv || !f_ext || !side_lengths))
    return fail(w->ctx, EGS_ERR_INVALID, "NULL array on the first egs_world_set_bodies");
  return guarded(w->ctx, [&]() -> egs_status {
    if (!w->prob) world_make_problem(w, w->jb0.data(), w->jb1.data(), (int)w->jb0.size());
    egs_status st = egs_problem_set_state(w->prob, pos, R, v, wv, Minv, f_ext);
    if (st != EGS_OK) return st;
    if (side_lengths) upload(w->dside, side_lengths, (size_t)w->n * 3, w->ctx->stream);
    w->have_bodies = true;
    return EGS_OK;
  });
}

egs_status egs_world_set_joints(egs_world *w, int32_t m_joints, const int32_t *body0, const int32_t *body1,
                                const double *data) {
  if (!w || m_joints < 0 || (m_joints > 0 && (!body0 || !body1 || !data))) return EGS_ERR_INVALID;
  w->jb0.assign(body0, body0 + m_joints);
  w->jb1.assign(body1, body1 + m_joints);
  w->jdata.assign(data, data + (size_t)m_joints * 7);
  return guarded(w->ctx, [&]() -> egs_status {
    w->djb0.alloc((size_t)m_joints); w->djb1.alloc((size_t)m_joints); w->djdata.alloc((size_t)m_joints * 7);
    if (m_joints > 0) {
      upload(w->djb0, body0, (size_t)m_joints, w->ctx->stream);
      upload(w->djb1, body1, (size_t)m_joints, w->ctx->stream);
      upload(w->djdata, data, (size_t)m_joints * 7, w->ctx->stream);
    }
    world_make_problem(w, w->jb0.data(), w->jb1.data(), m_joints);   // contacts are re-detected by the next step
    w->m_contacts = 0;
    return EGS_OK;
  });
}

egs_status egs_world_step(egs_world *w, double dt, double erp, const egs_solve_params *params,
                          int32_t detect_contacts, egs_solve_stats *stats) {
  if (!w) return EGS_ERR_INVALID;
  if (!w->have_bodies) return fail(w->ctx, EGS_ERR_INVALID, "egs_world_set_bodies first");
  if (!(dt > 0)) return fail(w->ctx, EGS_ERR_INVALID, "dt must be > 0");
  return guarded(w->ctx, [&]() -> egs_status {
    hipStream_t s = w->ctx->stream;
    const int mj = (int)w->jb0.size();
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    auto lap = [&](int k) {
      if (!w->trace) return;
      const auto t1 = clk::now();
      w->t_phase[k] += std::chrono::duration<double, std::micro>(t1 - t0).count();
      t0 = t1;
    };
    if (detect_contacts) {  // UpdateContacts + pruning on the device (ensembles.cc:393-394)
      const int mc = w->col.run(s, w->n, w->prob->pos.p, w->prob->R.p, w->dside.p, mj, w->djb0.p, w->djb1.p, w->djdata.p);
      lap(0);
      const size_t mt = (size_t)mj + (size_t)mc;
      if (mt > w->h_cap) {   // grow-only; the stream is idle here (col.run synchronised)
        w->topo_pinned.reset();
        w->h_cap = mt + mt / 4 + 64;
        w->h_b0 = static_cast<int32_t *>(w->topo_pinned.take(2 * w->h_cap * sizeof(int32_t)));
        w->h_b1 = w->h_b0 + w->h_cap;
      }
      std::copy(w->jb0.begin(), w->jb0.end(), w->h_b0);
      std::copy(w->jb1.begin(), w->jb1.end(), w->h_b1);
      if (mc > 0) {   // the GPU writes the 8 bytes per contact straight into page-locked host memory
        w->col.export_topology(s, mc, w->h_b0 + mj, w->h_b1 + mj);
        HIPCHK(hipStreamSynchronize(s));
      }
      const bool changed = mt != w->topo_b0.size() || (mt > 0 && (
                           std::memcmp(w->h_b0, w->topo_b0.data(), mt * sizeof(int32_t)) != 0 ||
                           std::memcmp(w->h_b1, w->topo_b1.data(), mt * sizeof(int32_t)) != 0));
      lap(1);
      if (changed) world_make_problem(w, w->h_b0, w->h_b1, (int)mt);  // host plan only on topology change
      lap(2);
      if (mc > 0)
        HIPCHK(hipMemcpyAsync(w->prob->data.p + (size_t)mj * 7, w->col.data(), (size_t)mc * 7 * sizeof(double),
                              hipMemcpyDeviceToDevice, s));
      w->m_contacts = mc;
    }
    egs_problem *p = w->prob;
    if (p->m > 0) {
      do_assemble(p, dt, erp);
      egs_status st = do_solve(p, params, stats);
      if (st != EGS_OK) return st;
      // the body state must not be advanced with a lambda that came out of a timed-out ordering
      // wait: look at the flag before integrating (one 4-byte read-back per step)
      HIPCHK(hipStreamSynchronize(s));
      if (stall_seen(p)) return report_stall(p);
    } else {  // no constraints: v_dot = M^-1 f (ensembles.cc:504-505)
      if (egs_status st = validate_params(w->ctx, params)) return st;
      HIPCHK(hipMemsetAsync(p->acc.p, 0, (size_t)(p->n > 0 ? p->n : 1) * 6 * p->real_size(), s));
      if (stats) { std::memset(stats, 0, sizeof *stats); fill_stats(p, stats); }
    }
    do_velocity(p, dt);
    launch_advance(p->n, p->pos.p, p->R.p, p->v.p, p->w.p, p->v6.p, dt, s);
    HIPCHK(hipGetLastError());
    if (w->trace) { HIPCHK(hipStreamSynchronize(s)); lap(3); w->t_phase[4] += 1; }
    return EGS_OK;
  });
}

egs_status egs_world_get_bodies(egs_world *w, double *pos, double *R, double *v, double *wv) {
  if (!w || !w->prob) return EGS_ERR_INVALID;
  return egs_problem_get_state(w->prob, pos, R, v, wv);
}

egs_status egs_world_get_contacts(egs_world *w, int32_t max_contacts, int32_t *m_out, int32_t *body0, int32_t *body1,
                                  double *data) {
  if (!w || !m_out) return EGS_ERR_INVALID;
  *m_out = w->m_contacts;
  if (w->m_contacts > max_contacts) return fail(w->ctx, EGS_ERR_INVALID, "max_contacts too small");
  return guarded(w->ctx, [&]() -> egs_status {
    const size_t mc = (size_t)w->m_contacts;
    hipStream_t s = w->ctx->stream;
    if (mc && body0) HIPCHK(hipMemcpyAsync(body0, w->col.body0(), mc * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (mc && body1) HIPCHK(hipMemcpyAsync(body1, w->col.body1(), mc * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (mc && data) HIPCHK(hipMemcpyAsync(data, w->col.data(), mc * 7 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return EGS_OK;
  });
}

egs_status egs_world_get_lambda(egs_world *w, int32_t max_rows, int32_t *rows_out, double *lambda) {
  if (!w || !w->prob || !rows_out) return EGS_ERR_INVALID;
  *rows_out = 3 * w->prob->m;
  if (3 * w->prob->m > max_rows) return fail(w->ctx, EGS_ERR_INVALID, "max_rows too small");
  if (w->prob->m == 0) return EGS_OK;
  return egs_problem_get_lambda(w->prob, lambda);
}

egs_status egs_world_info(egs_world *w, int32_t *n_constraints, int32_t *n_contacts, int32_t *replans) {
  if (!w) return EGS_ERR_INVALID;
  if (n_constraints) *n_constraints = w->prob ? w->prob->m : 0;
  if (n_contacts) *n_contacts = w->m_contacts;
  if (replans) *replans = w->replans;
  return EGS_OK;
}

}  // extern "C"
