// matvec.h -- launch interface of the stand-alone mat-vec kernels (matvec.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "matvec_plan.h"

namespace egs {

template <typename REAL>
struct MatvecArgs {
  const MvLane *lanes;
  const MvTile *tiles;
  const MvSlot *slots;
  const uint16_t *ents;
  const MvBoundary *boundary;
  int32_t n_boundary;
  int32_t max_slots;
  const REAL *Minv;          // [n][36]
  const REAL *J0, *J1;       // [m][18]
  const REAL *x;             // [3m]
  REAL *y;                   // [3m]
  REAL *T;                   // [n_shared_entries][6]  J^T x of the sides on shared bodies
  REAL eps, scale;
  int32_t accumulate;        // y += part instead of y = part
  int32_t stream_nt;         // 1: J0 / J1 with non-temporal loads (EGS_MV_NT=0 turns it off)
};

// part: 1 = strict lower, 2 = strict upper, 4 = diagonal, 8 = the full product.
// Enqueues the shared-body pre-pass (when the schedule has shared bodies) and the tile kernel.
template <typename REAL>
void launch_matvec(const MatvecArgs<REAL> &a, int part, int n_tiles, int block, hipStream_t s);

size_t matvec_lds_bytes(int block, int max_slots, size_t real_size);

}  // namespace egs
