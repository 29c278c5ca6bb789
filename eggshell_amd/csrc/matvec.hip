// matvec.hip -- gfx950 kernels of the stand-alone block-sparse products
//   sparse::CalculateSparseJMJtX   y = (J W J^T + eps I) x    sparse_iterations_utils.cc:624-695
//   sparse::CalculateSparseLx/Ux   strict lower / upper part   :427-493, :495-561
//   sparse::CalculateSparseDx      ((diag + eps) * scale) x    :571-603
// (LxUx / UxDx / LxDx, :563-569 and :606-622, are two launches, the second accumulating).
//
// The reference visits all O(m^2) constraint pairs; here the product goes through the
// per-body sums u_b = sum_j J_jb^T x_j (list order) and a_b = W_b u_b, i.e. O(nnz):
//   matvec_tile_kernel      one workgroup per tile of the schedule (matvec_plan.h):
//                           J0 / J1 rows of the tile streamed HBM -> LDS with 16-byte
//                           coalesced loads (a tile's constraints are ascending runs of the
//                           list), each lane then owns one constraint: J^T x into the LDS
//                           entry array, per-body ordered sums and W u by 6 lanes per body,
//                           gather y_i = J_i0 a_b0 + J_i1 a_b1 (+ eps x_i).  For a tile whose
//                           bodies are all private J is read from HBM exactly once.
//   matvec_boundary_kernel  pre-pass for SHARED bodies only: the constraint sides that touch
//                           them publish J^T x in a global entry array.
// Built with -ffp-contract=off; operation order = oracle/matvec_fast.inc (bit-comparable).
#include "matvec.h"
#include "solve_device.h"

namespace egs {

namespace {

template <typename REAL> struct Vec2;
template <> struct Vec2<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct Vec2<float> { typedef float type __attribute__((ext_vector_type(2))); };

// own 3x3 block D = J0 W0 J0^T + J1 W1 J1^T in the order of load_cons (solve_device.h):
// B = W J^T column by column, then the k-chains, then d0 + d1
template <typename REAL>
__device__ __forceinline__ void own_block(const REAL *J0, const REAL *J1, bool has0, bool has1,
                                          const REAL *W0, const REAL *W1, REAL *D) {
  REAL B0[18], B1[18];
#pragma unroll
  for (int k = 0; k < 18; ++k) { B0[k] = REAL(0); B1[k] = REAL(0); }
  if (has0) {
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) {
      REAL Wr[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) Wr[k] = W0[6 * cc + k];
#pragma unroll
      for (int r = 0; r < 3; ++r) B0[3 * cc + r] = dot6(Wr, J0 + 6 * r);
    }
  }
  if (has1) {
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) {
      REAL Wr[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) Wr[k] = W1[6 * cc + k];
#pragma unroll
      for (int r = 0; r < 3; ++r) B1[3 * cc + r] = dot6(Wr, J1 + 6 * r);
    }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      REAL d0 = J0[6 * r] * B0[q];
#pragma unroll
      for (int k = 1; k < 6; ++k) d0 = tfma(J0[6 * r + k], B0[3 * k + q], d0);
      REAL d1 = J1[6 * r] * B1[q];
#pragma unroll
      for (int k = 1; k < 6; ++k) d1 = tfma(J1[6 * r + k], B1[3 * k + q], d1);
      D[3 * r + q] = d0 + d1;
    }
}

// t[k] = sum_r J[r][k] x_r, rows in order 0,1,2
template <typename REAL>
__device__ __forceinline__ void jt_x(const REAL *J, const REAL *x, REAL *t) {
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    REAL v = J[k] * x[0];
    v = tfma(J[6 + k], x[1], v);
    v = tfma(J[12 + k], x[2], v);
    t[k] = v;
  }
}

template <typename REAL>
__global__ void __launch_bounds__(256) matvec_boundary_kernel(const MatvecArgs<REAL> A) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= A.n_boundary) return;
  const MvBoundary b = A.boundary[i];
  REAL x[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) x[r] = A.x[(size_t)b.cidx * 3 + r];
  for (int side = 0; side < 2; ++side) {
    const int t = side ? b.t1 : b.t0;
    if (t < 0) continue;
    const REAL *Jg = (side ? A.J1 : A.J0) + (size_t)b.cidx * 18;
    REAL J[18], tt[6];
#pragma unroll
    for (int k = 0; k < 18; ++k) J[k] = Jg[k];
    jt_x(J, x, tt);
#pragma unroll
    for (int k = 0; k < 6; ++k) A.T[(size_t)t * 6 + k] = tt[k];
  }
}

// PART: 1 = L, 2 = U, 4 = D, 8 = full product
template <typename REAL, int BLOCK, int PART>
__global__ void __launch_bounds__(BLOCK) matvec_tile_kernel(const MatvecArgs<REAL> A) {
  typedef typename Vec2<REAL>::type V2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [stage0: BLOCK x 18][stage1: BLOCK x 18][a: max_slots x 6][cidx: BLOCK][slot body: max_slots]
  REAL *s_stage0 = reinterpret_cast<REAL *>(smem);
  REAL *s_stage1 = s_stage0 + BLOCK * 18;
  REAL *s_a = s_stage1 + BLOCK * 18;
  int *s_cidx = reinterpret_cast<int *>(s_a + (size_t)A.max_slots * 6);
  int *s_body = s_cidx + BLOCK;
  REAL *s_t = s_stage0;      // entry array, aliases stage0 once the rows are in registers
  REAL *s_pre = s_stage1;    // W (partial body sum) per entry (L / U), aliases stage1

  const int tile = blockIdx.x, tid = threadIdx.x;
  const MvTile T = A.tiles[tile];
  const MvSlot *slots = A.slots + T.slot_off;
  const MvLane d = A.lanes[(size_t)tile * BLOCK + tid];
  const bool active = d.cidx >= 0;
  const bool has0 = active && d.slot0 != 0, has1 = active && d.slot1 != 0;
  s_cidx[tid] = d.cidx;

  // the body pass: 6 lanes per slot, lane r owns component r; its first slot's W row and
  // descriptor are requested now, before the J stream
  constexpr int kGroups = BLOCK / 6;
  const int grp = tid / 6, comp = tid - 6 * grp;
  const bool in_pass = PART != 4 && grp < kGroups;
  MvSlot sd{-1, 0, 0, -1, -1};
  REAL Wrow[6] = {REAL(0), REAL(0), REAL(0), REAL(0), REAL(0), REAL(0)};
  if (in_pass && 1 + grp < T.nslots) {
    sd = slots[1 + grp];
#pragma unroll
    for (int k = 0; k < 6; ++k) Wrow[k] = A.Minv[(size_t)sd.body * 36 + 6 * comp + k];
  }
  for (int s = tid; s < T.nslots; s += BLOCK) s_body[s] = slots[s].body;
  if (PART == 8 && tid < 6) s_a[tid] = REAL(0);   // slot 0 = the world
  REAL x[3] = {REAL(0), REAL(0), REAL(0)};
  if (active) {
#pragma unroll
    for (int r = 0; r < 3; ++r) x[r] = A.x[(size_t)d.cidx * 3 + r];
  }
  __syncthreads();

  // J0, J1 rows of the tile: HBM -> registers -> LDS, 9 two-element units per row, consecutive
  // lanes on consecutive units (ascending constraint runs make these long contiguous reads)
  {
    const V2 *g0 = reinterpret_cast<const V2 *>(A.J0), *g1 = reinterpret_cast<const V2 *>(A.J1);
    V2 *st0 = reinterpret_cast<V2 *>(s_stage0), *st1 = reinterpret_cast<V2 *>(s_stage1);
    V2 r0[9], r1[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int e = q * BLOCK + tid, c = e / 9, u = e - 9 * c;
      const int ci = s_cidx[c];
      const V2 z = {REAL(0), REAL(0)};
      // J is read exactly once per product: streaming (non-temporal) loads keep it out of the caches' way
      r0[q] = ci >= 0 ? (A.stream_nt ? __builtin_nontemporal_load(g0 + (size_t)ci * 9 + u) : g0[(size_t)ci * 9 + u]) : z;
      r1[q] = ci >= 0 ? (A.stream_nt ? __builtin_nontemporal_load(g1 + (size_t)ci * 9 + u) : g1[(size_t)ci * 9 + u]) : z;
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      st0[q * BLOCK + tid] = r0[q];
      st1[q * BLOCK + tid] = r1[q];
    }
  }
  __syncthreads();
  REAL J0[18], J1[18];
  {
    const V2 *st0 = reinterpret_cast<const V2 *>(s_stage0) + tid * 9, *st1 = reinterpret_cast<const V2 *>(s_stage1) + tid * 9;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const V2 a = st0[q], b = st1[q];
      J0[2 * q] = has0 ? a.x : REAL(0); J0[2 * q + 1] = has0 ? a.y : REAL(0);   // a world side counts as zero
      J1[2 * q] = has1 ? b.x : REAL(0); J1[2 * q + 1] = has1 ? b.y : REAL(0);
    }
  }

  REAL y[3] = {REAL(0), REAL(0), REAL(0)};
  if (PART == 4) {
    if (active) {
      REAL D[9];
      own_block(J0, J1, has0, has1, A.Minv + (size_t)(has0 ? s_body[d.slot0] : 0) * 36,
                A.Minv + (size_t)(has1 ? s_body[d.slot1] : 0) * 36, D);
#pragma unroll
      for (int r = 0; r < 3; ++r) y[r] = ((D[4 * r] + A.eps) * A.scale) * x[r];
    }
  } else {
    __syncthreads();   // every lane holds its rows: the stage buffers become the entry arrays
    if (active) {
      REAL t0[6], t1[6];
      jt_x(J0, x, t0);
      jt_x(J1, x, t1);
      if (has0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s_t[(int)d.e0 * 6 + k] = t0[k];
      }
      if (has1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s_t[(int)d.e1 * 6 + k] = t1[k];
      }
    }
    __syncthreads();
    if (in_pass) {
      for (int s = 1 + grp; s < T.nslots; s += kGroups) {
        if (s != 1 + grp) {
          sd = slots[s];
#pragma unroll
          for (int k = 0; k < 6; ++k) Wrow[k] = A.Minv[(size_t)sd.body * 36 + 6 * comp + k];
        }
        REAL run[6] = {REAL(0), REAL(0), REAL(0), REAL(0), REAL(0), REAL(0)};
        if (sd.t_off < 0) {          // private body: its sides are LDS entries seg .. seg + cnt - 1
          for (int kk = 0; kk < sd.cnt; ++kk) {
            const int e = sd.seg + (PART == 2 ? sd.cnt - 1 - kk : kk);
            if (PART != 8) s_pre[e * 6 + comp] = dot6(Wrow, run);
#pragma unroll
            for (int q = 0; q < 6; ++q) run[q] = run[q] + s_t[e * 6 + q];
          }
        } else {                     // shared body: every side was published by the pre-pass
          const REAL *Tg = A.T + (size_t)sd.t_off * 6;
          const uint16_t *ids = A.ents + sd.ents_off;
          for (int kk = 0; kk < sd.cnt; ++kk) {
            const int k = PART == 2 ? sd.cnt - 1 - kk : kk;
            if (PART != 8) {
              const int e = ids[k];
              if (e != kMvRemote) s_pre[e * 6 + comp] = dot6(Wrow, run);
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) run[q] = run[q] + Tg[(size_t)k * 6 + q];
          }
        }
        if (PART == 8) s_a[s * 6 + comp] = dot6(Wrow, run);
      }
    }
    __syncthreads();
    if (active) {
      REAL a0[6], a1[6];
      const REAL *p0 = PART == 8 ? s_a + (int)d.slot0 * 6 : s_pre + (int)d.e0 * 6;
      const REAL *p1 = PART == 8 ? s_a + (int)d.slot1 * 6 : s_pre + (int)d.e1 * 6;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        a0[k] = (PART == 8 || has0) ? p0[k] : REAL(0);
        a1[k] = (PART == 8 || has1) ? p1[k] : REAL(0);
      }
      if (PART == 8) {
#pragma unroll
        for (int r = 0; r < 3; ++r) y[r] = tfma(A.eps, x[r], row_dot(J0 + 6 * r, a0, J1 + 6 * r, a1));
      } else {
        REAL D[9];
        own_block(J0, J1, has0, has1, A.Minv + (size_t)(has0 ? s_body[d.slot0] : 0) * 36,
                  A.Minv + (size_t)(has1 ? s_body[d.slot1] : 0) * 36, D);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const REAL sd2 = row_dot(J0 + 6 * r, a0, J1 + 6 * r, a1);
          REAL own = REAL(0);
          if (PART == 1) {
#pragma unroll
            for (int l = 0; l < r; ++l) own = tfma(D[3 * r + l], x[l], own);
          } else {
#pragma unroll
            for (int l = r + 1; l < 3; ++l) own = tfma(D[3 * r + l], x[l], own);
          }
          y[r] = sd2 + own;
        }
      }
    }
  }
  if (active) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      REAL *o = A.y + (size_t)d.cidx * 3 + r;
      *o = A.accumulate ? *o + y[r] : y[r];
    }
  }
}

}  // namespace

size_t matvec_lds_bytes(int block, int max_slots, size_t real_size) {
  return (size_t)block * 36 * real_size + (size_t)max_slots * 6 * real_size + (size_t)block * sizeof(int) +
         (size_t)max_slots * sizeof(int);
}

template <typename REAL>
void launch_matvec(const MatvecArgs<REAL> &a, int part, int n_tiles, int block, hipStream_t s) {
  if (n_tiles <= 0) return;
  if (a.n_boundary > 0 && part != 4)
    hipLaunchKernelGGL((matvec_boundary_kernel<REAL>), dim3((a.n_boundary + 255) / 256), dim3(256), 0, s, a);
  const size_t lds = matvec_lds_bytes(block, a.max_slots, sizeof(REAL));
  const dim3 g(n_tiles), b(block);
  // more than 64 KiB of dynamic LDS has to be allowed per kernel function
#define EGS_MV1(BLK, P)                                                                               \
  {                                                                                                   \
    static bool raised = false;                                                                       \
    if (!raised && lds > 48 * 1024) {                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&matvec_tile_kernel<REAL, BLK, P>),    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);              \
      raised = true;                                                                                  \
    }                                                                                                 \
    hipLaunchKernelGGL((matvec_tile_kernel<REAL, BLK, P>), g, b, lds, s, a);                          \
  }
#define EGS_MV(BLK)                                                                                   \
  switch (part) {                                                                                     \
    case 1: EGS_MV1(BLK, 1) break;                                                                    \
    case 2: EGS_MV1(BLK, 2) break;                                                                    \
    case 4: EGS_MV1(BLK, 4) break;                                                                    \
    default: EGS_MV1(BLK, 8) break;                                                                   \
  }
  if (block == 128) { EGS_MV(128) } else { EGS_MV(256) }
#undef EGS_MV
#undef EGS_MV1
}

template void launch_matvec<double>(const MatvecArgs<double> &, int, int, int, hipStream_t);
template void launch_matvec<float>(const MatvecArgs<float> &, int, int, int, hipStream_t);

}  // namespace egs
