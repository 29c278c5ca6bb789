// Micro-benchmark of the 64 x 64 diagonal-tile factorisation (chol_diag_tile of eggshell_amd/csrc/dense_lcp.hip):
// the latency chain that bounds every panel step of the blocked Cholesky.  Variants are timed inside one launch
// (reps repetitions on a tile staged in LDS) and checked against a host Cholesky + inverse.
//   hipcc -O3 --offload-arch=gfx950 -o diag_bench diag_bench.hip && ./diag_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

constexpr int NB = 64;
constexpr int kStageLd = NB + 2;
typedef double double4_t __attribute__((ext_vector_type(4)));

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rsqrt_refined(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  const double e = __builtin_fma(-d * y, y, 1.0);
  const double t = __builtin_fma(0.375, e, 0.5) * e;
  return __builtin_fma(y, t, y);
}
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- V0: the round-3 tile routine (five wavefronts, one barrier per column) --------------------------
template <bool INVERSE>
__device__ void tile_v0(const double *sB, double (*sCol)[NB], double *sRinv, double *Tout, int ld, double *inv, int *fail) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < 4) {
    double a[NB / 4];
#pragma unroll
    for (int m = 0; m < NB / 4; ++m) {
      const int c = 4 * m + wave;
      a[m] = (c <= lane) ? sB[lane * kStageLd + c] : 0.0;
    }
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int mo = j >> 2;
      if (wave == (j & 3)) {
        double d = readlane_f64(a[mo], j);
        if (!(d > 0.0)) { bad = true; d = 1.0; }
        const double rinv = rsqrt_refined(d);
        const double l = (lane >= j) ? a[mo] * rinv : 0.0;
        sCol[j & 1][lane] = l;
        if (lane == j) sRinv[j & 1] = rinv;
        if (lane >= j) Tout[(size_t)lane * ld + j] = l;
      }
      lds_barrier();
      if (j + 1 < NB) {
        const double lrow = sCol[j & 1][lane];
#pragma unroll
        for (int m = mo; m < NB / 4; ++m) a[m] = __builtin_fma(-lrow, sCol[j & 1][4 * m + wave], a[m]);
      }
    }
    if (bad && lane == 0) atomicOr(fail, 1);
  } else {
    double a[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) a[c] = 0.0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      lds_barrier();
      if (INVERSE) {
        const double xj = (((lane == j) ? 1.0 : 0.0) - a[j]) * sRinv[j & 1];
        inv[j * NB + lane] = xj;
        double col[NB];
#pragma unroll
        for (int k = j + 1; k < NB; ++k) col[k] = sCol[j & 1][k];
#pragma unroll
        for (int k = j + 1; k < NB; ++k) a[k] = __builtin_fma(col[k], xj, a[k]);
      }
    }
  }
}

// ---- V2: GW columns per barrier ------------------------------------------------------------------------
// Wavefront w owns the column groups g = 4 m + w (columns GW g .. GW g + GW - 1), lane i = row i.  The owner
// factors its GW columns inside the wavefront (readlane broadcasts, no LDS), publishes them, ONE barrier, and every
// wavefront applies the rank-GW update to the groups it still owns.
template <int GW>
__device__ void tile_v2(const double *sB, double *sPanRaw /*[2][GW][NB]*/, double *Tout, int ld, int *fail) {
  constexpr int NG = NB / GW;          // groups
  constexpr int MG = NG / 4;           // groups per wavefront
  double (*sPan)[GW][NB] = reinterpret_cast<double (*)[GW][NB]>(sPanRaw);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= 4) {
#pragma unroll
    for (int g = 0; g < NG; ++g) lds_barrier();
    return;
  }
  double a[MG][GW];
#pragma unroll
  for (int m = 0; m < MG; ++m)
#pragma unroll
    for (int q = 0; q < GW; ++q) {
      const int c = GW * (4 * m + wave) + q;
      a[m][q] = (c <= lane) ? sB[lane * kStageLd + c] : 0.0;
    }
  bool bad = false;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int mo = g >> 2, c0 = GW * g;
    if (wave == (g & 3)) {
      double l[GW];
#pragma unroll
      for (int q = 0; q < GW; ++q) {
        double d = readlane_f64(a[mo][q], c0 + q);
        if (!(d > 0.0)) { bad = true; d = 1.0; }
        const double rinv = rsqrt_refined(d);
        l[q] = (lane >= c0 + q) ? a[mo][q] * rinv : 0.0;
#pragma unroll
        for (int q2 = q + 1; q2 < GW; ++q2) a[mo][q2] = __builtin_fma(-l[q], readlane_f64(l[q], c0 + q2), a[mo][q2]);
        sPan[g & 1][q][lane] = l[q];
        if (lane >= c0 + q) Tout[(size_t)lane * ld + c0 + q] = l[q];
      }
    }
    lds_barrier();
    if (g + 1 < NG) {
      double lrow[GW];
#pragma unroll
      for (int q = 0; q < GW; ++q) lrow[q] = sPan[g & 1][q][lane];
#pragma unroll
      for (int m = mo; m < MG; ++m) {
        // group 4 m + wave is still open iff 4 m + wave > g; for m == mo that depends on the wavefront
        if (m == mo && wave <= (g & 3)) continue;
        const int cg = GW * (4 * m + wave);
#pragma unroll
        for (int q2 = 0; q2 < GW; ++q2) {
          double acc = a[m][q2];
#pragma unroll
          for (int q = 0; q < GW; ++q) acc = __builtin_fma(-lrow[q], sPan[g & 1][q][cg + q2], acc);
          a[m][q2] = acc;
        }
      }
    }
  }
  if (bad && lane == 0) atomicOr(fail, 1);
}


// ---- V3: GW columns per barrier, L kept in LDS (column-major, every column in its own place: no double buffer, no
// global store on the chain), no masks (entries above the diagonal are scratch), the four 16 x 16 diagonal blocks of
// the inverse by the fifth wavefront as the columns appear, the rest of the inverse by MFMA products afterwards. ----
__device__ long long g_stamp[8];
__device__ long long g_trace[4][16][4];
#define TR(k) do { } while (0)
#define STAMP(k) do { if (threadIdx.x == 0) g_stamp[k] = __builtin_readcyclecounter(); } while (0)
constexpr int LS = NB + 2;      // column stride of sL
constexpr int XS = NB + 2;      // row stride of sX
constexpr int QS = 34;          // row stride of the 32 x 32 scratch

__device__ __forceinline__ double4_t mfma4(double a, double b, double4_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void inverse_rest_and_store(const double *sL, double *sX, double *sQ, double *Tout, int ld, double *inv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  STAMP(1);
  lds_barrier();
  STAMP(2);
  const int li = lane & 15, lk = lane >> 4;
  // level 1: X(2p+1, 2p) = -X(2p+1, 2p+1) (L(2p+1, 2p) X(2p, 2p)), p = wavefront 0, 1
  if (wave < 2) {
    const int o = 32 * wave;
    double4_t P = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) P = mfma4(sL[(o + 4 * kk + lk) * LS + o + 16 + li], sX[(o + 4 * kk + lk) * XS + o + li], P);
    double *sP = sQ + wave * 16 * QS;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) sP[(lk + 4 * reg) * QS + li] = P[reg];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    double4_t D = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) D = mfma4(sX[(o + 16 + li) * XS + o + 16 + 4 * kk + lk], sP[(4 * kk + lk) * QS + li], D);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) sX[(o + 16 + lk + 4 * reg) * XS + o + li] = -D[reg];
  }
  lds_barrier();
  // level 2: X_BL = -X_BR (L_BL X_TL), 32 x 32 blocks, one 16 x 16 tile per wavefront
  if (wave < 4) {
    const int ti = wave >> 1, tj = wave & 1;
    double4_t Q = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) Q = mfma4(sL[(4 * kk + lk) * LS + 32 + 16 * ti + li], sX[(4 * kk + lk) * XS + 16 * tj + li], Q);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) sQ[(16 * ti + lk + 4 * reg) * QS + 16 * tj + li] = Q[reg];
  }
  lds_barrier();
  if (wave < 4) {
    const int ti = wave >> 1, tj = wave & 1;
    double4_t D = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) D = mfma4(sX[(32 + 16 * ti + li) * XS + 32 + 4 * kk + lk], sQ[(4 * kk + lk) * QS + 16 * tj + li], D);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) sX[(32 + 16 * ti + lk + 4 * reg) * XS + 16 * tj + li] = -D[reg];
  }
  lds_barrier();
  STAMP(3);
  for (int i = threadIdx.x; i < NB * NB; i += blockDim.x) {
    const int r = i >> 6, c = i & 63;
    inv[i] = sX[r * XS + c];
    if (c <= r) Tout[(size_t)r * ld + c] = sL[c * LS + r];
  }
}

template <int GW>
__device__ void tile_v3(const double *sB, double *sL, double *sRv, double *sX, double *sQ, double *Tout, int ld, double *inv, int *fail) {
  constexpr int NG = NB / GW, MG = NG / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the inverse starts as zero above its diagonal blocks
  for (int i = threadIdx.x; i < NB * XS; i += blockDim.x) sX[i] = 0.0;
  if (wave < 4) {
    double a[MG][GW];
#pragma unroll
    for (int m = 0; m < MG; ++m)
#pragma unroll
      for (int q = 0; q < GW; ++q) a[m][q] = sB[lane * kStageLd + GW * (4 * m + wave) + q];
    bool bad = false;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int mo = g >> 2, c0 = GW * g;
      if (wave == (g & 3)) {
        double l[GW];
#pragma unroll
        for (int q = 0; q < GW; ++q) {
          const double d = readlane_f64(a[mo][q], c0 + q);
          bad |= !(d > 0.0);
          const double rinv = rsqrt_refined(d);
          l[q] = a[mo][q] * rinv;
#pragma unroll
          for (int q2 = q + 1; q2 < GW; ++q2) a[mo][q2] = __builtin_fma(-l[q], readlane_f64(l[q], c0 + q2), a[mo][q2]);
          sL[(c0 + q) * LS + lane] = l[q];
          if (lane == 0) sRv[c0 + q] = rinv;
        }
      }
      lds_barrier();
      if (g + 1 < NG) {
        double lrow[GW];
#pragma unroll
        for (int q = 0; q < GW; ++q) lrow[q] = sL[(c0 + q) * LS + lane];
#pragma unroll
        for (int m = mo; m < MG; ++m) {
          // (a group that is already factored takes the update too: its registers are dead, and the code stays free of
          //  wavefront-dependent branches)
          const int cg = GW * (4 * m + wave);
#pragma unroll
          for (int q2 = 0; q2 < GW; ++q2) {
            double acc = a[m][q2];
#pragma unroll
            for (int q = 0; q < GW; ++q) acc = __builtin_fma(-lrow[q], sL[(c0 + q) * LS + cg + q2], acc);
            a[m][q2] = acc;
          }
        }
      }
    }
    if (bad && lane == 0) atomicOr(fail, 1);
  } else {
    // lane c < 16: column c of the inverse of the current 16 x 16 diagonal block, rows as their L columns appear
    double acc[16];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      lds_barrier();
#pragma unroll
      for (int q = 0; q < GW; ++q) {
        const int R = GW * g + q, b = R >> 4, r = R & 15;
        if (r == 0) {
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[k] = 0.0;
        }
        if (lane < 16) {
          const double x = (((lane == r) ? 1.0 : 0.0) - acc[r]) * sRv[R];
          sX[R * XS + 16 * b + lane] = x;
#pragma unroll
          for (int k2 = r + 1; k2 < 16; ++k2) acc[k2] = __builtin_fma(sL[R * LS + 16 * b + k2], x, acc[k2]);
        }
      }
    }
  }
  inverse_rest_and_store(sL, sX, sQ, Tout, ld, inv);
}

// ---- V4: as V3 with GW = 4, but the 4 x 4 diagonal mini-block is read out of the owner's lanes once (readlane ->
// scalar registers) and factored by every lane redundantly; each lane then solves its own row against it.  No
// cross-lane traffic inside the column chain. ----
template <bool NEWTON1>
__device__ __forceinline__ double rsq_pick(double d) {
  if (!NEWTON1) return rsqrt_refined(d);
  const double y = __builtin_amdgcn_rsq(d);
  const double e = __builtin_fma(-d * y, y, 1.0);
  return __builtin_fma(0.5 * y, e, y);
}

template <bool NEWTON1, bool SCHED>
__device__ __forceinline__ void tile_v4(const double *sB, double *sL, double *sRv, double *sX, double *sQ, double *Tout, int ld, double *inv, int *fail) {
  constexpr int GW = 4, NG = NB / GW, MG = NG / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < NB * XS; i += blockDim.x) sX[i] = 0.0;
  STAMP(0);
  if (wave < 4) {
    double a[MG][GW];
#pragma unroll
    for (int m = 0; m < MG; ++m)
#pragma unroll
      for (int q = 0; q < GW; ++q) a[m][q] = sB[lane * kStageLd + GW * (4 * m + wave) + q];
    bool bad = false;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int mo = g >> 2, c0 = GW * g;
      TR(0);
      if (wave == (g & 3)) {
        // the mini-block, lower triangle: m[r][c] from lane c0 + r
        const double m00 = readlane_f64(a[mo][0], c0);
        const double m10 = readlane_f64(a[mo][0], c0 + 1), m11 = readlane_f64(a[mo][1], c0 + 1);
        const double m20 = readlane_f64(a[mo][0], c0 + 2), m21 = readlane_f64(a[mo][1], c0 + 2), m22 = readlane_f64(a[mo][2], c0 + 2);
        const double m30 = readlane_f64(a[mo][0], c0 + 3), m31 = readlane_f64(a[mo][1], c0 + 3), m32 = readlane_f64(a[mo][2], c0 + 3),
                     m33 = readlane_f64(a[mo][3], c0 + 3);
        const double r0 = rsq_pick<NEWTON1>(m00);
        const double x0 = a[mo][0] * r0;
        const double l10 = m10 * r0, l20 = m20 * r0, l30 = m30 * r0;
        const double d1 = __builtin_fma(-l10, l10, m11);
        const double r1 = rsq_pick<NEWTON1>(d1);
        const double x1 = __builtin_fma(-x0, l10, a[mo][1]) * r1;
        const double l21 = __builtin_fma(-l20, l10, m21) * r1, l31 = __builtin_fma(-l30, l10, m31) * r1;
        const double d2 = __builtin_fma(-l21, l21, __builtin_fma(-l20, l20, m22));
        const double r2 = rsq_pick<NEWTON1>(d2);
        const double x2 = __builtin_fma(-x1, l21, __builtin_fma(-x0, l20, a[mo][2])) * r2;
        const double l32 = __builtin_fma(-l31, l21, __builtin_fma(-l30, l20, m32)) * r2;
        const double d3 = __builtin_fma(-l32, l32, __builtin_fma(-l31, l31, __builtin_fma(-l30, l30, m33)));
        const double r3 = rsq_pick<NEWTON1>(d3);
        const double x3 = __builtin_fma(-x2, l32, __builtin_fma(-x1, l31, __builtin_fma(-x0, l30, a[mo][3]))) * r3;
        bad |= !(m00 > 0.0) | !(d1 > 0.0) | !(d2 > 0.0) | !(d3 > 0.0);
        sL[(c0 + 0) * LS + lane] = x0; sL[(c0 + 1) * LS + lane] = x1; sL[(c0 + 2) * LS + lane] = x2; sL[(c0 + 3) * LS + lane] = x3;
        if (lane == 0) { sRv[c0] = r0; sRv[c0 + 1] = r1; sRv[c0 + 2] = r2; sRv[c0 + 3] = r3; }
      }
      TR(1);
      lds_barrier();
      TR(2);
      if (g + 1 < NG) {
        double lrow[GW];
#pragma unroll
        for (int q = 0; q < GW; ++q) lrow[q] = sL[(c0 + q) * LS + lane];
#pragma unroll
        for (int m = mo; m < MG; ++m) {
          // (a group that is already factored takes the update too: its registers are dead, and the code stays free of
          //  wavefront-dependent branches)
          const int cg = GW * (4 * m + wave);
#pragma unroll
          for (int q2 = 0; q2 < GW; ++q2) {
            double acc = a[m][q2];
#pragma unroll
            for (int q = 0; q < GW; ++q) acc = __builtin_fma(-lrow[q], sL[(c0 + q) * LS + cg + q2], acc);
            a[m][q2] = acc;
          }
          if (SCHED) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (bad && lane == 0) atomicOr(fail, 1);
  } else {
    double acc[16];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      lds_barrier();
#pragma unroll
      for (int q = 0; q < GW; ++q) {
        const int R = GW * g + q, b = R >> 4, r = R & 15;
        if (r == 0) {
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[k] = 0.0;
        }
        if (lane < 16) {
          const double x = (((lane == r) ? 1.0 : 0.0) - acc[r]) * sRv[R];
          sX[R * XS + 16 * b + lane] = x;
#pragma unroll
          for (int k2 = r + 1; k2 < 16; ++k2) acc[k2] = __builtin_fma(sL[R * LS + 16 * b + k2], x, acc[k2]);
        }
      }
    }
  }
  inverse_rest_and_store(sL, sX, sQ, Tout, ld, inv);
}

__device__ __forceinline__ void inverse_rest_and_store_v6(const double *sL, double *sX, double *sQ, const double *sRv, double *Tout, int ld, double *inv, int *fail) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  STAMP(1);
  lds_barrier();
  STAMP(2);
  // L is final: its stores drain while the matrix cores finish the inverse; a pivot that was not positive left a NaN or
  // an infinity in its reciprocal square root
  if (wave == 4) {      // the fifth wavefront has nothing else left to do
    for (int r = 0; r < NB; ++r) if (lane <= r) Tout[(size_t)r * ld + lane] = sL[lane * LS + r];
  }
  if (threadIdx.x < NB) { const double rv = sRv[threadIdx.x]; if (!(rv > 0.0) || !(rv < 1.0e300)) atomicOr(fail, 1); }
  const int li = lane & 15, lk = lane >> 4;
  // level 1: X(2p+1, 2p) = -X(2p+1, 2p+1) (L(2p+1, 2p) X(2p, 2p)), p = wavefront 0, 1
  if (wave < 2) {
    const int o = 32 * wave;
    double4_t P = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) P = mfma4(sL[(o + 4 * kk + lk) * LS + o + 16 + li], sX[(o + 4 * kk + lk) * XS + o + li], P);
    double *sP = sQ + wave * 16 * QS;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) sP[(lk + 4 * reg) * QS + li] = P[reg];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    double4_t D = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) D = mfma4(sX[(o + 16 + li) * XS + o + 16 + 4 * kk + lk], sP[(4 * kk + lk) * QS + li], D);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) sX[(o + 16 + lk + 4 * reg) * XS + o + li] = -D[reg];
  }
  lds_barrier();
  // level 2: X_BL = -X_BR (L_BL X_TL), 32 x 32 blocks, one 16 x 16 tile per wavefront
  if (wave < 4) {
    const int ti = wave >> 1, tj = wave & 1;
    double4_t Q = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) Q = mfma4(sL[(4 * kk + lk) * LS + 32 + 16 * ti + li], sX[(4 * kk + lk) * XS + 16 * tj + li], Q);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) sQ[(16 * ti + lk + 4 * reg) * QS + 16 * tj + li] = Q[reg];
  }
  lds_barrier();
  if (wave < 4) {
    const int ti = wave >> 1, tj = wave & 1;
    double4_t D = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) D = mfma4(sX[(32 + 16 * ti + li) * XS + 32 + 4 * kk + lk], sQ[(4 * kk + lk) * QS + 16 * tj + li], D);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) sX[(32 + 16 * ti + lk + 4 * reg) * XS + 16 * tj + li] = -D[reg];
  }
  lds_barrier();
  STAMP(3);
  for (int i = threadIdx.x; i < NB * NB; i += blockDim.x) {
    const int r = i >> 6, c = i & 63;
    inv[i] = sX[r * XS + c];
  }
}


template <bool NEWTON1, bool SCHED>
__device__ __forceinline__ void tile_v6(const double *sB, double *sL, double *sRv, double *sX, double *sQ, double *Tout, int ld, double *inv, int *fail) {
  constexpr int GW = 4, NG = NB / GW, MG = NG / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < NB * XS; i += blockDim.x) sX[i] = 0.0;
  STAMP(0);
  if (wave < 4) {
    double a[MG][GW];
#pragma unroll
    for (int m = 0; m < MG; ++m)
#pragma unroll
      for (int q = 0; q < GW; ++q) a[m][q] = sB[lane * kStageLd + GW * (4 * m + wave) + q];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int mo = g >> 2, c0 = GW * g;
      TR(0);
      if (wave == (g & 3)) {
        // the mini-block, lower triangle: m[r][c] from lane c0 + r
        const double m00 = readlane_f64(a[mo][0], c0);
        const double m10 = readlane_f64(a[mo][0], c0 + 1), m11 = readlane_f64(a[mo][1], c0 + 1);
        const double m20 = readlane_f64(a[mo][0], c0 + 2), m21 = readlane_f64(a[mo][1], c0 + 2), m22 = readlane_f64(a[mo][2], c0 + 2);
        const double m30 = readlane_f64(a[mo][0], c0 + 3), m31 = readlane_f64(a[mo][1], c0 + 3), m32 = readlane_f64(a[mo][2], c0 + 3),
                     m33 = readlane_f64(a[mo][3], c0 + 3);
        const double r0 = rsq_pick<NEWTON1>(m00);
        const double x0 = a[mo][0] * r0;
        const double l10 = m10 * r0, l20 = m20 * r0, l30 = m30 * r0;
        const double d1 = __builtin_fma(-l10, l10, m11);
        const double r1 = rsq_pick<NEWTON1>(d1);
        const double x1 = __builtin_fma(-x0, l10, a[mo][1]) * r1;
        const double l21 = __builtin_fma(-l20, l10, m21) * r1, l31 = __builtin_fma(-l30, l10, m31) * r1;
        const double d2 = __builtin_fma(-l21, l21, __builtin_fma(-l20, l20, m22));
        const double r2 = rsq_pick<NEWTON1>(d2);
        const double x2 = __builtin_fma(-x1, l21, __builtin_fma(-x0, l20, a[mo][2])) * r2;
        const double l32 = __builtin_fma(-l31, l21, __builtin_fma(-l30, l20, m32)) * r2;
        const double d3 = __builtin_fma(-l32, l32, __builtin_fma(-l31, l31, __builtin_fma(-l30, l30, m33)));
        const double r3 = rsq_pick<NEWTON1>(d3);
        const double x3 = __builtin_fma(-x2, l32, __builtin_fma(-x1, l31, __builtin_fma(-x0, l30, a[mo][3]))) * r3;
        sL[(c0 + 0) * LS + lane] = x0; sL[(c0 + 1) * LS + lane] = x1; sL[(c0 + 2) * LS + lane] = x2; sL[(c0 + 3) * LS + lane] = x3;
        if (lane == 0) { sRv[c0] = r0; sRv[c0 + 1] = r1; sRv[c0 + 2] = r2; sRv[c0 + 3] = r3; }
      }
      TR(1);
      lds_barrier();
      TR(2);
      if (g + 1 < NG) {
        double lrow[GW];
#pragma unroll
        for (int q = 0; q < GW; ++q) lrow[q] = sL[(c0 + q) * LS + lane];
#pragma unroll
        for (int m = mo; m < MG; ++m) {
          // (a group that is already factored takes the update too: its registers are dead, and the code stays free of
          //  wavefront-dependent branches)
          const int cg = GW * (4 * m + wave);
#pragma unroll
          for (int q2 = 0; q2 < GW; ++q2) {
            double acc = a[m][q2];
#pragma unroll
            for (int q = 0; q < GW; ++q) acc = __builtin_fma(-lrow[q], sL[(c0 + q) * LS + cg + q2], acc);
            a[m][q2] = acc;
          }
          if (SCHED) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else {
    double acc[16];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      lds_barrier();
#pragma unroll
      for (int q = 0; q < GW; ++q) {
        const int R = GW * g + q, b = R >> 4, r = R & 15;
        if (r == 0) {
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[k] = 0.0;
        }
        if (lane < 16) {
          const double x = (((lane == r) ? 1.0 : 0.0) - acc[r]) * sRv[R];
          sX[R * XS + 16 * b + lane] = x;
#pragma unroll
          for (int k2 = r + 1; k2 < 16; ++k2) acc[k2] = __builtin_fma(sL[R * LS + 16 * b + k2], x, acc[k2]);
        }
      }
    }
  }
  inverse_rest_and_store_v6(sL, sX, sQ, sRv, Tout, ld, inv, fail);
}

// ---- V5: V4 with the sixteen steps rolled into four rounds of four (one round = each wavefront factors one of its
// groups; its registers are then shifted down so that the current group is always a[0]): a quarter of the code, and
// rounds 2..4 run from a warm instruction cache. ----
template <bool NEWTON1>
__device__ __forceinline__ void tile_v5(const double *sB, double *sL, double *sRv, double *sX, double *sQ, double *Tout, int ld, double *inv, int *fail) {
  constexpr int GW = 4, MG = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < NB * XS; i += blockDim.x) sX[i] = 0.0;
  STAMP(0);
  if (wave < 4) {
    double a[MG][GW];
#pragma unroll
    for (int m = 0; m < MG; ++m)
#pragma unroll
      for (int q = 0; q < GW; ++q) a[m][q] = sB[lane * kStageLd + GW * (4 * m + wave) + q];
    bool bad = false;
#pragma unroll 1
    for (int round = 0; round < 4; ++round) {
#pragma unroll
      for (int ow = 0; ow < 4; ++ow) {
        const int c0 = GW * (4 * round + ow);
        if (wave == ow) {
          const double m00 = readlane_f64(a[0][0], c0);
          const double m10 = readlane_f64(a[0][0], c0 + 1), m11 = readlane_f64(a[0][1], c0 + 1);
          const double m20 = readlane_f64(a[0][0], c0 + 2), m21 = readlane_f64(a[0][1], c0 + 2), m22 = readlane_f64(a[0][2], c0 + 2);
          const double m30 = readlane_f64(a[0][0], c0 + 3), m31 = readlane_f64(a[0][1], c0 + 3), m32 = readlane_f64(a[0][2], c0 + 3),
                       m33 = readlane_f64(a[0][3], c0 + 3);
          const double r0 = rsq_pick<NEWTON1>(m00);
          const double x0 = a[0][0] * r0;
          const double l10 = m10 * r0, l20 = m20 * r0, l30 = m30 * r0;
          const double d1 = __builtin_fma(-l10, l10, m11);
          const double r1 = rsq_pick<NEWTON1>(d1);
          const double x1 = __builtin_fma(-x0, l10, a[0][1]) * r1;
          const double l21 = __builtin_fma(-l20, l10, m21) * r1, l31 = __builtin_fma(-l30, l10, m31) * r1;
          const double d2 = __builtin_fma(-l21, l21, __builtin_fma(-l20, l20, m22));
          const double r2 = rsq_pick<NEWTON1>(d2);
          const double x2 = __builtin_fma(-x1, l21, __builtin_fma(-x0, l20, a[0][2])) * r2;
          const double l32 = __builtin_fma(-l31, l21, __builtin_fma(-l30, l20, m32)) * r2;
          const double d3 = __builtin_fma(-l32, l32, __builtin_fma(-l31, l31, __builtin_fma(-l30, l30, m33)));
          const double r3 = rsq_pick<NEWTON1>(d3);
          const double x3 = __builtin_fma(-x2, l32, __builtin_fma(-x1, l31, __builtin_fma(-x0, l30, a[0][3]))) * r3;
          bad |= !(m00 > 0.0) | !(d1 > 0.0) | !(d2 > 0.0) | !(d3 > 0.0);
          sL[(c0 + 0) * LS + lane] = x0; sL[(c0 + 1) * LS + lane] = x1; sL[(c0 + 2) * LS + lane] = x2; sL[(c0 + 3) * LS + lane] = x3;
          if (lane == 0) { sRv[c0] = r0; sRv[c0 + 1] = r1; sRv[c0 + 2] = r2; sRv[c0 + 3] = r3; }
        }
        lds_barrier();
        double lrow[GW];
#pragma unroll
        for (int q = 0; q < GW; ++q) lrow[q] = sL[(c0 + q) * LS + lane];
        // a[m] = group 4 (round + m) + wave; the factored ones (a[0] of wavefronts <= ow) take the update too, dead values
#pragma unroll
        for (int m = 0; m < MG; ++m) {
          if (round + m < 4) {
            const int cg = GW * (4 * (round + m) + wave);
#pragma unroll
            for (int q2 = 0; q2 < GW; ++q2) {
              double acc = a[m][q2];
#pragma unroll
              for (int q = 0; q < GW; ++q) acc = __builtin_fma(-lrow[q], sL[(c0 + q) * LS + cg + q2], acc);
              a[m][q2] = acc;
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int m = 0; m + 1 < MG; ++m)
#pragma unroll
        for (int q = 0; q < GW; ++q) a[m][q] = a[m + 1][q];
    }
    if (bad && lane == 0) atomicOr(fail, 1);
  } else {
    // lane c < 16: column c of the inverse of the round's 16 x 16 diagonal block
#pragma unroll 1
    for (int round = 0; round < 4; ++round) {
      double acc[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] = 0.0;
#pragma unroll
      for (int ow = 0; ow < 4; ++ow) {
        lds_barrier();
#pragma unroll
        for (int q = 0; q < GW; ++q) {
          const int r = GW * ow + q, R = 16 * round + r;
          if (lane < 16) {
            const double x = (((lane == r) ? 1.0 : 0.0) - acc[r]) * sRv[R];
            sX[R * XS + 16 * round + lane] = x;
#pragma unroll
            for (int k2 = r + 1; k2 < 16; ++k2) acc[k2] = __builtin_fma(sL[R * LS + 16 * round + k2], x, acc[k2]);
          }
        }
      }
    }
  }
  inverse_rest_and_store(sL, sX, sQ, Tout, ld, inv);
}

template <int VARIANT>
__global__ void __launch_bounds__(320) bench_kernel(const double *A, double *Tout, double *inv, int *fail, int reps, long long *cycles) {
  __shared__ __attribute__((aligned(16))) double sB[NB * kStageLd];
  __shared__ __attribute__((aligned(16))) double sCol[2][NB];
  __shared__ __attribute__((aligned(16))) double sPan[2 * 8 * NB];
  __shared__ double sRinv[2];
  __shared__ __attribute__((aligned(16))) double sL[NB * LS + 128], sX[NB * XS], sQ[32 * QS], sRv[NB];
  for (int i = threadIdx.x; i < NB * NB; i += blockDim.x) sB[(i / NB) * kStageLd + (i % NB)] = A[i];
  __syncthreads();
  const long long t0 = wall_clock64();
  STAMP(0);
  {
    if (VARIANT == 0) tile_v0<true>(sB, sCol, sRinv, Tout, NB, inv, fail);
    if (VARIANT == 1) tile_v0<false>(sB, sCol, sRinv, Tout, NB, inv, fail);
    if (VARIANT == 2) tile_v2<4>(sB, sPan, Tout, NB, fail);
    if (VARIANT == 3) tile_v2<2>(sB, sPan, Tout, NB, fail);
    if (VARIANT == 4) tile_v2<8>(sB, sPan, Tout, NB, fail);
    if (VARIANT == 5) tile_v3<4>(sB, sL, sRv, sX, sQ, Tout, NB, inv, fail);
    if (VARIANT == 6) tile_v3<8>(sB, sL, sRv, sX, sQ, Tout, NB, inv, fail);
    if (VARIANT == 7) tile_v3<16>(sB, sL, sRv, sX, sQ, Tout, NB, inv, fail);
    if (VARIANT == 8) tile_v4<false, true>(sB, sL, sRv, sX, sQ, Tout, NB, inv, fail);
    if (VARIANT == 11) tile_v4<false, false>(sB, sL, sRv, sX, sQ, Tout, NB, inv, fail);
    if (VARIANT == 12) tile_v4<true, true>(sB, sL, sRv, sX, sQ, Tout, NB, inv, fail);
    if (VARIANT == 13) tile_v6<false, false>(sB, sL, sRv, sX, sQ, Tout, NB, inv, fail);
    if (VARIANT == 9) tile_v5<false>(sB, sL, sRv, sX, sQ, Tout, NB, inv, fail);
    if (VARIANT == 10) tile_v5<true>(sB, sL, sRv, sX, sQ, Tout, NB, inv, fail);
    __syncthreads();
  }
  STAMP(4);
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) *cycles = t1 - t0;
}

int main() {
  std::vector<double> M(NB * NB), A(NB * NB, 0.0), L(NB * NB, 0.0), X(NB * NB, 0.0);
  srand(1);
  for (auto &v : M) v = rand() / (double)RAND_MAX * 2 - 1;
  for (int i = 0; i < NB; ++i)
    for (int j = 0; j < NB; ++j) {
      double s = (i == j) ? 0.5 : 0.0;
      for (int k = 0; k < NB; ++k) s += M[i * NB + k] * M[j * NB + k];
      A[i * NB + j] = s;
    }
  // host reference
  for (int j = 0; j < NB; ++j) {
    double d = A[j * NB + j];
    for (int k = 0; k < j; ++k) d -= L[j * NB + k] * L[j * NB + k];
    L[j * NB + j] = std::sqrt(d);
    for (int i = j + 1; i < NB; ++i) {
      double s = A[i * NB + j];
      for (int k = 0; k < j; ++k) s -= L[i * NB + k] * L[j * NB + k];
      L[i * NB + j] = s / L[j * NB + j];
    }
  }
  for (int c = 0; c < NB; ++c)
    for (int r = 0; r < NB; ++r) {
      double s = (r == c) ? 1.0 : 0.0;
      for (int k = 0; k < r; ++k) s -= L[r * NB + k] * X[k * NB + c];
      X[r * NB + c] = s / L[r * NB + r];
    }
  double *dA, *dT, *dI; int *dF; long long *dC;
  CHK(hipMalloc(&dA, NB * NB * 8)); CHK(hipMalloc(&dT, NB * NB * 8)); CHK(hipMalloc(&dI, NB * NB * 8));
  CHK(hipMalloc(&dF, 4)); CHK(hipMalloc(&dC, 8));
  CHK(hipMemcpy(dA, A.data(), NB * NB * 8, hipMemcpyHostToDevice));
  const int reps = 30;
  auto run = [&](int variant, const char *name, bool has_inv) {
    CHK(hipMemset(dT, 0, NB * NB * 8)); CHK(hipMemset(dI, 0, NB * NB * 8)); CHK(hipMemset(dF, 0, 4));
    long long best = 1LL << 60, sum = 0;
    for (int pass = 0; pass < reps; ++pass) {
      if (variant == 0) hipLaunchKernelGGL(bench_kernel<0>, dim3(1), dim3(320), 0, 0, dA, dT, dI, dF, reps, dC);
      if (variant == 1) hipLaunchKernelGGL(bench_kernel<1>, dim3(1), dim3(320), 0, 0, dA, dT, dI, dF, reps, dC);
      if (variant == 2) hipLaunchKernelGGL(bench_kernel<2>, dim3(1), dim3(256), 0, 0, dA, dT, dI, dF, reps, dC);
      if (variant == 3) hipLaunchKernelGGL(bench_kernel<3>, dim3(1), dim3(256), 0, 0, dA, dT, dI, dF, reps, dC);
      if (variant == 4) hipLaunchKernelGGL(bench_kernel<4>, dim3(1), dim3(256), 0, 0, dA, dT, dI, dF, reps, dC);
      if (variant == 5) hipLaunchKernelGGL(bench_kernel<5>, dim3(1), dim3(320), 0, 0, dA, dT, dI, dF, reps, dC);
      if (variant == 6) hipLaunchKernelGGL(bench_kernel<6>, dim3(1), dim3(320), 0, 0, dA, dT, dI, dF, reps, dC);
      if (variant == 7) hipLaunchKernelGGL(bench_kernel<7>, dim3(1), dim3(320), 0, 0, dA, dT, dI, dF, reps, dC);
      if (variant == 8) hipLaunchKernelGGL(bench_kernel<8>, dim3(1), dim3(320), 0, 0, dA, dT, dI, dF, reps, dC);
      if (variant == 9) hipLaunchKernelGGL(bench_kernel<9>, dim3(1), dim3(320), 0, 0, dA, dT, dI, dF, reps, dC);
      if (variant == 11) hipLaunchKernelGGL(bench_kernel<11>, dim3(1), dim3(320), 0, 0, dA, dT, dI, dF, reps, dC);
      if (variant == 12) hipLaunchKernelGGL(bench_kernel<12>, dim3(1), dim3(320), 0, 0, dA, dT, dI, dF, reps, dC);
      if (variant == 13) hipLaunchKernelGGL(bench_kernel<13>, dim3(1), dim3(320), 0, 0, dA, dT, dI, dF, reps, dC);
      if (variant == 10) hipLaunchKernelGGL(bench_kernel<10>, dim3(1), dim3(320), 0, 0, dA, dT, dI, dF, reps, dC);
      CHK(hipDeviceSynchronize());
      long long c1;
      CHK(hipMemcpy(&c1, dC, 8, hipMemcpyDeviceToHost));
      if (pass > 0) { best = c1 < best ? c1 : best; sum += c1; }
    }
    long long cyc = sum; int f;
    std::vector<double> T(NB * NB), Iv(NB * NB);
    CHK(hipMemcpy(&f, dF, 4, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(T.data(), dT, NB * NB * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(Iv.data(), dI, NB * NB * 8, hipMemcpyDeviceToHost));
    double eL = 0, eI = 0;
    for (int i = 0; i < NB; ++i)
      for (int j = 0; j <= i; ++j) {
        eL = std::fmax(eL, std::fabs(T[i * NB + j] - L[i * NB + j]));
        eI = std::fmax(eI, std::fabs(Iv[i * NB + j] - X[i * NB + j]));
      }
    long long st[8];
    CHK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof st));
    if (false) {
      long long tr[4][16][4];
      CHK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_trace), sizeof tr));
      for (int g = 0; g < 16; ++g) {
        std::printf("   g %2d owner %d:", g, g & 3);
        for (int w = 0; w < 4; ++w) std::printf("  w%d top->pre-barrier %5lld, barrier %5lld |", w, tr[w][g][1] - tr[w][g][0], tr[w][g][2] - tr[w][g][1]);
        if (g) std::printf("  owner cycle %lld", tr[g & 3][g][2] - tr[(g - 1) & 3][g - 1][2]);
        std::printf("\n");
      }
    }
    if (variant >= 5) std::printf("   cycles: chain %lld, wait for wave 4 %lld, rest of inverse %lld, stores %lld\n", st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3]);
    std::printf("%-40s mean %7.3f  best %7.3f us per tile   max|L err| %.2e   max|inv err| %.2e   fail %d\n", name, cyc / 100.0 / (reps - 1),
                best / 100.0, eL, has_inv ? eI : -1.0, f);
  };
  run(0, "v0 five wavefronts, 1 col/barrier", true);
  run(1, "v1 = v0 without the inverse", false);
  run(3, "v2 two cols/barrier, no inverse", false);
  run(2, "v2 four cols/barrier, no inverse", false);
  run(4, "v2 eight cols/barrier, no inverse", false);
  run(5, "v3 GW=4, LDS-resident L, MFMA inverse", true);
  run(6, "v3 GW=8", true);
  run(7, "v3 GW=16", true);
  run(8, "v4 replicated 4x4 mini-block", true);
  run(9, "v5 = v4 rolled into 4 rounds", true);
  run(10, "v5 + one-Newton rsqrt", true);
  run(11, "v4 without sched barriers", true);
  run(12, "v4 + one-Newton rsqrt", true);
  run(13, "v6 = v4, no per-step checks, L stored early", true);
  return 0;
}
