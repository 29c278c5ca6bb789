// collide.hip -- contact generation on the GPU: Ensemble::UpdateContacts
// (ensembles.cc:445-480) + the contact-vs-contact pruning of
// CheckAndCorrectEnsembleState (ensembles.cc:241-329), i.e. the step that feeds
// the constraint solve.
//
//   box-ground   collision.cc:408-436   (8 vertices, z < 0)
//   box-box      collision.cc:166-388   (15-axis SAT, face clipping, edge-edge)
//
// The contact LIST ORDER is part of the result (the solver sweeps in list
// order): all ground contacts by body, then body pairs i < j lexicographically,
// each pair's contacts in the order the reference emits them.  The GPU keeps it
// with counts + exclusive scans instead of push_back:
//   1. ground_count / cand_kernel     per body: ground contacts; candidate j > i
//      (bounding spheres overlap; ordered by wave ballot).  From 2048 bodies up
//      the candidates come from a hashed uniform grid instead of all pairs
//      (cand_grid_kernel; same sphere test, same ascending order)
//   2. flatten candidates (scan)      -> pair list in (i, j) order
//   3. narrow_kernel<false>           per pair: SAT + clipping, count kept contacts
//   4. scan, narrow_kernel<true> + ground emit  -> final arrays
// Arithmetic follows oracle/collision.c operation by operation (fp64,
// -ffp-contract=off), so the contact set is bit-identical to the CPU oracle's.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "collide.h"

namespace egs {

namespace {

constexpr int KMAX = 64;      // candidate partners j > i per body
constexpr int MAXC = 16;      // contacts per pair before pruning

struct HipErr : std::runtime_error { using std::runtime_error::runtime_error; };
void chk(hipError_t e, const char *what) {
  if (e != hipSuccess) throw HipErr(std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(call) chk((call), #call)

template <typename T>
struct Buf {
  T *p = nullptr;
  explicit Buf(size_t n) { if (n) HIPCHK(hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T))); }
  ~Buf() { if (p) (void)hipFree(p); }
  Buf(const Buf &) = delete;
  Buf &operator=(const Buf &) = delete;
};

struct Box { double c[3]; double R[9]; double h[3]; };
struct V2 { double x, y; };

__device__ __forceinline__ double dot3(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
__device__ __forceinline__ void cross3(const double *a, const double *b, double *o) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
__device__ __forceinline__ void mat3_vec(const double *A, const double *v, double *o) {
  double x = (A[0] * v[0] + A[1] * v[1]) + A[2] * v[2];
  double y = (A[3] * v[0] + A[4] * v[1]) + A[5] * v[2];
  double z = (A[6] * v[0] + A[7] * v[1]) + A[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
__device__ __forceinline__ void mat3_mul(const double *A, const double *B, double *O) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) O[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void mat3_T(const double *A, double *O) {
  O[0] = A[0]; O[1] = A[3]; O[2] = A[6]; O[3] = A[1]; O[4] = A[4]; O[5] = A[7]; O[6] = A[2]; O[7] = A[5]; O[8] = A[8];
}
__device__ __forceinline__ void colv(const double *R, int j, double *o) { o[0] = R[j]; o[1] = R[3 + j]; o[2] = R[6 + j]; }
__device__ __forceinline__ double sgn(double a) { return (a >= 0) ? 1.0 : -1.0; }

// collision.cc:408-436; returns the count, optionally writes [<=8][7]
__device__ int ground_contacts(const double *c, const double *R, const double *side, double *out) {
  int n = 0;
  double c0[3], c1[3], c2[3];
  colv(R, 0, c0); colv(R, 1, c1); colv(R, 2, c2);
  for (int x = -1; x <= 1; x += 2)
    for (int y = -1; y <= 1; y += 2)
      for (int z = -1; z <= 1; z += 2) {
        double v[3];
        for (int k = 0; k < 3; ++k)
          v[k] = ((c[k] + c0[k] * side[0] * 0.5 * x) + c1[k] * side[1] * 0.5 * y) + c2[k] * side[2] * 0.5 * z;
        if (v[2] < 0) {
          if (out) {
            double *o = out + 7 * n;
            o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = 0; o[4] = 0; o[5] = 1; o[6] = -v[2];
          }
          ++n;
        }
      }
  return n;
}

__device__ void line_closest_approach(const double *pa, const double *ua, const double *pb, const double *ub,
                                      double *alpha, double *beta) {  // collision.cc:53-69
  double p[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
  double uaub = dot3(ua, ub), q1 = dot3(ua, p), q2 = -dot3(ub, p);
  double d = 1 - uaub * uaub;
  if (d == 0) { *alpha = 0; *beta = 0; }
  else { *alpha = (q1 + uaub * q2) / d; *beta = (uaub * q1 + q2) / d; }
}

__device__ int seg_line(V2 p1, V2 p2, V2 nrm, double d, V2 *p) {  // collision.cc:77-87
  double k1 = (nrm.x * p1.x + nrm.y * p1.y) + d;
  double k2 = (nrm.x * p2.x + nrm.y * p2.y) + d;
  if (k1 * k2 < 0) {
    double t = k1 / (k2 - k1);
    p->x = p1.x - t * (p2.x - p1.x);
    p->y = p1.y - t * (p2.y - p1.y);
    return 1;
  }
  return 0;
}

__device__ int clip_poly(const V2 *poly, int n, V2 nrm, double d, V2 *out) {  // collision.cc:91-106
  int k = 0;
  for (int i = 0; i < n; ++i) {
    if ((nrm.x * poly[i].x + nrm.y * poly[i].y) + d >= 0) out[k++] = poly[i];
    V2 np;
    if (seg_line(poly[i], poly[(i + 1) % n], nrm, d, &np)) out[k++] = np;
  }
  return k;
}

__device__ int box_rect(const Box *B, const Box *Rc, V2 *poly) {  // collision.cc:112-164
  const double kTol = 1e-9;
  double Bc[3] = {B->c[0] - Rc->c[0], B->c[1] - Rc->c[1], B->c[2] - Rc->c[2]};
  int n = 4;
  poly[0] = V2{-Rc->h[0], -Rc->h[1]};
  poly[1] = V2{-Rc->h[0], Rc->h[1]};
  poly[2] = V2{Rc->h[0], Rc->h[1]};
  poly[3] = V2{Rc->h[0], -Rc->h[1]};
  V2 tmp[24];
  double Rn[3], r0[3], r1[3];
  colv(Rc->R, 2, Rn); colv(Rc->R, 0, r0); colv(Rc->R, 1, r1);
  for (int i = 0; i < 3; ++i) {
    double Bn[3], cr[3];
    colv(B->R, i, Bn);
    double BnBc = dot3(Bn, Bc);
    cross3(Bn, Rn, cr);
    double crossn = sqrt(dot3(cr, cr));
    for (int j = -1; j <= 1; j += 2) {
      double Bd = -j * BnBc - B->h[i];
      if (crossn < kTol) {
        if (Bd <= 0) continue;
        return 0;
      }
      V2 H = {dot3(r0, Bn), dot3(r1, Bn)};
      V2 Hn = {-j * H.x, -j * H.y};
      int k = clip_poly(poly, n, Hn, -Bd, tmp);
      for (int q = 0; q < k; ++q) poly[q] = tmp[q];
      n = k;
      if (n == 0) return 0;
    }
  }
  return n;
}

// collision.cc:166-388.  contacts [<= MAXC][7]; returns the number generated.
__device__ int collide_boxes(const double *c1, const double *R1, const double *s1, const double *c2,
                             const double *R2, const double *s2, double *contacts) {
  const double kAlign = 0.9962, kTol = 1e-9;
  Box box1, box2;
  for (int k = 0; k < 3; ++k) { box1.c[k] = c1[k]; box2.c[k] = c2[k]; box1.h[k] = s1[k] * 0.5; box2.h[k] = s2[k] * 0.5; }
  for (int k = 0; k < 9; ++k) { box1.R[k] = R1[k]; box2.R[k] = R2[k]; }
  double R1t[9], R[9], Q[9], p[3], dc[3];
  mat3_T(R1, R1t);
  mat3_mul(R1t, R2, R);
  for (int k = 0; k < 3; ++k) dc[k] = c2[k] - c1[k];
  mat3_vec(R1t, dc, p);
  for (int k = 0; k < 9; ++k) Q[k] = fabs(R[k]);
  int aacount = 0;
  for (int j = 0; j < 3; ++j) {
    double mx = Q[j];
    if (Q[3 + j] > mx) mx = Q[3 + j];
    if (Q[6 + j] > mx) mx = Q[6 + j];
    aacount += (mx > kAlign);
  }
  const double *H1 = box1.h, *H2 = box2.h;
  double min_FN = -DBL_MAX, sep_FN[3] = {0, 0, 0};
  int code_FN = 0;
#define RR(i, j) R[3 * (i) + (j)]
#define QQ(i, j) Q[3 * (i) + (j)]
#define SEPF(e1expr, e2expr, Rsrc, col, thecode)                                  \
  {                                                                               \
    double e1 = (e1expr);                                                         \
    double separation = fabs(e1) - (e2expr);                                      \
    if (separation > 0) return 0;                                                 \
    if (separation > min_FN) {                                                    \
      min_FN = separation;                                                        \
      double nn[3]; colv(Rsrc, col, nn);                                          \
      double sg = sgn(e1);                                                        \
      sep_FN[0] = sg * nn[0]; sep_FN[1] = sg * nn[1]; sep_FN[2] = sg * nn[2];      \
      code_FN = (thecode);                                                        \
    }                                                                             \
  }
  SEPF(p[0], H1[0] + ((H2[0] * QQ(0, 0) + H2[1] * QQ(0, 1)) + H2[2] * QQ(0, 2)), R1, 0, 1)
  SEPF(p[1], H1[1] + ((H2[0] * QQ(1, 0) + H2[1] * QQ(1, 1)) + H2[2] * QQ(1, 2)), R1, 1, 2)
  SEPF(p[2], H1[2] + ((H2[0] * QQ(2, 0) + H2[1] * QQ(2, 1)) + H2[2] * QQ(2, 2)), R1, 2, 3)
  SEPF((RR(0, 0) * p[0] + RR(1, 0) * p[1]) + RR(2, 0) * p[2], ((H1[0] * QQ(0, 0) + H1[1] * QQ(1, 0)) + H1[2] * QQ(2, 0)) + H2[0], R2, 0, 4)
  SEPF((RR(0, 1) * p[0] + RR(1, 1) * p[1]) + RR(2, 1) * p[2], ((H1[0] * QQ(0, 1) + H1[1] * QQ(1, 1)) + H1[2] * QQ(2, 1)) + H2[1], R2, 1, 5)
  SEPF((RR(0, 2) * p[0] + RR(1, 2) * p[1]) + RR(2, 2) * p[2], ((H1[0] * QQ(0, 2) + H1[1] * QQ(1, 2)) + H1[2] * QQ(2, 2)) + H2[2], R2, 2, 6)
#undef SEPF
  double min_EE = -DBL_MAX, sep_EE[3] = {0, 0, 0};
  int code_EE = 0;
#define SEPE(e1expr, e2expr, n0, n1, n2, thecode)                                 \
  {                                                                               \
    double nv[3] = {(n0), (n1), (n2)};                                            \
    double len = sqrt(dot3(nv, nv));                                              \
    if (len > kTol) {                                                             \
      double e1 = (e1expr);                                                       \
      double separation = fabs(e1) - (e2expr);                                    \
      if (separation > 0) return 0;                                               \
      separation /= len;                                                          \
      if (separation > min_EE) {                                                  \
        min_EE = separation;                                                      \
        double dn = sgn(e1) * len;                                                \
        sep_EE[0] = nv[0] / dn; sep_EE[1] = nv[1] / dn; sep_EE[2] = nv[2] / dn;    \
        code_EE = (thecode);                                                      \
      }                                                                           \
    }                                                                             \
  }
  SEPE(p[2] * RR(1, 0) - p[1] * RR(2, 0), (H1[1] * QQ(2, 0) + H1[2] * QQ(1, 0) + H2[1] * QQ(0, 2) + H2[2] * QQ(0, 1)), 0, -RR(2, 0), RR(1, 0), 7)
  SEPE(p[2] * RR(1, 1) - p[1] * RR(2, 1), (H1[1] * QQ(2, 1) + H1[2] * QQ(1, 1) + H2[0] * QQ(0, 2) + H2[2] * QQ(0, 0)), 0, -RR(2, 1), RR(1, 1), 8)
  SEPE(p[2] * RR(1, 2) - p[1] * RR(2, 2), (H1[1] * QQ(2, 2) + H1[2] * QQ(1, 2) + H2[0] * QQ(0, 1) + H2[1] * QQ(0, 0)), 0, -RR(2, 2), RR(1, 2), 9)
  SEPE(p[0] * RR(2, 0) - p[2] * RR(0, 0), (H1[0] * QQ(2, 0) + H1[2] * QQ(0, 0) + H2[1] * QQ(1, 2) + H2[2] * QQ(1, 1)), RR(2, 0), 0, -RR(0, 0), 10)
  SEPE(p[0] * RR(2, 1) - p[2] * RR(0, 1), (H1[0] * QQ(2, 1) + H1[2] * QQ(0, 1) + H2[0] * QQ(1, 2) + H2[2] * QQ(1, 0)), RR(2, 1), 0, -RR(0, 1), 11)
  SEPE(p[0] * RR(2, 2) - p[2] * RR(0, 2), (H1[0] * QQ(2, 2) + H1[2] * QQ(0, 2) + H2[0] * QQ(1, 1) + H2[1] * QQ(1, 0)), RR(2, 2), 0, -RR(0, 2), 12)
  SEPE(p[1] * RR(0, 0) - p[0] * RR(1, 0), (H1[0] * QQ(1, 0) + H1[1] * QQ(0, 0) + H2[1] * QQ(2, 2) + H2[2] * QQ(2, 1)), -RR(1, 0), RR(0, 0), 0, 13)
  SEPE(p[1] * RR(0, 1) - p[0] * RR(1, 1), (H1[0] * QQ(1, 1) + H1[1] * QQ(0, 1) + H2[0] * QQ(2, 2) + H2[2] * QQ(2, 0)), -RR(1, 1), RR(0, 1), 0, 14)
  SEPE(p[1] * RR(0, 2) - p[0] * RR(1, 2), (H1[0] * QQ(1, 2) + H1[1] * QQ(0, 2) + H2[0] * QQ(2, 1) + H2[1] * QQ(2, 0)), -RR(1, 2), RR(0, 2), 0, 15)
#undef SEPE
#undef RR
#undef QQ
  {
    double t[3];
    mat3_vec(R1, sep_EE, t);
    sep_EE[0] = t[0]; sep_EE[1] = t[1]; sep_EE[2] = t[2];
  }
  const int best_FN = (code_EE == 0) ? 1 : (min_FN > min_EE);
  int n = 0;
  if (aacount == 0 && !best_FN) {  // edge-edge, collision.cc:278-301
    double pa[3], pb[3];
    for (int k = 0; k < 3; ++k) { pa[k] = c1[k]; pb[k] = c2[k]; }
    for (int j = 0; j < 3; ++j) {
      double a1[3], a2[3];
      colv(R1, j, a1); colv(R2, j, a2);
      double sa = sgn(dot3(sep_EE, a1)), sb = sgn(dot3(sep_EE, a2));
      for (int k = 0; k < 3; ++k) { pa[k] += sa * H1[j] * a1[k]; pb[k] -= sb * H2[j] * a2[k]; }
    }
    double ua[3], ub[3], alpha, beta;
    colv(R1, (code_EE - 7) / 3, ua);
    colv(R2, (code_EE - 7) % 3, ub);
    line_closest_approach(pa, ua, pb, ub, &alpha, &beta);
    for (int k = 0; k < 3; ++k) {
      contacts[k] = (pa[k] + ua[k] * alpha + pb[k] + ub[k] * beta) * 0.5;
      contacts[3 + k] = sep_EE[k];
    }
    contacts[6] = -min_EE;
    return 1;
  }
  const Box *A = (code_FN <= 3) ? &box1 : &box2;
  Box B = (code_FN <= 3) ? box2 : box1;
  const double sgnA = (code_FN <= 3) ? 1.0 : -1.0;
  double An[3] = {sep_FN[0] * sgnA, sep_FN[1] * sgnA, sep_FN[2] * sgnA};
  double BRt[9], nf[3];
  mat3_T(B.R, BRt);
  mat3_vec(BRt, An, nf);
  int nfi = 0;
  {
    double best = fabs(nf[0]);
    if (fabs(nf[1]) > best) { best = fabs(nf[1]); nfi = 1; }
    if (fabs(nf[2]) > best) { best = fabs(nf[2]); nfi = 2; }
  }
  double Bn[3], bcol[3];
  colv(B.R, nfi, bcol);
  for (int k = 0; k < 3; ++k) Bn[k] = -sgn(nf[nfi]) * bcol[k];
  {
    double BR[9], a0[3], a1[3], a2[3];
    for (int k = 0; k < 3; ++k) B.c[k] += Bn[k] * B.h[nfi];
    colv(B.R, (nfi + 1) % 3, a0); colv(B.R, (nfi + 2) % 3, a1); colv(B.R, nfi, a2);
    for (int k = 0; k < 3; ++k) { BR[3 * k] = a0[k]; BR[3 * k + 1] = a1[k]; BR[3 * k + 2] = a2[k]; }
    const double h0 = B.h[(nfi + 1) % 3], h1 = B.h[(nfi + 2) % 3];
    for (int k = 0; k < 9; ++k) B.R[k] = BR[k];
    B.h[0] = h0; B.h[1] = h1; B.h[2] = 0;
  }
  double Afc[3];
  for (int k = 0; k < 3; ++k) Afc[k] = A->c[k] + An[k] * A->h[(code_FN - 1) % 3];
  const double Ad = -dot3(An, Afc);
  V2 poly[24];
  const int np = box_rect(A, &B, poly);
  double b0[3], b1[3];
  colv(B.R, 0, b0); colv(B.R, 1, b1);
  for (int i = 0; i < np && n < MAXC; ++i) {
    double pos[3];
    for (int k = 0; k < 3; ++k) pos[k] = (B.c[k] + b0[k] * poly[i].x) + b1[k] * poly[i].y;
    const double depth = -(dot3(An, pos) + Ad);
    if (fabs(depth) > kTol || aacount >= 2) {
      double *o = contacts + 7 * n++;
      for (int k = 0; k < 3; ++k) { o[k] = pos[k]; o[3 + k] = sep_FN[k]; }
      o[6] = depth;
    }
  }
  if (n == 0) {  // collision.cc:378-386
    for (int k = 0; k < 3; ++k) { contacts[k] = c2[k]; contacts[3 + k] = sep_FN[k]; }
    contacts[6] = -min_FN;
    n = 1;
  }
  return n;
}

// ---- kernels ---------------------------------------------------------------
__global__ void __launch_bounds__(256) ground_kernel(int n, const double *pos, const double *R, const double *side,
                                                     const int *off, int *count, int *b0, int *b1, double *data) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n) return;
  double buf[8 * 7];
  const int c = ground_contacts(pos + 3 * (size_t)b, R + 9 * (size_t)b, side + 3 * (size_t)b, off ? buf : nullptr);
  if (!off) { count[b] = c; return; }
  for (int k = 0; k < c; ++k) {
    const size_t o = (size_t)off[b] + k;
    b0[o] = -1; b1[o] = b;  // contact.h:13-15: ground contacts are (null, body)
    for (int q = 0; q < 7; ++q) data[o * 7 + q] = buf[7 * k + q];
  }
}

// One wavefront per body i: candidates j > i whose bounding spheres overlap, in
// ascending j (ballot keeps the order).
__global__ void __launch_bounds__(64) cand_kernel(int n, const double *pos, const double *side, int *cand, int *count,
                                                  int *overflow) {
  const int i = blockIdx.x, lane = threadIdx.x;
  const double ci[3] = {pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2]};
  const double ri = 0.5 * sqrt(dot3(side + 3 * (size_t)i, side + 3 * (size_t)i));
  int found = 0;
  for (int base = i + 1; base < n; base += 64) {
    const int j = base + lane;
    bool hit = false;
    if (j < n) {
      const double d[3] = {pos[3 * (size_t)j] - ci[0], pos[3 * (size_t)j + 1] - ci[1], pos[3 * (size_t)j + 2] - ci[2]};
      const double rj = 0.5 * sqrt(dot3(side + 3 * (size_t)j, side + 3 * (size_t)j));
      const double rr = (ri + rj) * 1.0000001 + 1e-12;  // conservative: never drops a touching pair
      hit = dot3(d, d) <= rr * rr;
    }
    const unsigned long long mask = __ballot(hit);
    if (hit) {
      const int k = found + __popcll(mask & ((1ull << lane) - 1ull));
      if (k < KMAX) cand[(size_t)i * KMAX + k] = j;
    }
    found += __popcll(mask);
  }
  if (lane == 0) {
    if (found > KMAX) { atomicOr(overflow, 1); found = KMAX; }
    count[i] = found;
  }
}

// The same test without the KMAX cap, for the rare scene in which some body has more than KMAX
// partners (a big plate under many small boxes): FILL = false counts, FILL = true writes the pairs of
// body i at off[i] in ascending j -- the order of the capped kernels, so the contact list is unchanged.
template <bool FILL>
__global__ void __launch_bounds__(64) cand_all_kernel(int n, const double *pos, const double *side, const int *off,
                                                      int *count, int *pi, int *pj) {
  const int i = blockIdx.x, lane = threadIdx.x;
  const double ci[3] = {pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2]};
  const double ri = 0.5 * sqrt(dot3(side + 3 * (size_t)i, side + 3 * (size_t)i));
  int found = 0;
  for (int base = i + 1; base < n; base += 64) {
    const int j = base + lane;
    bool hit = false;
    if (j < n) {
      const double d[3] = {pos[3 * (size_t)j] - ci[0], pos[3 * (size_t)j + 1] - ci[1], pos[3 * (size_t)j + 2] - ci[2]};
      const double rj = 0.5 * sqrt(dot3(side + 3 * (size_t)j, side + 3 * (size_t)j));
      const double rr = (ri + rj) * 1.0000001 + 1e-12;
      hit = dot3(d, d) <= rr * rr;
    }
    const unsigned long long mask = __ballot(hit);
    if (FILL && hit) {
      const int k = off[i] + found + __popcll(mask & ((1ull << lane) - 1ull));
      pi[k] = i; pj[k] = j;
    }
    found += __popcll(mask);
  }
  if (!FILL && lane == 0) count[i] = found;
}

// ---- uniform-grid broad phase (large n) ---------------------------------------
// Cell edge >= the largest rr of the sphere test, so a touching pair is always in
// adjacent cells.  Bodies are binned into a hashed table (count -> scan -> fill);
// one wavefront per body i then visits its 27 neighbour cells, keeps j > i that
// pass the SAME sphere test as cand_kernel, and sorts the (<= 64) survivors
// ascending -- so the candidate list, hence the contact order, is identical to
// the all-pairs kernel's.
constexpr int kCellClamp = 1 << 20;

__device__ __forceinline__ unsigned cell_hash(int x, int y, int z) {
  return (unsigned)x * 73856093u ^ (unsigned)y * 19349663u ^ (unsigned)z * 83492791u;
}

__global__ void __launch_bounds__(1024) cell_size_kernel(int n, const double *side, double *cell) {
  __shared__ double red[1024];
  double r = 0.0;
  for (int b = threadIdx.x; b < n; b += 1024) r = fmax(r, 0.5 * sqrt(dot3(side + 3 * (size_t)b, side + 3 * (size_t)b)));
  red[threadIdx.x] = r;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
  if (threadIdx.x == 0) cell[0] = 2.0 * red[0] * 1.000001 + 1e-9;
}

__global__ void __launch_bounds__(256) cell_bin_kernel(int n, const double *pos, const double *cell, int table_mask,
                                                       int *coords, int *bucket_of, int *arrival, int *tcount) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n) return;
  const double c = cell[0];
  int q[3];
  for (int k = 0; k < 3; ++k) {
    double f = floor(pos[3 * (size_t)b + k] / c);
    f = fmin(fmax(f, (double)-kCellClamp), (double)kCellClamp);   // monotone, so neighbours stay neighbours
    q[k] = (int)f;
    coords[3 * (size_t)b + k] = q[k];
  }
  const int bkt = (int)(cell_hash(q[0], q[1], q[2]) & (unsigned)table_mask);
  bucket_of[b] = bkt;
  arrival[b] = atomicAdd(&tcount[bkt], 1);   // order inside a bucket is irrelevant: candidates are sorted later
}

__global__ void __launch_bounds__(256) cell_fill_kernel(int n, const int *bucket_of, const int *arrival, const int *toff,
                                                        int *sorted) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b < n) sorted[toff[bucket_of[b]] + arrival[b]] = b;
}

__global__ void __launch_bounds__(64) cand_grid_kernel(int n, const double *pos, const double *side, const int *coords,
                                                       int table_mask, const int *toff, const int *tcount,
                                                       const int *sorted, int *cand, int *count, int *overflow) {
  __shared__ int list[KMAX];
  const int i = blockIdx.x, lane = threadIdx.x;
  const double ci[3] = {pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2]};
  const double ri = 0.5 * sqrt(dot3(side + 3 * (size_t)i, side + 3 * (size_t)i));
  const int cx = coords[3 * (size_t)i], cy = coords[3 * (size_t)i + 1], cz = coords[3 * (size_t)i + 2];
  int found = 0;
  for (int nb = 0; nb < 27; ++nb) {
    const int x = cx + nb % 3 - 1, y = cy + (nb / 3) % 3 - 1, z = cz + nb / 9 - 1;
    const int bkt = (int)(cell_hash(x, y, z) & (unsigned)table_mask);
    const int beg = toff[bkt], cnt = tcount[bkt];
    for (int base = 0; base < cnt; base += 64) {
      const int k = base + lane;
      bool hit = false;
      int j = -1;
      if (k < cnt) {
        j = sorted[beg + k];
        // exact cell match: a bucket shared by two cells (hash collision) is never counted twice
        if (j > i && coords[3 * (size_t)j] == x && coords[3 * (size_t)j + 1] == y && coords[3 * (size_t)j + 2] == z) {
          const double d[3] = {pos[3 * (size_t)j] - ci[0], pos[3 * (size_t)j + 1] - ci[1], pos[3 * (size_t)j + 2] - ci[2]};
          const double rj = 0.5 * sqrt(dot3(side + 3 * (size_t)j, side + 3 * (size_t)j));
          const double rr = (ri + rj) * 1.0000001 + 1e-12;
          hit = dot3(d, d) <= rr * rr;
        }
      }
      const unsigned long long mask = __ballot(hit);
      if (hit) {
        const int idx = found + __popcll(mask & ((1ull << lane) - 1ull));
        if (idx < KMAX) list[idx] = j;
      }
      found += __popcll(mask);
    }
  }
  __syncthreads();
  const int nf = found < KMAX ? found : KMAX;
  const int c = lane < nf ? list[lane] : 0x7fffffff;
  int rank = 0;
  for (int k = 0; k < nf; ++k) rank += (__shfl(c, k) < c) ? 1 : 0;   // candidates are distinct bodies
  if (lane < nf) cand[(size_t)i * KMAX + rank] = c;
  if (lane == 0) {
    if (found > KMAX) atomicOr(overflow, 1);
    count[i] = nf;
  }
}

__global__ void __launch_bounds__(256) flatten_kernel(int n, const int *cand, const int *count, const int *off, int *pi,
                                                      int *pj) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < count[i]; ++k) { pi[off[i] + k] = i; pj[off[i] + k] = cand[(size_t)i * KMAX + k]; }
}

// Narrow phase + contact-vs-contact pruning within the pair
// (ensembles.cc:308-316, kMinConstraintDistance = 1e-6: a contact is deleted if
// ANY earlier contact of the same pair lies within 1e-6 of it).
// Joints between the same two bodies prune contacts within 1e-6 of the joint
// position (ensembles.cc:296-306; Joint::GetConstraintPosition, joints.cc:57-75).
struct JointList { int mj; const int *b0, *b1; const double *data; };

template <bool EMIT>
__global__ void __launch_bounds__(64) narrow_kernel(int npairs, const int *pi, const int *pj, const double *pos,
                                                    const double *R, const double *side, const int *off, int base,
                                                    int *count, int *b0, int *b1, double *data, JointList jl) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= npairs) return;
  const int i = pi[t], j = pj[t];
  double cs[MAXC * 7];
  const int nc = collide_boxes(pos + 3 * (size_t)i, R + 9 * (size_t)i, side + 3 * (size_t)i, pos + 3 * (size_t)j,
                               R + 9 * (size_t)j, side + 3 * (size_t)j, cs);
  int kept = 0;
  for (int a = 0; a < nc; ++a) {
    bool del = false;
    for (int q = 0; q < jl.mj; ++q) {
      const int a0 = jl.b0[q], a1 = jl.b1[q];
      if (!((a0 == i && a1 == j) || (a0 == j && a1 == i))) continue;
      double r0[3], r1[3];
      mat3_vec(R + 9 * (size_t)a0, jl.data + 7 * (size_t)q, r0);
      mat3_vec(R + 9 * (size_t)a1, jl.data + 7 * (size_t)q + 3, r1);
      double d[3];
      for (int k = 0; k < 3; ++k) {
        const double jp = ((pos[3 * (size_t)a0 + k] + r0[k]) + (pos[3 * (size_t)a1 + k] + r1[k])) / 2;
        d[k] = jp - cs[7 * a + k];
      }
      if (sqrt(dot3(d, d)) < 1e-6) del = true;
    }
    for (int b = 0; b < a; ++b) {
      const double d[3] = {cs[7 * b] - cs[7 * a], cs[7 * b + 1] - cs[7 * a + 1], cs[7 * b + 2] - cs[7 * a + 2]};
      if (sqrt(dot3(d, d)) < 1e-6) del = true;
    }
    if (del) continue;
    if (EMIT) {
      const size_t o = (size_t)base + off[t] + kept;
      b0[o] = i; b1[o] = j;
      for (int q = 0; q < 7; ++q) data[o * 7 + q] = cs[7 * a + q];
    }
    ++kept;
  }
  if (!EMIT) count[t] = kept;
}

// Copies the contact topology into device-visible (page-locked host) memory: a
// small PCIe write by the GPU is far cheaper than two DMA-engine copies.
__global__ void __launch_bounds__(256) export_pairs_kernel(int m, const int *b0, const int *b1, int *o0, int *o1) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) { o0[i] = b0[i]; o1[i] = b1[i]; }
}

// ---- exclusive scan (int32), three small kernels ----------------------------
constexpr int SCAN_CHUNK = 2048;  // elements per block (256 threads x 8)
__global__ void __launch_bounds__(256) scan_reduce_kernel(int n, const int *in, int *block_sums) {
  __shared__ int red[256];
  const int base = blockIdx.x * SCAN_CHUNK;
  int s = 0;
  for (int k = threadIdx.x; k < SCAN_CHUNK; k += 256) if (base + k < n) s += in[base + k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0];
}
__global__ void scan_blocks_kernel(int nblocks, int *block_sums, int *total) {  // one thread: nblocks is small
  int run = 0;
  for (int b = 0; b < nblocks; ++b) { const int v = block_sums[b]; block_sums[b] = run; run += v; }
  *total = run;
}
__global__ void __launch_bounds__(256) scan_apply_kernel(int n, const int *in, const int *block_sums, int *out) {
  __shared__ int part[256];
  const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * 8;
  int v[8], s = 0;
  for (int k = 0; k < 8; ++k) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { int run = block_sums[blockIdx.x]; for (int t = 0; t < 256; ++t) { const int x = part[t]; part[t] = run; run += x; } }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int k = 0; k < 8; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
}

// n <= 16384: the whole scan in one 1024-thread workgroup (one launch instead of three)
constexpr int SCAN_SMALL = 16384;
__global__ void __launch_bounds__(1024) scan_small_kernel(int n, const int *in, int *out, int *total) {
  __shared__ int part[1024];
  const int per = (n + 1023) / 1024;            // contiguous elements per thread, <= 16
  const int base = threadIdx.x * per;
  int v[16], sum = 0;
  for (int k = 0; k < per; ++k) { v[k] = (base + k < n) ? in[base + k] : 0; sum += v[k]; }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {          // inclusive Hillis-Steele over the 1024 partial sums
    const int add = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  int run = part[threadIdx.x] - sum;            // exclusive prefix of this thread's chunk
  for (int k = 0; k < per; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
  if (threadIdx.x == 1023) *total = part[1023];
}

void exclusive_scan_async(hipStream_t s, int n, const int *in, int *out, int *scratch_blocks, int *total_d) {
  const int nblocks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  if (n <= 0) return;
  if (n <= SCAN_SMALL) {
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, s, n, in, out, total_d);
    return;
  }
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(nblocks), dim3(256), 0, s, n, in, scratch_blocks);
  hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1), 0, s, nblocks, scratch_blocks, total_d);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nblocks), dim3(256), 0, s, n, in, scratch_blocks, out);
}

int exclusive_scan(hipStream_t s, int n, const int *in, int *out, int *scratch_blocks, int *total_d) {
  if (n <= 0) return 0;
  exclusive_scan_async(s, n, in, out, scratch_blocks, total_d);
  int total = 0;
  HIPCHK(hipMemcpyAsync(&total, total_d, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return total;
}

}  // namespace

// ---- device-resident collider ---------------------------------------------
namespace {
constexpr int kGridMinBodies = 2048;   // below this the all-pairs wave scan is a single cheap launch
bool use_grid(int n) {
  const char *e = std::getenv("EGS_BROADPHASE");   // "grid" / "pairs" force one broad phase (tests)
  if (e && !std::strcmp(e, "grid")) return true;
  if (e && !std::strcmp(e, "pairs")) return false;
  return n >= kGridMinBodies;
}
}  // namespace

struct Collider::Impl {
  // growable device buffers
  template <typename T>
  struct G {
    T *p = nullptr;
    size_t cap = 0;
    void need(size_t n) {
      if (n <= cap) return;
      if (p) (void)hipFree(p);
      p = nullptr;
      cap = n + n / 4 + 64;
      HIPCHK(hipMalloc(reinterpret_cast<void **>(&p), cap * sizeof(T)));
    }
    ~G() { if (p) (void)hipFree(p); }
  };
  G<int> gcount, goff, ccount, coff, cand, flags, blocks, pi, pj, pcount, poff, blocks2, b0, b1;
  G<int> coords, bucket_of, arrival, tcount, toff, sorted, blocks3, scratch;   // uniform grid
  G<double> data, cell;
};

Collider::Collider() : impl_(new Impl) {}
Collider::~Collider() { delete impl_; }
const int32_t *Collider::body0() const { return impl_->b0.p; }
const int32_t *Collider::body1() const { return impl_->b1.p; }
const double *Collider::data() const { return impl_->data.p; }

int Collider::run(hipStream_t s, int n, const double *dpos, const double *dR, const double *dside, int mj,
                  const int32_t *djb0, const int32_t *djb1, const double *djdata) {
  const JointList jl{mj, djb0, djb1, djdata};
  Impl &I = *impl_;
  n_ground_ = 0; n_pairs_ = 0;
  if (n <= 0) return 0;
  const size_t nn = (size_t)n;
  I.gcount.need(nn); I.goff.need(nn); I.ccount.need(nn); I.coff.need(nn); I.cand.need(nn * KMAX); I.flags.need(4);
  I.blocks.need((nn + SCAN_CHUNK - 1) / SCAN_CHUNK + 8); I.blocks2.need((nn + SCAN_CHUNK - 1) / SCAN_CHUNK + 8);
  // flags: [0] candidate overflow, [1] ground contacts, [2] candidate pairs, [3] pair contacts
  HIPCHK(hipMemsetAsync(I.flags.p, 0, 4 * sizeof(int), s));
  const int gb = (n + 255) / 256;
  hipLaunchKernelGGL(ground_kernel, dim3(gb), dim3(256), 0, s, n, dpos, dR, dside, (const int *)nullptr, I.gcount.p,
                     (int *)nullptr, (int *)nullptr, (double *)nullptr);
  exclusive_scan_async(s, n, I.gcount.p, I.goff.p, I.blocks.p, I.flags.p + 1);
  if (use_grid(n)) {
    int table = 1024;
    while (table < 2 * n) table <<= 1;
    I.cell.need(1); I.coords.need(nn * 3); I.bucket_of.need(nn); I.arrival.need(nn); I.sorted.need(nn);
    I.tcount.need((size_t)table); I.toff.need((size_t)table); I.blocks3.need((size_t)table / SCAN_CHUNK + 8);
    I.scratch.need(1);
    HIPCHK(hipMemsetAsync(I.tcount.p, 0, (size_t)table * sizeof(int), s));
    hipLaunchKernelGGL(cell_size_kernel, dim3(1), dim3(1024), 0, s, n, dside, I.cell.p);
    hipLaunchKernelGGL(cell_bin_kernel, dim3(gb), dim3(256), 0, s, n, dpos, I.cell.p, table - 1, I.coords.p, I.bucket_of.p,
                       I.arrival.p, I.tcount.p);
    exclusive_scan_async(s, table, I.tcount.p, I.toff.p, I.blocks3.p, I.scratch.p);
    hipLaunchKernelGGL(cell_fill_kernel, dim3(gb), dim3(256), 0, s, n, I.bucket_of.p, I.arrival.p, I.toff.p, I.sorted.p);
    hipLaunchKernelGGL(cand_grid_kernel, dim3(n), dim3(64), 0, s, n, dpos, dside, I.coords.p, table - 1, I.toff.p,
                       I.tcount.p, I.sorted.p, I.cand.p, I.ccount.p, I.flags.p);
  } else {
    hipLaunchKernelGGL(cand_kernel, dim3(n), dim3(64), 0, s, n, dpos, dside, I.cand.p, I.ccount.p, I.flags.p);
  }
  exclusive_scan_async(s, n, I.ccount.p, I.coff.p, I.blocks2.p, I.flags.p + 2);
  int totals[4] = {0, 0, 0, 0};   // ONE read-back for overflow, G and C
  HIPCHK(hipMemcpyAsync(totals, I.flags.p, 3 * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  const bool spill = totals[0] != 0;   // some body has more than KMAX partners: uncapped count -> scan -> fill
  if (spill) {
    hipLaunchKernelGGL((cand_all_kernel<false>), dim3(n), dim3(64), 0, s, n, dpos, dside, (const int *)nullptr, I.ccount.p,
                       (int *)nullptr, (int *)nullptr);
    totals[2] = exclusive_scan(s, n, I.ccount.p, I.coff.p, I.blocks2.p, I.flags.p + 2);
  }
  const int G = totals[1], C = totals[2];
  int P = 0;
  if (C > 0) {
    I.pi.need(C); I.pj.need(C); I.pcount.need(C); I.poff.need(C); I.blocks2.need(((size_t)C + SCAN_CHUNK - 1) / SCAN_CHUNK + 8);
    if (spill) hipLaunchKernelGGL((cand_all_kernel<true>), dim3(n), dim3(64), 0, s, n, dpos, dside, I.coff.p, (int *)nullptr, I.pi.p, I.pj.p);
    else
    hipLaunchKernelGGL(flatten_kernel, dim3(gb), dim3(256), 0, s, n, I.cand.p, I.ccount.p, I.coff.p, I.pi.p, I.pj.p);
    hipLaunchKernelGGL((narrow_kernel<false>), dim3((C + 63) / 64), dim3(64), 0, s, C, I.pi.p, I.pj.p, dpos, dR, dside,
                       (const int *)nullptr, 0, I.pcount.p, (int *)nullptr, (int *)nullptr, (double *)nullptr, jl);
    P = exclusive_scan(s, C, I.pcount.p, I.poff.p, I.blocks2.p, I.flags.p + 3);
  }
  const int m = G + P;
  n_ground_ = G; n_pairs_ = C;
  if (m == 0) return 0;
  I.b0.need(m); I.b1.need(m); I.data.need((size_t)m * 7);
  hipLaunchKernelGGL(ground_kernel, dim3(gb), dim3(256), 0, s, n, dpos, dR, dside, I.goff.p, (int *)nullptr, I.b0.p,
                     I.b1.p, I.data.p);
  if (C > 0)
    hipLaunchKernelGGL((narrow_kernel<true>), dim3((C + 63) / 64), dim3(64), 0, s, C, I.pi.p, I.pj.p, dpos, dR, dside,
                       I.poff.p, G, (int *)nullptr, I.b0.p, I.b1.p, I.data.p, jl);
  HIPCHK(hipGetLastError());
  return m;
}

void Collider::export_topology(hipStream_t s, int m, int32_t *mapped_b0, int32_t *mapped_b1) const {
  if (m <= 0) return;
  hipLaunchKernelGGL(export_pairs_kernel, dim3((m + 255) / 256), dim3(256), 0, s, m, impl_->b0.p, impl_->b1.p, mapped_b0,
                     mapped_b1);
  HIPCHK(hipGetLastError());
}

int update_contacts(hipStream_t s, int n, const double *pos, const double *R, const double *side, int max_contacts,
                    int32_t *body0, int32_t *body1, double *data, int *n_ground, int *n_pairs, int mj,
                    const int32_t *jb0, const int32_t *jb1, const double *jdata) {
  if (n_ground) *n_ground = 0;
  if (n_pairs) *n_pairs = 0;
  if (n <= 0) return 0;
  const size_t nn = (size_t)n;
  Buf<double> dpos(nn * 3), dR(nn * 9), dside(nn * 3);
  HIPCHK(hipMemcpyAsync(dpos.p, pos, nn * 3 * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(dR.p, R, nn * 9 * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(dside.p, side, nn * 3 * sizeof(double), hipMemcpyHostToDevice, s));
  Buf<int> djb0((size_t)mj), djb1((size_t)mj);
  Buf<double> djdata((size_t)mj * 7);
  if (mj > 0) {
    HIPCHK(hipMemcpyAsync(djb0.p, jb0, (size_t)mj * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(djb1.p, jb1, (size_t)mj * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(djdata.p, jdata, (size_t)mj * 7 * sizeof(double), hipMemcpyHostToDevice, s));
  }
  Collider col;
  const int m = col.run(s, n, dpos.p, dR.p, dside.p, mj, djb0.p, djb1.p, djdata.p);
  if (n_ground) *n_ground = col.n_ground();
  if (n_pairs) *n_pairs = col.n_pairs();
  if (m > max_contacts) throw std::invalid_argument("update_contacts: max_contacts too small (" + std::to_string(m) + " needed)");
  if (m == 0) return 0;
  HIPCHK(hipMemcpyAsync(body0, col.body0(), (size_t)m * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(body1, col.body1(), (size_t)m * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(data, col.data(), (size_t)m * 7 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return m;
}

}  // namespace egs
