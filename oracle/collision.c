/* collision.c -- CPU ORACLE (test infrastructure): restatement of the
 * reference's box-ground and box-box contact generation,
 * eggshell/collision.cc:408-436 and :166-388 (with helpers :53-164), used to
 * produce and to cross-check the synthetic contact sets.  R is row-major here;
 * col(R,j) is the j-th column (a box axis in world coordinates). */
#include <float.h>
#include <stdlib.h>
#include <string.h>

#include "egs_oracle.h"
#include "linalg.h"

static inline void colv(const double *R, int j, double *o) {
  o[0] = R[j]; o[1] = R[3 + j]; o[2] = R[6 + j];
}
static inline double sgn(double a) { return (a >= 0) ? 1.0 : -1.0; } /* :29-31 */

/* collision.cc:408-436 */
int orc_collide_box_ground(const double c[3], const double R[9],
                           const double side[3], double *contacts) {
  int n = 0;
  double c0[3], c1[3], c2[3];
  colv(R, 0, c0); colv(R, 1, c1); colv(R, 2, c2);
  for (int x = -1; x <= 1; x += 2)
    for (int y = -1; y <= 1; y += 2)
      for (int z = -1; z <= 1; z += 2) {
        double v[3];
        for (int k = 0; k < 3; ++k)
          v[k] = ((c[k] + c0[k] * side[0] * 0.5 * x) + c1[k] * side[1] * 0.5 * y) +
                 c2[k] * side[2] * 0.5 * z;
        if (v[2] < 0) {
          double *o = contacts + 7 * n++;
          o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
          o[3] = 0; o[4] = 0; o[5] = 1;
          o[6] = -v[2];
        }
      }
  return n;
}

typedef struct { double center[3]; double R[9]; double half[3]; } box_t;
typedef struct { double x, y; } v2;

/* :53-69 */
static void line_closest_approach(const double *pa, const double *ua,
                                  const double *pb, const double *ub,
                                  double *alpha, double *beta) {
  double p[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
  double uaub = dot3(ua, ub);
  double q1 = dot3(ua, p);
  double q2 = -dot3(ub, p);
  double d = 1 - uaub * uaub;
  if (d == 0) { *alpha = 0; *beta = 0; }
  else { *alpha = (q1 + uaub * q2) / d; *beta = (uaub * q1 + q2) / d; }
}

/* :77-87 */
static int seg_line(v2 p1, v2 p2, v2 nrm, double d, v2 *p) {
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

/* :91-106 */
static int clip_poly(const v2 *poly, int n, v2 nrm, double d, v2 *out) {
  int k = 0;
  for (int i = 0; i < n; ++i) {
    if ((nrm.x * poly[i].x + nrm.y * poly[i].y) + d >= 0) out[k++] = poly[i];
    v2 np;
    if (seg_line(poly[i], poly[(i + 1) % n], nrm, d, &np)) out[k++] = np;
  }
  return k;
}

/* :112-164: intersect box B with rectangle Rc; polygon in Rc's frame. */
static int box_rect(const box_t *B, const box_t *Rc, v2 *poly) {
  const double kTol = 1e-9;
  double Bc[3] = {B->center[0] - Rc->center[0], B->center[1] - Rc->center[1],
                  B->center[2] - Rc->center[2]};
  int n = 4;
  poly[0] = (v2){-Rc->half[0], -Rc->half[1]};
  poly[1] = (v2){-Rc->half[0], Rc->half[1]};
  poly[2] = (v2){Rc->half[0], Rc->half[1]};
  poly[3] = (v2){Rc->half[0], -Rc->half[1]};
  v2 tmp[32];
  double Rn[3], r0[3], r1[3];
  colv(Rc->R, 2, Rn); colv(Rc->R, 0, r0); colv(Rc->R, 1, r1);
  for (int i = 0; i < 3; ++i) {
    double Bn[3], cr[3];
    colv(B->R, i, Bn);
    double BnBc = dot3(Bn, Bc);
    cross3(Bn, Rn, cr);
    double crossn = sqrt(dot3(cr, cr));
    for (int j = -1; j <= 1; j += 2) {
      double Bd = -j * BnBc - B->half[i];
      if (crossn < kTol) {
        if (Bd <= 0) continue;
        return 0;
      }
      v2 H = {dot3(r0, Bn), dot3(r1, Bn)};
      v2 Hn = {-j * H.x, -j * H.y};
      int k = clip_poly(poly, n, Hn, -Bd, tmp);
      for (int q = 0; q < k; ++q) poly[q] = tmp[q];
      n = k;
      if (n == 0) return 0;
    }
  }
  return n;
}

/* test hooks: the helpers above, so that tests/test_oracle_collision.py can restate
 * the reference's own tests of them (collision.cc:527-681) */
void orc_line_closest_approach(const double pa[3], const double ua[3], const double pb[3],
                               const double ub[3], double *alpha, double *beta) {
  line_closest_approach(pa, ua, pb, ub, alpha, beta);
}
int orc_clip_polygon(const double *poly_xy, int n, const double normal[2], double d, double *out_xy) {
  v2 in[32], out[64];
  if (n > 32) return -1;
  for (int i = 0; i < n; ++i) in[i] = (v2){poly_xy[2 * i], poly_xy[2 * i + 1]};
  int k = clip_poly(in, n, (v2){normal[0], normal[1]}, d, out);
  for (int i = 0; i < k; ++i) { out_xy[2 * i] = out[i].x; out_xy[2 * i + 1] = out[i].y; }
  return k;
}
/* box B vs rectangle (centre, R, half[0..1]); polygon [<=32][2] in the rectangle's frame */
int orc_box_rectangle(const double bc[3], const double bR[9], const double bhalf[3], const double rc[3],
                      const double rR[9], const double rhalf[2], double *poly_xy) {
  box_t B, Rc;
  for (int k = 0; k < 3; ++k) { B.center[k] = bc[k]; B.half[k] = bhalf[k]; Rc.center[k] = rc[k]; }
  memcpy(B.R, bR, sizeof B.R); memcpy(Rc.R, rR, sizeof Rc.R);
  Rc.half[0] = rhalf[0]; Rc.half[1] = rhalf[1]; Rc.half[2] = 0;
  v2 poly[32];
  int n = box_rect(&B, &Rc, poly);
  for (int i = 0; i < n; ++i) { poly_xy[2 * i] = poly[i].x; poly_xy[2 * i + 1] = poly[i].y; }
  return n;
}

/* collision.cc:166-388 */
static int collide_boxes_impl(const double c1[3], const double R1[9],
                              const double s1[3], const double c2[3],
                              const double R2[9], const double s2[3], double *contacts,
                              int max_contacts, int *code_out, double *info /* axis[3], depth; or NULL */) {
  const double kAlign = 0.9962, kTol = 1e-9;
  box_t box1, box2;
  for (int k = 0; k < 3; ++k) {
    box1.center[k] = c1[k]; box2.center[k] = c2[k];
    box1.half[k] = s1[k] * 0.5; box2.half[k] = s2[k] * 0.5;
  }
  memcpy(box1.R, R1, sizeof box1.R);
  memcpy(box2.R, R2, sizeof box2.R);
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
  const double *H1 = box1.half, *H2 = box2.half;
  double min_FN = -DBL_MAX, sep_FN[3] = {0, 0, 0};
  int code_FN = 0;
#define RR(i, j) R[3 * (i) + (j)]
#define QQ(i, j) Q[3 * (i) + (j)]
#define SEPF(e1expr, e2expr, Rsrc, col, thecode)                       \
  {                                                                    \
    double e1 = (e1expr);                                              \
    double separation = fabs(e1) - (e2expr);                           \
    if (separation > 0) { if (code_out) *code_out = 0; return 0; }     \
    if (separation > min_FN) {                                         \
      min_FN = separation;                                             \
      double nn[3]; colv(Rsrc, col, nn);                               \
      double sg = sgn(e1);                                             \
      sep_FN[0] = sg * nn[0]; sep_FN[1] = sg * nn[1]; sep_FN[2] = sg * nn[2]; \
      code_FN = (thecode);                                             \
    }                                                                  \
  }
  /* H2.dot(Q.row(i)), H1.dot(Q.col(j)), R.col(j).dot(p) */
  SEPF(p[0], H1[0] + ((H2[0] * QQ(0, 0) + H2[1] * QQ(0, 1)) + H2[2] * QQ(0, 2)), R1, 0, 1)
  SEPF(p[1], H1[1] + ((H2[0] * QQ(1, 0) + H2[1] * QQ(1, 1)) + H2[2] * QQ(1, 2)), R1, 1, 2)
  SEPF(p[2], H1[2] + ((H2[0] * QQ(2, 0) + H2[1] * QQ(2, 1)) + H2[2] * QQ(2, 2)), R1, 2, 3)
  SEPF((RR(0, 0) * p[0] + RR(1, 0) * p[1]) + RR(2, 0) * p[2],
       ((H1[0] * QQ(0, 0) + H1[1] * QQ(1, 0)) + H1[2] * QQ(2, 0)) + H2[0], R2, 0, 4)
  SEPF((RR(0, 1) * p[0] + RR(1, 1) * p[1]) + RR(2, 1) * p[2],
       ((H1[0] * QQ(0, 1) + H1[1] * QQ(1, 1)) + H1[2] * QQ(2, 1)) + H2[1], R2, 1, 5)
  SEPF((RR(0, 2) * p[0] + RR(1, 2) * p[1]) + RR(2, 2) * p[2],
       ((H1[0] * QQ(0, 2) + H1[1] * QQ(1, 2)) + H1[2] * QQ(2, 2)) + H2[2], R2, 2, 6)
#undef SEPF
  double min_EE = -DBL_MAX, sep_EE[3] = {0, 0, 0};
  int code_EE = 0;
#define SEPE(e1expr, e2expr, n0, n1, n2, thecode)                      \
  {                                                                    \
    double nv[3] = {(n0), (n1), (n2)};                                 \
    double len = sqrt(dot3(nv, nv));                                   \
    if (len > kTol) {                                                  \
      double e1 = (e1expr);                                            \
      double separation = fabs(e1) - (e2expr);                         \
      if (separation > 0) { if (code_out) *code_out = 0; return 0; }   \
      separation /= len;                                               \
      if (separation > min_EE) {                                       \
        min_EE = separation;                                           \
        double dn = sgn(e1) * len;                                     \
        sep_EE[0] = nv[0] / dn; sep_EE[1] = nv[1] / dn; sep_EE[2] = nv[2] / dn; \
        code_EE = (thecode);                                           \
      }                                                                \
    }                                                                  \
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
  /* The reference CHECKs code_FN != 0 && code_EE != 0 (:264) and would Panic
   * for exactly axis-aligned boxes (every edge x edge axis degenerate).  The
   * oracle instead treats "no valid EE axis" as "FN is best". */
  {
    double t[3];
    mat3_vec(R1, sep_EE, t);
    sep_EE[0] = t[0]; sep_EE[1] = t[1]; sep_EE[2] = t[2];
  }
  int best_FN = (code_EE == 0) ? 1 : (min_FN > min_EE);
  if (info) { /* CollisionInfo, collision.cc:283-291 */
    const double *ax = best_FN ? sep_FN : sep_EE;
    info[0] = ax[0]; info[1] = ax[1]; info[2] = ax[2];
    info[3] = best_FN ? -min_FN : -min_EE;
  }
  int n = 0;
  if (aacount == 0 && !best_FN) { /* edge-edge, :278-301 */
    if (code_out) *code_out = code_EE;
    double pa[3], pb[3];
    for (int k = 0; k < 3; ++k) { pa[k] = c1[k]; pb[k] = c2[k]; }
    for (int j = 0; j < 3; ++j) {
      double a1[3], a2[3];
      colv(R1, j, a1); colv(R2, j, a2);
      double sa = sgn(dot3(sep_EE, a1)), sb = sgn(dot3(sep_EE, a2));
      for (int k = 0; k < 3; ++k) {
        pa[k] += sa * H1[j] * a1[k];
        pb[k] -= sb * H2[j] * a2[k];
      }
    }
    double ua[3], ub[3], alpha, beta;
    colv(R1, (code_EE - 7) / 3, ua);
    colv(R2, (code_EE - 7) % 3, ub);
    line_closest_approach(pa, ua, pb, ub, &alpha, &beta);
    double *o = contacts;
    for (int k = 0; k < 3; ++k) {
      o[k] = (pa[k] + ua[k] * alpha + pb[k] + ub[k] * beta) * 0.5;
      o[3 + k] = sep_EE[k];
    }
    o[6] = -min_EE;
    return 1;
  }
  /* face-something, :303-388 */
  if (code_out) *code_out = code_FN;
  const box_t *A = (code_FN <= 3) ? &box1 : &box2;
  box_t B = (code_FN <= 3) ? box2 : box1;
  double sgnA = (code_FN <= 3) ? 1.0 : -1.0;
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
    for (int k = 0; k < 3; ++k) B.center[k] += Bn[k] * B.half[nfi];
    colv(B.R, (nfi + 1) % 3, a0); colv(B.R, (nfi + 2) % 3, a1); colv(B.R, nfi, a2);
    for (int k = 0; k < 3; ++k) { BR[3 * k] = a0[k]; BR[3 * k + 1] = a1[k]; BR[3 * k + 2] = a2[k]; }
    double h0 = B.half[(nfi + 1) % 3], h1 = B.half[(nfi + 2) % 3];
    memcpy(B.R, BR, sizeof BR);
    B.half[0] = h0; B.half[1] = h1; B.half[2] = 0;
  }
  double Afc[3];
  for (int k = 0; k < 3; ++k) Afc[k] = A->center[k] + An[k] * A->half[(code_FN - 1) % 3];
  double Ad = -dot3(An, Afc);
  v2 poly[32];
  int np = box_rect(A, &B, poly);
  double b0[3], b1[3];
  colv(B.R, 0, b0); colv(B.R, 1, b1);
  for (int i = 0; i < np && n < max_contacts; ++i) {
    double pos[3];
    for (int k = 0; k < 3; ++k) pos[k] = (B.center[k] + b0[k] * poly[i].x) + b1[k] * poly[i].y;
    double depth = -(dot3(An, pos) + Ad);
    if (fabs(depth) > kTol || aacount >= 2) {
      double *o = contacts + 7 * n++;
      for (int k = 0; k < 3; ++k) { o[k] = pos[k]; o[3 + k] = sep_FN[k]; }
      o[6] = depth;
    }
  }
  if (n == 0) { /* :378-386 */
    double *o = contacts;
    for (int k = 0; k < 3; ++k) { o[k] = c2[k]; o[3 + k] = sep_FN[k]; }
    o[6] = -min_FN;
    if (code_out) *code_out = 16;
    n = 1;
  }
  return n;
}

int orc_collide_boxes(const double c1[3], const double R1[9],
                      const double s1[3], const double c2[3],
                      const double R2[9], const double s2[3], double *contacts,
                      int max_contacts, int *code_out) {
  return collide_boxes_impl(c1, R1, s1, c2, R2, s2, contacts, max_contacts, code_out, 0);
}

/* the same, also returning CollisionInfo {separating_axis, depth} (collision.h:29-38) */
int orc_collide_boxes_info(const double c1[3], const double R1[9],
                           const double s1[3], const double c2[3],
                           const double R2[9], const double s2[3], double *contacts,
                           int max_contacts, int *code_out, double info[4]) {
  return collide_boxes_impl(c1, R1, s1, c2, R2, s2, contacts, max_contacts, code_out, info);
}
