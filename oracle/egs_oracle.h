/*
 * egs_oracle.h -- CPU ORACLE for the eggshell constraint-solve hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it.  The product
 * (eggshell_amd/) never links, imports or calls anything in oracle/.
 *
 * It is a from-scratch restatement, in plain C99 without Eigen, of the
 * reference algorithm (teenylasers/eggshell, snapshot 2025-05-09).  Each
 * function cites the reference file:line it follows.
 *
 * PARITY PINNING.  The reference cannot be compiled in this image (Eigen 3.3.8
 * is an un-vendored dependency and is absent), so the
 * oracle is pinned by
 *   - the reference's own literal test vectors (Murty 5x5 KAT, lcp.cc:348-389;
 *     utils literals, utils.cc:398-497),
 *   - the reference's own property tests restated in tests/ (matrix-free ==
 *     dense products, CheckMixedConstraintSolutions, Ax=b+w), and
 *   - an independent numpy dense implementation.
 * No golden lambda vector / iteration count / trajectory exists in the
 * reference, so bit-level parity of the iterative path against the reference
 * BINARY is "parity unpinned"; it is pinned at the algorithm level only.
 *
 * Conventions: all matrices row-major.  A constraint always has 3 rows
 * (joints.cc:18-19, contact.cc:103-105).  Body index -1 is the world.
 */
#ifndef EGS_ORACLE_H
#define EGS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constraint descriptors (what Constraint subclasses hold) ------------ */
enum { ORC_JOINT_BALL = 0, ORC_CONTACT_BOX = 1 };

/* Body state, SoA of plain arrays:  p[n][3], R[n][9] (row-major), v[n][3],
 * w[n][3], mass[n], I_body[n][9].                                            */

/* joints.cc:3-35 / contact.cc:14-117.  kind[m]; body0/body1[m];
 * data[m][7]: joint  = c0(3), c1(3), unused
 *             contact = position(3), normal(3), depth.
 * Outputs J0,J1 [m][18]; is_eq[3m]; lo,hi[3m]; err[3m].                      */
void orc_assemble(int n, const double *p, const double *R, int m,
                  const int32_t *kind, const int32_t *body0,
                  const int32_t *body1, const double *data, double *J0,
                  double *J1, uint8_t *is_eq, double *lo, double *hi,
                  double *err);

/* ensembles.cc:202-212: per-body 6x6 block of M^-1, [n][36]. */
void orc_minv_blocks(int n, const double *R, const double *mass,
                     const double *I_body, double *Minv);
/* ensembles.cc:214-222: f_ext[n][6] = (m g, -w x (I_g w)). */
void orc_external_force(int n, const double *R, const double *w,
                        const double *mass, const double *I_body,
                        double *f_ext);
/* ensembles.cc:569-570: rhs = -(erp/dt^2) err - J (v/dt + M^-1 f_ext). */
void orc_ode_rhs(int n, const double *v, const double *w, const double *Minv,
                 const double *f_ext, int m, const int32_t *body0,
                 const int32_t *body1, const double *J0, const double *J1,
                 const double *err, double dt, double erp, double *rhs);
/* ensembles.cc:535,572: vnew = v + dt M^-1 (f_ext + J^T lambda); [n][6]. */
void orc_velocity_update(int n, const double *v, const double *w,
                         const double *Minv, const double *f_ext, int m,
                         const int32_t *body0, const int32_t *body1,
                         const double *J0, const double *J1,
                         const double *lambda, double dt, double *vnew);
/* ensembles.cc:577-591 + utils.cc:82-89: midpoint position / rotation. */
void orc_position_update(int n, double *p, double *R, const double *v6_old,
                         const double *v6_new, double dt);

/* utils.cc:233-237 (Eigen Quaternion::FromTwoVectors + toRotationMatrix). */
void orc_align_vectors(const double a[3], const double b[3], double Rout[9]);
/* utils.cc:82-89. */
void orc_w_to_R(const double w[3], double dt, double Rout[9]);
/* ensembles.cc:668-707: Chain(num_links, anchor).  Fills p,R,v,w,mass,I_body
 * for n=num_links bodies and kind/body0/body1/data for m=num_links joints. */
void orc_chain(int num_links, const double anchor[3], double *p, double *R,
               double *v, double *w, double *mass, double *I_body,
               int32_t *kind, int32_t *body0, int32_t *body1, double *data);

/* ---- literal O(m^2) matrix-free algorithms (sparse_iterations_utils.cc) -- */
typedef struct {
  int n, m;
  const double *Minv;   /* [n][36] */
  const int32_t *body0, *body1;
  const double *J0, *J1; /* [m][18] */
  const uint8_t *is_eq;  /* [3m] */
  const double *lo, *hi; /* [3m] */
} orc_system;

void orc_lit_Lx(const orc_system *s, const double *x, double *out);    /* :427-493 */
void orc_lit_Ux(const orc_system *s, const double *x, double *out);    /* :495-561 */
void orc_lit_Dx(const orc_system *s, const double *x, double eps, double scale,
                double *out);                                          /* :571-603 */
void orc_lit_JMJtX(const orc_system *s, const double *x, double eps,
                   double *out);                                       /* :624-695 */
void orc_lit_solve_diag(const orc_system *s, const double *rhs, double eps,
                        double scale, double *x);                      /* :67-108 */
/* quirks!=0 reproduces Q1 (projection uses the neighbour's type/bounds). */
void orc_lit_solve_lower(const orc_system *s, const double *rhs, double eps,
                         double scale, int quirks, double *x);         /* :159-243 */
void orc_lit_solve_upper(const orc_system *s, const double *rhs, double eps,
                         double scale, int quirks, double *x);         /* :292-373 */
double orc_lit_residual(const orc_system *s, const double *rhs, const double *x,
                        double cfm);                 /* sparse_iterations.cc:51-69 */
/* sparse_iterations.cc:148-226. method 0 Jacobi, 1 GS, 2 SOR(backward).
 * Returns the number of iterations done. tol<=0: run exactly max_iters. */
int orc_lit_iterate(const orc_system *s, const double *rhs, double cfm,
                    int method, double omega, int max_iters, double tol,
                    int quirks, double *x, double *residual_out);

/* dense J M^-1 J^T + eps I, [3m][3m] row-major (ensembles.cc:510). */
void orc_dense_JMJt(const orc_system *s, double eps, double *A);

/* ---- fast O(nnz) sequential projected Jacobi/GS/SOR ---------------------- */
/* Same mathematics as orc_lit_iterate (corrected semantics, no Q1), body
 * accumulators a_b = M_b^-1 sum_i J_ib^T x_i.  The operation order below is
 * the order the HIP kernels use, so results are comparable bit for bit.
 * a_out (may be NULL): final accumulators [n][6].                           */
int orc_fast_iterate_f64(const orc_system *s, const double *rhs, double cfm,
                         int method, double omega, int max_iters, double tol,
                         int check_every, double *x, double *a_out,
                         double *residual_out);
int orc_fast_iterate_f32(int n, int m, const float *Minv, const int32_t *body0,
                         const int32_t *body1, const float *J0, const float *J1,
                         const uint8_t *is_eq, const float *lo, const float *hi,
                         const float *rhs, float cfm, int method, float omega,
                         int max_iters, float tol, int check_every, float *x,
                         float *a_out, float *residual_out);

/* w = A x - rhs row by row (sparse_iterations.cc:57) from x and the accumulators a[n][6]
 * orc_fast_iterate_f64 returned: the expression of the solve kernels' epilogue. */
void orc_fast_wres_f64(const orc_system *s, const double *rhs, double cfm, const double *x,
                       const double *a, double *w);

/* O(nnz) twin of the matrix-free products CalculateSparse{Lx,Ux,Dx,LxUx,UxDx,LxDx,JMJtX}
 * (sparse_iterations_utils.cc:427-695) in the operation order of the HIP mat-vec
 * kernels.  parts: bit 0 = L, bit 1 = U, bit 2 = D (sums as the reference adds
 * them); 8 = the full product with eps on the diagonal.                       */
void orc_fast_matvec_f64(const orc_system *s, const double *x, int parts, double eps,
                         double scale, double *y);
void orc_fast_matvec_f32(int n, int m, const float *Minv, const int32_t *body0, const int32_t *body1,
                         const float *J0, const float *J1, const float *x, int parts, float eps,
                         float scale, float *y);

/* ---- dense direct LCP (lcp.cc) ------------------------------------------- */
/* lcp.cc:157-274.  Returns 1 on success. pivots_out may be NULL. */
int orc_murty(int dim, const double *A, const double *b, const double *lo,
              const double *hi, double *x, double *w, int *pivots_out);
/* lcp.cc:20-103 (err as in the reference: 0 -> 1e-9). */
int orc_check_murty(int dim, const double *A, const double *b, const double *x,
                    const double *w, uint8_t *S, double *C, const double *lo,
                    const double *hi, double err);
/* lcp.cc:276-336: bounds are accepted and ignored (quirk Q3) unless
 * use_bounds!=0 (corrected box semantics). */
int orc_mixed_constraints(int dim, const double *A, const double *b,
                          const uint8_t *C, const double *lo, const double *hi,
                          int use_bounds, double *x, double *w,
                          int *pivots_out);

/* ---- collision (collision.cc), used to build the synthetic contact sets -- */
/* collision.cc:408-436. contacts: [<=8][7] = pos(3), normal(3), depth. */
/* ---- the iterations on an explicit dense matrix (dense_iter.c): sparse_iterations.cc:35-49, 72-144 and the dense
 *      twins of sparse_iterations_utils.cc:25-40, 110-128, 245-262.  A row-major n x n. */
double orc_dense_residual(int n, const double *A, const double *b, const double *x, const uint8_t *C, const double *lo,
                          const double *hi);
int orc_dense_iterate(int n, const double *A, const double *b, const uint8_t *C, const double *lo, const double *hi,
                      int method, double omega, int max_iters, double tol, double *x, double *residual_out);

/* ---- toolkit/lcp.cc: incremental-factor box LCP (lcp_toolkit.c).  Row-major n x n, lower triangle only. */
int otk_cholesky(double *L, int n);
void otk_lsolve(const double *L, int n, int m, double *x);
void otk_ltsolve(const double *L, int n, int m, double *x);
void otk_lltsolve(const double *L, int n, int m, double *x);
int otk_rank_update(double *L, int n, int i0, int p, const double *vec, double sigma, double *temp);
int otk_add_cholesky_row(const double *A, int n, int m, double *L);
int otk_swap_cholesky_rows(const double *A, int n, int i, int m, double *L, double *work);
void otk_swap_rows_and_columns(double *A, int n, int i, int j, int *perm);
int otk_box_dantzig(int n, double *A, const double *b, const double *lo, const double *hi,
                    double *x, double *w, int *perm_out, int *pivots);
int otk_box_murty(int n, double *A, const double *b, const double *lo, const double *hi, int max_iterations,
                  double *x, double *w, int *perm_out, int *iters);
/* SolveLCP_BoxSchur (toolkit/lcp.cc:627-747); nub_arg = -1 scans, >= 0 is the reference's test hook */
int otk_box_schur(int n, double *A, const double *b, const double *lo, const double *hi, int algorithm,
                  int max_iterations, int nub_arg, int q6, double *x, double *w, int *perm_out, int *nub_out, int *iters);

int orc_collide_box_ground(const double c[3], const double R[9],
                           const double side[3], double *contacts);
/* collision.cc:166-388. contacts [<=16][7]; code_out may be NULL. */
int orc_collide_boxes(const double c1[3], const double R1[9],
                      const double s1[3], const double c2[3],
                      const double R2[9], const double s2[3], double *contacts,
                      int max_contacts, int *code_out);

/* the same with CollisionInfo {separating_axis[3], depth} (collision.h) */
int orc_collide_boxes_info(const double c1[3], const double R1[9],
                           const double s1[3], const double c2[3],
                           const double R2[9], const double s2[3], double *contacts,
                           int max_contacts, int *code_out, double info[4]);
/* test hooks for collision.cc:53-164 */
void orc_line_closest_approach(const double pa[3], const double ua[3], const double pb[3],
                               const double ub[3], double *alpha, double *beta);
int orc_clip_polygon(const double *poly_xy, int n, const double normal[2], double d, double *out_xy);
int orc_box_rectangle(const double bc[3], const double bR[9], const double bhalf[3], const double rc[3],
                      const double rR[9], const double rhalf[2], double *poly_xy);

#ifdef __cplusplus
}
#endif
#endif
