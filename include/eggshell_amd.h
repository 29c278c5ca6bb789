/*
 * eggshell_amd.h -- C ABI of the MI355X (gfx950) constraint-solve library,
 * libeggshell_amd.so.  Plain pointers and sizes only; all arrays are host
 * memory, row-major, IEEE fp64 unless stated; the library owns device memory.
 *
 * It is a drop-in for ONE path of teenylasers/eggshell: the per-step
 * constraint solve.  Each entry point names the reference interface it
 * replaces (file:line relative to the reference tree).
 *
 * Conventions (as in the reference):
 *   - a body has 6 velocity coordinates [v_lin; omega] (ensembles.cc:429-436);
 *   - every constraint has exactly 3 rows (joints.cc:18-19, contact.cc:103);
 *   - body index -1 is the world (constraints.h:42-43);
 *   - rows are ordered as the ConstraintsList is (ensembles.cc:234-239);
 *   - is_eq[r] != 0 marks an equality row ("C" in the reference); inequality
 *     rows are clamped to [lo, hi], +-inf allowed
 *     (sparse_iterations_utils.cc:12-21).
 *
 * Error convention: every function returns an egs_status; nothing ever exits
 * the process (the reference Panics: toolkit/error.cc:92-100).  The message
 * for the last failure is egs_last_error().  Non-convergence of the iterative
 * solver is NOT an error, exactly as in the reference
 * (sparse_iterations.cc:208-225 returns the last iterate silently).
 */
#ifndef EGGSHELL_AMD_H
#define EGGSHELL_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct egs_context egs_context; /* one per (process, GPU): device + stream */
typedef struct egs_problem egs_problem; /* one ensemble (or batch) resident on the GPU */

typedef enum {
  EGS_OK = 0,
  EGS_ERR_INVALID = 1,     /* bad argument (the reference would CHECK/Panic) */
  EGS_ERR_NO_DEVICE = 2,   /* no gfx950 device / HIP runtime unusable */
  EGS_ERR_HIP = 3,         /* a HIP call failed */
  EGS_ERR_STALL = 4,       /* device-side ordering wait timed out (bug guard).  The device flag is
                              sticky: a stall in an asynchronous call (step / solve without stats)
                              is returned by the next call on the problem that enqueues work once
                              the flag has landed, and at the latest by the next call that
                              synchronises (get_lambda / get_velocity / get_state / get_stats ...);
                              reporting clears it.  egs_world_step checks before it integrates. */
  EGS_ERR_UNSUPPORTED = 5, /* feature not built yet */
  EGS_ERR_LCP_FAILED = 6,  /* dense LCP did not reach a solution (lcp.cc:250) */
  EGS_ERR_INTERNAL = 7     /* a library invariant failed / host allocation failed */
} egs_status;

/* sparse_iterations.cc:21-26 */
typedef enum { EGS_JACOBI = 0, EGS_GAUSS_SEIDEL = 1, EGS_SOR = 2 } egs_method;
typedef enum { EGS_F64 = 0, EGS_F32 = 1 } egs_precision;
/* constraint descriptor kinds for device-side assembly */
typedef enum { EGS_JOINT_BALL = 0, EGS_CONTACT_BOX = 1 } egs_constraint_kind;

/* Runtime form of the reference's compile-time constants:
 *   omega = 1.5 and max_iters = 500 (sparse_iterations.cc:15-19),
 *   tol = kAllowNumericalError = 1e-9 (constants.h:5),
 *   cfm = the `cfm` argument of sparse::*Iteration (sparse_iterations.h:26-34).
 * tol <= 0 runs exactly max_iters sweeps (the benchmark mode).
 * check_every = k evaluates the residual every k sweeps; k = 1 reproduces the
 * reference's stopping rule exactly. */
typedef struct {
  int32_t method;      /* egs_method */
  int32_t max_iters;
  int32_t check_every;
  int32_t reserved;
  double omega;
  double cfm;
  double tol;
} egs_solve_params;

typedef struct {
  int32_t iterations;  /* sweeps performed */
  int32_t status;      /* egs_status of the device run */
  double residual;     /* sparse_iterations.cc:51-69 metric of the returned x */
  int32_t n_islands;   /* connected components of the constraint graph */
  int32_t n_tiles;     /* workgroup-resident tiles */
  int32_t n_global;    /* constraints solved by the cross-workgroup path */
  int32_t reserved;    /* 1 = latency schedule (4 lanes per constraint) in use */
  int32_t schedule;    /* which kernels ran the last GS / SOR solve, egs_schedule_flags */
  int32_t tile_constraints; /* constraints per workgroup tile of that schedule */
} egs_solve_stats;

typedef enum {
  EGS_SCHED_QUAD = 1,          /* quad_solve_kernel: 4 lanes per constraint */
  EGS_SCHED_ISO = 2,           /* tile_solve_kernel, isotropic-body variant (no stored M^-1 J^T) */
  EGS_SCHED_QUAD_PATCHES = 4,  /* oversize islands: body patches on the 4-lane kernel */
  EGS_SCHED_LANE_PATCHES = 8,  /* oversize islands: body patches on the 1-lane kernel */
  EGS_SCHED_ALL_GLOBAL = 16,   /* oversize islands: the all-global kernel */
  EGS_SCHED_STATIC = 32,       /* step_solve_kernel / step_quad_kernel (with EGS_SCHED_QUAD): the plan's sweep on
                                  its static timetable (one workgroup barrier per time step) instead of tickets */
  EGS_SCHED_LEAN = 64          /* lean_step_kernel: the timetable sweep in 128 VGPRs (one linear block for both sides,
                                  constants parked in LDS): four 256-constraint tiles per CU (fp64, isotropic bodies) */
} egs_schedule_flags;

void egs_default_params(egs_solve_params *p); /* GS, 500, 1, omega 1.5, cfm 0, tol 1e-9 */

/* ---- context ------------------------------------------------------------ */
egs_status egs_context_create(int device_index, egs_context **out);
void egs_context_destroy(egs_context *ctx);
const char *egs_last_error(const egs_context *ctx);
egs_status egs_context_synchronize(egs_context *ctx);
/* hipEvent pair on the context's stream (the stream every kernel below is
 * launched on).  stop() synchronises and returns elapsed milliseconds. */
egs_status egs_timer_start(egs_context *ctx);
egs_status egs_timer_stop(egs_context *ctx, float *elapsed_ms);
/* Sum and count of the solve-kernel launch durations (hipEvents around each
 * launch) since the last reset; synchronises. */
egs_status egs_kernel_time(egs_context *ctx, double *sum_ms, int64_t *launches,
                           int reset);

/* ---- entry 1: replaces sparse::{Jacobi,GaussSeidel,SOR}Iteration
 *      (const ConstraintsList&, const MatrixXd& M_inverse, const VectorXd& rhs,
 *       double cfm)   sparse_iterations.h:26-34 / sparse_iterations.cc:148-286.
 * The C++ adapter flattens the ConstraintsList with m ComputeJ calls:
 *   Minv  [n][36]  the 6x6 diagonal blocks M_inverse.block<6,6>(6b,6b)
 *   body0/1 [m]    Constraint::i0_, i1_ (constraints.h:42-43)
 *   J0,J1 [m][18]  the 3x6 blocks ComputeJ returns (constraints.h:22-24)
 *   is_eq, lo, hi [3m], rhs [3m];  out x [3m].
 * x0 = rhs, stop at residual <= tol or max_iters sweeps.                     */
egs_status egs_solve_blocks(egs_context *ctx, int32_t n_bodies,
                            const double *Minv, int32_t m,
                            const int32_t *body0, const int32_t *body1,
                            const double *J0, const double *J1,
                            const uint8_t *is_eq, const double *lo,
                            const double *hi, const double *rhs,
                            const egs_solve_params *params, int32_t precision,
                            double *x, egs_solve_stats *stats);

/* ---- device-resident form of the same path (what bench.py times) -------- */
/* Analyses the constraint graph (islands -> workgroup tiles) and allocates
 * device storage.  The topology (body0/body1) is fixed for the problem's life;
 * values can be re-uploaded any number of times. */
egs_status egs_problem_create(egs_context *ctx, int32_t n_bodies, int32_t m,
                              const int32_t *body0, const int32_t *body1,
                              int32_t precision, egs_problem **out);
/* Batched form (BASELINE config 4: 1024 independent 64-body
 * ensembles): E ensembles in ONE problem.  body0/body1 hold the E constraint
 * lists back to back with ensemble-LOCAL body indices (-1 = world); the
 * returned offset tables ([E+1] each, may be NULL) say where ensemble e's
 * bodies and constraints live in the concatenated arrays every other
 * egs_problem_* call takes.  Ensembles never share a body, so each is its own
 * set of islands and results equal E separate solves bit for bit; one launch
 * sweeps them all. */
egs_status egs_problem_create_batch(egs_context *ctx, int32_t n_ensembles,
                                    const int32_t *n_bodies, const int32_t *n_constraints,
                                    const int32_t *body0, const int32_t *body1,
                                    int32_t precision, egs_problem **out,
                                    int32_t *body_offset, int32_t *constraint_offset);
void egs_problem_destroy(egs_problem *p);
/* upload the flat system of entry 1 (any pointer may be NULL = keep) */
egs_status egs_problem_set_blocks(egs_problem *p, const double *Minv,
                                  const double *J0, const double *J1,
                                  const uint8_t *is_eq, const double *lo,
                                  const double *hi, const double *rhs);
/* asynchronous on the context stream */
egs_status egs_problem_solve(egs_problem *p, const egs_solve_params *params,
                             egs_solve_stats *stats);
egs_status egs_problem_get_lambda(egs_problem *p, double *x /*[3m]*/);
/* a_b = Minv_b sum_i J_ib^T lambda_i, [n][6]: the solver's by-product,
 * so that v_dot = Minv f_ext + a (ensembles.cc:535) needs no extra pass. */
egs_status egs_problem_get_accumulators(egs_problem *p, double *a /*[n][6]*/);

/* w = A lambda - rhs of the last solve, [3m]: what GetResidualError
 * (sparse_iterations.cc:51-69) reduces; written by the solve kernels' epilogue. */
egs_status egs_problem_get_wres(egs_problem *p, double *w /*[3m]*/);

/* ---- the matrix-free products: replace sparse::CalculateSparse{JMJtX,Lx,Ux,
 *      LxUx,Dx,UxDx,LxDx}(const ConstraintsList&, const MatrixXd& M_inverse,
 *      const VectorXd& x, double epsilon_diagonal, double scale_diagonal)
 *      sparse_iterations_utils.h / sparse_iterations_utils.cc:427-695.
 * With A = J M^-1 J^T (3m x 3m, never formed):
 *   EGS_MV_FULL   y = (A + eps I) x                          (:624-695; scale unused)
 *   EGS_MV_LOWER  y = strictLower(A) x  -- the strict lower triangle of a
 *                 constraint's own 3x3 block belongs to it   (:427-493, :484-486)
 *   EGS_MV_UPPER  y = strictUpper(A) x                       (:495-561, :522-524)
 *   EGS_MV_DIAG   y_r = ((A_rr + eps) * scale) x_r           (:571-603, :594)
 * and the sums the reference offers, added in its order: LOWER|UPPER (:563-569),
 * UPPER|DIAG (:606-613), LOWER|DIAG (:615-622).  The reference walks all O(m^2)
 * constraint pairs; here the cost is O(m).
 * Uses the blocks the problem holds (set_blocks or assemble).  x = NULL takes
 * the device-resident lambda of the last solve; y = NULL leaves the result on
 * the device (egs_problem_get_matvec) and makes the call asynchronous.        */
typedef enum { EGS_MV_LOWER = 1, EGS_MV_UPPER = 2, EGS_MV_DIAG = 4, EGS_MV_FULL = 8 } egs_matvec_part;
egs_status egs_problem_matvec(egs_problem *p, int32_t parts, double eps, double scale,
                              const double *x /*[3m] or NULL*/, double *y /*[3m] or NULL*/);
egs_status egs_problem_get_matvec(egs_problem *p, double *y /*[3m]*/);
/* one-shot form over flat host arrays, as egs_solve_blocks (the adapter's
 * sparse::CalculateSparse* functions call this) */
egs_status egs_matvec_blocks(egs_context *ctx, int32_t n_bodies, const double *Minv, int32_t m,
                             const int32_t *body0, const int32_t *body1, const double *J0,
                             const double *J1, int32_t parts, double eps, double scale,
                             int32_t precision, const double *x, double *y);

/* ---- entry 2: replaces the assembly half of Ensemble::StepVelocities_ODE
 *      (ensembles.cc:563-575): ComputeJ (ensembles.cc:38-87 ->
 *      joints.cc:13-35, contact.cc:38-117), ComputePositionConstraintError
 *      (ensembles.cc:156-171 -> joints.cc:3-11, contact.cc:14-22), the rhs
 *      (ensembles.cc:569-570) and, after the solve, the velocity update
 *      (ensembles.cc:535, 572).  Body state and constraint descriptors:
 *   pos [n][3], R [n][9] row-major, v [n][3], w [n][3] (global frame),
 *   Minv [n][36], f_ext [n][6]  (ensembles.cc:202-222; frozen at Init, Q5)
 *   kind [m] (egs_constraint_kind),
 *   data [m][7]: joint   = c0(3), c1(3) (c1 = world point if body1 = -1), 0
 *                contact = position(3), normal(3), depth (collision.h:12-27) */
egs_status egs_problem_set_state(egs_problem *p, const double *pos,
                                 const double *R, const double *v,
                                 const double *w, const double *Minv,
                                 const double *f_ext);
/* The compact form of Minv: per body 1/m and the 3x3 inverse of the
 * global-frame inertia, row-major -- all ConstructMassInertiaMatrixInverse
 * (ensembles.cc:202-212) ever stores; the 6x6 blocks are built from it. */
egs_status egs_problem_set_mass(egs_problem *p, const double *inv_mass,
                                const double *inv_inertia);
egs_status egs_problem_set_constraints(egs_problem *p, const int32_t *kind,
                                       const double *data);
/* J, err, bounds, rhs = -(erp/dt^2) err - J (v/dt + Minv f_ext) on device */
egs_status egs_problem_assemble(egs_problem *p, double dt, double erp);
/* assemble + solve + v_new = v + dt (Minv f_ext + a): one whole hot-path pass */
egs_status egs_problem_step(egs_problem *p, double dt, double erp,
                            const egs_solve_params *params,
                            egs_solve_stats *stats);
egs_status egs_problem_get_blocks(egs_problem *p, double *J0, double *J1,
                                  uint8_t *is_eq, double *lo, double *hi,
                                  double *rhs, double *err);
egs_status egs_problem_get_velocity(egs_problem *p, double *v6 /*[n][6]*/);
/* Ensemble::StepPositions_ODE (ensembles.cc:577-591) on the device, after
 * egs_problem_step: p += dt (v + v_new)/2, R = WtoQ((w + w_new)/2, dt) R
 * (utils.cc:82-89), then v, w <- v_new, so the body state never leaves the
 * GPU between steps while the contact set is unchanged. */
egs_status egs_problem_advance(egs_problem *p, double dt);
egs_status egs_problem_get_state(egs_problem *p, double *pos, double *R, double *v, double *w);
/* fills stats->residual/iterations of the last solve (synchronises) */
egs_status egs_problem_get_stats(egs_problem *p, egs_solve_stats *stats);

/* ---- entry 3: replaces Lcp::MixedConstraintsSolver(A, b, C, x_lo, x_hi, x, w)
 *      lcp.h:21-23 / lcp.cc:276-336 (and Lcp::MurtyPrincipalPivot,
 *      lcp.cc:157-274).  A [N][N], b, C, lo, hi [N] -> x, w [N]; *ok as the
 *      reference's bool.  use_bounds is a bit mask:
 *        0      the reference: bounds ignored (quirk Q3), single-index pivots,
 *               cap min(1000, 2^n) pivots (lcp.cc:168)
 *        bit 0  honour x_lo/x_hi (the true box problem)
 *        bit 1  block principal pivoting instead of the single-index rule: same
 *               solution, tens of factorisations instead of hundreds, no cap --
 *               the reference's rule cannot finish N >~ 1200 mixed problems.    */
egs_status egs_mixed_constraints_solve(egs_context *ctx, int32_t N,
                                       const double *A, const double *b,
                                       const uint8_t *C, const double *lo,
                                       const double *hi, int32_t use_bounds,
                                       double *x, double *w, int32_t *ok,
                                       int32_t *pivots);

/* The same with the give-up limits of lcp::Settings (toolkit/lcp.h:161-167): max_pivots > 0
 * / max_seconds > 0 stop the pivoting loop there and the call reports *ok = 0
 * (EGS_ERR_LCP_FAILED), as SolveLCP returns false.  0 = the library's own cap.   */
egs_status egs_mixed_constraints_solve_limits(egs_context *ctx, int32_t N, const double *A, const double *b,
                                              const uint8_t *C, const double *lo, const double *hi,
                                              int32_t use_bounds, int32_t max_pivots, double max_seconds,
                                              double *x, double *w, int32_t *ok, int32_t *pivots);

/* Replaces sparse::JacobiIteration / GaussSeidelIteration / SORIteration on an EXPLICIT matrix:
 *   VectorXd sparse::XIteration(const MatrixXd& A, const VectorXd& b)                                    sparse_iterations.h:13-24
 *   VectorXd sparse::XIteration(const MatrixXd& A, const VectorXd& b, const ArrayXb& C, x_lo, x_hi)
 * i.e. BaseIteration(A, b, ...) of sparse_iterations.cc:72-144 with its dense solves
 * (sparse_iterations_utils.cc:25-40, 110-128, 245-262) and stopping test (:35-49, 128-141): x0 = b,
 * one sweep, one residual, stop at err <= tol or after max_iters sweeps.  A [N][N] row-major (general,
 * need not be symmetric; non-zero diagonal), N <= 1024; C / lo / hi all NULL = the 2-argument form
 * (every row an equality).  params: method, omega, max_iters, tol as for entry 1 (cfm unused: it is in A).
 * stats: iterations, residual.  The reference's spectral-radius gate (:113-121, Panic when
 * rho(M^-1 N) >= 1) is not evaluated: a diverging splitting runs to max_iters and reports its residual. */
egs_status egs_dense_iterate(egs_context *ctx, int32_t N, const double *A, const double *b, const uint8_t *C,
                             const double *lo, const double *hi, const egs_solve_params *params, double *x,
                             egs_solve_stats *stats);

/* ---- the dense front half of Ensemble::ComputeVDot (ensembles.cc:498-538) on
 *      the device, for the sizes the reference's dense solver is meant for
 *      (Chain, Cairn): the problem's blocks (assemble or set_blocks) -> dense
 *      A = J M^-1 J^T + cfm I (ensembles.cc:510, 513-521), [3m][3m] row-major,
 *      kept on the device; A (host) may be NULL.  fp64 problems only.          */
egs_status egs_problem_dense_system(egs_problem *p, double cfm, double *A /*[3m][3m] or NULL*/);
/* Replaces CheckMatrixCondition / GetConditionNumber (ensembles.cc:514 -> utils.cc:256-287): the
 * reference takes sigma_max / sigma_min from a JacobiSVD (Eigen, absent here).  For the symmetric
 * positive definite A that is lambda_max / lambda_min: power iteration on A and inverse iteration
 * with A's Cholesky factor, on the device (up to 3m = 1024 rows; within a few per cent, never above
 * the true value; beyond 1024 rows the pivot bound (max L_ii / min L_ii)^2, a lower bound) -- +inf
 * when A is not positive definite.  The caller compares with kGoodConditionNumber = 1e7
 * (constants.h:12) and picks the cfm of the step, as ensembles.cc:513-521.                      */
egs_status egs_problem_dense_condition(egs_problem *p, double cfm, double *estimate);
/* The same for a caller's own symmetric positive definite matrix (A [N][N] row-major, host): replaces
 * GetConditionNumber(A) / CheckMatrixCondition(A) of utils.cc:256-287 as such.  *pivot_bound (may be
 * NULL) receives the cheap lower bound (max L_ii / min L_ii)^2 beside the estimate.               */
egs_status egs_dense_condition(egs_context *ctx, int32_t N, const double *A, double *estimate, double *pivot_bound);
/* One StepVelocities_ODE through the reference's LIVE dense path (ensembles.cc:563-575,
 * 498-538): assemble, dense system with the cfm the caller decided, then
 * Lcp::MixedConstraintsSolver on the device matrix (use_bounds as in entry 3), v update.
 * Nothing but 3m row types / bounds crosses PCIe; lambda, accumulators and v_new are
 * read with egs_problem_get_lambda / get_accumulators / get_velocity.             */
egs_status egs_problem_step_dense(egs_problem *p, double dt, double erp, double cfm,
                                  int32_t use_bounds, int32_t *ok, int32_t *pivots);

/* ---- contact generation ("next" row 1): replaces Ensemble::UpdateContacts
 *      (ensembles.cc:445-480 -> CollideBoxAndGround collision.cc:408-436,
 *      CollideBoxes collision.cc:166-388) and the contact-vs-contact pruning of
 *      CheckAndCorrectEnsembleState (ensembles.cc:308-328, 1e-6).
 * pos [n][3], R [n][9], side_lengths [n][3] -> the contact list in the
 * reference's order (ground contacts by body, then body pairs i < j):
 * body0/body1 [m] (-1 = ground), data [m][7] = position, normal, depth, ready
 * for egs_problem_set_constraints with kind = EGS_CONTACT_BOX.             */
egs_status egs_update_contacts(egs_context *ctx, int32_t n_bodies,
                               const double *pos, const double *R,
                               const double *side_lengths, int32_t max_contacts,
                               int32_t *m_out, int32_t *body0, int32_t *body1,
                               double *data);

/* The same with the joint-vs-contact check of CheckAndCorrectEnsembleState
 * (ensembles.cc:296-306): a contact within 1e-6 of a joint between the same two
 * bodies (Joint::GetConstraintPosition, joints.cc:57-75) is dropped.
 * jb0/jb1 [m_joints], jdata [m_joints][7] = c0, c1 as in set_constraints.     */
egs_status egs_update_contacts_joints(egs_context *ctx, int32_t n_bodies,
                                      const double *pos, const double *R,
                                      const double *side_lengths, int32_t m_joints,
                                      const int32_t *jb0, const int32_t *jb1,
                                      const double *jdata, int32_t max_contacts,
                                      int32_t *m_out, int32_t *body0,
                                      int32_t *body1, double *data);

/* ---- the whole Ensemble::Step on the device ------------------------------
 * Ensemble::Step(dt, OPEN_DYNAMICS_ENGINE) (ensembles.cc:390-427) with the
 * sparse switch on: UpdateContacts + contact pruning, StepVelocities_ODE
 * (assembly, rhs, projected solve, velocity update) and StepPositions_ODE all
 * run on the GPU and the body state stays there between steps.  Only the
 * contact TOPOLOGY (body index pairs) is read back each step; the schedule is
 * re-planned on the host only when it changed.  Constraint list order as in
 * the reference: joints first, then contacts (ensembles.cc:234-239).         */
typedef struct egs_world egs_world;
egs_status egs_world_create(egs_context *ctx, int32_t n_bodies, int32_t precision, egs_world **out);
void egs_world_destroy(egs_world *w);
/* pos [n][3], R [n][9], v, w [n][3], Minv [n][36], f_ext [n][6] (frozen as
 * Ensemble::Init leaves them, quirk Q5), side_lengths [n][3] (body.h:91).
 * The first call needs every array; later calls may pass NULL for any of them
 * to keep what the device holds (typically Minv, f_ext, side_lengths). */
egs_status egs_world_set_bodies(egs_world *w, const double *pos, const double *R, const double *v,
                                const double *w_ang, const double *Minv, const double *f_ext,
                                const double *side_lengths);
/* ball joints: body0/body1 [m], data [m][7] = c0, c1 (world point if body1 = -1), 0 */
egs_status egs_world_set_joints(egs_world *w, int32_t m_joints, const int32_t *body0, const int32_t *body1,
                                const double *data);
egs_status egs_world_step(egs_world *w, double dt, double erp, const egs_solve_params *params,
                          int32_t detect_contacts, egs_solve_stats *stats);
egs_status egs_world_get_bodies(egs_world *w, double *pos, double *R, double *v, double *w_ang);
egs_status egs_world_get_contacts(egs_world *w, int32_t max_contacts, int32_t *m_out, int32_t *body0,
                                  int32_t *body1, double *data);
egs_status egs_world_get_lambda(egs_world *w, int32_t max_rows, int32_t *rows_out, double *lambda);
egs_status egs_world_info(egs_world *w, int32_t *n_constraints, int32_t *n_contacts, int32_t *replans);

/* Replaces lcp::SolveLCP_BoxDantzig (toolkit/lcp.cc:444-619; reached from lcp::SolveLCP with
 * Settings.algorithm = COTTLE_DANTZIG, box_lcp = true, schur_complement = false, toolkit/lcp.cc:776-779):
 * Cottle-Dantzig principal pivoting on A x = b + w with lo <= x <= hi, the Cholesky factor of the active
 * set kept up to date row by row (AddCholeskyRow / SwapCholeskyRows, toolkit/lcp.cc:91-157; O(n^2) per
 * pivot).  A [n][n] row-major: only the lower triangle is read, and it is PERMUTED IN PLACE exactly as the
 * reference leaves it (toolkit/lcp.h:170-171); perm[k] = original index of the final row k (may be NULL).
 * Requires lo <= 0 <= hi and lo < hi (toolkit/lcp.cc:448-450: EGS_ERR_INVALID otherwise) and
 * 1 <= n <= 1024 (n <= 96: one wavefront, both matrices in LDS; above: four wavefronts, A permuted in place
 * in device memory).  max_steps > 0 gives up after that many pivot steps, 0 = the library's cap of
 * 20 n + 1000 (*ok = 0, EGS_ERR_LCP_FAILED) -- the device loop always has an exit, where the reference's
 * has none (toolkit/lcp.cc:493) -- as does a non-positive pivot (A not positive definite).                  */
egs_status egs_box_lcp_dantzig(egs_context *ctx, int32_t n, double *A, const double *b, const double *lo,
                               const double *hi, int32_t max_steps, double *x, double *w, int32_t *perm,
                               int32_t *ok, int32_t *pivots);

/* Replaces lcp::SolveLCP_BoxMurty (toolkit/lcp.cc:380-442; with lo = 0, hi = +inf / DBL_MAX it is
 * SolveLCP_Murty, :333-378) on its LinearReducer (:213-328): principal pivoting that moves the first
 * violated index (in the caller's order) in or out of the index set and keeps the set's Cholesky factor
 * up to date row by row.  Same arguments, limits (n <= 1024, lo <= 0 <= hi) and in-place permutation of A's
 * lower triangle as egs_box_lcp_dantzig; max_iterations > 0 = Settings::max_iterations (the call then
 * reports *ok = 0, EGS_ERR_LCP_FAILED, as the reference returns false, toolkit/lcp.cc:438-441).       */
egs_status egs_box_lcp_murty(egs_context *ctx, int32_t n, double *A, const double *b, const double *lo,
                             const double *hi, int32_t max_iterations, double *x, double *w, int32_t *perm,
                             int32_t *ok, int32_t *iterations);

/* The same two solvers (algorithm 0 = SolveLCP_BoxMurty, 1 = SolveLCP_BoxDantzig) on `count` independent
 * problems in ONE launch, a workgroup per problem -- what a batch of ensembles hands lcp::SolveLCP
 * (toolkit/lcp.h:172-174), which the reference calls once per problem.  Problem k has n[k] rows; its matrix
 * sits at A + sum_{j<k} n[j]^2 (row-major, lower triangle read and permuted in place), its vectors (b, lo, hi,
 * x, w, perm) at sum_{j<k} n[j].  ok[k] / pivots[k]: the problem's own outcome and step count -- the same as
 * its single call.  max_steps / max_seconds: Settings::max_iterations / max_time per problem (0 = the
 * library's cap / none).  Returns EGS_OK when every problem was run (see ok[]); EGS_ERR_INVALID for a bad
 * size or bounds that break lo <= 0 <= hi (lo < hi for Dantzig).  perm, pivots may be NULL.               */
egs_status egs_box_lcp_batch(egs_context *ctx, int32_t algorithm, int32_t count, const int32_t *n, double *A,
                             const double *b, const double *lo, const double *hi, int32_t max_steps,
                             double max_seconds, double *x, double *w, int32_t *perm, int32_t *ok,
                             int32_t *pivots);

/* Replaces lcp::SolveLCP_BoxSchur (toolkit/lcp.cc:627-747), what lcp::SolveLCP (toolkit/lcp.h:172-174,
 * toolkit/lcp.cc:752-785) runs under its default Settings (schur_complement = true, box_lcp = true):
 * the two-pointer partition that brings the unbounded rows (lo = -infinity, hi = +infinity; "infinity" =
 * DBL_MAX or the real one) to the front, Z = L L' on them, the Schur complement R = C - B Z^-1 B' and its
 * right-hand side (:714-727), the box LCP on R by SolveLCP_BoxMurty (algorithm 0) or SolveLCP_BoxDantzig
 * (1), and y = Z^-1 (c - B' z).  A: row-major n x n, ONLY THE LOWER TRIANGLE IS READ OR WRITTEN
 * (toolkit/lcp.h:73); it is permuted in place as the reference leaves it: by the partition and, when no row
 * is unbounded, by the inner solver's pivoting (:695-700).  perm[k] = original index of row k after the
 * partition (may be NULL).  nub >= 0 is the reference's test hook (:623-626: that many leading indexes are
 * taken as unbounded unseen), -1 scans the bounds; *nub_out = test_nub_from_SolveLCP_BoxSchur.
 * reference_quirks != 0 keeps the literal tests `hi < -DBL_MAX` of :664, 669 (the lower
 * bound alone decides); 0 tests hi against +DBL_MAX.  max_iterations / max_seconds as above.  Bounded parts
 * beyond 1024 rows are solved by block principal pivoting (same solution).                                  */
egs_status egs_box_lcp_schur(egs_context *ctx, int32_t n, double *A, const double *b, const double *lo,
                             const double *hi, int32_t algorithm, int32_t nub, int32_t reference_quirks,
                             int32_t max_iterations, double max_seconds, double *x, double *w, int32_t *perm,
                             int32_t *ok, int32_t *nub_out, int32_t *pivots);

/* ---- diagnostics (host only, needs no GPU) -------------------------------
 * The schedule the solver derives from the constraint graph: islands, the
 * workgroup tile each constraint lands in (-1 = cross-workgroup path) and the
 * per-body tickets (rank/count of the constraint among its body's, list
 * order) that make the parallel sweep reproduce the reference's list order
 * (sparse_iterations_utils.cc:159-243, 292-373).  Arrays [m], may be NULL.   */
/* Diagnostics: with EGS_TRACE_UPDATES=1 in the environment the 4-lane patch kernel stamps every update with the
 * device's 100 MHz wall clock; this copies the stamps of the problem's last such launch, [sweeps][m] (0 = not
 * written), for tools/trace_patches.py, which walks the critical chain of the sweep pipeline.                  */
egs_status egs_problem_debug_trace(egs_problem *p, uint64_t *out, int64_t count, int32_t *sweeps);
/* The body patches an oversize island is cut into (plan.cpp::build_patches): per constraint its patch (-1: none),
 * its lane in the patch, and per side bit 0 = the list-order predecessor on that body sits in another patch, bit 1 =
 * the successor does (the hand-offs that cross global memory).  Arrays [m], may be NULL.                      */
egs_status egs_debug_plan_patches(int32_t n_bodies, int32_t m, const int32_t *body0, const int32_t *body1,
                                  int32_t *n_patches, int32_t *cons_patch, int32_t *cons_lane, int32_t *remote0,
                                  int32_t *remote1);
egs_status egs_debug_plan(int32_t n_bodies, int32_t m, const int32_t *body0,
                          const int32_t *body1, int32_t tile_size,
                          int32_t *n_islands, int32_t *n_tiles,
                          int32_t *n_global, int32_t *cons_tile,
                          int32_t *pos0, int32_t *cnt0, int32_t *pos1,
                          int32_t *cnt1);

/* The LDS slot numbers of the same schedule: per constraint the slot of each side
 * (0 = the world), and per constraint's tile the number of slots it allocates.
 * Slot numbers decide LDS banks; a body keeps one slot in its
 * tile.  Arrays [m], may be NULL; constraints on the cross-workgroup path get -1.  */
egs_status egs_debug_plan_slots(int32_t n_bodies, int32_t m, const int32_t *body0,
                                const int32_t *body1, int32_t tile_size,
                                int32_t *lane, int32_t *slot0, int32_t *slot1,
                                int32_t *tile_nslots);

/* The static timetable of the same schedule (step_solve.hip): constraint c
 * runs its update of sweep s at time step level[c] + period * s of its tile, level = depth in the
 * list-order dependency DAG of one sweep (sparse_iterations_utils.cc:159-243: a constraint reads
 * what the previous constraint of each of its bodies wrote), period = the largest level span of a
 * body in the tile, depth = levels in the tile.  Arrays [m] (period / depth: the values of the
 * constraint's tile), may be NULL; -1 on the cross-workgroup path.
 * tile_size = 0: the 4-lanes-per-constraint plan with its automatic tile size.
 * *runs = 1 (4-lane plan only): every aligned group of four consecutive constraints joins the same two bodies (the four
 * contact points of a box face) and is ONE node of the timetable: level / period / depth then count
 * groups, and the group's four updates run back to back inside its time step, in list order
 * (reversed in a backward sweep), handing the accumulators from lane to lane.                     */
egs_status egs_debug_plan_timetable(int32_t n_bodies, int32_t m, const int32_t *body0,
                                    const int32_t *body1, int32_t tile_size,
                                    int32_t *level, int32_t *period, int32_t *depth, int32_t *runs);

/* Which kernel takes islands larger than a workgroup in a GS / SOR solve (host only, the
 * chooser the library itself uses): 0 = body patches on the 4-lanes-per-constraint kernel,
 * 1 = body patches on the 1-lane kernel, 2 = the all-global kernel.  Patches wait on each
 * other, so their count must not exceed (workgroups per CU of the kernel, as
 * hipOccupancyMaxActiveBlocksPerMultiprocessor reports for the exact instantiation) x CUs.  */
int32_t egs_debug_choose_oversize_schedule(int32_t n_patch_tiles, int32_t quad_per_cu,
                                           int32_t patch_per_cu, int32_t cu_count,
                                           int32_t patches_enabled, int32_t quad_patches_enabled);

/* The schedule of the matrix-free products (host only): constraints are
 * partitioned into workgroup tiles; a body touched from more than one tile is
 * "shared" and its constraints form the boundary list that a pre-pass publishes.
 * cons_tile / cons_lane [m], may be NULL.  tile_size = 128 or 256.               */
egs_status egs_debug_matvec_plan(int32_t n_bodies, int32_t m, const int32_t *body0,
                                 const int32_t *body1, int32_t tile_size, int32_t *n_tiles,
                                 int32_t *n_islands, int32_t *n_shared_bodies,
                                 int32_t *n_boundary, int32_t *cons_tile, int32_t *cons_lane);

#ifdef __cplusplus
}
#endif
#endif
