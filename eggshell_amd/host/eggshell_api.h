// eggshell_api.h -- host-side C++ adapter: the reference's Body / Constraint /
// Joint / Contact / Ensemble API for the constraint-solve path, unchanged in
// names, argument meaning and result types, with the arithmetic behind
//   sparse::{Jacobi,GaussSeidel,SOR}Iteration   (sparse_iterations.h:26-34)
//   Lcp::MixedConstraintsSolver                 (lcp.h:21-23)
//   Ensemble::Step / StepVelocities_ODE         (ensembles.cc:390-427, 563-575)
// routed to the MI355X library through the C ABI (include/eggshell_amd.h).
// What is mirrored here and from where:
//   Body                body.h:13-96 (state p,v,m,R,w,I; side 0.3)
//   Constraint          constraints.h:14-48
//   Joint, BallAndSocketJoint   joints.h:12-50, joints.cc:3-35
//   ContactGeometry     collision.h:12-27
//   Contact             contact.h:11-56, contact.cc:14-117 (FrictionModel::BOX)
//   Ensemble, Chain     ensembles.h:25-186, ensembles.cc:24-87,156-171,202-239,
//                       429-443, 563-591, 668-707
// Error convention differs on purpose: the reference Panics (_exit(1)); these
// functions throw egs::Error carrying the C ABI status and message.
#ifndef EGGSHELL_API_H
#define EGGSHELL_API_H

#include <array>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/eggshell_amd.h"
#include "eigen_lite.h"

namespace egs {
struct Error : std::runtime_error {
  int status;
  Error(int st, const std::string &msg) : std::runtime_error(msg), status(st) {}
};
// Process-wide context on device 0 (created on first use).
egs_context *DefaultContext();
}  // namespace egs

// utils.h
Matrix3d CrossMat(const Vector3d &a);                      // utils.cc:16-24
Matrix3d AlignVectors(const Vector3d &a, const Vector3d &b);  // utils.cc:233-237
Matrix3d WtoR(const Vector3d &w, double dt);               // utils.cc:82-89 (as a matrix)

class Body {  // body.h:13-96
 public:
  Body() : m_(1.0), R_(Matrix3d::Identity()) { I_ = CalculateInertia(m_); }
  Body(const Vector3d &p, const Vector3d &v, const Matrix3d &R, const Vector3d &w)
      : p_(p), v_(v), m_(1.0), R_(R), w_(w) { I_ = CalculateInertia(m_); }
  Body(const Vector3d &p, const Vector3d &v, double m, const Matrix3d &R, const Vector3d &w, const Matrix3d &I)
      : p_(p), v_(v), m_(m), R_(R), w_(w), I_(I) {}
  const Vector3d &p() const { return p_; }
  const Vector3d &v() const { return v_; }
  double m() const { return m_; }
  const Matrix3d &R() const { return R_; }
  const Vector3d &w_g() const { return w_; }
  const Matrix3d &I_b() const { return I_; }
  Matrix3d I_g() const { return R_ * I_ * R_.transpose(); }
  void SetP(const Vector3d &p) { p_ = p; }
  void SetV(const Vector3d &v) { v_ = v; }
  void SetR(const Matrix3d &R) { R_ = R; }
  void SetW_GlobalFrame(const Vector3d &w) { w_ = w; }
  Vector3d GetSideLengths() const { return Vector3d(0.3, 0.3, 0.3); }  // body.h:91

 private:
  Vector3d p_, v_;
  double m_;
  Matrix3d R_;
  Vector3d w_;
  Matrix3d I_;
  Matrix3d CalculateInertia(double m) const;  // body.cc:19-36
};

class Constraint {  // constraints.h:14-48
 public:
  Constraint(std::shared_ptr<Body> b0, int i0, std::shared_ptr<Body> b1, int i1)
      : i0_(i0), i1_(i1), b0_(std::move(b0)), b1_(std::move(b1)) {}
  virtual ~Constraint() = default;
  virtual VectorXd ComputeError() const = 0;
  virtual void ComputeJ(MatrixXd *J_b0, MatrixXd *J_b1, ArrayXb *constraint_type, VectorXd *constraint_lo,
                        VectorXd *constraint_hi) const = 0;
  // Device-assembly descriptor (egs_constraint_kind + 7 doubles); returns false
  // for constraint types the library cannot assemble itself (then the solver
  // falls back to flattening ComputeJ on the host, entry 1).
  virtual bool Describe(int32_t *kind, double data[7]) const { (void)kind; (void)data; return false; }
  // Get the global frame constraint position (constraints.h:38-39).
  virtual Vector3d GetConstraintPosition() const = 0;
  int i0_ = -1;
  int i1_ = -1;

 protected:
  const std::shared_ptr<Body> b0_;
  const std::shared_ptr<Body> b1_;
};

class Joint : public Constraint {  // joints.h:12-28
 public:
  Joint(std::shared_ptr<Body> b0, int i0, const Vector3d &c0, const Vector3d &c1)
      : Constraint(std::move(b0), i0, nullptr, -1), c0_(c0), c1_(c1) {}
  Joint(std::shared_ptr<Body> b0, int i0, const Vector3d &c0, std::shared_ptr<Body> b1, int i1, const Vector3d &c1)
      : Constraint(std::move(b0), i0, std::move(b1), i1), c0_(c0), c1_(c1) {}

 protected:
  Vector3d c0_, c1_;
};

class BallAndSocketJoint : public Joint {  // joints.h:31-50
 public:
  using Joint::Joint;
  VectorXd ComputeError() const override;                                   // joints.cc:3-11
  void ComputeJ(MatrixXd *J_b0, MatrixXd *J_b1, ArrayXb *constraint_type, VectorXd *constraint_lo,
                VectorXd *constraint_hi) const override;                    // joints.cc:13-35
  bool Describe(int32_t *kind, double data[7]) const override;
  Vector3d GetConstraintPosition() const override;                          // joints.cc:57-75
};

struct ContactGeometry {  // collision.h:12-27
  Vector3d position, normal;
  double depth = 0;
  ContactGeometry() {}
  ContactGeometry(const Vector3d &p, const Vector3d &n, double d) : position(p), normal(n), depth(d) {}
};

class Contact : public Constraint {  // contact.h:11-56
 public:
  Contact(std::shared_ptr<Body> b, int index, const ContactGeometry &cg)
      : Constraint(nullptr, -1, std::move(b), index), cg_(cg) {}
  Contact(std::shared_ptr<Body> b0, int i0, std::shared_ptr<Body> b1, int i1, const ContactGeometry &cg)
      : Constraint(std::move(b0), i0, std::move(b1), i1), cg_(cg) {}
  VectorXd ComputeError() const override;                                   // contact.cc:14-22
  void ComputeJ(MatrixXd *J_b0, MatrixXd *J_b1, ArrayXb *C, VectorXd *x_lo, VectorXd *x_hi) const override;  // :38-117
  bool Describe(int32_t *kind, double data[7]) const override;
  Vector3d GetConstraintPosition() const override { return cg_.position; }  // contact.cc:166-168

 private:
  const ContactGeometry cg_;
};

typedef std::vector<std::shared_ptr<Body>> ComponentsList;     // ensembles.h:19-22
typedef std::vector<std::shared_ptr<Joint>> JointsList;
typedef std::vector<std::shared_ptr<Contact>> ContactsList;
typedef std::vector<std::shared_ptr<Constraint>> ConstraintsList;

namespace sparse {  // sparse_iterations.h:13-34
// on an explicit matrix (sparse_iterations.h:13-24, sparse_iterations.cc:72-144): the 2-argument forms treat every
// row as an equality, the 5-argument forms project the rows with C = false onto [x_lo, x_hi]
VectorXd JacobiIteration(const MatrixXd &A, const VectorXd &b);
VectorXd JacobiIteration(const MatrixXd &A, const VectorXd &b, const ArrayXb &C, const VectorXd &x_lo, const VectorXd &x_hi);
VectorXd GaussSeidelIteration(const MatrixXd &A, const VectorXd &b);
VectorXd GaussSeidelIteration(const MatrixXd &A, const VectorXd &b, const ArrayXb &C, const VectorXd &x_lo, const VectorXd &x_hi);
VectorXd SORIteration(const MatrixXd &A, const VectorXd &b);
VectorXd SORIteration(const MatrixXd &A, const VectorXd &b, const ArrayXb &C, const VectorXd &x_lo, const VectorXd &x_hi);
// matrix-free, on an ensemble's constraints (sparse_iterations.h:26-34)
VectorXd JacobiIteration(const ConstraintsList &constraints, const MatrixXd &M_inverse, const VectorXd &rhs,
                         double cfm = 0.0);
VectorXd GaussSeidelIteration(const ConstraintsList &constraints, const MatrixXd &M_inverse, const VectorXd &rhs,
                              double cfm = 0.0);
VectorXd SORIteration(const ConstraintsList &constraints, const MatrixXd &M_inverse, const VectorXd &rhs,
                      double cfm = 0.0);
// sparse_iterations_utils.cc:697-720
void ConstructMixedConstraints(const ConstraintsList &constraints, ArrayXb *C, VectorXd *x_lo, VectorXd *x_hi);
// The matrix-free products (sparse_iterations_utils.h:60-109, sparse_iterations_utils.cc:427-695):
// strict lower / strict upper / diagonal parts of (J M^-1 J^T) x and the full product with
// epsilon on the diagonal.  Same names, arguments and defaults; O(m) on the GPU
// (egs_matvec_blocks) instead of the reference's O(m^2) pair loops.
VectorXd CalculateSparseDx(const ConstraintsList &constraints, const MatrixXd &M_inverse, const VectorXd &x,
                           double epsilon = 0.0, double scale = 1.0);
VectorXd CalculateSparseLx(const ConstraintsList &constraints, const MatrixXd &M_inverse, const VectorXd &x,
                           double epsilon_diagonal = 0.0, double scale_diagonal = 1.0);
VectorXd CalculateSparseUx(const ConstraintsList &constraints, const MatrixXd &M_inverse, const VectorXd &x,
                           double epsilon_diagonal = 0.0, double scale_diagonal = 1.0);
VectorXd CalculateSparseLxUx(const ConstraintsList &constraints, const MatrixXd &M_inverse, const VectorXd &x,
                             double epsilon_diagonal = 0.0, double scale_diagonal = 1.0);
VectorXd CalculateSparseLxDx(const ConstraintsList &constraints, const MatrixXd &M_inverse, const VectorXd &x,
                             double epsilon_diagonal = 0.0, double scale_diagonal = 1.0);
VectorXd CalculateSparseUxDx(const ConstraintsList &constraints, const MatrixXd &M_inverse, const VectorXd &x,
                             double epsilon_diagonal = 0.0, double scale_diagonal = 1.0);
VectorXd CalculateSparseJMJtX(const ConstraintsList &constraints, const MatrixXd &M_inverse, const VectorXd &x,
                              double epsilon_diagonal = 0.0);
// Iteration count and residual of the most recent *Iteration call (the
// reference prints the count to stdout, sparse_iterations.cc:223-224).
struct LastSolve { int iterations; double residual; int n_islands, n_tiles, n_global; };
LastSolve GetLastSolve();
}  // namespace sparse

namespace Lcp {  // lcp.h:21-23
bool MixedConstraintsSolver(const MatrixXd &A, const VectorXd &b, const ArrayXb &C, const VectorXd &x_lo,
                            const VectorXd &x_hi, VectorXd &x, VectorXd &w);
}

// toolkit/lcp.h:104-174 -- the "adjacent" solver family the north star names
// (lcp::SolveLCP; not linked into eggshell).  Box LCP
// A x = b + w with lo <= 0 <= hi.  What is kept of the contract:
//   * the dispatch of toolkit/lcp.cc:752-785, including its two refusals (Schur complement
//     or Cottle-Dantzig without box_lcp: the reference Panics, here egs::Error / EGS_ERR_INVALID);
//   * only the LOWER TRIANGLE of A is ever read or written (toolkit/lcp.h:73; the reference's tests hand
//     over lower triangles, toolkit/lcp.cc:808, 880-881, 913, 957, 1109-1110);
//   * schur_complement (the default): SolveLCP_BoxSchur (toolkit/lcp.cc:627-747) on the device
//     (egs_box_lcp_schur): unbounded rows (lo = -inf or -DBL_MAX and hi = +inf or DBL_MAX) come first by
//     its two-pointer partition, Z = L L', R = C - B Z^-1 B', the box LCP on R by SolveLCP_BoxMurty /
//     SolveLCP_BoxDantzig (Settings::algorithm), back-substitution; A IS PERMUTED IN PLACE as the reference
//     leaves it (the partition; plus the inner solver's pivoting order when no row is unbounded, :695-700);
//     quirk Q6 (toolkit/lcp.cc:664, 669 test `hi < -DBL_MAX`, so a row with lo = -inf and a
//     FINITE hi is classed unbounded) is reproduced when reference_quirks is set;
//   * max_iterations and max_time: the solve gives up and returns false (toolkit/lcp.h:161-167); the device
//     loops also carry a cap of their own (20 n + 1000 steps) where the reference would loop for ever;
//   * box_lcp = false: lo = 0, hi = +inf whatever the vectors hold (toolkit/lcp.h:152-154).
//   * schur_complement = false (toolkit/lcp.cc:768-781): algorithm = COTTLE_DANTZIG runs SolveLCP_BoxDantzig
//     and algorithm = MURTY runs SolveLCP_BoxMurty / SolveLCP_Murty on a LinearReducer themselves on the
//     device (egs_box_lcp_batch: the incremental Cholesky factor of AddCholeskyRow / SwapCholeskyRows,
//     toolkit/lcp.cc:91-157), and A's lower triangle is permuted in place by their pivoting order, as in
//     the reference.
// What differs, by design: a bounded part beyond 1024 rows goes through block principal
// pivoting with a single-index safeguard (fresh blocked factorisations on the matrix cores; same unique
// solution), so there A is left in BoxSchur's order or untouched instead of carrying the pivoting order of
// the inner solver.
namespace lcp {
enum Algorithm { MURTY, COTTLE_DANTZIG };
struct Settings {
  Algorithm algorithm = MURTY;
  bool box_lcp = true;
  bool schur_complement = true;
  int max_iterations = __INT_MAX__;
  double max_time = __DBL_MAX__;
  bool reference_quirks = false;   // not in the reference: reproduce Q6 (see above)
};
int LastSolvePivots();             // principal pivots of the most recent SolveLCP
bool SolveLCP(const Settings &settings, MatrixXd &A, const VectorXd &b, const VectorXd &lo, const VectorXd &hi,
              VectorXd *x, VectorXd *w);
}  // namespace lcp

class Ensemble {  // ensembles.h:25-186
 public:
  virtual ~Ensemble();
  virtual void Init();                                                   // ensembles.cc:24-29
  enum struct Integrator { EXPLICIT_EULER = 0, OPEN_DYNAMICS_ENGINE, IMPLICIT_MIDPOINT };
  // Step with the switch the reference left open (kSparseImplementation,
  // ensembles.cc:17-21) turned on: contacts as given (UpdateContacts is
  // the caller's until the collision row is built), velocities by the
  // matrix-free projected SOR on the GPU, positions by the midpoint rule.
  virtual void Step(double dt, Integrator g = Integrator::OPEN_DYNAMICS_ENGINE);
  // When Ensemble is first initialized, check for position errors, correct them
  // with StepPositionRelaxation (ensembles.cc:602-622); PostStabilize brings
  // positions AND velocities back to the constraint manifold (ensembles.cc:624-646).
  void InitStabilize();
  void PostStabilize(int max_steps = 500);
  const MatrixXd &M_inverse() const { return M_inverse_; }
  const ConstraintsList constraints() const { return CombineConstraintsLists(); }
  const ComponentsList &components() const { return components_; }
  void SetContacts(const ContactsList &c) { contacts_ = c; }
  // Find all contacts between Bodies or between Body and ground, clear and
  // update contacts_ (ensembles.cc:445-480), then drop contacts closer than
  // 1e-6 to an earlier contact of the same pair (ensembles.cc:308-328): on the GPU.
  void UpdateContacts();
  // Check whether there exist constraint pairs that are too close to each other (ensembles.cc:241-329):
  // conflicting joints between one pair of components are an error (the reference Panics; here
  // egs::Error with EGS_ERR_INVALID), a contact closer than 1e-6 to a joint or to an earlier contact
  // of the same pair is dropped.
  void CheckAndCorrectEnsembleState();
  bool use_device_step = true;   // false: the explicit UpdateContacts / StepVelocities_ODE / StepPositions_ODE calls
  // true: StepVelocities_ODE goes through the reference's LIVE dense path on the device (ComputeVDot,
  // ensembles.cc:498-538: dense J M^-1 J^T, condition check, conditional cfm, Lcp::MixedConstraintsSolver)
  // instead of the matrix-free sweeps; implies the explicit Step path.
  bool use_dense_solver = false;
  double last_condition_estimate = 0;
  bool detect_contacts = true;   // Step() calls UpdateContacts() as the reference does (ensembles.cc:393)
  const VectorXd GetVelocities() const;                                  // ensembles.cc:429-436
  VectorXd ComputePositionConstraintError() const;                       // ensembles.cc:156-171
  // solver parameters (compile-time constants in the reference)
  egs_solve_params solver_params;
  double cfm_coeff = 0.01;                                               // kCfmCoeff, ensembles.cc:14
  VectorXd last_lambda;
  int last_stabilize_steps = 0;

 protected:
  Ensemble();
  int n_ = 0;
  ComponentsList components_;
  JointsList joints_;
  ContactsList contacts_;
  MatrixXd M_inverse_;
  VectorXd external_force_torque_;

 private:
  void ConstructMassInertiaMatrixInverse();                              // ensembles.cc:202-212
  void InitializeExternalForceTorqueVector();                            // ensembles.cc:214-222
  ConstraintsList CombineConstraintsLists() const;                       // ensembles.cc:234-239
  void UpdateComponentsVelocities(const VectorXd &v);                    // ensembles.cc:438-443
  VectorXd StepVelocities_ODE(double dt, const VectorXd &v, double error_reduction_param = 0.2);  // :563-575
  void StepPositions_ODE(double dt, const VectorXd &v, const VectorXd &v_new);                    // :577-591
  bool StepOnDevice(double dt);
  // -step_scale * J^T (J J^T)^-1 err (ensembles.cc:659-666); the SPD(-semidefinite)
  // solve runs matrix-free on the GPU: the same projected sweep with M^-1 = I and
  // every row an equality.
  VectorXd CalculateVelocityRelaxation(double step_scale) const;
  void StepPositions_ExplicitEuler(double dt, const VectorXd &v);          // ensembles.cc:553-561
  void StepPositionRelaxation(double dt, double step_scale = 0.2);         // ensembles.cc:648-651
  void StepPostStabilization(double dt, double step_scale = 0.2);          // ensembles.cc:653-658
  egs_world *world_ = nullptr;
  int world_joints_ = -1;
  std::vector<int32_t> world_jb0_, world_jb1_;   // the joints the device world holds
  std::vector<double> world_jdata_;
  egs_problem *problem_ = nullptr;
  std::vector<int32_t> plan_b0_, plan_b1_;   // topology the cached device problem was planned for
};

class Chain : public Ensemble {  // ensembles.h:188-198, ensembles.cc:668-707
 public:
  Chain(int num_links, const Vector3d &anchor_position);
};

// ensembles.h:191-200, ensembles.cc:708-728: num_rocks boxes (m = 1, I = 0.1 I3) suspended at random
// positions inside the bounds with random rotations, velocities (|v_k| <= 1) and spins.  The
// reference draws from Eigen's Random (std::rand); so does this, through the same formulas
// (Vector3d::Random: 2 rand()/RAND_MAX - 1 per coefficient; Quaterniond::UnitRandom), so the
// scenario family is the reference's -- not its exact numbers, which depend on Eigen's call order.
class Cairn : public Ensemble {
 public:
  Cairn(int num_rocks, const std::array<double, 2> &x_bound, const std::array<double, 2> &y_bound,
        const std::array<double, 2> &z_bound);

 private:
  const double max_init_v_ = 1;
  const double max_init_w_ = 1;
};

#endif
