// eggshell_api.cpp -- see eggshell_api.h.  Host C++ only; all heavy arithmetic
// goes through the C ABI to the HIP library.  There is no CPU solver here: if
// the library cannot reach a GPU the calls throw egs::Error.
#include "eggshell_api.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>

namespace egs {
egs_context *DefaultContext() {
  static egs_context *ctx = nullptr;
  static std::once_flag once;
  static egs_status st = EGS_OK;
  std::call_once(once, [] { st = egs_context_create(0, &ctx); });
  if (!ctx) throw Error(st, "egs_context_create failed: no usable MI355X (there is no CPU fallback)");
  return ctx;
}
static void check(egs_status st) {
  if (st != EGS_OK) throw Error(st, egs_last_error(DefaultContext()));
}
}  // namespace egs

// ---- utils ---------------------------------------------------------------
Matrix3d CrossMat(const Vector3d &a) {  // utils.cc:16-24
  Matrix3d m;
  m(0, 1) = -a(2); m(0, 2) = a(1);
  m(1, 0) = a(2);  m(1, 2) = -a(0);
  m(2, 0) = -a(1); m(2, 1) = a(0);
  return m;
}

static Matrix3d QuatToR(double w, double x, double y, double z) {
  double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
  double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  Matrix3d R;
  R(0, 0) = 1.0 - (tyy + tzz); R(0, 1) = txy - twz;         R(0, 2) = txz + twy;
  R(1, 0) = txy + twz;         R(1, 1) = 1.0 - (txx + tzz); R(1, 2) = tyz - twx;
  R(2, 0) = txz - twy;         R(2, 1) = tyz + twx;         R(2, 2) = 1.0 - (txx + tyy);
  return R;
}

Matrix3d AlignVectors(const Vector3d &a, const Vector3d &b) {  // utils.cc:233-237
  auto normalized = [](const Vector3d &v) { double n2 = v.dot(v); return n2 > 0 ? v / std::sqrt(n2) : v; };
  Vector3d v0 = normalized(a), v1 = normalized(b);
  double c = v1.dot(v0);
  if (c < -1.0 + 1e-12) {  // deterministic axis instead of Eigen's SVD
    if (c < -1.0) c = -1.0;
    double ax = std::fabs(v0[0]), ay = std::fabs(v0[1]), az = std::fabs(v0[2]);
    Vector3d e = (ax <= ay && ax <= az) ? Vector3d(1, 0, 0) : (ay <= az ? Vector3d(0, 1, 0) : Vector3d(0, 0, 1));
    Vector3d axis = v0.cross(e);
    axis = axis / axis.norm();
    double w2 = (1.0 + c) * 0.5, sv = std::sqrt(1.0 - w2);
    return QuatToR(std::sqrt(w2), axis[0] * sv, axis[1] * sv, axis[2] * sv);
  }
  Vector3d axis = v0.cross(v1);
  double s = std::sqrt((1.0 + c) * 2.0), invs = 1.0 / s;
  return QuatToR(s * 0.5, axis[0] * invs, axis[1] * invs, axis[2] * invs);
}

Matrix3d WtoR(const Vector3d &w, double dt) {  // utils.cc:82-89
  double n2 = w.dot(w), nrm = std::sqrt(n2);
  Vector3d ax = n2 > 0 ? w / nrm : w;
  double half = 0.5 * (nrm * dt), s = std::sin(half), c = std::cos(half);
  return QuatToR(c, s * ax[0], s * ax[1], s * ax[2]);
}

Matrix3d Body::CalculateInertia(double m) const {  // body.cc:19-36
  Vector3d s = GetSideLengths();
  Matrix3d I;
  I(0, 0) = m / 12 * (s[1] * s[1] + s[2] * s[2]);
  I(1, 1) = m / 12 * (s[0] * s[0] + s[2] * s[2]);
  I(2, 2) = m / 12 * (s[0] * s[0] + s[1] * s[1]);
  return I;
}

// ---- joints.cc -----------------------------------------------------------
static VectorXd ToX(const Vector3d &v) { VectorXd x(3); for (int k = 0; k < 3; ++k) x(k) = v[k]; return x; }

VectorXd BallAndSocketJoint::ComputeError() const {  // joints.cc:3-11
  if (b1_ == nullptr) return ToX(b0_->p() + b0_->R() * c0_ - c1_);
  return ToX(b0_->p() + b0_->R() * c0_ - b1_->p() - b1_->R() * c1_);
}

static void SetBlock(MatrixXd *J, const Matrix3d &lin, const Matrix3d &ang) {
  J->resize(3, 6);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) { (*J)(r, c) = lin(r, c); (*J)(r, 3 + c) = ang(r, c); }
}

void BallAndSocketJoint::ComputeJ(MatrixXd *J_b0, MatrixXd *J_b1, ArrayXb *ct, VectorXd *lo, VectorXd *hi) const {
  SetBlock(J_b0, Matrix3d::Identity(), -1.0 * CrossMat(b0_->R() * c0_));  // joints.cc:21-22
  if (b1_ == nullptr) SetBlock(J_b1, Matrix3d::Zero(), Matrix3d::Zero());
  else SetBlock(J_b1, -1.0 * Matrix3d::Identity(), CrossMat(b1_->R() * c1_));
  ct->resize(3); lo->resize(3); hi->resize(3);
  for (int k = 0; k < 3; ++k) (*ct)(k) = 1;  // joints.cc:31-34
}

bool BallAndSocketJoint::Describe(int32_t *kind, double data[7]) const {
  *kind = EGS_JOINT_BALL;
  for (int k = 0; k < 3; ++k) { data[k] = c0_[k]; data[3 + k] = c1_[k]; }
  data[6] = 0;
  return true;
}

Vector3d BallAndSocketJoint::GetConstraintPosition() const {  // joints.cc:57-75
  const Vector3d p0 = b0_->p() + b0_->R() * c0_;
  if (b1_ == nullptr) return p0;
  return (p0 + (b1_->p() + b1_->R() * c1_)) / 2;
}

// ---- contact.cc ----------------------------------------------------------
VectorXd Contact::ComputeError() const {  // contact.cc:14-22
  VectorXd e(3);
  e(2) = -cg_.depth;
  return e;
}

void Contact::ComputeJ(MatrixXd *J_b0, MatrixXd *J_b1, ArrayXb *C, VectorXd *x_lo, VectorXd *x_hi) const {
  Matrix3d R = AlignVectors(cg_.normal, Vector3d(0, 0, 1));  // contact.cc:53-54
  if (b0_ == nullptr) SetBlock(J_b0, Matrix3d::Zero(), Matrix3d::Zero());
  else SetBlock(J_b0, R * (-1.0 * Matrix3d::Identity()), R * CrossMat(cg_.position - b0_->p()));
  if (b1_ == nullptr) SetBlock(J_b1, Matrix3d::Zero(), Matrix3d::Zero());
  else SetBlock(J_b1, R * Matrix3d::Identity(), R * (-1.0 * CrossMat(cg_.position - b1_->p())));
  C->resize(3); x_lo->resize(3); x_hi->resize(3);  // FrictionModel::BOX, contact.cc:103-113
  (*x_lo)(0) = -1; (*x_lo)(1) = -1; (*x_lo)(2) = 0;
  (*x_hi)(0) = 1; (*x_hi)(1) = 1; (*x_hi)(2) = std::numeric_limits<double>::infinity();
}

bool Contact::Describe(int32_t *kind, double data[7]) const {
  *kind = EGS_CONTACT_BOX;
  for (int k = 0; k < 3; ++k) { data[k] = cg_.position[k]; data[3 + k] = cg_.normal[k]; }
  data[6] = cg_.depth;
  return true;
}

// ---- flattening: ConstraintsList -> the arrays of entry 1 -------------------
namespace {
struct Flat {
  int n = 0, m = 0;
  std::vector<double> Minv, J0, J1, lo, hi;
  std::vector<int32_t> body0, body1;
  std::vector<uint8_t> is_eq;
};

Flat Flatten(const ConstraintsList &constraints, const MatrixXd &M_inverse) {
  Flat f;
  f.n = M_inverse.rows() / 6;
  f.m = (int)constraints.size();
  f.Minv.resize((size_t)f.n * 36);
  for (int b = 0; b < f.n; ++b)  // only block<6,6>(6b,6b) is ever read
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) f.Minv[(size_t)b * 36 + 6 * r + c] = M_inverse(6 * b + r, 6 * b + c);
  f.J0.assign((size_t)f.m * 18, 0.0); f.J1.assign((size_t)f.m * 18, 0.0);
  f.lo.resize((size_t)f.m * 3); f.hi.resize((size_t)f.m * 3); f.is_eq.resize((size_t)f.m * 3);
  f.body0.resize(f.m); f.body1.resize(f.m);
  for (int i = 0; i < f.m; ++i) {  // m ComputeJ calls (the reference makes O(m^2) per pass)
    MatrixXd j0, j1; ArrayXb ct; VectorXd lo, hi;
    constraints[i]->ComputeJ(&j0, &j1, &ct, &lo, &hi);
    if (j0.rows() != 3 || ct.size() != 3)
      throw egs::Error(EGS_ERR_INVALID, "only 3-row constraints are supported (joints.cc:18, contact.cc:103)");
    f.body0[i] = constraints[i]->i0_; f.body1[i] = constraints[i]->i1_;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 6; ++c) { f.J0[(size_t)i * 18 + 6 * r + c] = j0(r, c); f.J1[(size_t)i * 18 + 6 * r + c] = j1(r, c); }
      f.lo[(size_t)i * 3 + r] = lo(r); f.hi[(size_t)i * 3 + r] = hi(r); f.is_eq[(size_t)i * 3 + r] = ct(r) ? 1 : 0;
    }
  }
  return f;
}

sparse::LastSolve g_last = {0, 0.0, 0, 0, 0};

VectorXd Iterate(const ConstraintsList &constraints, const MatrixXd &M_inverse, const VectorXd &rhs, double cfm,
                 egs_method method) {
  if (constraints.empty()) return VectorXd(0);  // sparse_iterations.cc:152-154
  Flat f = Flatten(constraints, M_inverse);
  if (rhs.size() != 3 * f.m) throw egs::Error(EGS_ERR_INVALID, "rhs size != 3 * constraints");
  egs_solve_params prm;
  egs_default_params(&prm);  // 500 sweeps, tol 1e-9, omega 1.5: sparse_iterations.cc:15-19
  prm.method = method;
  prm.cfm = cfm;
  egs_solve_stats st;
  VectorXd x(3 * f.m);
  egs::check(egs_solve_blocks(egs::DefaultContext(), f.n, f.Minv.data(), f.m, f.body0.data(), f.body1.data(),
                              f.J0.data(), f.J1.data(), f.is_eq.data(), f.lo.data(), f.hi.data(), rhs.data(), &prm,
                              EGS_F64, x.data(), &st));
  g_last = {st.iterations, st.residual, st.n_islands, st.n_tiles, st.n_global};
  return x;
}
}  // namespace

VectorXd sparse::JacobiIteration(const ConstraintsList &c, const MatrixXd &M, const VectorXd &rhs, double cfm) {
  return Iterate(c, M, rhs, cfm, EGS_JACOBI);
}
VectorXd sparse::GaussSeidelIteration(const ConstraintsList &c, const MatrixXd &M, const VectorXd &rhs, double cfm) {
  return Iterate(c, M, rhs, cfm, EGS_GAUSS_SEIDEL);
}
VectorXd sparse::SORIteration(const ConstraintsList &c, const MatrixXd &M, const VectorXd &rhs, double cfm) {
  return Iterate(c, M, rhs, cfm, EGS_SOR);
}
sparse::LastSolve sparse::GetLastSolve() { return g_last; }

// ---- the same three on an explicit matrix (sparse_iterations.cc:72-144, 229-267) ----------------
namespace {
VectorXd IterateDense(const MatrixXd &A, const VectorXd &b, const ArrayXb *C, const VectorXd *x_lo, const VectorXd *x_hi, int method) {
  // CHECK(A.rows() == A.cols() && A.rows() == b.size()) (:77): the reference Panics
  if (A.rows() != A.cols() || A.rows() != b.size()) throw egs::Error(EGS_ERR_INVALID, "BaseIteration: A must be square, b of its size");
  if (C && (C->size() != b.size() || x_lo->size() != b.size() || x_hi->size() != b.size()))
    throw egs::Error(EGS_ERR_INVALID, "BaseIteration: C, x_lo, x_hi of b's size");
  const int n = b.size();
  VectorXd x(n);
  if (n == 0) return x;                            // :79-81
  egs_solve_params prm;
  egs_default_params(&prm);                       // omega 1.5, 500 sweeps, tol 1e-9: sparse_iterations.cc:15-19, constants.h:5
  prm.method = method;
  egs_solve_stats st;
  egs_status rc = egs_dense_iterate(egs::DefaultContext(), n, A.data(), b.data(), C ? C->data() : nullptr, C ? x_lo->data() : nullptr,
                                    C ? x_hi->data() : nullptr, &prm, x.data(), &st);
  if (rc != EGS_OK) throw egs::Error(rc, egs_last_error(egs::DefaultContext()));
  g_last = sparse::LastSolve{st.iterations, st.residual, 0, 0, 0};
  return x;
}
}  // namespace
VectorXd sparse::JacobiIteration(const MatrixXd &A, const VectorXd &b) { return IterateDense(A, b, nullptr, nullptr, nullptr, EGS_JACOBI); }
VectorXd sparse::JacobiIteration(const MatrixXd &A, const VectorXd &b, const ArrayXb &C, const VectorXd &lo, const VectorXd &hi) {
  return IterateDense(A, b, &C, &lo, &hi, EGS_JACOBI);
}
VectorXd sparse::GaussSeidelIteration(const MatrixXd &A, const VectorXd &b) { return IterateDense(A, b, nullptr, nullptr, nullptr, EGS_GAUSS_SEIDEL); }
VectorXd sparse::GaussSeidelIteration(const MatrixXd &A, const VectorXd &b, const ArrayXb &C, const VectorXd &lo, const VectorXd &hi) {
  return IterateDense(A, b, &C, &lo, &hi, EGS_GAUSS_SEIDEL);
}
VectorXd sparse::SORIteration(const MatrixXd &A, const VectorXd &b) { return IterateDense(A, b, nullptr, nullptr, nullptr, EGS_SOR); }
VectorXd sparse::SORIteration(const MatrixXd &A, const VectorXd &b, const ArrayXb &C, const VectorXd &lo, const VectorXd &hi) {
  return IterateDense(A, b, &C, &lo, &hi, EGS_SOR);
}

// ---- sparse_iterations_utils.cc:427-695 ------------------------------------------
namespace {
VectorXd Product(const ConstraintsList &constraints, const MatrixXd &M_inverse, const VectorXd &x, int32_t parts,
                 double eps, double scale) {
  if (constraints.empty()) return VectorXd(0);
  Flat f = Flatten(constraints, M_inverse);
  if (x.size() != 3 * f.m) throw egs::Error(EGS_ERR_INVALID, "x size != 3 * constraints");
  VectorXd y(3 * f.m);
  egs::check(egs_matvec_blocks(egs::DefaultContext(), f.n, f.Minv.data(), f.m, f.body0.data(), f.body1.data(), f.J0.data(),
                               f.J1.data(), parts, eps, scale, EGS_F64, x.data(), y.data()));
  return y;
}
}  // namespace

VectorXd sparse::CalculateSparseDx(const ConstraintsList &c, const MatrixXd &M, const VectorXd &x, double epsilon, double scale) {
  return Product(c, M, x, EGS_MV_DIAG, epsilon, scale);
}
VectorXd sparse::CalculateSparseLx(const ConstraintsList &c, const MatrixXd &M, const VectorXd &x, double, double) {
  return Product(c, M, x, EGS_MV_LOWER, 0.0, 1.0);   // epsilon / scale unused (sparse_iterations_utils.h:69-71)
}
VectorXd sparse::CalculateSparseUx(const ConstraintsList &c, const MatrixXd &M, const VectorXd &x, double, double) {
  return Product(c, M, x, EGS_MV_UPPER, 0.0, 1.0);
}
VectorXd sparse::CalculateSparseLxUx(const ConstraintsList &c, const MatrixXd &M, const VectorXd &x, double, double) {
  return Product(c, M, x, EGS_MV_LOWER | EGS_MV_UPPER, 0.0, 1.0);
}
VectorXd sparse::CalculateSparseLxDx(const ConstraintsList &c, const MatrixXd &M, const VectorXd &x, double eps, double scale) {
  return Product(c, M, x, EGS_MV_LOWER | EGS_MV_DIAG, eps, scale);
}
VectorXd sparse::CalculateSparseUxDx(const ConstraintsList &c, const MatrixXd &M, const VectorXd &x, double eps, double scale) {
  return Product(c, M, x, EGS_MV_UPPER | EGS_MV_DIAG, eps, scale);
}
VectorXd sparse::CalculateSparseJMJtX(const ConstraintsList &c, const MatrixXd &M, const VectorXd &x, double eps) {
  return Product(c, M, x, EGS_MV_FULL, eps, 1.0);
}

void sparse::ConstructMixedConstraints(const ConstraintsList &constraints, ArrayXb *C, VectorXd *x_lo,
                                       VectorXd *x_hi) {  // sparse_iterations_utils.cc:697-720
  const int m = (int)constraints.size();
  C->resize(3 * m); x_lo->resize(3 * m); x_hi->resize(3 * m);
  for (int i = 0; i < m; ++i) {
    MatrixXd j0, j1; ArrayXb ct; VectorXd lo, hi;
    constraints[i]->ComputeJ(&j0, &j1, &ct, &lo, &hi);
    for (int r = 0; r < 3; ++r) { (*C)(3 * i + r) = ct(r); (*x_lo)(3 * i + r) = lo(r); (*x_hi)(3 * i + r) = hi(r); }
  }
}

bool Lcp::MixedConstraintsSolver(const MatrixXd &A, const VectorXd &b, const ArrayXb &C, const VectorXd &x_lo,
                                 const VectorXd &x_hi, VectorXd &x, VectorXd &w) {  // lcp.cc:276-336
  const int N = b.size();
  if (A.rows() != N || A.cols() != N || C.size() != N) throw egs::Error(EGS_ERR_INVALID, "dimension mismatch");
  x.resize(N); w.resize(N);
  int32_t ok = 0, pivots = 0;
  egs_status st = egs_mixed_constraints_solve(egs::DefaultContext(), N, A.data(), b.data(), C.data(), x_lo.data(),
                                              x_hi.data(), /*use_bounds=*/0, x.data(), w.data(), &ok, &pivots);
  if (st != EGS_OK && st != EGS_ERR_LCP_FAILED) egs::check(st);
  return ok != 0;
}

// ---- ensembles.cc ----------------------------------------------------------
Ensemble::Ensemble() {
  egs_default_params(&solver_params);
  solver_params.method = EGS_SOR;
}
Ensemble::~Ensemble() {
  if (problem_) egs_problem_destroy(problem_);
  if (world_) egs_world_destroy(world_);
}

void Ensemble::Init() {  // ensembles.cc:24-29
  if (world_) { egs_world_destroy(world_); world_ = nullptr; }   // M^-1 / f_ext are re-sent to a fresh world
  ConstructMassInertiaMatrixInverse();
  InitializeExternalForceTorqueVector();
  VectorXd err = ComputePositionConstraintError();  // CheckInitialConditions, :224-232
  for (int k = 0; k < err.size(); ++k)
    if (!(std::fabs(err(k)) <= 1e-9)) throw egs::Error(EGS_ERR_INVALID, "Check initial conditions failed.");
  CheckAndCorrectEnsembleState();                   // ensembles.cc:28
}

void Ensemble::ConstructMassInertiaMatrixInverse() {  // ensembles.cc:202-212
  M_inverse_ = MatrixXd::Zero(n_ * 6, n_ * 6);
  for (int i = 0; i < n_; ++i) {
    const auto &b = components_.at(i);
    Matrix3d Iinv = b->I_g().inverse();
    for (int k = 0; k < 3; ++k) M_inverse_(6 * i + k, 6 * i + k) = 1.0 / b->m();
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) M_inverse_(6 * i + 3 + r, 6 * i + 3 + c) = Iinv(r, c);
  }
}

void Ensemble::InitializeExternalForceTorqueVector() {  // ensembles.cc:214-222
  external_force_torque_ = VectorXd::Zero(n_ * 6);
  const Vector3d g(0, 0, -9.8);  // constants.h:8
  for (int i = 0; i < n_; ++i) {
    const auto &b = components_.at(i);
    Vector3d tq = ((-1.0 * CrossMat(b->w_g())) * b->I_g()) * b->w_g();
    for (int k = 0; k < 3; ++k) {
      external_force_torque_(6 * i + k) = b->m() * g[k];
      external_force_torque_(6 * i + 3 + k) = tq[k];
    }
  }
}

ConstraintsList Ensemble::CombineConstraintsLists() const {  // ensembles.cc:234-239
  ConstraintsList c;
  c.insert(c.end(), joints_.begin(), joints_.end());
  c.insert(c.end(), contacts_.begin(), contacts_.end());
  return c;
}

VectorXd Ensemble::ComputePositionConstraintError() const {  // ensembles.cc:156-171
  ConstraintsList cs = CombineConstraintsLists();
  VectorXd e(3 * (int)cs.size());
  for (size_t i = 0; i < cs.size(); ++i) {
    VectorXd ei = cs[i]->ComputeError();
    for (int k = 0; k < 3; ++k) e(3 * (int)i + k) = ei(k);
  }
  return e;
}

const VectorXd Ensemble::GetVelocities() const {  // ensembles.cc:429-436
  VectorXd v(n_ * 6);
  for (int i = 0; i < n_; ++i)
    for (int k = 0; k < 3; ++k) { v(6 * i + k) = components_[i]->v()[k]; v(6 * i + 3 + k) = components_[i]->w_g()[k]; }
  return v;
}

void Ensemble::UpdateComponentsVelocities(const VectorXd &v) {  // ensembles.cc:438-443
  for (int i = 0; i < n_; ++i) {
    components_[i]->SetV(Vector3d(v(6 * i), v(6 * i + 1), v(6 * i + 2)));
    components_[i]->SetW_GlobalFrame(Vector3d(v(6 * i + 3), v(6 * i + 4), v(6 * i + 5)));
  }
}

// ensembles.cc:563-575 with the sparse switch on.  If every constraint can
// describe itself the whole velocity step (assembly, rhs, solve, v update) runs
// on the GPU (entry 2); otherwise ComputeJ is flattened on the host (entry 1).
VectorXd Ensemble::StepVelocities_ODE(double dt, const VectorXd &v, double erp) {
  ConstraintsList cs = CombineConstraintsLists();
  const int m = (int)cs.size();
  if (m == 0) {  // ensembles.cc:504-505: v_dot = M^-1 f
    VectorXd v_new(n_ * 6);
    for (int i = 0; i < n_; ++i)
      for (int r = 0; r < 6; ++r) {
        double s = 0;
        for (int c = 0; c < 6; ++c) s += M_inverse_(6 * i + r, 6 * i + c) * external_force_torque_(6 * i + c);
        v_new(6 * i + r) = v(6 * i + r) + dt * s;
      }
    UpdateComponentsVelocities(v_new);
    return v_new;
  }
  std::vector<int32_t> kind(m), b0(m), b1(m);
  std::vector<double> data((size_t)m * 7);
  bool describable = true;
  for (int i = 0; i < m; ++i) {
    b0[i] = cs[i]->i0_; b1[i] = cs[i]->i1_;
    describable = describable && cs[i]->Describe(&kind[i], &data[(size_t)i * 7]);
  }
  egs_solve_params prm = solver_params;
  prm.cfm = cfm_coeff;
  VectorXd v_new(n_ * 6);
  if (describable) {
    if (!problem_ || b0 != plan_b0_ || b1 != plan_b1_) {  // re-plan only when the contact topology changed
      if (problem_) egs_problem_destroy(problem_);
      problem_ = nullptr;
      egs::check(egs_problem_create(egs::DefaultContext(), n_, m, b0.data(), b1.data(), EGS_F64, &problem_));
      plan_b0_ = b0; plan_b1_ = b1;
    }
    std::vector<double> pos((size_t)n_ * 3), R((size_t)n_ * 9), vl((size_t)n_ * 3), w((size_t)n_ * 3), Minv((size_t)n_ * 36);
    for (int i = 0; i < n_; ++i) {
      for (int k = 0; k < 3; ++k) { pos[3 * i + k] = components_[i]->p()[k]; vl[3 * i + k] = v(6 * i + k); w[3 * i + k] = v(6 * i + 3 + k); }
      for (int k = 0; k < 9; ++k) R[9 * i + k] = components_[i]->R().d[k];
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) Minv[(size_t)i * 36 + 6 * r + c] = M_inverse_(6 * i + r, 6 * i + c);
    }
    egs::check(egs_problem_set_state(problem_, pos.data(), R.data(), vl.data(), w.data(), Minv.data(), external_force_torque_.data()));
    egs::check(egs_problem_set_constraints(problem_, kind.data(), data.data()));
    egs_solve_stats st;
    if (use_dense_solver) {   // ComputeVDot (ensembles.cc:498-538) on the device
      egs::check(egs_problem_assemble(problem_, dt, erp));
      egs::check(egs_problem_dense_condition(problem_, 0.0, &last_condition_estimate));
      const double cfm = last_condition_estimate < 1e7 ? 0.0 : cfm_coeff;   // kGoodConditionNumber, constants.h:12
      int32_t ok = 0, pivots = 0;
      egs_status rc = egs_problem_step_dense(problem_, dt, erp, cfm, /*use_bounds=*/0, &ok, &pivots);
      if (rc != EGS_OK || !ok)   // the reference Panics here (ensembles.cc:531-534)
        throw egs::Error(rc != EGS_OK ? rc : EGS_ERR_LCP_FAILED, "Lcp::MixedConstraintsSolver exited without reaching a solution.");
    } else
    egs::check(egs_problem_step(problem_, dt, erp, &prm, &st));
    last_lambda.resize(3 * m);
    egs::check(egs_problem_get_lambda(problem_, last_lambda.data()));
    egs::check(egs_problem_get_velocity(problem_, v_new.data()));
  } else {
    Flat f = Flatten(cs, M_inverse_);
    VectorXd err = ComputePositionConstraintError();
    VectorXd rhs(3 * m);
    const double k = -erp / dt / dt;
    auto u_of = [&](int b, double *u) {
      for (int r = 0; r < 6; ++r) {
        double s = 0;
        for (int c = 0; c < 6; ++c) s += M_inverse_(6 * b + r, 6 * b + c) * external_force_torque_(6 * b + c);
        u[r] = v(6 * b + r) / dt + s;
      }
    };
    for (int i = 0; i < m; ++i) {
      double u0[6] = {0}, u1[6] = {0};
      if (f.body0[i] >= 0) u_of(f.body0[i], u0);
      if (f.body1[i] >= 0) u_of(f.body1[i], u1);
      for (int r = 0; r < 3; ++r) {
        double ju = 0;
        for (int c = 0; c < 6; ++c) ju += f.J0[(size_t)i * 18 + 6 * r + c] * u0[c] + f.J1[(size_t)i * 18 + 6 * r + c] * u1[c];
        rhs(3 * i + r) = k * err(3 * i + r) - ju;
      }
    }
    egs_solve_stats st;
    last_lambda.resize(3 * m);
    egs::check(egs_solve_blocks(egs::DefaultContext(), f.n, f.Minv.data(), f.m, f.body0.data(), f.body1.data(), f.J0.data(),
                                f.J1.data(), f.is_eq.data(), f.lo.data(), f.hi.data(), rhs.data(), &prm, EGS_F64,
                                last_lambda.data(), &st));
    std::vector<double> g((size_t)n_ * 6);
    for (int i = 0; i < n_ * 6; ++i) g[i] = external_force_torque_(i);
    for (int i = 0; i < m; ++i)
      for (int side = 0; side < 2; ++side) {
        const int b = side ? f.body1[i] : f.body0[i];
        if (b < 0) continue;
        const double *J = (side ? f.J1.data() : f.J0.data()) + (size_t)i * 18;
        for (int c = 0; c < 6; ++c)
          for (int r = 0; r < 3; ++r) g[(size_t)b * 6 + c] += J[6 * r + c] * last_lambda(3 * i + r);
      }
    for (int b = 0; b < n_; ++b)
      for (int r = 0; r < 6; ++r) {
        double s = 0;
        for (int c = 0; c < 6; ++c) s += M_inverse_(6 * b + r, 6 * b + c) * g[(size_t)b * 6 + c];
        v_new(6 * b + r) = v(6 * b + r) + dt * s;
      }
  }
  UpdateComponentsVelocities(v_new);
  return v_new;
}

void Ensemble::StepPositions_ODE(double dt, const VectorXd &v, const VectorXd &v_new) {  // ensembles.cc:577-591
  for (int i = 0; i < n_; ++i) {
    Vector3d vm, wm;
    for (int k = 0; k < 3; ++k) { vm[k] = (v(6 * i + k) + v_new(6 * i + k)) / 2.0; wm[k] = (v(6 * i + 3 + k) + v_new(6 * i + 3 + k)) / 2.0; }
    components_[i]->SetP(components_[i]->p() + dt * vm);
    components_[i]->SetR(WtoR(wm, dt) * components_[i]->R());
  }
}

void Ensemble::UpdateContacts() {  // ensembles.cc:445-480 (+ :308-328)
  contacts_.clear();
  if (n_ == 0) return;
  std::vector<double> pos((size_t)n_ * 3), R((size_t)n_ * 9), side((size_t)n_ * 3);
  for (int i = 0; i < n_; ++i) {
    const Vector3d sl = components_[i]->GetSideLengths();
    for (int k = 0; k < 3; ++k) { pos[3 * i + k] = components_[i]->p()[k]; side[3 * i + k] = sl[k]; }
    for (int k = 0; k < 9; ++k) R[9 * i + k] = components_[i]->R().d[k];
  }
  const int cap = 64 * n_ + 64;
  std::vector<int32_t> b0(cap), b1(cap);
  std::vector<double> data((size_t)cap * 7);
  int32_t m = 0;
  // The reference's Step prunes right after detection (CheckAndCorrectEnsembleState, ensembles.cc:394):
  // contact-vs-contact always, joint-vs-contact for the joints between the same two components.  Joints
  // that can describe themselves are pruned against on the device, the others on the host below.
  std::vector<int32_t> jb0, jb1;
  std::vector<double> jdata;
  std::vector<std::shared_ptr<Joint>> host_joints;
  for (const auto &j : joints_) {
    if (j->i0_ < 0 || j->i1_ < 0) continue;   // the pair scan never visits the ground (quirk Q4)
    int32_t kind = 0;
    double d7[7];
    if (j->Describe(&kind, d7)) { jb0.push_back(j->i0_); jb1.push_back(j->i1_); jdata.insert(jdata.end(), d7, d7 + 7); }
    else host_joints.push_back(j);
  }
  egs::check(egs_update_contacts_joints(egs::DefaultContext(), n_, pos.data(), R.data(), side.data(), (int32_t)jb0.size(),
                                        jb0.data(), jb1.data(), jdata.data(), cap, &m, b0.data(), b1.data(), data.data()));
  for (int k = 0; k < m; ++k) {
    const double *d = &data[(size_t)k * 7];
    ContactGeometry cg(Vector3d(d[0], d[1], d[2]), Vector3d(d[3], d[4], d[5]), d[6]);
    bool keep = true;
    for (const auto &j : host_joints) {        // ensembles.cc:296-306, kMinConstraintDistance = 1e-6
      const bool same_pair = (j->i0_ == b0[k] && j->i1_ == b1[k]) || (j->i0_ == b1[k] && j->i1_ == b0[k]);
      if (same_pair && (j->GetConstraintPosition() - cg.position).norm() < 1e-6) keep = false;
    }
    if (!keep) continue;
    if (b0[k] < 0) contacts_.push_back(std::make_shared<Contact>(components_[b1[k]], b1[k], cg));
    else contacts_.push_back(std::make_shared<Contact>(components_[b0[k]], b0[k], components_[b1[k]], b1[k], cg));
  }
}

// ensembles.cc:241-329.  UpdateContacts already returns a pruned contact list (the device does the
// contact-vs-contact and joint-vs-contact passes in the reference's order), so what is left for
// caller-supplied contacts (SetContacts) is the same scan on the host, and the joint-vs-joint check.
void Ensemble::CheckAndCorrectEnsembleState() {
  if (M_inverse_.rows() != 6 * n_ || M_inverse_.cols() != 6 * n_) throw egs::Error(EGS_ERR_INVALID, "M_inverse_ dimensions are incorrect.");
  if (external_force_torque_.size() != 6 * n_) throw egs::Error(EGS_ERR_INVALID, "external_force_torque_ dimensions are incorrect.");
  auto key_of = [](int a, int b) { return a < b ? std::make_pair(a, b) : std::make_pair(b, a); };
  auto close = [](const Constraint &c1, const Constraint &c2) {   // CheckConstraintPair, ensembles.cc:376-388
    return (c1.GetConstraintPosition() - c2.GetConstraintPosition()).norm() < 1e-6;
  };
  // joint vs joint: conflict or overconstraint -> the reference Panics (ensembles.cc:280-289)
  for (size_t a = 0; a < joints_.size(); ++a)
    for (size_t b = a + 1; b < joints_.size(); ++b) {
      if (joints_[a]->i0_ < 0 || joints_[a]->i1_ < 0) continue;   // pairs 0 <= i < j only (quirk Q4)
      if (key_of(joints_[a]->i0_, joints_[a]->i1_) != key_of(joints_[b]->i0_, joints_[b]->i1_)) continue;
      if (close(*joints_[a], *joints_[b]))
        throw egs::Error(EGS_ERR_INVALID, "Joint constraints between components " + std::to_string(key_of(joints_[a]->i0_, joints_[a]->i1_).first) +
                                              " and " + std::to_string(key_of(joints_[a]->i0_, joints_[a]->i1_).second) +
                                              " conflict or cause overconstraint.");
    }
  // joint vs contact, then contact vs contact (the later one goes), pair by pair (the reference's
  // pairwise maps, ensembles.cc:331-374); erased in descending order
  std::map<std::pair<int, int>, std::vector<int>> pair_joints, pair_contacts;
  for (size_t j = 0; j < joints_.size(); ++j)
    if (joints_[j]->i0_ >= 0 && joints_[j]->i1_ >= 0) pair_joints[key_of(joints_[j]->i0_, joints_[j]->i1_)].push_back((int)j);
  for (size_t c = 0; c < contacts_.size(); ++c)
    if (contacts_[c]->i0_ >= 0 && contacts_[c]->i1_ >= 0) pair_contacts[key_of(contacts_[c]->i0_, contacts_[c]->i1_)].push_back((int)c);
  std::vector<char> drop(contacts_.size(), 0);
  for (const auto &pc : pair_contacts) {
    const auto pj = pair_joints.find(pc.first);
    for (size_t a = 0; a < pc.second.size(); ++a) {
      const int c = pc.second[a];
      if (pj != pair_joints.end())
        for (int j : pj->second) if (close(*joints_[j], *contacts_[c])) drop[c] = 1;
      for (size_t e = 0; e < a; ++e)   // every earlier contact of the pair takes part, dropped or not (ensembles.cc:308-322)
        if (close(*contacts_[pc.second[e]], *contacts_[c])) drop[c] = 1;
    }
  }
  for (size_t c = contacts_.size(); c-- > 0;)
    if (drop[c]) contacts_.erase(contacts_.begin() + (long)c);
}

// The whole Step through egs_world (include/eggshell_amd.h): possible when every
// permanent constraint can describe itself (ball joints) -- contacts always can.
// Body objects are the interface, so their state is pushed before and pulled
// after the step; inside the step nothing but the contact topology leaves the GPU.
bool Ensemble::StepOnDevice(double dt) {
  if (!use_device_step || use_dense_solver) return false;
  const int mj = (int)joints_.size();
  std::vector<int32_t> jb0(mj), jb1(mj), kind(1);
  std::vector<double> jdata((size_t)mj * 7);
  for (int i = 0; i < mj; ++i) {
    jb0[i] = joints_[i]->i0_; jb1[i] = joints_[i]->i1_;
    if (!joints_[i]->Describe(&kind[0], &jdata[(size_t)i * 7])) return false;
  }
  if (!detect_contacts && !contacts_.empty()) return false;   // caller-supplied contacts: use the explicit path
  egs_context *ctx = egs::DefaultContext();
  const bool first = !world_;
  if (!world_) {
    egs::check(egs_world_create(ctx, n_, EGS_F64, &world_));
    world_joints_ = -1;
  }
  std::vector<double> pos((size_t)n_ * 3), R((size_t)n_ * 9), vl((size_t)n_ * 3), w((size_t)n_ * 3), Minv((size_t)n_ * 36),
      side((size_t)n_ * 3);
  for (int i = 0; i < n_; ++i) {
    const Vector3d sl = components_[i]->GetSideLengths();
    for (int k = 0; k < 3; ++k) {
      pos[3 * i + k] = components_[i]->p()[k]; vl[3 * i + k] = components_[i]->v()[k];
      w[3 * i + k] = components_[i]->w_g()[k]; side[3 * i + k] = sl[k];
    }
    for (int k = 0; k < 9; ++k) R[9 * i + k] = components_[i]->R().d[k];
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) Minv[(size_t)i * 36 + 6 * r + c] = M_inverse_(6 * i + r, 6 * i + c);
  }
  // M^-1, the external force and the box sizes are frozen at Init() (Q5): sent once
  egs::check(egs_world_set_bodies(world_, pos.data(), R.data(), vl.data(), w.data(), first ? Minv.data() : nullptr,
                                  first ? external_force_torque_.data() : nullptr, first ? side.data() : nullptr));
  // joints are permanent in the reference (ensembles.cc:331-334); a caller that edits them all the same
  // (same count, other bodies or anchors) must not step against the stale device copy
  if (world_joints_ != mj || jb0 != world_jb0_ || jb1 != world_jb1_ || jdata != world_jdata_) {
    egs::check(egs_world_set_joints(world_, mj, jb0.data(), jb1.data(), jdata.data()));
    world_joints_ = mj;
    world_jb0_ = jb0; world_jb1_ = jb1; world_jdata_ = jdata;
  }
  egs_solve_params prm = solver_params;
  prm.cfm = cfm_coeff;
  egs_solve_stats st;
  egs::check(egs_world_step(world_, dt, /*erp=*/0.2, &prm, detect_contacts ? 1 : 0, &st));
  egs::check(egs_world_get_bodies(world_, pos.data(), R.data(), vl.data(), w.data()));
  for (int i = 0; i < n_; ++i) {
    components_[i]->SetP(Vector3d(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
    components_[i]->SetV(Vector3d(vl[3 * i], vl[3 * i + 1], vl[3 * i + 2]));
    components_[i]->SetW_GlobalFrame(Vector3d(w[3 * i], w[3 * i + 1], w[3 * i + 2]));
    Matrix3d Rm;
    for (int k = 0; k < 9; ++k) Rm.d[k] = R[9 * i + k];
    components_[i]->SetR(Rm);
  }
  int32_t mcons = 0, mc = 0, replans = 0;
  egs::check(egs_world_info(world_, &mcons, &mc, &replans));
  contacts_.clear();
  if (mc > 0) {   // the contact list the step used (for constraints(), Draw(), ...)
    std::vector<int32_t> b0(mc), b1(mc);
    std::vector<double> data((size_t)mc * 7);
    int32_t got = 0;
    egs::check(egs_world_get_contacts(world_, mc, &got, b0.data(), b1.data(), data.data()));
    for (int k = 0; k < mc; ++k) {
      const double *d = &data[(size_t)k * 7];
      ContactGeometry cg(Vector3d(d[0], d[1], d[2]), Vector3d(d[3], d[4], d[5]), d[6]);
      if (b0[k] < 0) contacts_.push_back(std::make_shared<Contact>(components_[b1[k]], b1[k], cg));
      else contacts_.push_back(std::make_shared<Contact>(components_[b0[k]], b0[k], components_[b1[k]], b1[k], cg));
    }
  }
  last_lambda.resize(3 * mcons);
  if (mcons > 0) {
    int32_t rows = 0;
    egs::check(egs_world_get_lambda(world_, 3 * mcons, &rows, last_lambda.data()));
  }
  return true;
}

void Ensemble::Step(double dt, Integrator g) {  // ensembles.cc:390-427
  // The reference cannot complete a step with either of the other two: IMPLICIT_MIDPOINT Panics outright
  // (ensembles.cc:403-405) and EXPLICIT_EULER reaches ComputeJDotV_Joints(), whose first statement is a Panic
  // (ensembles.cc:94-97). There is no behaviour to reproduce, so both are refused with the reference's reason.
  if (g == Integrator::IMPLICIT_MIDPOINT)
    throw egs::Error(EGS_ERR_UNSUPPORTED,
                     "Implicit midpoint integrator is not properly implemented and tested (ensembles.cc:403-405)");
  if (g != Integrator::OPEN_DYNAMICS_ENGINE)
    throw egs::Error(EGS_ERR_UNSUPPORTED,
                     "Integrator::EXPLICIT_EULER panics in the reference (ComputeJDotV_Joints, ensembles.cc:94-97); "
                     "only Integrator::OPEN_DYNAMICS_ENGINE completes a step");
  if (StepOnDevice(dt)) return;   // collide -> solve -> integrate in one resident pipeline
  const VectorXd v = GetVelocities();
  if (detect_contacts) UpdateContacts();
  CheckAndCorrectEnsembleState();   // ensembles.cc:394
  VectorXd v_new = StepVelocities_ODE(dt, v);
  StepPositions_ODE(dt, v, v_new);
}

Chain::Chain(int num_links, const Vector3d &anchor) {  // ensembles.cc:668-707
  if (num_links <= 0) throw egs::Error(EGS_ERR_INVALID, "num_links > 0");
  n_ = num_links;
  const double az = 0.95531661812451, ax = M_PI / 4;
  double qz_w = std::cos(az / 2), qz_z = std::sin(az / 2), qx_w = std::cos(ax / 2), qx_x = std::sin(ax / 2);
  Matrix3d R = QuatToR(qz_w * qx_w, qz_w * qx_x, qz_z * qx_x, qz_z * qx_w);
  for (int i = 0; i < n_; ++i) {
    Vector3d p(std::sqrt(3.0) * 0.3 * i + anchor[0], 0 + anchor[1], 0 + anchor[2]);
    components_.push_back(std::make_shared<Body>(p, Vector3d::Zero(), R, Vector3d::Zero()));
  }
  const Vector3d c1(0.15, -0.15, 0.15), c2(-0.15, 0.15, -0.15);
  for (int i = 0; i < n_ - 1; ++i)
    joints_.push_back(std::make_shared<BallAndSocketJoint>(components_[i], i, c1, components_[i + 1], i + 1, c2));
  joints_.push_back(std::make_shared<BallAndSocketJoint>(components_[0], 0, Vector3d::Zero(), components_[0]->p()));
}

// ---- Cairn (ensembles.cc:708-728) ------------------------------------------------
namespace {
double Rand01() { return double(std::rand()) / double(RAND_MAX); }          // Eigen internal::random<double>(0, 1)
double RandPm1() { return -1.0 + 2.0 * Rand01(); }                          // ... (-1, 1): Vector3d::Random() per coefficient
Vector3d RandomVector() { double a = RandPm1(), b = RandPm1(), c = RandPm1(); return Vector3d(a, b, c); }
}  // namespace

Cairn::Cairn(int num_rocks, const std::array<double, 2> &xb, const std::array<double, 2> &yb, const std::array<double, 2> &zb) {
  if (num_rocks < 0) throw egs::Error(EGS_ERR_INVALID, "num_rocks >= 0");
  n_ = num_rocks;
  const Matrix3d I = Matrix3d::Identity() * 0.1;
  for (int i = 0; i < num_rocks; ++i) {
    // RandomPosition, utils.cc:26-38
    Vector3d u = (RandomVector() + Vector3d(1, 1, 1)) / 2;
    Vector3d p(u[0] * std::fabs(xb[1] - xb[0]) + std::min(xb[0], xb[1]), u[1] * std::fabs(yb[1] - yb[0]) + std::min(yb[0], yb[1]),
               u[2] * std::fabs(zb[1] - zb[0]) + std::min(zb[0], zb[1]));
    // RandomRotationViaQuaternion -> Quaterniond::UnitRandom (utils.cc:52-55)
    const double u1 = Rand01(), u2 = 2 * M_PI * Rand01(), u3 = 2 * M_PI * Rand01();
    const double a = std::sqrt(1 - u1), b = std::sqrt(u1);
    const Matrix3d R = QuatToR(a * std::sin(u2), a * std::cos(u2), b * std::sin(u3), b * std::cos(u3));
    const Vector3d v = RandomVector() * max_init_v_, w = RandomVector() * max_init_w_;   // utils.cc:40-48
    components_.push_back(std::make_shared<Body>(p, v, 1.0, R, w, I));
  }
}
