// stabilize.cpp -- Ensemble::InitStabilize / PostStabilize and their helpers
// (ensembles.cc:602-666), part of the C++ adapter (see eggshell_api.h).
// The reference solves (J J^T) y = err with a dense LDLT; here the same system
// goes through the matrix-free GPU sweep (entry 1 of the C ABI) with M^-1 = I
// and all rows equalities.  J J^T is only positive SEMI-definite when contacts
// are redundant; the correction J^T y is unique whenever err is consistent, and
// the sweep converges to it.
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#include "eggshell_api.h"

namespace {
constexpr double kAllowNumericalError = 1e-9;  // constants.h:5
constexpr double kSimTimeStep = 0.001;         // constants.h:6
}  // namespace

VectorXd Ensemble::CalculateVelocityRelaxation(double step_scale) const {  // ensembles.cc:659-666
  const ConstraintsList cs = constraints();
  const int m = (int)cs.size();
  VectorXd corr(6 * n_);
  if (m == 0) return corr;
  std::vector<double> Minv((size_t)n_ * 36, 0.0), J0((size_t)m * 18), J1((size_t)m * 18), lo((size_t)m * 3, 0.0),
      hi((size_t)m * 3, 0.0), x((size_t)m * 3);
  std::vector<int32_t> b0(m), b1(m);
  std::vector<uint8_t> is_eq((size_t)m * 3, 1);
  for (int b = 0; b < n_; ++b)
    for (int k = 0; k < 6; ++k) Minv[(size_t)b * 36 + 7 * k] = 1.0;
  for (int i = 0; i < m; ++i) {
    MatrixXd j0, j1; ArrayXb ct; VectorXd clo, chi;
    cs[i]->ComputeJ(&j0, &j1, &ct, &clo, &chi);
    b0[i] = cs[i]->i0_; b1[i] = cs[i]->i1_;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 6; ++c) { J0[(size_t)i * 18 + 6 * r + c] = j0(r, c); J1[(size_t)i * 18 + 6 * r + c] = j1(r, c); }
  }
  const VectorXd err = ComputePositionConstraintError();
  egs_solve_params prm;
  egs_default_params(&prm);
  prm.method = EGS_SOR; prm.cfm = 0.0; prm.tol = 1e-11; prm.max_iters = 20000; prm.check_every = 10;
  egs_solve_stats st;
  egs_status rc = egs_solve_blocks(egs::DefaultContext(), n_, Minv.data(), m, b0.data(), b1.data(), J0.data(), J1.data(),
                                   is_eq.data(), lo.data(), hi.data(), err.data(), &prm, EGS_F64, x.data(), &st);
  if (rc != EGS_OK) throw egs::Error(rc, egs_last_error(egs::DefaultContext()));
  for (int i = 0; i < m; ++i)
    for (int side = 0; side < 2; ++side) {
      const int b = side ? b1[i] : b0[i];
      if (b < 0) continue;
      const double *J = (side ? J1.data() : J0.data()) + (size_t)i * 18;
      for (int c = 0; c < 6; ++c)
        for (int r = 0; r < 3; ++r) corr(6 * b + c) += J[6 * r + c] * x[(size_t)i * 3 + r];
    }
  return corr * (-1.0 * step_scale);
}

void Ensemble::StepPositions_ExplicitEuler(double dt, const VectorXd &v) {  // ensembles.cc:553-561
  for (int i = 0; i < n_; ++i) {
    components_[i]->SetP(components_[i]->p() + dt * Vector3d(v(6 * i), v(6 * i + 1), v(6 * i + 2)));
    components_[i]->SetR(WtoR(Vector3d(v(6 * i + 3), v(6 * i + 4), v(6 * i + 5)), dt) * components_[i]->R());
  }
}

void Ensemble::StepPositionRelaxation(double dt, double step_scale) {  // ensembles.cc:648-651
  StepPositions_ExplicitEuler(dt, CalculateVelocityRelaxation(step_scale));
}

void Ensemble::StepPostStabilization(double dt, double step_scale) {  // ensembles.cc:653-658
  const VectorXd relax = CalculateVelocityRelaxation(step_scale);
  StepPositions_ExplicitEuler(dt, relax);
  const VectorXd v = GetVelocities();
  for (int i = 0; i < n_; ++i) {
    components_[i]->SetV(Vector3d(v(6 * i) + relax(6 * i), v(6 * i + 1) + relax(6 * i + 1), v(6 * i + 2) + relax(6 * i + 2)));
    components_[i]->SetW_GlobalFrame(Vector3d(v(6 * i + 3) + relax(6 * i + 3), v(6 * i + 4) + relax(6 * i + 4), v(6 * i + 5) + relax(6 * i + 5)));
  }
}

static double SquaredNorm(const VectorXd &e) { double s = 0; for (int k = 0; k < e.size(); ++k) s += e(k) * e(k); return s; }

void Ensemble::InitStabilize() {  // ensembles.cc:602-622
  if (detect_contacts) UpdateContacts();
  double err_sq = SquaredNorm(ComputePositionConstraintError());
  const int max_steps = 100;
  int step_counter = 0;
  while (err_sq > kAllowNumericalError && step_counter < max_steps) {
    StepPositionRelaxation(kSimTimeStep * 500);
    if (detect_contacts) UpdateContacts();
    err_sq = SquaredNorm(ComputePositionConstraintError());
    ++step_counter;
  }
  last_stabilize_steps = step_counter;
}

void Ensemble::PostStabilize(int max_steps) {  // ensembles.cc:624-646
  double err_sq = SquaredNorm(ComputePositionConstraintError());
  int step_counter = 0;
  while (err_sq > kAllowNumericalError && step_counter < max_steps) {
    StepPostStabilization(kSimTimeStep * 100);
    err_sq = SquaredNorm(ComputePositionConstraintError());
    ++step_counter;
  }
  last_stabilize_steps = step_counter;
}

// ---- toolkit/lcp.h:172-174, toolkit/lcp.cc:627-785 ------------------------------
namespace { int g_last_lcp_pivots = 0; }
int lcp::LastSolvePivots() { return g_last_lcp_pivots; }

bool lcp::SolveLCP(const Settings &settings, MatrixXd &A, const VectorXd &b, const VectorXd &lo, const VectorXd &hi,
                   VectorXd *x, VectorXd *w) {
  const int N = b.size();
  // CHECKs of toolkit/lcp.cc:756-760 (the reference Panics)
  if (A.rows() != A.cols() || A.rows() <= 0 || A.rows() != N || lo.size() != N || hi.size() != N || !x || !w)
    throw egs::Error(EGS_ERR_INVALID, "SolveLCP: A must be square and non-empty, b / lo / hi of its size");
  // the dispatch of toolkit/lcp.cc:762-784 and its refusals
  if (settings.schur_complement && !settings.box_lcp)
    throw egs::Error(EGS_ERR_INVALID, "Schur complement solver only available for box LCP");
  if (!settings.schur_complement && settings.algorithm == COTTLE_DANTZIG && !settings.box_lcp)
    throw egs::Error(EGS_ERR_INVALID, "Cottle Dantzig solver only available for box LCP");
  if (settings.algorithm != MURTY && settings.algorithm != COTTLE_DANTZIG) throw egs::Error(EGS_ERR_INVALID, "Unknown LCP solver selection");
  const double inf = std::numeric_limits<double>::infinity();
  const int max_it = settings.max_iterations >= __INT_MAX__ ? 0 : std::max(settings.max_iterations, 1);
  const double max_sec = settings.max_time >= __DBL_MAX__ ? 0.0 : settings.max_time;
  const int algorithm = settings.algorithm == COTTLE_DANTZIG ? 1 : 0;
  x->resize(N); w->resize(N);
  int32_t ok = 0, pivots = 0;
  if (settings.schur_complement) {
    // SolveLCP_BoxSchur (toolkit/lcp.cc:627-747) on the device.  Like the reference it reads and writes ONLY the lower
    // triangle of A (toolkit/lcp.h:73; the reference's own test hands over a lower triangle, toolkit/lcp.cc:1109-1110) and
    // leaves A permuted: unbounded rows first, and the inner solver's pivoting order when nothing is unbounded.
    int32_t nub = 0;
    egs_status st = egs_box_lcp_schur(egs::DefaultContext(), N, A.data(), b.data(), lo.data(), hi.data(), algorithm, /*nub: scan*/ -1,
                                      settings.reference_quirks ? 1 : 0, max_it, max_sec, x->data(), w->data(), nullptr, &ok, &nub, &pivots);
    g_last_lcp_pivots = pivots;
    if (st != EGS_OK && st != EGS_ERR_LCP_FAILED) throw egs::Error(st, egs_last_error(egs::DefaultContext()));
    return ok != 0;
  }
  VectorXd l(N), h(N);
  for (int i = 0; i < N; ++i) {
    l(i) = settings.box_lcp ? lo(i) : 0.0;    // toolkit/lcp.h:152-154
    h(i) = settings.box_lcp ? hi(i) : inf;
    // "infinity" is DBL_MAX or the real infinity (toolkit/lcp.h:149-150)
    if (l(i) <= -__DBL_MAX__) l(i) = -inf;
    if (h(i) >= __DBL_MAX__) h(i) = inf;
  }
  if (N <= 1024) {
    // Without the Schur complement the reference runs SolveLCP_BoxDantzig (toolkit/lcp.cc:444-619) or
    // SolveLCP_BoxMurty / SolveLCP_Murty on a LinearReducer (:213-442): both on the device with their
    // incremental Cholesky factor; A's lower triangle carries the pivoting order afterwards, as in the
    // reference.  Their preconditions (lo <= 0 <= hi; lo < hi for Dantzig, :448-450) come back as
    // EGS_ERR_INVALID, the iteration / time limit as false.
    const int32_t n32 = N;
    egs_status st = egs_box_lcp_batch(egs::DefaultContext(), algorithm, 1, &n32, A.data(), b.data(), l.data(), h.data(), max_it, max_sec,
                                      x->data(), w->data(), nullptr, &ok, &pivots);
    g_last_lcp_pivots = pivots;
    if (st != EGS_OK) throw egs::Error(st, egs_last_error(egs::DefaultContext()));
    return ok != 0;
  }
  // beyond the incremental solvers: block principal pivoting on the symmetric matrix the lower triangle stands for
  // (same solution for SPD A; A keeps its order)
  MatrixXd Asym(N, N);
  ArrayXb C(N);
  for (int r = 0; r < N; ++r)
    for (int c = 0; c <= r; ++c) { Asym(r, c) = A(r, c); Asym(c, r) = A(r, c); }
  egs_status st = egs_mixed_constraints_solve_limits(egs::DefaultContext(), N, Asym.data(), b.data(), C.data(), l.data(), h.data(),
                                                     /*bounds + block pivoting*/ 3, max_it, max_sec, x->data(), w->data(), &ok, &pivots);
  g_last_lcp_pivots = pivots;
  if (st != EGS_OK && st != EGS_ERR_LCP_FAILED) throw egs::Error(st, egs_last_error(egs::DefaultContext()));
  return ok != 0;
}
