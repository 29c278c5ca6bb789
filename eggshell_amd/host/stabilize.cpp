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
  VectorXd l(N), h(N);
  ArrayXb C(N);
  std::vector<char> unbounded(N, 0);
  for (int i = 0; i < N; ++i) {
    l(i) = settings.box_lcp ? lo(i) : 0.0;    // toolkit/lcp.h:152-154
    h(i) = settings.box_lcp ? hi(i) : inf;
    // "infinity" is DBL_MAX or the real infinity (toolkit/lcp.h:149-150).  A row is unbounded when both
    // bounds are infinite; the reference tests `hi < -DBL_MAX` (quirk Q6, toolkit/lcp.cc:664, 669), i.e.
    // the lower bound alone decides.
    const bool lo_inf = l(i) <= -__DBL_MAX__, hi_inf = h(i) >= __DBL_MAX__;
    unbounded[i] = settings.schur_complement && (settings.reference_quirks ? lo_inf : (lo_inf && hi_inf));
    if (lo_inf) l(i) = -inf;
    if (hi_inf || (unbounded[i] && settings.reference_quirks)) h(i) = inf;   // Q6: the finite hi of such a row is never looked at
    C(i) = unbounded[i] ? 1 : 0;
  }
  if (settings.schur_complement) {
    // SolveLCP_BoxSchur permutes the problem so that the unbounded rows come first, and A stays permuted
    // (toolkit/lcp.h:170-171, toolkit/lcp.cc:656-683): the same two-pointer partition, applied to A's rows
    // and columns.  The solve below works on the unpermuted copy; x and w come back in the caller's order,
    // as BoxSchur's Unpermute leaves them.
    std::vector<char> ub = unbounded;
    const MatrixXd A0 = A;
    std::vector<int> perm(N);
    for (int i = 0; i < N; ++i) perm[i] = i;
    int nub = 0, nb = N - 1;
    while (true) {
      for (; nub <= nb; ++nub) if (!ub[nub]) break;
      for (; nb >= nub; --nb) if (ub[nb]) break;
      if (nub > nb) break;
      std::swap(ub[nub], ub[nb]);
      std::swap(perm[nub], perm[nb]);
    }
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < N; ++c) A(r, c) = A0(perm[r], perm[c]);
    x->resize(N); w->resize(N);
    int32_t ok = 0, pivots = 0;
    const int max_piv = settings.max_iterations >= __INT_MAX__ ? 0 : std::max(settings.max_iterations, 1);
    const double max_sec = settings.max_time >= __DBL_MAX__ ? 0.0 : settings.max_time;
    egs_status st = egs_mixed_constraints_solve_limits(egs::DefaultContext(), N, A0.data(), b.data(), C.data(), l.data(), h.data(),
                                                       /*bounds + block pivoting*/ 3, max_piv, max_sec, x->data(), w->data(), &ok, &pivots);
    g_last_lcp_pivots = pivots;
    if (st != EGS_OK && st != EGS_ERR_LCP_FAILED) throw egs::Error(st, egs_last_error(egs::DefaultContext()));
    return ok != 0;
  }
  x->resize(N); w->resize(N);
  int32_t ok = 0, pivots = 0;
  const int max_piv = settings.max_iterations >= __INT_MAX__ ? 0 : std::max(settings.max_iterations, 1);
  const double max_sec = settings.max_time >= __DBL_MAX__ ? 0.0 : settings.max_time;
  if (N <= 96) {
    // Without the Schur complement the reference runs SolveLCP_BoxDantzig (toolkit/lcp.cc:444-619) or
    // SolveLCP_BoxMurty / SolveLCP_Murty on a LinearReducer (:213-442): both on the device with their
    // incremental Cholesky factor; A's lower triangle carries the pivoting order afterwards, as in the
    // reference.  Their preconditions (lo <= 0 <= hi; lo < hi for Dantzig, :448-450) come back as
    // EGS_ERR_INVALID, the iteration limit as false.
    const int lim = settings.max_iterations >= __INT_MAX__ ? 0 : std::max(settings.max_iterations, 1);
    egs_status st = settings.algorithm == COTTLE_DANTZIG
        ? egs_box_lcp_dantzig(egs::DefaultContext(), N, A.data(), b.data(), l.data(), h.data(), 0, x->data(), w->data(), nullptr, &ok, &pivots)
        : egs_box_lcp_murty(egs::DefaultContext(), N, A.data(), b.data(), l.data(), h.data(), lim, x->data(), w->data(), nullptr, &ok, &pivots);
    g_last_lcp_pivots = pivots;
    if (st != EGS_OK && st != EGS_ERR_LCP_FAILED) throw egs::Error(st, egs_last_error(egs::DefaultContext()));
    return ok != 0;
  }
  egs_status st = egs_mixed_constraints_solve_limits(egs::DefaultContext(), N, A.data(), b.data(), C.data(), l.data(), h.data(),
                                                     /*bounds + block pivoting*/ 3, max_piv, max_sec, x->data(), w->data(), &ok, &pivots);
  g_last_lcp_pivots = pivots;
  if (st != EGS_OK && st != EGS_ERR_LCP_FAILED) throw egs::Error(st, egs_last_error(egs::DefaultContext()));
  return ok != 0;
}
