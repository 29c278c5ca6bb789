// adapter_demo.cpp -- exercises the reference-shaped C++ API end to end on the
// GPU and prints the results for tests/test_gpu_adapter.py to compare with the
// oracle.  Reads like the reference's own ensemble tests
// (sparse_iterations.cc:515-748): build an ensemble, Init(), call
// sparse::*Iteration(en.constraints(), en.M_inverse(), rhs, cfm), Step().
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "eggshell_api.h"

static void print_vec(const char *name, const VectorXd &v) {
  std::printf("%s", name);
  for (int i = 0; i < v.size(); ++i) std::printf(" %.17g", v(i));
  std::printf("\n");
}

// An axis-aligned box pile; its contacts come from Ensemble::UpdateContacts
// (device collision), called by Step() as in the reference (ensembles.cc:393).
class BoxPile : public Ensemble {
 public:
  BoxPile(int nx, int ny, int nz, double sink = 1e-3, double gap = 1e-2) {
    const double side = 0.3, h = 0.15;
    const int ncol = nx * ny;
    n_ = ncol * nz;
    Matrix3d I = Matrix3d::Identity() * 0.1;  // ensembles.cc:719
    for (int k = 0; k < nz; ++k)
      for (int iy = 0; iy < ny; ++iy)
        for (int ix = 0; ix < nx; ++ix) {
          Vector3d p(ix * (side + gap), iy * (side + gap), (h - sink) + k * (side - sink));
          components_.push_back(std::make_shared<Body>(p, Vector3d::Zero(), 1.0, Matrix3d::Identity(), Vector3d::Zero(), I));
        }
  }
};


// Two boxes, one on the other, with an optional ball joint between them (the joint-vs-contact
// pruning of CheckAndCorrectEnsembleState, ensembles.cc:296-306) or two coincident joints
// (the joint-vs-joint check, ensembles.cc:280-289).
class JointedPair : public Ensemble {
 public:
  JointedPair() {
    const Matrix3d I = Matrix3d::Identity() * 0.1;
    n_ = 2;
    components_.push_back(std::make_shared<Body>(Vector3d(0, 0, 0.149), Vector3d::Zero(), 1.0, Matrix3d::Identity(), Vector3d::Zero(), I));
    components_.push_back(std::make_shared<Body>(Vector3d(0, 0, 0.448), Vector3d::Zero(), 1.0, Matrix3d::Identity(), Vector3d::Zero(), I));
  }
  void AddJointAt(const Vector3d &world) {   // both anchors on the world point: zero joint error
    joints_.push_back(std::make_shared<BallAndSocketJoint>(components_[0], 0, world - components_[0]->p(), components_[1], 1,
                                                          world - components_[1]->p()));
  }
  Vector3d ContactPosition(int k) const { return contacts_.at(k)->GetConstraintPosition(); }
  int NumContacts() const { return (int)contacts_.size(); }
  int PairContacts() const { int c = 0; for (const auto &k : contacts_) c += (k->i0_ >= 0 && k->i1_ >= 0); return c; }
};

static void print_bodies(const char *tag, const Ensemble &e) {
  const int n = (int)e.components().size();
  VectorXd p(3 * n), R(9 * n);
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 3; ++k) p(3 * i + k) = e.components()[i]->p()[k];
    for (int k = 0; k < 9; ++k) R(9 * i + k) = e.components()[i]->R().d[k];
  }
  char name[64];
  std::snprintf(name, sizeof name, "%s_p", tag); print_vec(name, p);
  std::snprintf(name, sizeof name, "%s_R", tag); print_vec(name, R);
  std::snprintf(name, sizeof name, "%s_v", tag); print_vec(name, e.GetVelocities());
}

// what round 2 added to the adapter: the matrix-free products, Cairn, the constraint-pair checks, the
// dense solver path, the lcp::SolveLCP contract
static void round2() {
  {  // sparse::CalculateSparse* (sparse_iterations_utils.cc:938-1052: Chain(4) at t = 0 and after steps)
    Chain chain(4, Vector3d(0, 0, 2));
    chain.Init();
    chain.solver_params.max_iters = 2000;
    chain.cfm_coeff = 0.0;
    for (int rep = 0; rep < 2; ++rep) {
      const int rows = 3 * (int)chain.constraints().size();
      VectorXd x(rows);
      for (int k = 0; k < rows; ++k) x(k) = ((k * 29 + 3 * rep) % 13 - 6) / 6.5;
      char tag[32];
      std::snprintf(tag, sizeof tag, "prod%d", rep);
      print_bodies(tag, chain);
      const ConstraintsList cs = chain.constraints();
      const MatrixXd &Mi = chain.M_inverse();
      auto out = [&](const char *what, const VectorXd &v) { char nm[64]; std::snprintf(nm, sizeof nm, "%s_%s", tag, what); print_vec(nm, v); };
      out("x", x);
      out("JMJtX", sparse::CalculateSparseJMJtX(cs, Mi, x, 0.01));
      out("Lx", sparse::CalculateSparseLx(cs, Mi, x));
      out("Ux", sparse::CalculateSparseUx(cs, Mi, x));
      out("LxUx", sparse::CalculateSparseLxUx(cs, Mi, x));
      out("Dx", sparse::CalculateSparseDx(cs, Mi, x, 0.01, 1.0 / 1.5));
      out("UxDx", sparse::CalculateSparseUxDx(cs, Mi, x, 0.01, 1.0 / 1.5));
      out("LxDx", sparse::CalculateSparseLxDx(cs, Mi, x, 0.01, 1.0 / 1.5));
      for (int s = 0; s < 5; ++s) chain.Step(0.001);
    }
    BoxPile pile(2, 2, 2);
    pile.Init();
    pile.UpdateContacts();
    const int rows = 3 * (int)pile.constraints().size();
    VectorXd x(rows);
    for (int k = 0; k < rows; ++k) x(k) = ((k * 31) % 17 - 8) / 8.5;
    print_vec("prodpile_x", x);
    print_vec("prodpile_JMJtX", sparse::CalculateSparseJMJtX(pile.constraints(), pile.M_inverse(), x, 0.01));
    print_vec("prodpile_LxDx", sparse::CalculateSparseLxDx(pile.constraints(), pile.M_inverse(), x, 0.01, 1.0 / 1.5));
    print_vec("prodpile_Ux", sparse::CalculateSparseUx(pile.constraints(), pile.M_inverse(), x));
  }
  {  // Chain(8) through the reference's LIVE dense path (ComputeVDot, ensembles.cc:498-538) on the device
    Chain chain(8, Vector3d(0, 0, 2));
    chain.Init();
    chain.use_dense_solver = true;
    for (int s = 0; s < 3; ++s) chain.Step(0.001);
    print_bodies("dense3", chain);
    print_vec("dense3_lambda", chain.last_lambda);
    std::printf("dense3_cond %.6g\n", chain.last_condition_estimate);
  }
  {  // Cairn (ensembles.cc:708-728) as the reference drives it: InitStabilize, then Step(5e-3) (model.cc:28-31, 80)
    std::srand(7);
    Cairn cairn(4, {-0.2, 0.2}, {-0.2, 0.2}, {1.0, 1.6});   // a short column: the rocks start interpenetrating
    print_bodies("cairn0", cairn);
    cairn.Init();
    cairn.InitStabilize();
    std::printf("cairn_stab_steps %d\n", cairn.last_stabilize_steps);
    print_bodies("cairn1", cairn);
    cairn.solver_params.method = EGS_SOR;
    cairn.solver_params.max_iters = 500;
    cairn.solver_params.tol = 1e-9;
    int seen = 0;
    for (int s = 0; s < 20; ++s) { cairn.Step(0.005); seen += (int)cairn.constraints().size(); }
    print_bodies("cairn2", cairn);
    std::printf("cairn_contacts %d\n", seen);
    std::srand(11);
    Cairn tall(4, {-0.2, 0.2}, {-0.2, 0.2}, {1.0, 8.0});     // the reference's own bounds (sparse_iterations.cc:621)
    print_bodies("tall0", tall);
    tall.Init();
    tall.InitStabilize();
    std::printf("tall_stab_steps %d\n", tall.last_stabilize_steps);
    print_bodies("tall1", tall);
    tall.solver_params = cairn.solver_params;
    int seen_tall = 0;
    for (int s = 0; s < 20; ++s) { tall.Step(0.005); seen_tall += (int)tall.constraints().size(); }
    print_bodies("tall2", tall);
    std::printf("tall_contacts %d\n", seen_tall);
  }
  {  // constraint-pair checks (ensembles.cc:241-329)
    JointedPair free_pair;
    free_pair.Init();
    free_pair.UpdateContacts();
    std::printf("jc_contacts_free %d %d\n", free_pair.NumContacts(), free_pair.PairContacts());
    const Vector3d at = free_pair.ContactPosition(free_pair.NumContacts() - 1);   // a box-box contact
    for (int device = 0; device < 2; ++device) {
      JointedPair jp;
      jp.AddJointAt(at);
      jp.Init();
      jp.use_device_step = device != 0;
      jp.solver_params.max_iters = 200;
      jp.Step(0.005);
      std::printf(device ? "jc_contacts_device %d %d\n" : "jc_contacts_explicit %d %d\n", jp.NumContacts(), jp.PairContacts());
      print_vec(device ? "jc_lambda_device" : "jc_lambda_explicit", jp.last_lambda);
    }
    int status = 0;
    try {
      JointedPair jj;
      jj.AddJointAt(at);
      jj.AddJointAt(at + Vector3d(0, 0, 5e-7));   // closer than kMinConstraintDistance = 1e-6 (ensembles.cc:16)
      jj.Init();
    } catch (const egs::Error &e) { status = e.status; }
    std::printf("jj_conflict_status %d\n", status);
    JointedPair apart;
    apart.AddJointAt(at);
    apart.AddJointAt(at + Vector3d(0.01, 0, 0));
    apart.Init();
    std::printf("jj_apart_ok 1\n");
  }
  {  // lcp::SolveLCP contract (toolkit/lcp.h:104-174, toolkit/lcp.cc:627-785)
    const int n = 12;
    MatrixXd M(n, n), A(n, n);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) M(i, j) = ((i * 7 + j * 13) % 17 - 8) / 9.0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        double sacc = (i == j) ? 1.0 : 0.0;
        for (int k = 0; k < n; ++k) sacc += M(k, i) * M(k, j);
        A(i, j) = sacc;
      }
    VectorXd b(n), lo(n), hi(n), x, w;
    for (int i = 0; i < n; ++i) { b(i) = ((i * 5) % 7 - 3) * 1.5; lo(i) = -0.25; hi(i) = 0.5; }
    {  // the iterations on an EXPLICIT matrix (sparse_iterations.h:13-24): 2-argument and 5-argument forms
      MatrixXd Ad = A;
      for (int i = 0; i < n; ++i) Ad(i, i) += 6.0;        // a little diagonal dominance, as the reference's tests add (:392)
      print_vec("dense_gs_x", sparse::GaussSeidelIteration(Ad, b));
      std::printf("dense_gs_iters %d\n", sparse::GetLastSolve().iterations);
      print_vec("dense_sor_x", sparse::SORIteration(Ad, b));
      ArrayXb Cm(n);
      for (int i = 0; i < n; ++i) Cm(i) = (i % 3 == 0) ? 1 : 0;
      print_vec("dense_gs_mixed_x", sparse::GaussSeidelIteration(Ad, b, Cm, lo, hi));
    }
    const double inf = std::numeric_limits<double>::infinity();
    lo(3) = -inf; hi(3) = inf;
    lo(8) = -__DBL_MAX__; hi(8) = __DBL_MAX__;
    lo(5) = -inf;                                  // lo = -inf with a FINITE hi: quirk Q6
    lcp::Settings st;
    MatrixXd A1 = A;
    bool ok = lcp::SolveLCP(st, A1, b, lo, hi, &x, &w);
    std::printf("lcp_default_ok %d\n", ok ? 1 : 0);
    print_vec("lcp_default_x", x); print_vec("lcp_default_w", w);
    VectorXd Aperm(n * n);
    for (int i = 0; i < n * n; ++i) Aperm(i) = A1.data()[i];
    print_vec("lcp_default_A", Aperm);             // permuted in place: unbounded rows first
    st.reference_quirks = true;
    MatrixXd A2 = A;
    ok = lcp::SolveLCP(st, A2, b, lo, hi, &x, &w);
    std::printf("lcp_q6_ok %d\n", ok ? 1 : 0);
    print_vec("lcp_q6_x", x); print_vec("lcp_q6_w", w);
    st.reference_quirks = false;
    st.schur_complement = false;
    MatrixXd A3 = A;
    ok = lcp::SolveLCP(st, A3, b, lo, hi, &x, &w);
    std::printf("lcp_noschur_ok %d\n", ok ? 1 : 0);
    print_vec("lcp_noschur_x", x);
    st.box_lcp = false;                            // lo = 0, hi = inf whatever the vectors hold
    MatrixXd A4 = A;
    ok = lcp::SolveLCP(st, A4, b, lo, hi, &x, &w);
    std::printf("lcp_nobox_ok %d\n", ok ? 1 : 0);
    print_vec("lcp_nobox_x", x); print_vec("lcp_nobox_w", w);
    {  // algorithm = COTTLE_DANTZIG without the Schur complement: SolveLCP_BoxDantzig itself (toolkit/lcp.cc:776-779)
      lcp::Settings dz;
      dz.algorithm = lcp::COTTLE_DANTZIG;
      dz.schur_complement = false;
      VectorXd lod(n), hid(n);
      for (int i = 0; i < n; ++i) { lod(i) = (i % 4 == 1) ? 0.0 : -0.25; hid(i) = (i % 5 == 2) ? inf : 0.5; }
      MatrixXd A6 = A;
      ok = lcp::SolveLCP(dz, A6, b, lod, hid, &x, &w);
      std::printf("lcp_dantzig_ok %d %d\n", ok ? 1 : 0, lcp::LastSolvePivots());
      print_vec("lcp_dantzig_x", x); print_vec("lcp_dantzig_w", w);
      VectorXd A6v(n * n);
      for (int i = 0; i < n * n; ++i) A6v(i) = A6.data()[i];
      print_vec("lcp_dantzig_A", A6v);             // lower triangle permuted in place by the pivoting order
    }
    int refused = 0;
    st.schur_complement = true;                    // Schur complement without box_lcp: the reference Panics
    try { MatrixXd A5 = A; (void)lcp::SolveLCP(st, A5, b, lo, hi, &x, &w); } catch (const egs::Error &e) { refused = e.status; }
    std::printf("lcp_refused %d\n", refused);
    // max_iterations: a 150-row problem needs more than one pivot
    const int N = 150;
    MatrixXd Mb(N, N), Ab(N, N);
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) Mb(i, j) = ((i * 37 + j * 101 + (i * j) % 7) % 29 - 14) / 14.0;
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        double sacc = (i == j) ? 0.5 : 0.0;
        for (int k = 0; k < N; ++k) sacc += Mb(k, i) * Mb(k, j);
        Ab(i, j) = sacc;
      }
    VectorXd bb(N), lb(N), hb(N);
    for (int i = 0; i < N; ++i) { bb(i) = ((i * 11) % 9 - 4) * 0.7; lb(i) = -0.1; hb(i) = 0.2; }
    lcp::Settings big;
    MatrixXd Ab1 = Ab;
    ok = lcp::SolveLCP(big, Ab1, bb, lb, hb, &x, &w);
    std::printf("lcp_big_ok %d %d\n", ok ? 1 : 0, lcp::LastSolvePivots());
    big.max_iterations = 1;
    MatrixXd Ab2 = Ab;
    ok = lcp::SolveLCP(big, Ab2, bb, lb, hb, &x, &w);
    std::printf("lcp_capped_ok %d %d\n", ok ? 1 : 0, lcp::LastSolvePivots());
    // The reference's own test shape (toolkit/lcp.cc:1105-1123, 1146-1199): ONLY THE LOWER TRIANGLE is handed over
    // (here the upper one holds a sentinel that must neither be read nor written), default Settings, n = 20 as the
    // reference's test and n = 300; every other row bounded, and Cottle-Dantzig as the inner algorithm on a third run.
    for (int pass = 0; pass < 3; ++pass) {
      const int m = pass == 1 ? 300 : 20;
      MatrixXd Mm(m, m), Am(m, m);
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) Mm(i, j) = ((i * 31 + j * 17 + (i * j) % 11) % 23 - 11) / 11.0;
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) {
          if (j > i) { Am(i, j) = 555.0; continue; }
          double sacc = (i == j) ? 0.25 : 0.0;
          for (int k = 0; k < m; ++k) sacc += Mm(k, i) * Mm(k, j);
          Am(i, j) = sacc;
        }
      VectorXd bm(m), lm(m), hm(m);
      for (int i = 0; i < m; ++i) {
        bm(i) = ((i * 13) % 11 - 5) * 0.3;
        const bool bounded = (i % 2 == 1) || (i % 7 == 0);
        lm(i) = bounded ? -0.05 * (1 + i % 3) : -__DBL_MAX__;
        hm(i) = bounded ? 0.04 * (1 + i % 4) : __DBL_MAX__;
      }
      lcp::Settings sl;
      if (pass == 2) sl.algorithm = lcp::COTTLE_DANTZIG;
      ok = lcp::SolveLCP(sl, Am, bm, lm, hm, &x, &w);
      char name[64];
      std::snprintf(name, sizeof name, "lcp_lower%d_ok", pass);
      std::printf("%s %d %d\n", name, ok ? 1 : 0, lcp::LastSolvePivots());
      std::snprintf(name, sizeof name, "lcp_lower%d_x", pass); print_vec(name, x);
      std::snprintf(name, sizeof name, "lcp_lower%d_w", pass); print_vec(name, w);
      VectorXd Av(m * m);
      for (int i = 0; i < m * m; ++i) Av(i) = Am.data()[i];
      std::snprintf(name, sizeof name, "lcp_lower%d_A", pass); print_vec(name, Av);
    }
  }
}

int main(int argc, char **argv) {
  try {
    if (argc > 1 && !std::strcmp(argv[1], "--round2")) { round2(); return 0; }
    Chain chain(8, Vector3d(0, 0, 2));
    chain.Init();
    const int rows = 3 * (int)chain.constraints().size();
    VectorXd rhs(rows);
    for (int k = 0; k < rows; ++k) rhs(k) = ((k * 37) % 11 - 5) / 7.0;
    print_vec("chain_sor", sparse::SORIteration(chain.constraints(), chain.M_inverse(), rhs, 0.1));
    std::printf("chain_sor_iters %d\n", sparse::GetLastSolve().iterations);
    print_vec("chain_gs", sparse::GaussSeidelIteration(chain.constraints(), chain.M_inverse(), rhs, 0.1));
    std::printf("chain_gs_iters %d\n", sparse::GetLastSolve().iterations);
    print_vec("chain_jacobi", sparse::JacobiIteration(chain.constraints(), chain.M_inverse(), rhs, 0.1));
    std::printf("chain_jacobi_iters %d\n", sparse::GetLastSolve().iterations);
    chain.solver_params.max_iters = 2000;
    chain.cfm_coeff = 0.0;
    for (int s = 0; s < 3; ++s) chain.Step(0.001);
    VectorXd p(24), v = chain.GetVelocities();
    for (int i = 0; i < 8; ++i)
      for (int k = 0; k < 3; ++k) p(3 * i + k) = chain.components()[i]->p()[k];
    print_vec("chain_step3_p", p);
    print_vec("chain_step3_v", v);
    print_vec("chain_step3_lambda", chain.last_lambda);

    BoxPile pile(2, 2, 3);
    pile.Init();
    pile.solver_params.method = EGS_GAUSS_SEIDEL;
    pile.solver_params.max_iters = 50;
    pile.solver_params.tol = 0.0;
    pile.Step(0.005);
    print_vec("pile_lambda", pile.last_lambda);
    print_vec("pile_v", pile.GetVelocities());
    {
      // A small drop test: the full Ensemble::Step loop (UpdateContacts on the
      // GPU, projected SOR on the GPU, midpoint positions) as model.cc:78-95
      // drives the reference's cairn, dt = 5e-3 (model.cc:80).
      BoxPile drop(1, 1, 3, /*sink=*/-0.05, /*gap=*/0.0);   // three boxes, 5 cm apart, 5 cm above ground
      drop.Init();
      drop.solver_params.method = EGS_SOR;
      drop.solver_params.max_iters = 500;
      drop.solver_params.tol = 1e-9;
      int contacts_seen = 0;
      for (int s = 0; s < 120; ++s) {
        drop.Step(0.005);
        contacts_seen += (int)drop.constraints().size();
      }
      VectorXd dp(9);
      for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) dp(3 * i + k) = drop.components()[i]->p()[k];
      print_vec("drop_p", dp);
      print_vec("drop_v", drop.GetVelocities());
      std::printf("drop_contacts %d\n", contacts_seen);
    }
    {
      // InitStabilize / PostStabilize (ensembles.cc:602-646): a Chain(4) whose
      // links were pulled off their joints relaxes back onto the constraint manifold.
      Chain bent(4, Vector3d(0, 0, 2));
      bent.Init();
      for (int i = 1; i < 4; ++i) {
        const Vector3d p = bent.components()[i]->p();
        bent.components()[i]->SetP(p + Vector3d(0.01 * i, -0.02 * i, 0.015 * i));
      }
      bent.InitStabilize();
      VectorXd sp(12);
      for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 3; ++k) sp(3 * i + k) = bent.components()[i]->p()[k];
      print_vec("stab_p", sp);
      print_vec("stab_err", bent.ComputePositionConstraintError());
      std::printf("stab_steps %d\n", bent.last_stabilize_steps);
      for (int i = 1; i < 4; ++i) {
        const Vector3d p = bent.components()[i]->p();
        bent.components()[i]->SetP(p + Vector3d(-0.02, 0.01 * i, 0.0));
        bent.components()[i]->SetV(Vector3d(0.1, 0.0, -0.2));
      }
      bent.PostStabilize();
      for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 3; ++k) sp(3 * i + k) = bent.components()[i]->p()[k];
      print_vec("post_p", sp);
      print_vec("post_v", bent.GetVelocities());
      std::printf("post_steps %d\n", bent.last_stabilize_steps);
    }
    if (argc > 1 && !std::strcmp(argv[1], "--dense")) {
      // Lcp::MixedConstraintsSolver on the reference's literal 5x5 (lcp.cc:369-376)
      const double a[25] = {2.1104, 1.4090, 1.5055, 1.3060, 1.1413, 1.4090, 1.9846, 1.7126, 1.0858, 1.9358, 1.5055, 1.7126, 2.1673,
                            1.3226, 1.5765, 1.3060, 1.0858, 1.3226, 1.2704, 0.8927, 1.1413, 1.9358, 1.5765, 0.8927, 2.1211};
      const double bb[5] = {0.6691, 0.1904, 0.3689, 0.4607, 0.9816};
      MatrixXd A(5, 5); VectorXd b(5), lo(5), hi(5), x, w; ArrayXb C(5);
      for (int i = 0; i < 25; ++i) A.data()[i] = a[i];
      for (int i = 0; i < 5; ++i) { b(i) = bb[i]; hi(i) = std::numeric_limits<double>::infinity(); }
      bool ok = Lcp::MixedConstraintsSolver(A, b, C, lo, hi, x, w);
      std::printf("dense_ok %d\n", ok ? 1 : 0);
      print_vec("dense_x", x);
      print_vec("dense_w", w);
      // lcp::SolveLCP (toolkit/lcp.h:172-174): a 12x12 box problem with two
      // unbounded rows, A = M^T M + I from a fixed rational pattern.
      const int n = 12;
      MatrixXd M(n, n), A2(n, n);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) M(i, j) = ((i * 7 + j * 13) % 17 - 8) / 9.0;
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
          double s = (i == j) ? 1.0 : 0.0;
          for (int k = 0; k < n; ++k) s += M(k, i) * M(k, j);
          A2(i, j) = s;
        }
      VectorXd b2(n), lo2(n), hi2(n), x2, w2;
      for (int i = 0; i < n; ++i) {
        b2(i) = ((i * 5) % 7 - 3) * 1.5;
        lo2(i) = -0.25; hi2(i) = 0.5;
      }
      lo2(3) = -std::numeric_limits<double>::infinity(); hi2(3) = std::numeric_limits<double>::infinity();
      lo2(8) = -__DBL_MAX__; hi2(8) = __DBL_MAX__;
      lcp::Settings settings;
      const bool ok2 = lcp::SolveLCP(settings, A2, b2, lo2, hi2, &x2, &w2);
      std::printf("solvelcp_ok %d\n", ok2 ? 1 : 0);
      print_vec("solvelcp_x", x2);
      print_vec("solvelcp_w", w2);
    }
  } catch (const egs::Error &e) {
    std::fprintf(stderr, "egs::Error %d: %s\n", e.status, e.what());
    return 2;
  }
  return 0;
}
