// adapter_demo.cpp -- exercises the reference-shaped C++ API end to end on the
// GPU and prints the results for tests/test_gpu_adapter.py to compare with the
// oracle.  Reads like the reference's own ensemble tests
// (sparse_iterations.cc:515-748): build an ensemble, Init(), call
// sparse::*Iteration(en.constraints(), en.M_inverse(), rhs, cfm), Step().
#include <cstdio>
#include <cstring>

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


int main(int argc, char **argv) {
  try {
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
