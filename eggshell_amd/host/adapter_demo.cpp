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

// An axis-aligned box pile with the contact list UpdateContacts would produce
// (ensembles.cc:445-480; analytic here, collision is a "next" row).
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
    ncol_ = ncol; nz_ = nz;
  }
  // What UpdateContacts would find (the reference calls it from Step, after
  // Init: contacts are not part of the initial-condition check).
  void MakeContacts() {
    const double side = 0.3, h = 0.15;
    const int ncol = ncol_, nz = nz_;
    ContactsList cs;
    for (int b = 0; b < ncol; ++b)
      for (int sx = -1; sx <= 1; sx += 2)
        for (int sy = -1; sy <= 1; sy += 2) {
          const Vector3d &p = components_[b]->p();
          Vector3d v(p[0] + side * 0.5 * sx, p[1] + side * 0.5 * sy, p[2] + side * 0.5 * -1);
          cs.push_back(std::make_shared<Contact>(components_[b], b, ContactGeometry(v, Vector3d(0, 0, 1), -v[2])));
        }
    for (int k = 0; k + 1 < nz; ++k)
      for (int c = 0; c < ncol; ++c) {
        const int i = k * ncol + c, j = (k + 1) * ncol + c;
        const Vector3d &pi = components_[i]->p(), &pj = components_[j]->p();
        const double zb = pj[2] + (-1.0) * h, ztop = pi[2] + h;
        const double px[4] = {-h, -h, h, h}, py[4] = {-h, h, h, -h};
        for (int q = 0; q < 4; ++q) {
          Vector3d pos(pj[0] + px[q], pj[1] + py[q], zb);
          cs.push_back(std::make_shared<Contact>(components_[i], i, components_[j], j,
                                                 ContactGeometry(pos, Vector3d(0, 0, 1), -(pos[2] + -ztop))));
        }
      }
    SetContacts(cs);
  }

 private:
  int ncol_ = 0, nz_ = 0;
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
    pile.MakeContacts();
    pile.solver_params.method = EGS_GAUSS_SEIDEL;
    pile.solver_params.max_iters = 50;
    pile.solver_params.tol = 0.0;
    pile.Step(0.005);
    print_vec("pile_lambda", pile.last_lambda);
    print_vec("pile_v", pile.GetVelocities());
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
    }
  } catch (const egs::Error &e) {
    std::fprintf(stderr, "egs::Error %d: %s\n", e.status, e.what());
    return 2;
  }
  return 0;
}
