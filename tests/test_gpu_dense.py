"""GPU (-m gpu): entry 3, the dense direct LCP (Lcp::MixedConstraintsSolver +
MurtyPrincipalPivot, lcp.cc:141-336) against the reference's literal KAT, its
property tests (lcp.cc:412-528) and the oracle's restatement."""
import numpy as np
import pytest

from oracle import oracle as orc
from test_oracle_lcp import A1, A2, B1, B2, W1, X1, _spd

pytestmark = pytest.mark.gpu
INF = np.inf


def test_murty_kat1(ctx):
    """lcp.cc:367-389 through the mixed entry with no equality rows."""
    ok, x, w, piv = ctx.mixed_constraints_solve(A1, B1, np.zeros(5, np.uint8), np.zeros(5), np.full(5, INF))
    assert ok and np.linalg.norm(x - X1) <= 5e-4 and np.linalg.norm(w - W1) <= 5e-4
    assert np.linalg.norm(A1 @ x - B1 - w) < 1e-9
    ok, x, w, piv = ctx.mixed_constraints_solve(A2, B2, np.zeros(5, np.uint8), np.zeros(5), np.full(5, INF))
    assert ok and np.linalg.norm(A2 @ x - B2 - w) < 1e-9


@pytest.mark.parametrize("dim", [1, 7, 50, 64, 65, 130, 200])
def test_mixed_no_bounds_vs_oracle(ctx, dim):
    """MixedConstraintsSolver_NoBounds (lcp.cc:467-497): returns true,
    ||Ax - b - w|| < 1e-9, inequality rows >= 0; same pivot count and the same
    answer as the oracle (1e-8: Cholesky here vs pivoted LDLT there)."""
    rng = np.random.default_rng(dim)
    for _ in range(3):
        A = _spd(rng, dim)
        b = rng.uniform(-1, 1, dim)
        Ceq = rng.integers(0, 2, dim).astype(np.uint8)
        ok, x, w, piv = ctx.mixed_constraints_solve(A, b, Ceq, np.zeros(dim), np.full(dim, INF))
        oko, xo, wo, pivo = orc.mixed_constraints(A, b, Ceq, np.zeros(dim), np.full(dim, INF))
        assert ok and oko
        assert np.linalg.norm(A @ x - b - w) < 1e-9
        eq = Ceq.astype(bool)
        assert (eq | (x >= 0)).all() and (w[eq] == 0).all()
        assert piv == pivo
        scale = max(1.0, np.abs(xo).max())
        assert np.abs(x - xo).max() <= 1e-8 * scale and np.abs(w - wo).max() <= 1e-8 * scale


def test_all_equality_and_all_inequality(ctx):
    rng = np.random.default_rng(5)
    dim = 90
    A = _spd(rng, dim)
    b = rng.uniform(-1, 1, dim)
    ok, x, w, piv = ctx.mixed_constraints_solve(A, b, np.ones(dim, np.uint8), np.zeros(dim), np.full(dim, INF))
    assert ok and piv == 0 and not w.any()
    assert np.abs(x - np.linalg.solve(A, b)).max() <= 1e-8 * max(1.0, np.abs(x).max())
    ok, x, w, piv = ctx.mixed_constraints_solve(A, b, np.zeros(dim, np.uint8), np.zeros(dim), np.full(dim, INF))
    oko, xo, wo, pivo = orc.murty(A, b)
    assert ok and oko and piv == pivo and np.abs(x - xo).max() < 1e-8


def test_bounds_ignored_like_reference_and_true_box(ctx):
    """use_bounds=0: x_lo/x_hi ignored (quirk Q3, lcp.cc:298); use_bounds=1
    solves the genuine box problem (checked by its KKT conditions)."""
    rng = np.random.default_rng(6)
    dim = 80
    A = _spd(rng, dim) + 0.5 * np.eye(dim)
    b = rng.uniform(-3, 3, dim)
    Ceq = rng.integers(0, 2, dim).astype(np.uint8)
    lo, hi = np.full(dim, -0.3), np.full(dim, 0.4)
    r0 = ctx.mixed_constraints_solve(A, b, Ceq, lo, hi, use_bounds=0)
    r1 = ctx.mixed_constraints_solve(A, b, Ceq, np.zeros(dim), np.full(dim, INF), use_bounds=0)
    assert r0[0] and r1[0] and np.array_equal(r0[1], r1[1])
    ok, x, w, piv = ctx.mixed_constraints_solve(A, b, Ceq, lo, hi, use_bounds=1)
    assert ok and np.linalg.norm(A @ x - b - w) < 1e-9
    ineq = ~Ceq.astype(bool)
    assert (x[ineq] >= lo[ineq]).all() and (x[ineq] <= hi[ineq]).all()
    inside = ineq & (x > lo) & (x < hi)
    assert np.abs(w[inside]).max(initial=0) < 1e-9
    assert (w[ineq & (x == lo)] >= -1e-9).all() and (w[ineq & (x == hi)] <= 1e-9).all()
    oko, xo, wo, _ = orc.mixed_constraints(A, b, Ceq, lo, hi, use_bounds=1)
    assert oko and np.abs(x - xo).max() < 1e-8


def test_dense_on_ensemble_matrix(ctx):
    """The live path of Ensemble::ComputeVDot (ensembles.cc:498-538): dense
    J M^-1 J^T of Chain(8) (all equality rows) -> lambda = A^-1 rhs."""
    from eggshell_amd import scenes
    from helpers import ode_rhs_from_scene, system_from_scene
    sc = scenes.chain(8)
    s, err = system_from_scene(sc)
    rhs, _ = ode_rhs_from_scene(sc, s, err, 1e-3)
    A = orc.dense_JMJt(s, 0.0)
    ok, x, w, piv = ctx.mixed_constraints_solve(A, rhs, s.is_eq, s.lo, s.hi)
    oko, xo, wo, _ = orc.mixed_constraints(A, rhs, s.is_eq, s.lo, s.hi)
    assert ok and oko and np.abs(x - xo).max() <= 1e-9 * max(1.0, np.abs(xo).max())


def test_not_positive_definite_reports_failure(ctx):
    A = np.array([[1.0, 2.0], [2.0, 1.0]])     # indefinite
    ok, x, w, piv = ctx.mixed_constraints_solve(A, np.ones(2), np.ones(2, np.uint8), np.zeros(2), np.full(2, INF))
    assert not ok


def test_config5_size_properties(ctx):
    """N = 1024 mixed problem (C5 is N = 2048; the bench tool reports it):
    Ax = b + w and complementarity, no oracle at this size."""
    rng = np.random.default_rng(8)
    N = 1024
    M = rng.uniform(-1, 1, (N, N))
    A = M.T @ M + 1e-3 * np.eye(N)
    b = rng.uniform(-1, 1, N)
    Ceq = rng.integers(0, 2, N).astype(np.uint8)
    ok, x, w, piv = ctx.mixed_constraints_solve(A, b, Ceq, np.zeros(N), np.full(N, INF))
    if ok:   # the reference's own cap (1000 pivots, lcp.cc:168) may bite at this size
        assert np.linalg.norm(A @ x - b - w) < 1e-6 * np.linalg.norm(b) * 1e3
        eq = Ceq.astype(bool)
        assert (x[~eq] >= 0).all() and (w[~eq] >= -1e-8).all()
        assert abs(x[~eq] @ w[~eq]) < 1e-6
    else:
        assert piv >= 1000


@pytest.mark.parametrize("dim", [2, 63, 111, 112, 113])
def test_single_workgroup_loop_boundary(ctx, dim):
    """Up to 112 inequality rows the whole pivot loop runs in one workgroup (one launch, one
    read-back); 113 takes the launch-per-pivot path.  Same pivot count and answer as the
    oracle on both sides of the switch; box semantics too."""
    rng = np.random.default_rng(300 + dim)
    A = _spd(rng, dim) + 0.2 * np.eye(dim)
    b = rng.uniform(-1, 1, dim)
    none = np.zeros(dim, np.uint8)
    ok, x, w, piv = ctx.mixed_constraints_solve(A, b, none, np.zeros(dim), np.full(dim, INF))
    oko, xo, wo, pivo = orc.murty(A, b)
    assert ok and oko and piv == pivo
    assert np.abs(x - xo).max() <= 1e-8 * max(1.0, np.abs(xo).max()) and np.abs(w - wo).max() <= 1e-8 * max(1.0, np.abs(wo).max())
    lo, hi = np.full(dim, -0.3), np.full(dim, 0.4)
    ok, x, w, piv = ctx.mixed_constraints_solve(A, b, none, lo, hi, use_bounds=1)
    oko, xo, wo, pivo = orc.mixed_constraints(A, b, none, lo, hi, use_bounds=1)
    assert ok and oko and piv == pivo and np.abs(x - xo).max() <= 1e-8 and np.abs(w - wo).max() <= 1e-8


def test_single_workgroup_loop_reports_indefinite_matrix(ctx):
    """A(S,S) not positive definite: failure, not a wrong answer (as on the large path)."""
    A = np.array([[1.0, 2.0], [2.0, 1.0]])       # symmetric, eigenvalues 3 and -1
    ok, x, w, piv = ctx.mixed_constraints_solve(A, np.array([1.0, 1.0]), np.zeros(2, np.uint8), np.zeros(2), np.full(2, INF))
    assert not ok


@pytest.mark.parametrize("dim", [5, 70, 300])
def test_asymmetric_matrix_is_rejected(ctx, dim):
    """The factorisations read the lower triangle only, so A must be symmetric: a single
    perturbed entry anywhere (tile interior, tile edge, last row) is reported as
    EGS_ERR_INVALID, a symmetric matrix of the same size is accepted."""
    from eggshell_amd import capi
    rng = np.random.default_rng(dim)
    A = _spd(rng, dim) + 0.2 * np.eye(dim)
    b = rng.uniform(-1, 1, dim)
    args = (b, np.zeros(dim, np.uint8), np.zeros(dim), np.full(dim, INF))
    assert ctx.mixed_constraints_solve(A, *args)[0]
    for (i, j) in ((1, 0), (dim - 1, dim // 2), (dim // 2, dim // 3)):
        Abad = A.copy()
        Abad[i, j] += 1e-3
        with pytest.raises(capi.EgsError) as e:
            ctx.mixed_constraints_solve(Abad, *args)
        assert e.value.status == capi.ERR_INVALID
