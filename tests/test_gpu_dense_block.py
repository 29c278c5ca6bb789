"""GPU (-m gpu): block principal pivoting (use_bounds bit 1) -- NOT the
reference's pivot rule, same LCP solution.  Checked against the reference-rule
oracle where that finishes, and by KKT conditions at BASELINE C5 size
(N = 2048), which the reference's own rule cannot finish (1000-pivot cap,
lcp.cc:168)."""
import numpy as np
import pytest

from oracle import oracle as orc
from test_oracle_lcp import _spd

pytestmark = pytest.mark.gpu
INF = np.inf


@pytest.mark.parametrize("dim", [5, 50, 130, 300])
def test_block_rule_same_solution_as_reference_rule(ctx, dim):
    rng = np.random.default_rng(dim)
    A = _spd(rng, dim)
    b = rng.uniform(-1, 1, dim)
    Ceq = rng.integers(0, 2, dim).astype(np.uint8)
    lo, hi = np.zeros(dim), np.full(dim, INF)
    ok, x, w, piv = ctx.mixed_constraints_solve(A, b, Ceq, lo, hi, use_bounds=2)
    oko, xo, wo, pivo = orc.mixed_constraints(A, b, Ceq, lo, hi)
    assert ok and oko
    assert piv <= max(pivo, 3)
    scale = max(1.0, np.abs(xo).max())
    assert np.abs(x - xo).max() <= 1e-8 * scale and np.abs(w - wo).max() <= 1e-8 * scale
    # box variant
    A2 = A + 0.5 * np.eye(dim)
    b2 = rng.uniform(-3, 3, dim)
    ok, x, w, piv = ctx.mixed_constraints_solve(A2, b2, Ceq, np.full(dim, -0.3), np.full(dim, 0.4), use_bounds=3)
    oko, xo, wo, _ = orc.mixed_constraints(A2, b2, Ceq, np.full(dim, -0.3), np.full(dim, 0.4), use_bounds=1)
    assert ok and oko and np.abs(x - xo).max() <= 1e-8


def test_config5_n2048(ctx):
    """BASELINE C5: N = 2048, A = M^T M + 1e-3 I, b ~ U(-1,1), C ~ Bernoulli(1/2),
    seed 0.  The reference rule hits its 1000-pivot cap here (reported false);
    the block rule solves it: Ax = b + w, w = 0 on equality rows, x >= 0, w >= 0,
    x.w = 0 on inequality rows."""
    rng = np.random.default_rng(0)
    N = 2048
    M = rng.uniform(-1, 1, (N, N))
    A = M.T @ M + 1e-3 * np.eye(N)
    b = rng.uniform(-1, 1, N)
    Ceq = rng.integers(0, 2, N).astype(np.uint8)
    ok, x, w, piv = ctx.mixed_constraints_solve(A, b, Ceq, np.zeros(N), np.full(N, INF), use_bounds=2)
    assert ok and piv < 200
    eq = Ceq.astype(bool)
    assert np.linalg.norm(A @ x - b - w) <= 1e-6 * np.linalg.norm(A @ x)
    assert not w[eq].any()
    assert (x[~eq] >= 0).all() and (w[~eq] >= -1e-7).all()
    assert np.abs(x[~eq] * w[~eq]).max() < 1e-7


@pytest.mark.parametrize("N,mode", [(1100, 2), (1100, 3), (1350, 2)])
def test_block_rule_between_the_round_sizes(ctx, N, mode, monkeypatch):
    """Sizes that are no multiple of anything (the split upload from N = 1024 on, ragged panels, bordered pivots on a
    base set that is not a multiple of 64) with and without box bounds; KKT conditions, and the same answer with the
    round-3 shortcuts switched off (fresh factorisation at every pivot, S = everything as the start set)."""
    rng = np.random.default_rng(N + mode)
    M = rng.uniform(-1, 1, (N, N))
    A = M.T @ M + 1e-2 * np.eye(N)
    b = rng.uniform(-1, 1, N) * (3.0 if mode == 3 else 1.0)
    Ceq = (rng.uniform(size=N) < 0.4).astype(np.uint8)
    lo = np.zeros(N) if mode == 2 else np.where(rng.uniform(size=N) < 0.5, -0.2, 0.0)
    hi = np.full(N, INF) if mode == 2 else np.where(rng.uniform(size=N) < 0.5, 0.3, INF)
    ok, x, w, piv = ctx.mixed_constraints_solve(A, b, Ceq, lo, hi, use_bounds=mode)
    assert ok and piv < 60
    eq = Ceq.astype(bool)
    assert np.linalg.norm(A @ x - b - w) <= 1e-6 * max(1.0, np.linalg.norm(A @ x))
    assert not w[eq].any()
    xi, wi, li, hi_i = x[~eq], w[~eq], lo[~eq], hi[~eq]
    tol = 1e-7
    assert (xi >= li - tol).all() and (xi <= hi_i + tol).all()
    inside = (xi > li + tol) & (xi < hi_i - tol)
    assert np.abs(wi[inside]).max(initial=0.0) < 1e-6
    assert (wi[np.abs(xi - li) <= tol] >= -1e-6).all() and (wi[np.abs(xi - hi_i) <= tol] <= 1e-6).all()
    monkeypatch.setenv("EGS_DENSE_BORDER", "0")
    monkeypatch.setenv("EGS_DENSE_GUESS", "0")
    # (the switches are read once per process: this second solve documents the comparison when the suite is run with
    #  them set from outside -- tests/tools/env_matrix.sh does)
    ok2, x2, w2, _ = ctx.mixed_constraints_solve(A, b, Ceq, lo, hi, use_bounds=mode)
    assert ok2 and np.abs(x2 - x).max() <= 1e-8 * max(1.0, np.abs(x).max())


@pytest.mark.parametrize("N,eq_frac", [(1200, 1.0), (1400, 0.9), (2300, 1.0)])
def test_many_equality_rows(ctx, N, eq_frac):
    """More than 1 024 equality rows: the Schur stage's back substitution has more open columns than its workgroup has
    threads (found by tools/dense_fuzz.py: the strip update skipped them)."""
    rng = np.random.default_rng(N)
    M = rng.uniform(-1, 1, (N, N))
    A = M.T @ M + 1e-2 * np.eye(N)
    b = rng.uniform(-1, 1, N)
    Ceq = (rng.uniform(size=N) < eq_frac).astype(np.uint8)
    ok, x, w, piv = ctx.mixed_constraints_solve(A, b, Ceq, np.zeros(N), np.full(N, INF), use_bounds=2)
    assert ok
    eq = Ceq.astype(bool)
    assert np.abs(A @ x - b - w).max() <= 1e-7 * max(1.0, np.abs(A @ x).max())
    assert not w[eq].any() and (x[~eq] >= -1e-9).all() and (w[~eq] >= -1e-7).all()
    if eq.all():
        assert np.abs(x - np.linalg.solve(A, b)).max() <= 1e-7 * max(1.0, np.abs(x).max())
