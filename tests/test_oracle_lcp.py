"""CPU: pin the oracle's dense LCP (oracle/lcp_dense.c) on the reference's own
literal vectors and property tests (eggshell/lcp.cc:346-528)."""
import numpy as np

from oracle import oracle as orc

# lcp.cc:350-357 / 369-376: the reference's literal 5x5 system
A1 = np.array([2.1104, 1.4090, 1.5055, 1.3060, 1.1413, 1.4090, 1.9846, 1.7126, 1.0858,
               1.9358, 1.5055, 1.7126, 2.1673, 1.3226, 1.5765, 1.3060, 1.0858, 1.3226,
               1.2704, 0.8927, 1.1413, 1.9358, 1.5765, 0.8927, 2.1211]).reshape(5, 5)
B1 = np.array([0.6691, 0.1904, 0.3689, 0.4607, 0.9816])
X1 = np.array([0.0942, 0, 0, 0, 0.4121])
W1 = np.array([0, 0.7401, 0.4226, 0.0302, 0])
# lcp.cc:391-397 (KAT 2; the reference's own assertions on it are commented out)
A2 = np.array([2.7345, 1.8859, 2.0785, 1.9442, 1.9567, 1.8859, 2.2340, 2.0461, 2.3164,
               2.0875, 2.0785, 2.0461, 2.7591, 2.4606, 1.9473, 1.9442, 2.3164, 2.4606,
               2.5848, 2.2768, 1.9567, 2.0875, 1.9473, 2.2768, 2.4853]).reshape(5, 5)
B2 = np.array([0.7577, 0.7431, 0.3922, 0.6555, 0.1712])


def test_check_murty_solution_kat():
    """lcp.cc:348-365: is-a-solution at 1e-4, and the counter example."""
    S = np.array([1, 0, 0, 0, 1], np.uint8)
    ok, _ = orc.check_murty(A1, B1, X1, W1, S, err=1e-4)
    assert ok
    x = np.array([0.0942, 0, 0.5678, 0, 0.4121])
    w = np.array([0, 0.7401, 0.4226, -0.0302, 0])
    ok, _ = orc.check_murty(A1, B1, x, w, S, err=1e-4)
    assert not ok


def test_murty_simple_kat1():
    """lcp.cc:367-389: x, w within 5e-4 of the 4-digit expected values."""
    ok, x, w, piv = orc.murty(A1, B1)
    assert ok
    assert np.linalg.norm(x - X1) <= 5e-4
    assert np.linalg.norm(w - W1) <= 5e-4
    assert np.linalg.norm(A1 @ x - B1 - w) < 1e-9


def test_murty_simple_kat2_returns():
    """lcp.cc:390-410: the reference only CHECKs the return value here."""
    ok, x, w, _ = orc.murty(A2, B2)
    assert ok
    assert np.linalg.norm(A2 @ x - B2 - w) < 1e-9
    assert (x >= 0).all() and (w >= -1e-9).all()


def _spd(rng, dim):
    """utils.cc:203-215 GenerateSPDMatrix: M^T M, cond < 1e7."""
    while True:
        M = rng.uniform(-1, 1, (dim, dim))
        A = M.T @ M
        if np.linalg.cond(A) < 1e7:
            return A


def test_murty_no_bounds_batch():
    """lcp.cc:412-438, fewer/smaller instances so the CPU suite stays fast."""
    rng = np.random.default_rng(1)
    for _ in range(20):
        dim = 30
        A = _spd(rng, dim)
        b = rng.uniform(-1, 1, dim)
        ok, x, w, _ = orc.murty(A, b)
        assert ok
        assert np.linalg.norm(A @ x - b - w) < 1e-9
        assert (x >= 0).all()
        assert abs(x @ w) < 1e-8


def test_mixed_constraints_no_bounds_batch():
    """lcp.cc:467-497: Ax = b + w, inequality rows within [0, inf)."""
    rng = np.random.default_rng(2)
    for _ in range(20):
        dim = 30
        A = _spd(rng, dim)
        b = rng.uniform(-1, 1, dim)
        Ceq = rng.integers(0, 2, dim).astype(np.uint8)
        ok, x, w, _ = orc.mixed_constraints(A, b, Ceq, np.zeros(dim), np.full(dim, np.inf))
        assert ok
        assert np.linalg.norm(A @ x - b - w) < 1e-9
        assert (Ceq.astype(bool) | (x >= 0)).all()
        assert (w[Ceq.astype(bool)] == 0).all()


def test_mixed_constraints_ignores_bounds_like_reference():
    """Quirk Q3 (lcp.cc:298): x_lo/x_hi are accepted and ignored."""
    rng = np.random.default_rng(3)
    dim = 20
    A = _spd(rng, dim)
    b = rng.uniform(-1, 1, dim)
    Ceq = rng.integers(0, 2, dim).astype(np.uint8)
    r1 = orc.mixed_constraints(A, b, Ceq, np.full(dim, -10.0), np.full(dim, 10.0))
    r2 = orc.mixed_constraints(A, b, Ceq, np.zeros(dim), np.full(dim, np.inf))
    assert r1[0] and r2[0]
    assert np.array_equal(r1[1], r2[1])


def test_mixed_constraints_true_box():
    """use_bounds=1 (not the reference): a genuine box LCP solution."""
    rng = np.random.default_rng(4)
    for _ in range(10):
        dim = 20
        A = _spd(rng, dim) + 0.5 * np.eye(dim)
        b = rng.uniform(-3, 3, dim)
        Ceq = rng.integers(0, 2, dim).astype(np.uint8)
        lo, hi = np.full(dim, -0.3), np.full(dim, 0.4)
        ok, x, w, _ = orc.mixed_constraints(A, b, Ceq, lo, hi, use_bounds=1)
        assert ok
        assert np.linalg.norm(A @ x - b - w) < 1e-9
        ineq = ~Ceq.astype(bool)
        assert (x[ineq] >= lo[ineq] - 1e-12).all() and (x[ineq] <= hi[ineq] + 1e-12).all()
        inside = ineq & (x > lo) & (x < hi)
        assert np.abs(w[inside]).max(initial=0) < 1e-9
        assert (w[ineq & (x == lo)] >= -1e-9).all()
        assert (w[ineq & (x == hi)] <= 1e-9).all()
