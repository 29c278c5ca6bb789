"""CPU: the oracle's restatement of the reference's iterations on an EXPLICIT matrix (oracle/dense_iter.c;
sparse_iterations.cc:72-144 with the dense solves of sparse_iterations_utils.cc:25-40, 110-128, 245-262) held against
the reference's own tests of them, restated with their recipes and tolerance (sparse_iterations.cc:355-513: ten
instances each, dimension 3..50, diagonally dominant or SPD + k I, |Ax - b| < 1e-9 or CheckMixedConstraintSolutions,
:308-353) and against an independent numpy loop."""
import numpy as np

from oracle import oracle as orc

TOL = 1e-9          # kAllowNumericalError, constants.h:5


def diag_dominant(rng, n):         # GenerateDiagonalDominantMatrix, utils.cc:217-231
    while True:
        A = rng.uniform(-1, 1, (n, n))
        A[np.diag_indices(n)] *= np.abs(A).max() / np.abs(A).min()
        if np.linalg.cond(A) <= 1e7:
            return A


def spd(rng, n, ridge):            # GenerateSPDMatrix, utils.cc:203-215
    while True:
        m = rng.uniform(-1, 1, (n, n))
        A = m.T @ m
        if np.linalg.cond(A) <= 1e7:
            return A + ridge * np.eye(n)


def check_mixed(A, b, x, C, lo, hi):        # CheckMixedConstraintSolutions, sparse_iterations.cc:308-353
    w = A @ x - b
    assert np.linalg.norm(w[C]) < TOL
    for i in np.where(~C)[0]:
        if lo[i] < x[i] < hi[i]:
            assert abs(w[i]) < TOL
        elif x[i] == lo[i]:
            assert w[i] > -TOL
        elif x[i] == hi[i]:
            assert w[i] < TOL
        else:
            raise AssertionError("x outside its bounds")


def numpy_reference(A, b, method, C, lo, hi, omega=1.5, cap=500):
    """plain scalar loops, written independently of dense_iter.c"""
    n = len(b)
    x = b.copy()
    k = 1.0 / omega

    def err(x):
        w = A @ x - b
        ine = ~C
        return (np.linalg.norm(w[C]) + np.linalg.norm(w[ine & (x == lo) & (w < 0)]) + np.linalg.norm(w[ine & (x == hi) & (w > 0)])
                + np.linalg.norm(w[ine & (x > lo) & (x < hi)]))
    it = 0
    while err(x) > TOL and it < cap:
        new = x.copy()
        order = range(n) if method != 2 else range(n - 1, -1, -1)
        src = x if method == 0 else new
        for i in order:
            s = b[i] - sum(A[i, j] * (src[j] if (method == 0 or (method == 1 and j < i) or (method == 2 and j > i)) else x[j]) for j in range(n) if j != i)
            d = A[i, i]
            if method == 2:
                v = (s - (1 - k) * d * x[i]) / (k * d)
            else:
                v = s / d
            new[i] = v if C[i] else min(max(v, lo[i]), hi[i])
        x = new
        it += 1
    return x, it


def test_reference_tests_restated():
    rng = np.random.default_rng(3)
    inf = np.inf
    for inst in range(10):
        n = int(rng.integers(3, 51))
        b = rng.uniform(-1, 1, n)
        A = diag_dominant(rng, n)
        for method in (orc.JACOBI, orc.GAUSS_SEIDEL, orc.SOR):            # :355-381, 464-476
            x, it, res = orc.dense_iterate(A, b, method)
            assert np.linalg.norm(A @ x - b) < TOL and it < 500
        for method, ridge in ((orc.GAUSS_SEIDEL, 1.0), (orc.SOR, 2.0)):   # :383-401, 478-490
            S = spd(rng, n, ridge)
            x, it, res = orc.dense_iterate(S, b, method)
            assert np.linalg.norm(S @ x - b) < TOL
        C = rng.integers(0, 2, n).astype(bool)
        S = spd(rng, n, 0.5)
        for lo, hi in ((np.full(n, -inf), np.full(n, inf)), (np.full(n, -0.5), np.full(n, 0.5))):      # :403-462
            x, it, res = orc.dense_iterate(S, b, orc.GAUSS_SEIDEL, C, lo, hi)
            check_mixed(S, b, x, C, lo, hi)
        S = spd(rng, n, 2.0)
        x, it, res = orc.dense_iterate(S, b, orc.SOR, C, np.full(n, -10.0), np.full(n, 10.0))             # :492-513
        check_mixed(S, b, x, C, np.full(n, -10.0), np.full(n, 10.0))


def test_against_an_independent_numpy_loop():
    rng = np.random.default_rng(4)
    for n in (3, 11, 24):
        S = spd(rng, n, 0.5)
        b = rng.uniform(-1, 1, n)
        C = rng.integers(0, 2, n).astype(bool)
        lo, hi = np.full(n, -0.3), np.full(n, 0.4)
        for method in (orc.JACOBI, orc.GAUSS_SEIDEL, orc.SOR):
            A = diag_dominant(rng, n) if method == orc.JACOBI else S
            x, it, res = orc.dense_iterate(A, b, method, C, lo, hi)
            xr, itr = numpy_reference(A, b, method, C, lo, hi)
            assert abs(it - itr) <= 1 and np.abs(x - xr).max() < 1e-9


def test_edges():
    x, it, res = orc.dense_iterate(np.zeros((0, 0)), np.zeros(0), orc.GAUSS_SEIDEL)      # dim 0 (:79-81)
    assert len(x) == 0 and it == 0
    A = np.array([[2.0]]); b = np.array([3.0])
    x, it, res = orc.dense_iterate(A, b, orc.SOR)
    assert abs(x[0] - 1.5) < 1e-9
    # x0 = b already solves it: zero sweeps
    x, it, res = orc.dense_iterate(np.eye(4), np.arange(4.0), orc.JACOBI)
    assert it == 0 and np.array_equal(x, np.arange(4.0))
