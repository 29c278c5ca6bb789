"""CPU: the oracle's restatement of toolkit/lcp.cc's incremental-factor box LCP (oracle/lcp_toolkit.c)
held against the reference's OWN property tests, restated with their sizes and tolerances
(toolkit/lcp.cc:799-1078: N = 7, 1000 random SPD problems; Cholesky / LSolve / AddCholeskyRow 1e-10,
SwapCholeskyRows against a refactorisation 1e-9, Dantzig box-LCP conditions and |Ax - b - w| < 1e-6),
plus agreement with the oracle's dense box Murty (lcp_dense.c), an independent route to the same
unique solution.  The reference holds no golden vectors for this path ("parity unpinned")."""
import numpy as np

from oracle import oracle as orc

N = 7


def spd(rng, n, ridge=0.0):
    A0 = rng.uniform(-1, 1, (n, n))
    return A0 @ A0.T + ridge * np.eye(n)


def lower(A):
    return np.tril(A)          # only the lower triangle is handed over, as the reference's tests do


def check_box_lcp(A, b, lo, hi, x, w):
    for i in range(len(b)):
        assert ((lo[i] <= x[i] <= hi[i]) and w[i] == 0) or (x[i] == lo[i] and w[i] >= 0) or (x[i] == hi[i] and w[i] <= 0), i
    assert np.linalg.norm(A @ x - b - w) < 1e-6


def test_cholesky_and_solves():      # toolkit/lcp.cc:1007-1035
    rng = np.random.default_rng(1)
    for _ in range(50):
        A = spd(rng, N)
        ok, L = orc.tk_cholesky(lower(A))
        assert ok and np.linalg.norm(np.tril(L) - np.linalg.cholesky(A)) < 1e-10
        for n in range(1, N):
            b = rng.uniform(-1, 1, n)
            x = orc.tk_lsolve(L, n, b)
            assert np.linalg.norm(np.tril(L)[:n, :n] @ x - b) < 1e-10
            y = orc.tk_lsolve(L, n, b, transpose=True)
            assert np.linalg.norm(np.tril(L)[:n, :n].T @ y - b) < 1e-10


def test_add_cholesky_row():         # toolkit/lcp.cc:1037-1049
    rng = np.random.default_rng(2)
    for n in range(1, N + 1):
        A = spd(rng, N)
        Lfull = np.linalg.cholesky(A)
        L = Lfull.copy()
        L[n - 1:, :] = 0.0
        L[:, n - 1:] = 0.0
        ok, L2 = orc.tk_add_cholesky_row(lower(A), n, L)
        assert ok and np.linalg.norm(np.tril(L2)[:n, :n] - Lfull[:n, :n]) < 1e-10


def test_swap_cholesky_rows_against_refactorisation():      # toolkit/lcp.cc:1051-1078
    rng = np.random.default_rng(3)
    for sz in range(N, N + 5):
        for n in range(N):
            A = spd(rng, sz)
            L = np.linalg.cholesky(A)
            ok, L2 = orc.tk_swap_cholesky_rows(lower(A), n, N, L)
            assert ok
            A2 = A.copy()
            A2[:, [n, N - 1]] = A2[:, [N - 1, n]]
            A2[[n, N - 1], :] = A2[[N - 1, n], :]
            Lref = np.linalg.cholesky(A2)
            assert np.linalg.norm(np.tril(L2)[:N - 1, :N - 1] - Lref[:N - 1, :N - 1]) < 1e-9, (sz, n)
            if sz > N:
                assert np.array_equal(L2[N:, :], L[N:, :])      # rows beyond the block are not touched


def test_swap_rows_and_columns_touches_the_lower_triangle_only():      # toolkit/lcp.cc:171-195
    rng = np.random.default_rng(4)
    n = 9
    A = spd(rng, n)
    marked = np.tril(A) + np.triu(np.full((n, n), 777.0), 1)
    for i, j in ((0, 8), (2, 5), (5, 2), (3, 4), (6, 6)):
        P, perm = orc.tk_swap_rows_and_columns(marked, i, j, np.arange(n))
        assert np.array_equal(np.triu(P, 1), np.triu(marked, 1))
        want = A[np.ix_(perm, perm)]
        assert np.array_equal(np.tril(P), np.tril(want))


def test_box_dantzig_reference_property_test():      # toolkit/lcp.cc:947-1003
    rng = np.random.default_rng(5)
    for it in range(1000):
        A = spd(rng, N, 0.001)
        b = rng.uniform(-1, 1, N)
        lo_range, hi_range = 10.0, 10.0
        v = it % 6
        if v == 1: lo_range, hi_range = 100.0, 100.0
        elif v == 2: lo_range = hi_range = 1e99
        elif v == 3: lo_range, hi_range = 1.0, 1.0
        elif v == 4: lo_range = 0.0
        elif v == 5: hi_range = 0.0
        lo = -rng.uniform(0, 1, N) * lo_range
        hi = rng.uniform(0, 1, N) * hi_range
        lo = lo + 0.0; hi = hi + 0.0          # no negative zeros
        for i in range(N):
            r = int(rng.integers(0, 100))
            if r == 0 and hi[i] != 0: lo[i] = 0.0
            elif r == 1 and lo[i] != 0: hi[i] = 0.0
        keep = ~((lo == 0) & (hi == 0))      # the algorithm requires lo < hi (toolkit/lcp.cc:448-450)
        if not keep.all():
            hi[~keep] = 1.0
        ok, x, w, Ap, perm, piv = orc.tk_box_dantzig(lower(A), b, lo, hi)
        assert ok
        check_box_lcp(A, b, lo, hi, x, w)
        # A is permuted in place (toolkit/lcp.h:170-171): lower triangle of P A P'
        assert np.allclose(np.tril(Ap), np.tril(A[np.ix_(perm, perm)]), rtol=0, atol=0)
        assert sorted(perm.tolist()) == list(range(N))


def test_box_dantzig_agrees_with_the_dense_box_murty():
    rng = np.random.default_rng(6)
    for n in (1, 2, 5, 12, 30, 64):
        for _ in range(10):
            A = spd(rng, n, 0.05)
            b = rng.uniform(-2, 2, n)
            lo = -rng.uniform(0.05, 2, n); hi = rng.uniform(0.05, 2, n)
            hi[rng.uniform(size=n) < 0.3] = np.inf
            lo[rng.uniform(size=n) < 0.2] = 0.0
            ok, x, w, Ap, perm, piv = orc.tk_box_dantzig(lower(A), b, lo, hi)
            assert ok
            check_box_lcp(A, b, lo, hi, x, w)
            ok2, x2, w2, _ = orc.mixed_constraints(A, b, np.zeros(n, np.uint8), lo, hi, 1)     # true box Murty
            assert ok2 and np.abs(x - x2).max() < 1e-8 and np.abs(w - w2).max() < 1e-8


def test_box_murty_reference_property_test():      # toolkit/lcp.cc:874-945
    rng = np.random.default_rng(7)
    for it in range(1000):
        A = spd(rng, N)
        b = rng.uniform(-1, 1, N)
        # the standard LCP = the box problem with lo = 0, hi = "infinity" (toolkit/lcp.h:149-154)
        lo = np.zeros(N); hi = np.full(N, np.finfo(float).max)
        ok, x, w, Ap, perm, iters = orc.tk_box_murty(lower(A), b, lo, hi)
        assert ok
        assert (x >= 0).all() and (w >= 0).all() and np.all(x * w == 0)
        assert np.linalg.norm(A @ x - b - w) < 1e-6
        assert np.array_equal(np.tril(Ap), np.tril(A[np.ix_(perm, perm)]))
        # random box
        lo = -rng.uniform(0, 1, N) * 10.0; hi = rng.uniform(0, 1, N) * 10.0
        for i in range(N):
            r = int(rng.integers(0, 100))
            if r == 0: lo[i] = 0.0
            elif r == 1: hi[i] = 0.0
        ok, x, w, Ap, perm, iters = orc.tk_box_murty(lower(A), b, lo, hi)
        assert ok
        check_box_lcp(A, b, lo, hi, x, w)


def test_box_murty_agrees_with_dantzig_and_the_dense_murty_and_honours_the_limit():
    rng = np.random.default_rng(8)
    for n in (1, 3, 10, 33, 64):
        for _ in range(8):
            A = spd(rng, n, 0.05)
            b = rng.uniform(-2, 2, n)
            lo = -rng.uniform(0.05, 2, n); hi = rng.uniform(0.05, 2, n)
            hi[rng.uniform(size=n) < 0.3] = np.inf
            ok, x, w, Ap, perm, iters = orc.tk_box_murty(lower(A), b, lo, hi)
            assert ok
            check_box_lcp(A, b, lo, hi, x, w)
            okd, xd, wd, *_ = orc.tk_box_dantzig(lower(A), b, lo, hi)
            assert okd and np.abs(x - xd).max() < 1e-8 and np.abs(w - wd).max() < 1e-8
            ok2, x2, w2, _ = orc.mixed_constraints(A, b, np.zeros(n, np.uint8), lo, hi, 1)
            assert ok2 and np.abs(x - x2).max() < 1e-8
            if iters > 1:      # max_iterations: gives up and returns false (toolkit/lcp.cc:438-441)
                assert not orc.tk_box_murty(lower(A), b, lo, hi, max_iterations=iters - 1)[0]


# ---- SolveLCP_BoxSchur, the reference's own test restated (toolkit/lcp.cc:1084-1200) ----------------------
BIG = np.finfo(np.float64).max       # __DBL_MAX__: the reference's "infinity" in this test


def test_box_schur_reference_test_restated():
    rng = np.random.default_rng(11)
    n = 20
    A = spd(rng, n)                                   # A0 * A0', A0 = Random(n, n)   (:1086-1087)
    b = rng.uniform(-1, 1, n)
    x_full = np.linalg.solve(A, b)
    assert np.linalg.norm(A @ x_full - b) < 1e-6       # :1099
    lo = np.full(n, -BIG); hi = np.full(n, BIG)
    # unbounded problem through the nub hook: nub == n (direct factor and solve) and nub == n / 2 (Schur complement);
    # only the lower triangle is handed over (:1109-1110, 1131), w must be exactly 0 (:1118-1121, 1140-1143)
    for nub in (n, n // 2):
        ok, x, w, Ap, perm, nub_out, it = orc.tk_box_schur(lower(A), b, lo, hi, nub=nub)
        assert ok and nub_out == nub
        assert np.linalg.norm(x - x_full) < 1e-6
        assert w.shape == x.shape and np.all(w == 0)
    # a full box LCP with nub = 0 and nub = n / 2 (:1146-1173)
    for start, end in ((0, n), (n // 4, n // 4 + n // 2)):
        lo = np.full(n, -BIG); hi = np.full(n, BIG)
        lo[start:end] = -rng.uniform(0, 1, end - start) * 10.0
        hi[start:end] = rng.uniform(0, 1, end - start) * 10.0
        ok, x, w, Ap, perm, nub_out, it = orc.tk_box_schur(lower(A), b, lo, hi)
        assert ok
        assert np.linalg.norm(A @ x - b - w) < 1e-6
        assert np.all(x >= lo) and np.all(x <= hi)
        assert nub_out == n - (end - start)
    # random nub (:1176-1199)
    for _ in range(100):
        lo = np.full(n, -BIG); hi = np.full(n, BIG)
        pick = rng.integers(0, 2, n) == 1
        lo[pick] = -rng.uniform(0, 1, pick.sum()) * 10.0
        hi[pick] = rng.uniform(0, 1, pick.sum()) * 10.0
        ok, x, w, Ap, perm, nub_out, it = orc.tk_box_schur(lower(A), b, lo, hi)
        assert ok
        assert np.linalg.norm(A @ x - b - w) < 1e-6
        assert np.all(x >= lo) and np.all(x <= hi)
        assert nub_out == n - pick.sum()
        # what the partition leaves in A: the symmetric permutation, lower triangle only; upper stays as given (0 here)
        assert np.array_equal(np.tril(Ap), np.tril(A[np.ix_(perm, perm)])) or nub_out == 0
        assert np.all(np.triu(Ap, 1) == 0)
        # the unbounded rows come first
        unb = ~pick
        assert np.all(unb[perm[:nub_out]]) and not np.any(unb[perm[nub_out:]])


def test_box_schur_both_inner_algorithms_and_the_box_conditions():
    rng = np.random.default_rng(12)
    for n in (5, 20, 60):
        for trial in range(10):
            A = spd(rng, n, 0.01)
            b = rng.uniform(-1, 1, n)
            lo = np.full(n, -np.inf); hi = np.full(n, np.inf)         # the real infinity is "infinity" too (toolkit/lcp.h:149-150)
            pick = rng.uniform(size=n) < 0.6
            lo[pick] = -rng.uniform(0.01, 1, pick.sum()); hi[pick] = rng.uniform(0.01, 1, pick.sum())
            res = []
            for alg in (0, 1):
                ok, x, w, Ap, perm, nub, it = orc.tk_box_schur(lower(A), b, lo, hi, algorithm=alg)
                assert ok and nub == n - pick.sum()
                check_box_lcp(A, b, lo, hi, x, w)
                res.append(x)
            assert np.abs(res[0] - res[1]).max() < 1e-9               # one solution (A is SPD)
            ok2, x2, w2, piv2 = orc.mixed_constraints(A, b, (~pick).astype(np.uint8), np.where(pick, lo, 0.0), np.where(pick, hi, 1.0), 1)
            assert ok2 and np.abs(res[0] - x2).max() < 1e-8           # the dense restatement (lcp_dense.c), an independent route


def test_box_schur_quirk_q6_classifies_by_the_lower_bound_alone():
    # toolkit/lcp.cc:664, 669 test `hi < -DBL_MAX` / `hi >= -DBL_MAX`: a row with lo = -inf and a FINITE hi counts as
    # unbounded in the reference (its hi is then never looked at); q6 = False classifies it as bounded
    rng = np.random.default_rng(13)
    n = 8
    A = spd(rng, n, 0.1)
    b = rng.uniform(0.5, 1, n) * 5
    lo = np.full(n, -BIG); hi = np.full(n, BIG)
    lo[:4] = -1.0; hi[:4] = 1.0
    hi[6] = 0.01                                       # lo = -inf, hi finite
    ok, x, w, Ap, perm, nub, it = orc.tk_box_schur(lower(A), b, lo, hi, q6=True)
    assert ok and nub == 4
    ok, x2, w2, Ap, perm, nub2, it = orc.tk_box_schur(lower(A), b, lo, hi, q6=False)
    assert ok and nub2 == 3 and x2[6] <= 0.01 + 1e-15
    check_box_lcp(A, b, lo, hi, x2, w2)
