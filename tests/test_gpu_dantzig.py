"""GPU (-m gpu): lcp::SolveLCP_BoxDantzig with the incremental Cholesky factor (toolkit/lcp.cc:91-157,
444-619) on the device, through the C ABI (egs_box_lcp_dantzig), against the oracle's restatement
(oracle/lcp_toolkit.c, pinned by the reference's own property tests in tests/test_oracle_lcp_toolkit.py):
the reference's Dantzig property test restated (N = 7, 6 bound variants, box-LCP conditions, |Ax - b - w| <
1e-6), the same pivot sequence and the same in-place permutation of A as the sequential algorithm, sizes up
to the 96-row limit of the in-LDS instantiation and beyond it (matrices in device memory, four wavefronts), the
batched entry (one workgroup per problem), the step cap every device loop carries, and the documented refusals."""
import numpy as np
import pytest

from eggshell_amd import capi
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def spd(rng, n, ridge):
    A0 = rng.uniform(-1, 1, (n, n))
    return A0 @ A0.T + ridge * np.eye(n)


def check_box_lcp(A, b, lo, hi, x, w):
    for i in range(len(b)):
        assert ((lo[i] <= x[i] <= hi[i]) and w[i] == 0) or (x[i] == lo[i] and w[i] >= 0) or (x[i] == hi[i] and w[i] <= 0), i
    assert np.linalg.norm(A @ x - b - w) < 1e-6


def test_reference_property_test_n7(ctx):      # toolkit/lcp.cc:947-1003
    rng = np.random.default_rng(5)
    N = 7
    for it in range(120):
        A = spd(rng, N, 0.001)
        b = rng.uniform(-1, 1, N)
        lo_range, hi_range = [(10, 10), (100, 100), (1e99, 1e99), (1, 1), (0, 10), (10, 0)][it % 6]
        lo = -rng.uniform(0, 1, N) * lo_range + 0.0
        hi = rng.uniform(0, 1, N) * hi_range + 0.0
        dead = (lo == 0) & (hi == 0)
        hi[dead] = 1.0
        ok, x, w, Ap, perm, piv = ctx.box_lcp_dantzig(np.tril(A), b, lo, hi)
        assert ok
        check_box_lcp(A, b, lo, hi, x, w)
        oko, xo, wo, Ao, permo, pivo = orc.tk_box_dantzig(np.tril(A), b, lo, hi)
        # the same steps as the sequential algorithm: pivot count, permutation and the matrix left behind
        assert piv == pivo and np.array_equal(perm, permo)
        assert np.array_equal(np.tril(Ap), np.tril(Ao))
        assert np.array_equal(np.tril(Ap), np.tril(A[np.ix_(perm, perm)]))
        assert np.abs(x - xo).max() < 1e-12 and np.abs(w - wo).max() < 1e-12


@pytest.mark.parametrize("n", [1, 2, 13, 40, 64, 65, 96])
def test_sizes_up_to_the_limit(ctx, n):
    rng = np.random.default_rng(100 + n)
    for trial in range(3):
        A = spd(rng, n, 0.05)
        b = rng.uniform(-2, 2, n)
        lo = -rng.uniform(0.05, 2, n); hi = rng.uniform(0.05, 2, n)
        hi[rng.uniform(size=n) < 0.3] = np.inf
        lo[rng.uniform(size=n) < 0.2] = 0.0
        upper_marked = np.tril(A) + np.triu(np.full((n, n), 555.0), 1)          # the upper triangle is never read ...
        ok, x, w, Ap, perm, piv = ctx.box_lcp_dantzig(upper_marked, b, lo, hi)
        assert ok
        assert np.array_equal(np.triu(Ap, 1), np.triu(upper_marked, 1))          # ... nor written
        check_box_lcp(A, b, lo, hi, x, w)
        oko, xo, wo, Ao, permo, pivo = orc.tk_box_dantzig(np.tril(A), b, lo, hi)
        assert oko and piv == pivo and np.array_equal(perm, permo)
        assert np.abs(x - xo).max() < 1e-10 and np.abs(w - wo).max() < 1e-10
        ok2, x2, w2, _ = orc.mixed_constraints(A, b, np.zeros(n, np.uint8), lo, hi, 1)   # independent route: dense box Murty
        assert ok2 and np.abs(x - x2).max() < 1e-8


def test_refusals_and_limits(ctx):
    rng = np.random.default_rng(9)
    n = 10
    A = spd(rng, n, 0.1); b = rng.uniform(-1, 1, n)
    lo = -np.ones(n); hi = np.ones(n)
    with pytest.raises(capi.EgsError) as e:      # lo must be <= 0 (toolkit/lcp.cc:448-450)
        ctx.box_lcp_dantzig(A, b, lo + 2.0, hi + 2.0)
    assert e.value.status == capi.ERR_INVALID
    with pytest.raises(capi.EgsError) as e:      # beyond the incremental solvers' limit (1024 rows)
        ctx.box_lcp_dantzig(np.eye(1025), np.zeros(1025), -np.ones(1025), np.ones(1025))
    assert e.value.status == capi.ERR_INVALID
    ok, x, w, Ap, perm, piv = ctx.box_lcp_dantzig(A, b, lo, hi, max_steps=1)      # max_iterations-style give-up
    full = ctx.box_lcp_dantzig(A, b, lo, hi)
    assert full[0] and (ok or piv > 1) and (not ok or full[5] <= 1)
    Abad = A.copy(); Abad[3, 3] = -5.0                                            # not positive definite
    ok, *_ = ctx.box_lcp_dantzig(Abad, b, lo, hi)
    oko = orc.tk_box_dantzig(Abad, b, lo, hi)[0]
    assert ok == oko


def test_box_murty_on_a_linear_reducer(ctx):      # toolkit/lcp.cc:874-945 restated, and the oracle's steps
    rng = np.random.default_rng(15)
    N = 7
    for it in range(60):
        A = spd(rng, N, 0.0 if it % 2 else 0.001)
        b = rng.uniform(-1, 1, N)
        # standard LCP = box with lo = 0, hi = DBL_MAX (SolveLCP_Murty)
        lo = np.zeros(N); hi = np.full(N, np.finfo(float).max)
        ok, x, w, Ap, perm, iters = ctx.box_lcp_murty(np.tril(A), b, lo, hi)
        assert ok and (x >= 0).all() and (w >= 0).all() and np.all(x * w == 0) and np.linalg.norm(A @ x - b - w) < 1e-6
        oko, xo, wo, Ao, permo, ito = orc.tk_box_murty(np.tril(A), b, lo, hi)
        assert iters == ito and np.array_equal(perm, permo) and np.array_equal(np.tril(Ap), np.tril(Ao))
        assert np.abs(x - xo).max() < 1e-12 and np.abs(w - wo).max() < 1e-12
        lo = -rng.uniform(0, 1, N) * 10.0; hi = rng.uniform(0, 1, N) * 10.0
        ok, x, w, Ap, perm, iters = ctx.box_lcp_murty(np.tril(A), b, lo, hi)
        assert ok
        check_box_lcp(A, b, lo, hi, x, w)
        oko, xo, wo, Ao, permo, ito = orc.tk_box_murty(np.tril(A), b, lo, hi)
        assert iters == ito and np.array_equal(perm, permo) and np.array_equal(np.tril(Ap), np.tril(Ao))


@pytest.mark.parametrize("n", [1, 2, 17, 64, 96])
def test_box_murty_sizes_and_limit(ctx, n):
    rng = np.random.default_rng(300 + n)
    A = spd(rng, n, 0.05)
    b = rng.uniform(-2, 2, n)
    lo = -rng.uniform(0.05, 2, n); hi = rng.uniform(0.05, 2, n)
    hi[rng.uniform(size=n) < 0.3] = np.inf
    ok, x, w, Ap, perm, iters = ctx.box_lcp_murty(np.tril(A), b, lo, hi)
    assert ok
    check_box_lcp(A, b, lo, hi, x, w)
    oko, xo, wo, Ao, permo, ito = orc.tk_box_murty(np.tril(A), b, lo, hi)
    assert oko and iters == ito and np.array_equal(perm, permo)
    assert np.abs(x - xo).max() < 1e-10 and np.abs(w - wo).max() < 1e-10
    okd, xd, wd, *_ = ctx.box_lcp_dantzig(np.tril(A), b, lo, hi)           # the other algorithm, same unique solution
    assert okd and np.abs(x - xd).max() < 1e-8
    if iters > 1:       # Settings.max_iterations: gives up, returns false (toolkit/lcp.cc:438-441)
        assert not ctx.box_lcp_murty(np.tril(A), b, lo, hi, max_iterations=iters - 1)[0]


def test_repeated_calls_are_independent(ctx):
    """Back-to-back calls with the same and with different inputs (device buffers are allocated per call)."""
    rng = np.random.default_rng(77)
    for n in (12, 48, 96):
        A = spd(rng, n, 0.05); b = rng.uniform(-1, 1, n)
        lo = -np.full(n, 0.2); hi = np.full(n, 0.3)
        for fn, ofn in ((ctx.box_lcp_dantzig, orc.tk_box_dantzig), (ctx.box_lcp_murty, orc.tk_box_murty)):
            o = ofn(np.tril(A), b, lo, hi)
            for rep in range(4):
                r = fn(np.tril(A), b, lo, hi)
                assert r[0] and r[5] == o[5] and np.array_equal(r[4], o[4]) and np.array_equal(np.tril(r[3]), np.tril(o[3]))


@pytest.mark.parametrize("n,alg", [(97, 1), (97, 0), (160, 1), (200, 0), (333, 1)])
def test_beyond_the_lds_limit_the_same_steps(ctx, n, alg):
    """n > 96: A is permuted in place in device memory, L in a work area, four wavefronts per problem -- the same
    pivot sequence, permutation and matrix as the sequential restatement."""
    rng = np.random.default_rng(500 + n)
    A = spd(rng, n, 0.05)
    b = rng.uniform(-2, 2, n)
    lo = -rng.uniform(0.05, 2, n); hi = rng.uniform(0.05, 2, n)
    hi[rng.uniform(size=n) < 0.3] = np.inf
    marked = np.tril(A) + np.triu(np.full((n, n), 555.0), 1)
    fn, ofn = ((ctx.box_lcp_murty, orc.tk_box_murty), (ctx.box_lcp_dantzig, orc.tk_box_dantzig))[alg]
    ok, x, w, Ap, perm, piv = fn(marked, b, lo, hi)
    oko, xo, wo, Ao, permo, pivo = ofn(np.tril(A), b, lo, hi)
    assert ok and oko and piv == pivo and np.array_equal(perm, permo)
    assert np.array_equal(np.triu(Ap, 1), np.triu(marked, 1))
    assert np.array_equal(np.tril(Ap), np.tril(Ao))
    assert np.abs(x - xo).max() < 1e-9 and np.abs(w - wo).max() < 1e-9
    check_box_lcp(A, b, lo, hi, x, w)


def test_batch_equals_the_single_calls(ctx):
    """egs_box_lcp_batch: one workgroup per problem, mixed sizes (both instantiations in one call); every problem's
    pivots, permutation, matrix and solution are those of its own single call and of the oracle."""
    rng = np.random.default_rng(808)
    sizes = [7, 24, 1, 96, 40, 130, 24, 12, 64, 97, 3]
    for alg, ofn in ((1, orc.tk_box_dantzig), (0, orc.tk_box_murty)):
        As, bs, los, his = [], [], [], []
        for n in sizes:
            As.append(np.tril(spd(rng, n, 0.05))); bs.append(rng.uniform(-2, 2, n))
            los.append(-rng.uniform(0.05, 2, n)); his.append(rng.uniform(0.05, 2, n))
        ok, x, w, Ap, perm, piv = ctx.box_lcp_batch(alg, As, bs, los, his)
        for k, n in enumerate(sizes):
            oko, xo, wo, Ao, permo, pivo = ofn(As[k], bs[k], los[k], his[k])
            assert ok[k] and oko and piv[k] == pivo and np.array_equal(perm[k], permo), (alg, k)
            assert np.array_equal(np.tril(Ap[k]), np.tril(Ao))
            assert np.abs(x[k] - xo).max() < 1e-10 and np.abs(w[k] - wo).max() < 1e-10


def test_batch_of_a_thousand_contact_sized_problems(ctx):
    """1 536 problems of 24 rows (8 contacts x 3) in one launch: spot-checked against the oracle, all KKT-checked."""
    rng = np.random.default_rng(909)
    cnt, n = 1536, 24
    As, bs, los, his = [], [], [], []
    for k in range(cnt):
        A = spd(rng, n, 0.1)
        As.append(np.tril(A)); bs.append(rng.uniform(-1, 1, n))
        los.append(np.tile([-1.0, -1.0, 0.0], n // 3)); his.append(np.tile([1.0, 1.0, np.inf], n // 3))   # the friction box of contact.cc:103-113
    ok, x, w, Ap, perm, piv = ctx.box_lcp_batch(1, As, bs, los, his)
    assert all(ok)
    for k in range(cnt):
        Af = As[k] + np.tril(As[k], -1).T
        check_box_lcp(Af, bs[k], los[k], his[k], x[k], w[k])
    for k in range(0, cnt, 97):
        oko, xo, wo, Ao, permo, pivo = orc.tk_box_dantzig(As[k], bs[k], los[k], his[k])
        assert piv[k] == pivo and np.array_equal(perm[k], permo) and np.abs(x[k] - xo).max() < 1e-11


def test_every_device_loop_has_an_exit(ctx):
    """A degenerate (singular PSD) matrix and a zero step budget: the call returns false / EGS_ERR_LCP_FAILED cleanly --
    the device loop is capped at 20 n + 1000 steps where the reference's `while (true)` (toolkit/lcp.cc:493) has no cap."""
    rng = np.random.default_rng(4242)
    n = 30
    V = rng.uniform(-1, 1, (n, 5))
    A = V @ V.T                                        # rank 5: positive SEMI-definite, most pivots are ~0
    b = rng.uniform(-1, 1, n)
    lo = -np.ones(n); hi = np.ones(n)
    for fn in (ctx.box_lcp_dantzig, ctx.box_lcp_murty):
        ok, x, w, Ap, perm, piv = fn(np.tril(A), b, lo, hi)
        assert piv <= 20 * n + 1001                    # whatever the outcome, the loop ended inside the cap
        if ok:
            check_box_lcp(A, b, lo, hi, x, w)
    A = spd(rng, n, 0.05)
    ok, x, w, Ap, perm, piv = ctx.box_lcp_batch(0, [np.tril(A)], [b], [lo], [hi], max_seconds=1e-9)     # Settings::max_time
    full = ctx.box_lcp_murty(np.tril(A), b, lo, hi)
    assert full[0] and (not ok[0] or full[5] <= 1)
