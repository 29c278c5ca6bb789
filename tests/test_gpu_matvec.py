"""GPU (-m gpu): the stand-alone matrix-free products (egs_problem_matvec /
egs_matvec_blocks) against
  * the LITERAL O(m^2) restatement of sparse::CalculateSparse{JMJtX,Lx,Ux,Dx}
    (oracle/sparse_literal.c, sparse_iterations_utils.cc:427-695) at 1e-9 -- the
    tolerance of the reference's own tests of these functions (:938-1052),
  * numpy's dense J W J^T (independent of the oracle's block code) at 1e-9,
  * the O(nnz) twin in the kernels' operation order (oracle/matvec_fast.inc): bit-exact.
Scenarios as in the reference's tests (Chain(4) over 20 steps) plus box stacks, random
topologies, a connected wall (shared bodies, the boundary pre-pass) and C3."""
import numpy as np
import pytest

from eggshell_amd import capi, scenes
from helpers import dense_numpy, ode_step, random_system, system_from_scene
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

L, U, D, FULL = capi.MV_LOWER, capi.MV_UPPER, capi.MV_DIAG, capi.MV_FULL
EPS, SCALE = 0.01, 1.0 / 1.5   # cfm 0.01 as the reference's tests (:945); scale = 1/omega (sparse_iterations.cc:193)


def literal(s, x, parts):
    if parts == FULL:
        return orc.lit_JMJtX(s, x, EPS)
    y = np.zeros(3 * s.m)
    first = True
    for bit, f in ((L, lambda: orc.lit_Lx(s, x)), (U, lambda: orc.lit_Ux(s, x)), (D, lambda: orc.lit_Dx(s, x, EPS, SCALE))):
        if parts & bit:
            y = f() if first else y + f()
            first = False
    return y


def check_all_parts(ctx, s, x, lit=True, dense=False):
    A = dense_numpy(s, 0.0)[0] if dense else None
    for parts in (FULL, L, U, D, L | U, U | D, L | D):
        y = ctx.matvec_blocks(s.Minv, s.body0, s.body1, s.J0, s.J1, x, parts, EPS, SCALE)
        twin = orc.fast_matvec(s, x, parts, EPS, SCALE)
        assert np.array_equal(y, twin), "parts %d: GPU differs from the twin by %g" % (parts, np.abs(y - twin).max())
        if lit:
            assert np.linalg.norm(y - literal(s, x, parts)) < 1e-9
        if dense:
            Lo, Up, Dg = np.tril(A, -1), np.triu(A, 1), np.diag(A)
            ref = {FULL: (A + EPS * np.eye(A.shape[0])) @ x, L: Lo @ x, U: Up @ x, D: (Dg + EPS) * SCALE * x,
                   L | U: Lo @ x + Up @ x, U | D: Up @ x + (Dg + EPS) * SCALE * x, L | D: Lo @ x + (Dg + EPS) * SCALE * x}[parts]
            assert np.linalg.norm(y - ref) < 1e-9


def test_chain4_trajectory(ctx):
    """sparse_iterations_utils.cc:938-1052: Chain(4,(0,0,2)) at t = 0 and after each of 20 ODE steps."""
    sc = scenes.chain(4)
    rng = np.random.default_rng(0)
    for step in range(21):
        s, _ = system_from_scene(sc)
        s = orc.Sys(sc.get("Minv0", s.Minv), s.body0, s.body1, s.J0, s.J1, s.is_eq, s.lo, s.hi)   # M^-1 frozen at Init (Q5)
        check_all_parts(ctx, s, rng.uniform(-1, 1, 3 * s.m), dense=True)
        ode_step(sc, 1e-3)


@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 2, 2), (3, 3, 3), (8, 8, 4)])
def test_box_stacks(ctx, shape):
    sc = scenes.box_stack(*shape, jitter=1e-3, seed=3)
    s, _ = system_from_scene(sc)
    x = np.random.default_rng(1).uniform(-1, 1, 3 * s.m)
    check_all_parts(ctx, s, x, lit=s.m <= 300, dense=s.m <= 300)


def test_random_topologies(ctx):
    rng = np.random.default_rng(11)
    for n, m, connected in [(2, 1, False), (5, 3, False), (30, 100, False), (40, 700, True), (300, 500, False), (6, 600, True)]:
        s, _ = random_system(rng, n, m, connected=connected)
        check_all_parts(ctx, s, rng.uniform(-1, 1, 3 * m), lit=m <= 200, dense=m <= 200)


def test_connected_wall_uses_shared_bodies(ctx):
    sc = scenes.brick_wall(12, 10)
    b0, b1, data = ctx.update_contacts(sc["p"], sc["R"])
    sc.update(kind=np.full(len(b0), capi.CONTACT_BOX, np.int32), body0=b0, body1=b1, data=data)
    s, _ = system_from_scene(sc)
    pl = capi.debug_matvec_plan(s.n, s.body0, s.body1, 256)
    assert pl["n_islands"] == 1 and pl["n_shared_bodies"] > 0 and pl["n_boundary"] > 0
    check_all_parts(ctx, s, np.random.default_rng(2).uniform(-1, 1, 3 * s.m), lit=False)
    y = ctx.matvec_blocks(s.Minv, s.body0, s.body1, s.J0, s.J1, np.ones(3 * s.m), FULL, EPS)
    assert np.linalg.norm(y - orc.lit_JMJtX(s, np.ones(3 * s.m), EPS)) < 1e-9 * max(1.0, np.abs(y).max())


@pytest.mark.parametrize("tile", ["128", "256"])
def test_both_tile_sizes(ctx, tile, monkeypatch):
    monkeypatch.setenv("EGS_MV_TILE", tile)
    sc = scenes.box_stack(5, 4, 6, jitter=1e-3, seed=4)
    s, _ = system_from_scene(sc)
    rng = np.random.default_rng(3)
    check_all_parts(ctx, s, rng.uniform(-1, 1, 3 * s.m), lit=False)
    s2, _ = random_system(rng, 20, 400, connected=True)
    check_all_parts(ctx, s2, rng.uniform(-1, 1, 3 * s2.m), lit=False)


def test_fp32(ctx):
    sc = scenes.box_stack(4, 4, 4, jitter=1e-3, seed=6)
    s, _ = system_from_scene(sc)
    x = np.random.default_rng(4).uniform(-1, 1, 3 * s.m)
    for parts in (FULL, L, U, D, L | U):
        y = ctx.matvec_blocks(s.Minv, s.body0, s.body1, s.J0, s.J1, x, parts, EPS, SCALE, precision=capi.F32)
        twin = orc.fast_matvec_f32(s, x, parts, EPS, SCALE)
        assert np.array_equal(y.astype(np.float32), twin)
        ref = orc.fast_matvec(s, x, parts, EPS, SCALE)
        assert np.abs(y - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())   # fp32 has no reference counterpart


def test_resident_problem_and_lambda_input(ctx):
    """x = NULL multiplies the device-resident lambda; A lambda - rhs must equal the solve's own w."""
    sc = scenes.box_stack(4, 3, 5, jitter=1e-3, seed=8)
    s, _ = system_from_scene(sc)
    rhs = np.random.default_rng(5).uniform(-1, 1, 3 * s.m)
    pr = capi.Problem(ctx, s.n, s.body0, s.body1)
    pr.set_blocks(s.Minv, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs)
    pr.solve(capi.params(method=capi.GAUSS_SEIDEL, max_iters=30, tol=0.0, cfm=EPS))
    lam, w = pr.lambda_(), pr.wres()
    y = pr.matvec(None, FULL, EPS)
    assert np.array_equal(y, orc.fast_matvec(s, lam, FULL, EPS))
    assert np.linalg.norm((y - rhs) - w) < 1e-9
    assert np.linalg.norm(w - (orc.lit_JMJtX(s, lam, EPS) - rhs)) < 1e-9     # element-wise, vs the literal product
    pr.matvec(lam, L | D, EPS, SCALE, fetch=False)                                   # asynchronous form
    assert np.array_equal(pr.matvec_result(), orc.fast_matvec(s, lam, L | D, EPS, SCALE))
    pr.close()


def test_c3_full_size(ctx):
    """BASELINE config 3 (16 384 contacts): bit-exact vs the twin; linearity as the size-independent property."""
    sc = scenes.box_stack(16, 16, 16, jitter=1e-3, seed=1)
    s, _ = system_from_scene(sc)
    rng = np.random.default_rng(6)
    x1, x2 = rng.uniform(-1, 1, 3 * s.m), rng.uniform(-1, 1, 3 * s.m)
    pr = capi.Problem(ctx, s.n, s.body0, s.body1)
    pr.set_blocks(s.Minv, s.J0, s.J1, s.is_eq, s.lo, s.hi, x1)
    y1, y2, y12 = pr.matvec(x1, FULL, EPS), pr.matvec(x2, FULL, EPS), pr.matvec(x1 + 2.0 * x2, FULL, EPS)
    assert np.array_equal(y1, orc.fast_matvec(s, x1, FULL, EPS))
    assert np.abs(y12 - (y1 + 2.0 * y2)).max() < 1e-9 * np.abs(y12).max()
    # L + U + D with scale 1 and the same eps is the full product
    parts = pr.matvec(x1, L | U, 0.0, 1.0) + pr.matvec(x1, D, EPS, 1.0)
    assert np.abs(parts - y1).max() < 1e-9 * np.abs(y1).max()
    pr.close()


def test_argument_errors(ctx):
    sc = scenes.box_stack(2, 2, 2)
    s, _ = system_from_scene(sc)
    pr = capi.Problem(ctx, s.n, s.body0, s.body1)
    with pytest.raises(capi.EgsError) as e:
        pr.matvec(np.zeros(3 * s.m), FULL)          # no blocks yet
    assert e.value.status == capi.ERR_INVALID
    pr.set_blocks(s.Minv, s.J0, s.J1, s.is_eq, s.lo, s.hi, np.zeros(3 * s.m))
    for bad in (0, 9, 16):
        with pytest.raises(capi.EgsError):
            pr.matvec(np.zeros(3 * s.m), bad)
    pr.close()


def test_edge_topologies(ctx):
    """No constraints; a constraint with the world on both sides (only eps x survives, D = 0); one body
    carrying 300 constraints (more than a tile: the body is shared by three tiles)."""
    pr = capi.Problem(ctx, 3, np.zeros(0, np.int32), np.zeros(0, np.int32))
    pr.set_blocks(np.tile(np.eye(6).reshape(36), (3, 1)), np.zeros((0, 18)), np.zeros((0, 18)), np.zeros(0, np.uint8),
                  np.zeros(0), np.zeros(0), np.zeros(0))
    assert pr.matvec(np.zeros(0), FULL, EPS).shape == (0,)
    pr.close()
    rng = np.random.default_rng(8)
    s, _ = random_system(rng, 4, 6)
    b0, b1 = s.body0.copy(), s.body1.copy()
    b0[2] = b1[2] = -1                                   # world on both sides
    J0, J1 = s.J0.copy(), s.J1.copy()
    J0[2] = 0.0; J1[2] = 0.0
    s2 = orc.Sys(s.Minv, b0, b1, J0, J1, s.is_eq, s.lo, s.hi)
    x = rng.uniform(-1, 1, 3 * s2.m)
    check_all_parts(ctx, s2, x, dense=True)
    y = ctx.matvec_blocks(s2.Minv, b0, b1, J0, J1, x, FULL, EPS)
    assert np.array_equal(y[6:9], EPS * x[6:9])
    m = 300
    star = orc.Sys(s.Minv[:2], np.where(rng.uniform(size=m) < 0.5, -1, 1).astype(np.int32), np.zeros(m, np.int32),
                   rng.uniform(-1, 1, (m, 18)), rng.uniform(-1, 1, (m, 18)), np.zeros(3 * m, np.uint8), np.zeros(3 * m), np.zeros(3 * m))
    star.J0[star.body0 < 0] = 0.0
    pl = capi.debug_matvec_plan(2, star.body0, star.body1, 128)
    assert pl["n_tiles"] == 3 and pl["n_shared_bodies"] >= 1
    check_all_parts(ctx, star, rng.uniform(-1, 1, 3 * m), lit=False)
    xs = rng.uniform(-1, 1, 3 * m)
    assert np.linalg.norm(ctx.matvec_blocks(star.Minv, star.body0, star.body1, star.J0, star.J1, xs, FULL, EPS) - orc.lit_JMJtX(star, xs, EPS)) < 1e-9 * 300


import glob
import os

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_products(ctx, path):
    """The committed golden vectors (tests/tools/make_golden.py): every product of x = rhs, bit for bit."""
    g = np.load(path)
    cfm, scale = float(g["cfm"]), 1.0 / 1.5
    for tag, parts in (("full", FULL), ("L", L), ("U", U), ("D", D), ("LU", L | U), ("UD", U | D), ("LD", L | D)):
        y = ctx.matvec_blocks(g["Minv"], g["body0"], g["body1"], g["J0"], g["J1"], g["rhs"], parts, cfm, scale)
        assert np.array_equal(y, g["mv_" + tag]), tag
