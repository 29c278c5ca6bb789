"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle on
the same inputs.  Bar: BIT-EXACT lambda and accumulators (the kernels perform
the oracle's fma chains in the oracle's order; integer/index work exact);
residuals, which are reduced in a different order, to 1e-12 relative."""
import glob
import os

import numpy as np
import pytest

from eggshell_amd import capi, scenes
from helpers import random_system, system_from_scene
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
TAGS = ((0, "jacobi"), (1, "gs"), (2, "sor"))


def gpu_solve(ctx, s, rhs, cfm, method, K, tol=0.0, precision=capi.F64, check_every=1):
    pr = capi.Problem(ctx, s.n, s.body0, s.body1, precision)
    pr.set_blocks(s.Minv, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs)
    st = pr.solve(capi.params(method=method, max_iters=K, tol=tol, cfm=cfm, check_every=check_every))
    x, a = pr.lambda_(), pr.accumulators()
    pr.close()
    return x, a, st


def same_bits(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_fixtures(ctx, path):
    g = np.load(path)
    s = orc.Sys(g["Minv"], g["body0"], g["body1"], g["J0"], g["J1"], g["is_eq"], g["lo"], g["hi"])
    for method, tag in TAGS:
        for K in (1, 10, 50):
            x, a, st = gpu_solve(ctx, s, g["rhs"], float(g["cfm"]), method, K)
            assert st.status == capi.OK and st.iterations == K
            assert same_bits(x, g["x_%s_%d" % (tag, K)]), (tag, K)
            assert same_bits(a, g["a_%s_%d" % (tag, K)]), (tag, K)
            ref = float(g["res_%s_%d" % (tag, K)])
            if np.isfinite(ref):
                assert abs(st.residual - ref) <= 1e-12 * max(1.0, ref)


@pytest.mark.parametrize("method", [capi.JACOBI, capi.GAUSS_SEIDEL, capi.SOR])
def test_one_shot_entry_chain_and_stacks(ctx, method):
    """egs_solve_blocks (the sparse::*Iteration replacement) on C1 and C2."""
    rng = np.random.default_rng(10)
    for sc, K in ((scenes.chain(8), 50), (scenes.box_stack(8, 8, 4), 50)):
        s, _ = system_from_scene(sc)
        rhs = rng.uniform(-1, 1, 3 * s.m)
        x, st = ctx.solve_blocks(s.Minv, s.body0, s.body1, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs,
                                 capi.params(method=method, max_iters=K, tol=0.0, cfm=0.01))
        xf, a, it, rf = orc.fast_iterate(s, rhs, 0.01, method, max_iters=K, tol=0.0)
        assert same_bits(x, xf)
        assert st.n_global == 0


@pytest.mark.parametrize("seed", range(6))
def test_random_topologies_bit_exact(ctx, seed):
    """Fuzz the schedule: random graphs (world sides, shared bodies, several
    islands per tile, islands in shuffled list order), mixed row types."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(3, 120))
    m = int(rng.integers(1, 240))
    s, rhs = random_system(rng, n, m, world_frac=float(rng.uniform(0, 0.5)))
    for method in (capi.GAUSS_SEIDEL, capi.SOR, capi.JACOBI):
        for K in (0, 1, 4, 23):
            x, a, st = gpu_solve(ctx, s, rhs, 0.05, method, K)
            xf, af, _, rf = orc.fast_iterate(s, rhs, 0.05, method, max_iters=K, tol=0.0)
            assert st.status == capi.OK
            assert same_bits(x, xf), (method, K)
            assert same_bits(a, af), (method, K)


@pytest.mark.parametrize("method", [capi.GAUSS_SEIDEL, capi.SOR, capi.JACOBI])
def test_cross_workgroup_path_bit_exact(ctx, method):
    """Islands larger than a workgroup tile: a 700-link chain (one island of
    700 constraints) next to small islands, and a dense random island."""
    rng = np.random.default_rng(7)
    sc = scenes.concat([scenes.chain(700), scenes.box_stack(2, 2, 3), scenes.chain(5)])
    s, _ = system_from_scene(sc)
    rhs = rng.uniform(-1, 1, 3 * s.m)
    for K in (0, 1, 6):
        x, a, st = gpu_solve(ctx, s, rhs, 0.05, method, K)
        assert st.status == capi.OK and st.n_global == 700
        xf, af, _, _ = orc.fast_iterate(s, rhs, 0.05, method, max_iters=K, tol=0.0)
        assert same_bits(x, xf) and same_bits(a, af), K
    s2, rhs2 = random_system(rng, 60, 900, connected=True)
    x, a, st = gpu_solve(ctx, s2, rhs2, 0.5, method, 3)
    assert st.n_global == 900 and st.status == capi.OK
    xf, af, _, _ = orc.fast_iterate(s2, rhs2, 0.5, method, max_iters=3, tol=0.0)
    assert same_bits(x, xf) and same_bits(a, af)


@pytest.mark.parametrize("method", [capi.GAUSS_SEIDEL, capi.SOR, capi.JACOBI])
def test_reference_stopping_rule(ctx, method):
    """tol = 1e-9, check_every = 1 (sparse_iterations.cc:206-222): same sweep
    count and the same bits as the oracle; the result passes the reference's
    CheckMixedConstraintSolutions (residual metric <= 1e-9)."""
    rng = np.random.default_rng(11)
    for sc in (scenes.chain(4), scenes.box_stack(2, 2, 2)):
        s, _ = system_from_scene(sc)
        if method == capi.JACOBI and sc["kind"][0] == 1:
            continue   # Jacobi diverges on contacts (sparse_iterations.cc:576-578)
        rhs = rng.uniform(-1, 1, 3 * s.m)
        x, a, st = gpu_solve(ctx, s, rhs, 0.1, method, 500, tol=1e-9)
        xf, af, it, rf = orc.fast_iterate(s, rhs, 0.1, method, max_iters=500, tol=1e-9)
        assert st.iterations == it and it < 500
        assert same_bits(x, xf)
        assert st.residual <= 1e-9
        assert orc.lit_residual(s, rhs, x, 0.1) <= 2e-9
    # already converged at x0 = rhs: zero sweeps (rhs = 0)
    s, _ = system_from_scene(scenes.chain(4))
    x, a, st = gpu_solve(ctx, s, np.zeros(3 * s.m), 0.1, method, 500, tol=1e-9)
    assert st.iterations == 0 and not x.any()


def test_check_every_chunks(ctx):
    """check_every = k stops at the first multiple of k that satisfies tol."""
    rng = np.random.default_rng(12)
    s, _ = system_from_scene(scenes.box_stack(2, 2, 3))
    rhs = rng.uniform(-1, 1, 3 * s.m)
    x, a, st = gpu_solve(ctx, s, rhs, 0.1, capi.GAUSS_SEIDEL, 500, tol=1e-9, check_every=7)
    xf, af, it, rf = orc.fast_iterate(s, rhs, 0.1, orc.GAUSS_SEIDEL, max_iters=500, tol=1e-9, check_every=7)
    assert st.iterations == it and it % 7 == 0
    assert same_bits(x, xf) and same_bits(a, af)


def test_fp32_mode(ctx):
    """C4 precision: bit-exact against the fp32 oracle; against fp64 within the
    stated tolerance 2e-3 relative (PGS 50 sweeps on a 4x4x4 stack)."""
    rng = np.random.default_rng(13)
    sc = scenes.box_stack(4, 4, 4, jitter=1e-3, seed=5)
    s, _ = system_from_scene(sc)
    rhs = rng.uniform(-1, 1, 3 * s.m)
    for method in (capi.GAUSS_SEIDEL, capi.SOR):
        x32, a32, st = gpu_solve(ctx, s, rhs, 0.01, method, 50, precision=capi.F32)
        xo, ao, _, _ = orc.fast_iterate_f32(s, rhs, 0.01, method, max_iters=50)
        assert same_bits(x32.astype(np.float32), xo)
        assert same_bits(a32.astype(np.float32), ao)
        x64, _, _, _ = orc.fast_iterate(s, rhs, 0.01, method, max_iters=50, tol=0.0)
        assert np.abs(x32 - x64).max() <= 2e-3 * max(1.0, np.abs(x64).max())


def test_empty_and_degenerate(ctx):
    """m = 0 (sparse_iterations.cc:152-154); a single world-anchored joint; a
    body no constraint touches."""
    x, st = ctx.solve_blocks(np.zeros((3, 36)), [], [], np.zeros((0, 18)), np.zeros((0, 18)), [], [], [], [],
                             capi.params(max_iters=5, tol=0.0))
    assert x.shape == (0,) and st.iterations == 0
    rng = np.random.default_rng(14)
    s, rhs = random_system(rng, 3, 1, world_frac=1.0)
    x, a, st = gpu_solve(ctx, s, rhs, 0.1, capi.GAUSS_SEIDEL, 5)
    xf, af, _, _ = orc.fast_iterate(s, rhs, 0.1, orc.GAUSS_SEIDEL, max_iters=5, tol=0.0)
    assert same_bits(x, xf) and same_bits(a, af)


def test_invalid_arguments_do_not_crash(ctx):
    """Precondition failures return EGS_ERR_INVALID (the reference Panics)."""
    with pytest.raises(capi.EgsError) as e:
        capi.Problem(ctx, 2, [0, 3], [1, 0])           # body index out of range
    assert e.value.status == capi.ERR_INVALID
    with pytest.raises(capi.EgsError) as e:
        capi.Problem(ctx, 2, [1], [1])                 # same body on both sides
    assert e.value.status == capi.ERR_INVALID
    big0 = np.arange(70000, dtype=np.int32) % 1000     # past the 4-lane schedule's size: the
    big1 = np.full(70000, -1, np.int32)                # 1-lane schedule is built lazily
    big0[69999] = 1000
    with pytest.raises(capi.EgsError) as e:
        capi.Problem(ctx, 1000, big0, big1)
    assert e.value.status == capi.ERR_INVALID
    pr = capi.Problem(ctx, 2, [0], [1])
    with pytest.raises(capi.EgsError) as e:
        pr.solve(capi.params())                        # nothing uploaded
    assert e.value.status == capi.ERR_INVALID
    s, rhs = random_system(np.random.default_rng(1), 2, 1, world_frac=0.0)
    pr.set_blocks(s.Minv, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs)
    with pytest.raises(capi.EgsError) as e:
        pr.solve(capi.params(method=capi.SOR, omega=2.5))
    assert e.value.status == capi.ERR_INVALID
    with pytest.raises(capi.EgsError) as e:
        pr.solve(capi.params(method=9))
    assert e.value.status == capi.ERR_INVALID
    pr.close()


@pytest.mark.parametrize("method", [capi.GAUSS_SEIDEL, capi.SOR])
def test_quad_and_single_lane_schedules_agree(ctx, method, monkeypatch):
    """The latency-optimised 4-lanes-per-constraint kernel (EGS_QUAD=1) and the
    1-lane tile kernel (EGS_QUAD=0) give the oracle's bits on the same inputs:
    C2 pile, Chain(8), random graphs with islands <= 64 constraints, fp32."""
    rng = np.random.default_rng(40)
    cases = [system_from_scene(scenes.box_stack(8, 8, 4))[0], system_from_scene(scenes.chain(8))[0],
             system_from_scene(scenes.concat([scenes.chain(int(k)) for k in rng.integers(1, 60, 12)]))[0]]
    for s in cases:
        rhs = rng.uniform(-1, 1, 3 * s.m)
        for K in (0, 1, 9, 50):
            xf, af, _, rf = orc.fast_iterate(s, rhs, 0.02, method, max_iters=K, tol=0.0)
            for quad in ("0", "1"):
                monkeypatch.setenv("EGS_QUAD", quad)
                x, a, st = gpu_solve(ctx, s, rhs, 0.02, method, K)
                assert st.status == capi.OK
                assert same_bits(x, xf) and same_bits(a, af), (quad, K)
                assert abs(st.residual - rf) <= 1e-12 * max(1.0, rf)
        monkeypatch.setenv("EGS_QUAD", "1")
        x, a, st = gpu_solve(ctx, s, rhs, 0.1, method, 500, tol=1e-9)
        xf, af, it, rf = orc.fast_iterate(s, rhs, 0.1, method, max_iters=500, tol=1e-9)
        assert st.iterations == it and same_bits(x, xf)
        x32, a32, st = gpu_solve(ctx, s, rhs, 0.02, method, 20, precision=capi.F32)
        xo, ao, _, _ = orc.fast_iterate_f32(s, rhs, 0.02, method, max_iters=20)
        assert same_bits(x32.astype(np.float32), xo) and same_bits(a32.astype(np.float32), ao)


@pytest.mark.parametrize("method", [capi.GAUSS_SEIDEL, capi.SOR])
def test_quad_schedule_with_256_constraint_tiles(ctx, method):
    """Islands of 65..256 constraints still run 4 lanes per constraint, in
    1024-thread tiles; same bits as the oracle (fp64 and fp32)."""
    if os.environ.get("EGS_QUAD") == "0":
        pytest.skip("the 4-lane schedule is switched off (tests/tools/env_matrix.sh)")
    rng = np.random.default_rng(41)
    cases = [system_from_scene(scenes.concat([scenes.chain(int(k)) for k in (65, 200, 256, 3, 130, 90, 17)]))[0],
             random_system(rng, 40, 230, world_frac=0.1)[0],
             system_from_scene(scenes.concat([scenes.chain(int(k)) for k in (65, 128, 100, 7, 128)]))[0]]   # 128-constraint tiles
    for s in cases:
        rhs = rng.uniform(-1, 1, 3 * s.m)
        for K in (1, 7, 60):
            x, a, st = gpu_solve(ctx, s, rhs, 0.02, method, K)
            assert st.status == capi.OK and st.reserved == 1 and st.n_global == 0
            xf, af, _, rf = orc.fast_iterate(s, rhs, 0.02, method, max_iters=K, tol=0.0)
            assert same_bits(x, xf) and same_bits(a, af), K
            assert abs(st.residual - rf) <= 1e-12 * max(1.0, rf)
        assert st.n_tiles < (s.m + 63) // 64          # 256-constraint tiles, not 64
        x, a, st = gpu_solve(ctx, s, rhs, 0.1, method, 500, tol=1e-9)
        xf, af, it, rf = orc.fast_iterate(s, rhs, 0.1, method, max_iters=500, tol=1e-9)
        assert st.iterations == it and same_bits(x, xf)
        x32, a32, st = gpu_solve(ctx, s, rhs, 0.02, method, 20, precision=capi.F32)
        xo, ao, _, _ = orc.fast_iterate_f32(s, rhs, 0.02, method, max_iters=20)
        assert same_bits(x32.astype(np.float32), xo) and same_bits(a32.astype(np.float32), ao)


@pytest.mark.parametrize("method", [capi.GAUSS_SEIDEL, capi.SOR, capi.JACOBI])
def test_islands_up_to_512_constraints_stay_in_one_workgroup(ctx, method):
    """An island of 257..512 constraints runs in ONE 512-thread tile (all
    hand-offs in LDS) instead of the cross-workgroup patch path."""
    if os.environ.get("EGS_TILE", "512") != "512":
        pytest.skip("a smaller tile size is forced (tests/tools/env_matrix.sh)")
    rng = np.random.default_rng(43)
    s, rhs = random_system(rng, 70, 430, world_frac=0.1, connected=True)
    for K in (1, 8, 40):
        x, a, st = gpu_solve(ctx, s, rhs, 0.02, method, K)
        assert st.status == capi.OK and st.n_global == 0 and st.n_tiles == 1
        xf, af, _, rf = orc.fast_iterate(s, rhs, 0.02, method, max_iters=K, tol=0.0)
        assert same_bits(x, xf) and same_bits(a, af), K


@pytest.mark.parametrize("method", [capi.GAUSS_SEIDEL, capi.SOR, capi.JACOBI])
def test_isotropic_fast_path_has_the_same_bits(ctx, method, monkeypatch):
    """Bodies whose M^-1 blocks are exactly diag(a,a,a,b,b,b) (every BASELINE pile) take
    the tile kernel that forms B = M^-1 J^T on the fly; EGS_ISO=0 keeps B in registers;
    one stray 1e-18 off the diagonal falls back by itself.  All three: the oracle's bits."""
    monkeypatch.setenv("EGS_QUAD", "0")
    rng = np.random.default_rng(44)
    s = system_from_scene(scenes.box_stack(6, 5, 4, jitter=1e-3, seed=3))[0]
    rhs = rng.uniform(-1, 1, 3 * s.m)
    for prec in (capi.F64, capi.F32):
        if prec == capi.F32:
            xf, af = orc.fast_iterate_f32(s, rhs, 0.02, method, max_iters=30)[:2]
        else:
            xf, af = orc.fast_iterate(s, rhs, 0.02, method, max_iters=30, tol=0.0)[:2]
        for iso in ("2", "0"):           # 2 forces the variant (by default it is chosen for large batches only)
            monkeypatch.setenv("EGS_ISO", iso)
            x, a, st = gpu_solve(ctx, s, rhs, 0.02, method, 30, precision=prec)
            if prec == capi.F32:
                x, a = x.astype(np.float32), a.astype(np.float32)
            assert st.status == capi.OK and same_bits(x, xf) and same_bits(a, af), (prec, iso)
    monkeypatch.setenv("EGS_ISO", "2")
    Minv = s.Minv.copy()
    Minv[7, 1] = Minv[7, 6] = 1e-18                      # body 7 is no longer exactly isotropic
    s2 = orc.Sys(Minv, s.body0, s.body1, s.J0, s.J1, s.is_eq, s.lo, s.hi)
    x, a, st = gpu_solve(ctx, s2, rhs, 0.02, method, 30)
    xf, af, _, _ = orc.fast_iterate(s2, rhs, 0.02, method, max_iters=30, tol=0.0)
    assert same_bits(x, xf) and same_bits(a, af)


@pytest.mark.parametrize("method", [capi.GAUSS_SEIDEL, capi.SOR, capi.JACOBI])
def test_tolerance_terminated_runs_stop_where_the_reference_stops(ctx, method, monkeypatch):
    """tol > 0 (the reference's default loop: one sweep, one residual, stop at the first
    err <= tol): sweeps run in recorded chunks on the device; sweep count, lambda,
    accumulators and the residual equal the oracle's sweep-by-sweep loop -- for the 4-lane
    and the 1-lane kernels, fp64 and fp32, check_every > 1, and runs that hit max_iters."""
    rng = np.random.default_rng(45)
    cases = [system_from_scene(scenes.chain(12))[0], system_from_scene(scenes.box_stack(3, 3, 4))[0],
             random_system(rng, 30, 120, world_frac=0.1)[0]]
    for quad in ("1", "0"):
        monkeypatch.setenv("EGS_QUAD", quad)
        for s in cases:
            rhs = rng.uniform(-1, 1, 3 * s.m)
            for cfm, tol, max_iters, every in ((0.5, 1e-9, 500, 1), (0.5, 1e-6, 500, 7), (0.05, 1e-9, 70, 1), (0.5, 1e-9, 65, 1)):
                x, a, st = gpu_solve(ctx, s, rhs, cfm, method, max_iters, tol=tol, check_every=every)
                xf, af, it, rf = orc.fast_iterate(s, rhs, cfm, method, max_iters=max_iters, tol=tol, check_every=every)
                assert st.iterations == it, (quad, cfm, tol, max_iters, every, st.iterations, it)
                assert same_bits(x, xf) and same_bits(a, af)
                # Jacobi may diverge: inf == inf, or NaN on both sides (which also ends the reference's loop)
                assert (np.isnan(st.residual) and np.isnan(rf)) or st.residual == rf or abs(st.residual - rf) <= 1e-12 * max(1.0, rf)
    monkeypatch.setenv("EGS_QUAD", "1")
    s = cases[1]
    rhs = rng.uniform(-1, 1, 3 * s.m)
    x32, a32, st = gpu_solve(ctx, s, rhs, 0.5, method, 300, tol=1e-4, precision=capi.F32)
    xo, ao, it, _ = orc.fast_iterate_f32(s, rhs, 0.5, method, max_iters=300, tol=1e-4)
    assert st.iterations == it and same_bits(x32.astype(np.float32), xo) and same_bits(a32.astype(np.float32), ao)


def test_one_shot_entry_is_stateless_although_it_reuses_its_schedule(ctx):
    """egs_solve_blocks keeps the last schedule and device buffers for the next call with the
    same constraint graph; results never depend on what was solved before."""
    rng = np.random.default_rng(46)
    sA, rhsA = random_system(rng, 12, 40, world_frac=0.2)
    sB, rhsB = random_system(rng, 9, 25, world_frac=0.2)
    prm = capi.params(method=capi.SOR, max_iters=30, tol=0.0, cfm=0.05)

    def one_shot(s, rhs):
        return ctx.solve_blocks(s.Minv, s.body0, s.body1, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs, prm)[0]

    ref = {}
    for key, (s, rhs) in {"A": (sA, rhsA), "B": (sB, rhsB), "A2": (sA, -0.5 * rhsA)}.items():
        ref[key] = gpu_solve(ctx, s, rhs, 0.05, capi.SOR, 30)[0]
    assert same_bits(one_shot(sA, rhsA), ref["A"])
    assert same_bits(one_shot(sA, -0.5 * rhsA), ref["A2"])      # same graph: schedule reused
    assert same_bits(one_shot(sB, rhsB), ref["B"])              # other graph: rebuilt
    assert same_bits(one_shot(sA, rhsA), ref["A"])


def test_get_stats_after_the_fact(ctx):
    """egs_problem_get_stats reports the residual of the last solve whether or not the solve
    call itself asked for statistics (fixed sweeps and tolerance-terminated)."""
    rng = np.random.default_rng(47)
    s, rhs = random_system(rng, 20, 60, world_frac=0.2)
    for tol, K in ((0.0, 15), (1e-9, 500)):
        pr = capi.Problem(ctx, s.n, s.body0, s.body1)
        pr.set_blocks(s.Minv, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs)
        pr.solve(capi.params(method=capi.SOR, max_iters=K, tol=tol, cfm=0.3), want_stats=False)
        st = pr.stats()
        pr.close()
        _, _, it, rf = orc.fast_iterate(s, rhs, 0.3, capi.SOR, max_iters=K, tol=tol)
        assert st.iterations == it and abs(st.residual - rf) <= 1e-12 * max(1.0, rf), (tol, st.residual, rf)
