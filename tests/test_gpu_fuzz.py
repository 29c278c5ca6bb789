"""GPU (-m gpu): randomised sweep over topologies, sizes, methods, precisions and schedules;
every case must give the oracle's bits.  Catches ordering bugs that the hand-picked scenes
miss (waits between wavefronts, tile boundaries, world-only constraints, bodies with many
constraints)."""
import numpy as np
import pytest

from eggshell_amd import capi
from helpers import grouped_system, random_system
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def run(ctx, s, rhs, method, K, cfm, precision, dirty=False):
    pr = capi.Problem(ctx, s.n, s.body0, s.body1, precision)
    if dirty:   # an earlier solve on the same object must leave nothing behind
        pr.set_blocks(s.Minv, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs[::-1].copy())
        pr.solve(capi.params(method=method, max_iters=3, tol=0.0, cfm=max(cfm, 0.01)))
    pr.set_blocks(s.Minv, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs)
    st = pr.solve(capi.params(method=method, max_iters=K, tol=0.0, cfm=cfm))
    x, a = pr.lambda_(), pr.accumulators()
    pr.close()
    return x, a, st


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("EGS_FUZZ_SEEDS", "6"))))
def test_random_systems_all_schedules(ctx, seed, monkeypatch):
    rng = np.random.default_rng(1000 + seed)
    for case in range(20):
        n = int(rng.integers(2, 160))
        m = int(rng.integers(1, 1500))
        s, rhs = random_system(rng, n, m, world_frac=float(rng.uniform(0, 0.5)), eq_frac=float(rng.uniform(0, 1)),
                               connected=bool(rng.integers(0, 2)))
        if case % 4 == 3:      # isotropic mass blocks: the tile kernel that forms B on the fly
            w = np.zeros((n, 6, 6))
            for b in range(n):
                a_, b_ = rng.uniform(0.2, 3.0, 2)
                w[b] = np.diag([a_, a_, a_, b_, b_, b_])
            s = orc.Sys(w.reshape(n, 36), s.body0, s.body1, s.J0, s.J1, s.is_eq, s.lo, s.hi)
        if case % 5 == 2 and m <= 400:     # groups of four constraints on the same two bodies: the 4-lane plan's runs
            s, rhs = grouped_system(rng, s, rhs, 4 if case % 10 == 2 else "ragged")    # ragged: chunks with placeholders
            m = s.body0.shape[0]
            monkeypatch.setenv("EGS_RUNS", "2")      # chunks wherever the padding allows, whatever the period test says
        else:
            monkeypatch.setenv("EGS_RUNS", "1")
        monkeypatch.setenv("EGS_ISO", "2" if case % 4 == 3 else "1")
        method = int(rng.choice([capi.JACOBI, capi.GAUSS_SEIDEL, capi.SOR]))
        K = int(rng.integers(0, 25))
        cfm = float(rng.choice([0.0, 0.01, 0.3]))
        xf, af, _, _ = orc.fast_iterate(s, rhs, cfm, method, max_iters=K, tol=0.0)
        xo, ao, _, _ = orc.fast_iterate_f32(s, rhs, cfm, method, max_iters=K)
        # 4-lane tiles on tickets / on the static timetable; 1-lane tiles on tickets / on the timetable (step_solve.hip), with either oversize path
        for quad, patch, qpatch, step in (("1", "1", "1", "0"), ("1", "1", "1", "1"), ("0", "1", "0", "0"), ("0", "1", "0", "1"), ("0", "0", "1", "1")):
            monkeypatch.setenv("EGS_STEP", step)
            monkeypatch.setenv("EGS_QUAD", quad)
            monkeypatch.setenv("EGS_PATCH", patch)
            monkeypatch.setenv("EGS_QUAD_PATCH", qpatch)
            x, a, st = run(ctx, s, rhs, method, K, cfm, capi.F64, dirty=case % 2 == 1)
            assert st.status == capi.OK
            assert np.array_equal(x, xf, equal_nan=True) and np.array_equal(a, af, equal_nan=True), (seed, case, n, m, method, K, quad, patch, step)
        # the library's OWN choices: every schedule switch unset, so that use_static_timetable(), the occupancy gate of the
        # 4-lane schedule, the isotropic cost model and the runs heuristic are fuzzed too (the legs above force each path)
        for var in ("EGS_STEP", "EGS_QUAD", "EGS_PATCH", "EGS_QUAD_PATCH", "EGS_RUNS", "EGS_ISO"):
            monkeypatch.delenv(var, raising=False)
        x, a, st = run(ctx, s, rhs, method, K, cfm, capi.F64, dirty=case % 2 == 0)
        assert st.status == capi.OK
        assert np.array_equal(x, xf, equal_nan=True) and np.array_equal(a, af, equal_nan=True), (seed, case, n, m, method, K, "defaults", st.schedule)
        monkeypatch.setenv("EGS_RUNS", "1"); monkeypatch.setenv("EGS_ISO", "2" if case % 4 == 3 else "1")
        if case % 4 == 3 and method != capi.JACOBI:
            # isotropic bodies: the 128-VGPR timetable kernel (lean_solve.hip), which needs J1_lin = -J0_lin -- true for the
            # reference's constraint kinds, so give the random system that shape (and check that the switch is refused otherwise)
            both = (s.body0 >= 0) & (s.body1 >= 0)
            J0a = s.J0.copy().reshape(-1, 3, 6)
            J0a[both, :, :3] = -s.J1.reshape(-1, 3, 6)[both, :, :3]
            sa = orc.Sys(s.Minv, s.body0, s.body1, J0a.reshape(s.J0.shape), s.J1, s.is_eq, s.lo, s.hi)
            xa, aa, _, _ = orc.fast_iterate(sa, rhs, cfm, method, max_iters=K, tol=0.0)
            monkeypatch.setenv("EGS_LEAN", "1"); monkeypatch.setenv("EGS_QUAD", "0"); monkeypatch.setenv("EGS_STEP", "1")
            for tile in ("256", "512"):
                monkeypatch.setenv("EGS_TILE", tile)
                x, a, st = run(ctx, sa, rhs, method, K, cfm, capi.F64)
                assert np.array_equal(x, xa, equal_nan=True) and np.array_equal(a, aa, equal_nan=True), (seed, case, "lean", tile, st.schedule)
                x, a, st2 = run(ctx, s, rhs, method, K, cfm, capi.F64)          # not antisymmetric: the switch must not apply
                assert not (st2.schedule & capi.SCHED_LEAN) or not both.any()
                assert np.array_equal(x, xf, equal_nan=True)
            for var in ("EGS_LEAN", "EGS_TILE", "EGS_STEP"):
                monkeypatch.delenv(var, raising=False)
        # the remaining legs alternate between the 4-lane kernel and the static timetable
        monkeypatch.setenv("EGS_QUAD", "1" if case % 2 == 0 else "0"); monkeypatch.setenv("EGS_PATCH", "1"); monkeypatch.setenv("EGS_QUAD_PATCH", "1")
        if case % 5 == 0 and cfm > 0:     # the reference's stopping loop (recorded chunks on the device)
            tol, cap, every = float(rng.choice([1e-3, 1e-7])), int(rng.integers(1, 150)), int(rng.choice([1, 1, 3]))
            pr = capi.Problem(ctx, s.n, s.body0, s.body1)
            pr.set_blocks(s.Minv, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs)
            st = pr.solve(capi.params(method=method, max_iters=cap, tol=tol, cfm=cfm, check_every=every))
            xt, at = pr.lambda_(), pr.accumulators()
            pr.close()
            xr, ar, it, rr = orc.fast_iterate(s, rhs, cfm, method, max_iters=cap, tol=tol, check_every=every)
            assert st.iterations == it, (seed, case, "tol", st.iterations, it)
            assert np.array_equal(xt, xr, equal_nan=True) and np.array_equal(at, ar, equal_nan=True), (seed, case, "tol")
        x, a, st = run(ctx, s, rhs, method, K, cfm, capi.F32, dirty=case % 2 == 0)
        assert np.array_equal(x.astype(np.float32), xo, equal_nan=True) and np.array_equal(a.astype(np.float32), ao, equal_nan=True), (seed, case, "f32")   # a diverging cfm = 0 run overflows to the same NaNs on both sides
