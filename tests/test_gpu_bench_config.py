"""GPU (-m gpu): parity AT THE BENCH CONFIGURATION.  bench.py's headline is 24 C3 piles in one
problem with the default switches, which runs the static-timetable kernel's isotropic-body variant
(step_solve_kernel, 256-constraint tiles, three per CU); 6 piles run the regular variant.  Every pile of the batch
must have the bits of its own sequential list-order solve (oracle fast O(nnz) port), and the
kernel's epilogue w = A lambda - rhs must equal the literal-product residual element by element."""
import os

import numpy as np
import pytest

import bench
from eggshell_amd import capi, scenes
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("batch", [6, 24])
def test_bench_config_parity(ctx, batch):
    nx, ny, nz, sweeps, prec, dt = bench.WORKLOADS["c3"]
    seeds = [b + 1 for b in range(batch)]                       # bench.py rank 0
    piles = [scenes.box_stack(nx, ny, nz, jitter=1e-3, seed=sd, origin=(0.0, 100.0 * k)) for k, sd in enumerate(seeds)]
    sc = scenes.concat(piles)
    pr, _ = bench.build_problem(ctx, sc, capi.F64)
    prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=sweeps, tol=0.0, cfm=0.01)
    pr.step(dt, 0.2, prm)
    st = pr.stats()
    assert st.status == capi.OK and st.n_global == 0 and st.n_islands == 256 * batch
    default_switches = not any(k in os.environ for k in ("EGS_QUAD", "EGS_ISO", "EGS_TILE", "EGS_STEP"))   # tests/tools/env_matrix.sh forces others
    if default_switches:
        assert st.reserved == 0 and not (st.schedule & capi.SCHED_QUAD)
        if batch == 24:    # the kernel bench.py's `value` is measured on
            assert st.schedule & capi.SCHED_ISO and st.tile_constraints == 256 and st.n_tiles == 1536
            assert st.schedule & capi.SCHED_STATIC       # regular columns: the static timetable (step_solve.hip)
    lam, wres, acc = pr.lambda_(), pr.wres(), pr.accumulators()
    J0, J1, is_eq, lo, hi, rhs, err = pr.blocks()
    Minv, f_ext = bench.host_mass_and_force(sc)
    m1, n1 = piles[0]["kind"].shape[0], piles[0]["p"].shape[0]
    for k in range(batch):
        cons, rows, bod = slice(k * m1, (k + 1) * m1), slice(3 * k * m1, 3 * (k + 1) * m1), slice(k * n1, (k + 1) * n1)
        s = orc.Sys(Minv[bod], np.where(sc["body0"][cons] >= 0, sc["body0"][cons] - k * n1, -1), sc["body1"][cons] - k * n1,
                    J0[cons], J1[cons], is_eq[rows], lo[rows], hi[rows])
        xf, af, _, _ = orc.fast_iterate(s, rhs[rows], 0.01, orc.GAUSS_SEIDEL, max_iters=sweeps, tol=0.0)
        assert np.array_equal(lam[rows], xf), "pile %d" % k
        assert np.array_equal(acc[bod], af)
        # element-wise w = A lambda - rhs: the fast oracle's own epilogue expression, bit for bit
        assert np.array_equal(wres[rows], orc.fast_wres(s, rhs[rows], 0.01, xf, af))
    # (the LITERAL O(m^2) product would take minutes per pile at this size: its leg is
    # test_wres_elementwise_small below and the stand-alone product in tests/test_gpu_matvec.py)
    pr.close()


def test_wres_elementwise_small(ctx):
    """w = A lambda - rhs element by element against the literal CalculateSparseJMJtX (1e-9) and the
    fast oracle's own epilogue expression (bit-exact), all three methods."""
    sc = scenes.box_stack(3, 3, 4, jitter=1e-3, seed=9)
    from helpers import system_from_scene
    s, _ = system_from_scene(sc)
    rhs = np.random.default_rng(3).uniform(-1, 1, 3 * s.m)
    for method in (capi.JACOBI, capi.GAUSS_SEIDEL, capi.SOR):
        pr = capi.Problem(ctx, s.n, s.body0, s.body1)
        pr.set_blocks(s.Minv, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs)
        pr.solve(capi.params(method=method, max_iters=25, tol=0.0, cfm=0.02))
        lam, w = pr.lambda_(), pr.wres()
        xf, af, _, _ = orc.fast_iterate(s, rhs, 0.02, method, max_iters=25, tol=0.0)
        assert np.array_equal(lam, xf) and np.array_equal(w, orc.fast_wres(s, rhs, 0.02, xf, af))
        scale = max(1.0, np.abs(w).max())     # 25 Jacobi sweeps on an unbounded normal row grow large
        assert np.linalg.norm(w - (orc.lit_JMJtX(s, lam, 0.02) - rhs)) < 1e-9 * scale
        # the solve kernels accumulate a_b = sum B dx over the sweeps, the product sums J^T lambda afresh:
        # same mathematics, different rounding
        assert np.abs(w - (orc.fast_matvec(s, lam, orc.MV_FULL, 0.02) - rhs)).max() < 1e-11 * scale
        pr.close()
