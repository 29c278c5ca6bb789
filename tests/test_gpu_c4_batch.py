"""GPU (-m gpu): BASELINE config C4 at full size -- 1024 independent 4x4x4
ensembles (64 bodies, 256 contacts each) batched in one launch, fp32, 50 sweeps."""
import numpy as np
import pytest

from eggshell_amd import capi, scenes
from helpers import system_from_scene
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def test_c4_full_batch_fp32(ctx):
    piles = [scenes.box_stack(4, 4, 4, jitter=1e-3, seed=k, origin=(0.0, 10.0 * k)) for k in range(1024)]
    sc = scenes.concat(piles)
    n, m = sc["p"].shape[0], sc["kind"].shape[0]
    assert (n, m) == (1024 * 64, 1024 * 256)
    Minv = np.zeros((n, 6, 6))
    Minv[:, [0, 1, 2], [0, 1, 2]] = 1.0
    Minv[:, [3, 4, 5], [3, 4, 5]] = 10.0          # I = 0.1 I3, R = identity
    f_ext = np.zeros((n, 6)); f_ext[:, 2] = -9.8
    pr = capi.Problem(ctx, n, sc["body0"], sc["body1"], capi.F32)
    pr.set_state(sc["p"], sc["R"], sc["v"], sc["w"], Minv.reshape(n, 36), f_ext)
    pr.set_constraints(sc["kind"], sc["data"])
    st = pr.step(5e-3, 0.2, capi.params(method=capi.GAUSS_SEIDEL, max_iters=50, tol=0.0, cfm=0.01), want_stats=True)
    lam = pr.lambda_()
    assert st.status == capi.OK and st.n_islands == 1024 * 16 and st.n_global == 0
    J0, J1, is_eq, lo, hi, rhs, err = pr.blocks()
    assert (lam >= lo).all() and (lam <= hi).all()
    # EVERY one of the 1 024 ensembles against the fp32 oracle, bit for bit (0.4 ms each on the host)
    Mi = Minv.reshape(n, 36)
    lam32 = lam.astype(np.float32)
    for k in range(1024):
        rows = slice(k * 256 * 3, (k + 1) * 256 * 3)
        cons = slice(k * 256, (k + 1) * 256)
        s = orc.Sys(Mi[k * 64:(k + 1) * 64], np.where(sc["body0"][cons] >= 0, sc["body0"][cons] - 64 * k, -1),
                    sc["body1"][cons] - 64 * k, J0[cons], J1[cons], is_eq[rows], lo[rows], hi[rows])
        xo, ao, _, _ = orc.fast_iterate_f32(s, rhs[rows], 0.01, orc.GAUSS_SEIDEL, max_iters=50)
        assert np.array_equal(lam32[rows], xo), k
        if k % 128 == 0 or k == 1023:
            # fp32 vs fp64 on the same inputs: stated tolerance 2e-3 relative
            x64, _, _, _ = orc.fast_iterate(s, rhs[rows], 0.01, orc.GAUSS_SEIDEL, max_iters=50, tol=0.0)
            assert np.abs(lam[rows] - x64).max() <= 2e-3 * max(1.0, np.abs(x64).max())
    pr.close()


def test_batched_creation_equals_separate_solves(ctx):
    """egs_problem_create_batch (ensemble-local indices + offset tables): every ensemble of
    the batch gets the bits of its own separate solve; bad local indices are rejected."""
    from helpers import system_from_scene
    rng = np.random.default_rng(7)
    ens = [scenes.box_stack(2, 2, 3, jitter=1e-3, seed=1), scenes.chain(9), scenes.box_stack(3, 2, 2), scenes.chain(1)]
    systems = [system_from_scene(e)[0] for e in ens]
    rhs = [rng.uniform(-1, 1, 3 * s.m) for s in systems]
    pr, boff, coff = capi.Problem.batch(ctx, [s.n for s in systems], [s.m for s in systems],
                                        np.concatenate([s.body0 for s in systems]),
                                        np.concatenate([s.body1 for s in systems]))
    assert boff.tolist() == np.concatenate([[0], np.cumsum([s.n for s in systems])]).tolist()
    assert coff.tolist() == np.concatenate([[0], np.cumsum([s.m for s in systems])]).tolist()
    cat = lambda name: np.concatenate([getattr(s, name) for s in systems])
    pr.set_blocks(cat("Minv"), cat("J0"), cat("J1"), cat("is_eq"), cat("lo"), cat("hi"), np.concatenate(rhs))
    prm = capi.params(method=capi.SOR, max_iters=40, tol=0.0, cfm=0.05)
    st = pr.solve(prm)
    x = pr.lambda_()
    pr.close()
    assert st.status == capi.OK
    for e, s in enumerate(systems):
        one = capi.Problem(ctx, s.n, s.body0, s.body1)
        one.set_blocks(s.Minv, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs[e])
        one.solve(prm)
        assert np.array_equal(one.lambda_(), x[3 * coff[e]:3 * coff[e + 1]])
        one.close()
    with pytest.raises(capi.EgsError) as err:
        capi.Problem.batch(ctx, [2, 2], [1, 1], [0, 2], [1, 0])      # local index 2 in a 2-body ensemble
    assert err.value.status == capi.ERR_INVALID
