"""GPU (-m gpu): device-side assembly (K1-K4), the whole step, and
size-independent properties at BASELINE's full sizes."""
import numpy as np
import pytest

from eggshell_amd import capi, scenes
from helpers import dense_numpy, system_from_scene
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def make_problem(ctx, sc, precision=capi.F64):
    Minv = orc.minv_blocks(sc["R"], sc["mass"], sc["I_body"])
    f_ext = orc.external_force(sc["R"], sc["w"], sc["mass"], sc["I_body"])
    pr = capi.Problem(ctx, sc["p"].shape[0], sc["body0"], sc["body1"], precision)
    pr.set_state(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext)
    pr.set_constraints(sc["kind"], sc["data"])
    return pr, Minv, f_ext


def random_contacts(rng, n, m):
    """Bodies with random poses/velocities; contacts with random unit normals
    (incl. the antiparallel -z case) mixed with ball joints."""
    from scipy.spatial.transform import Rotation
    p = rng.uniform(-1, 1, (n, 3))
    R = Rotation.random(n, random_state=int(rng.integers(1 << 30))).as_matrix().reshape(n, 9)
    kind = rng.integers(0, 2, m).astype(np.int32)
    body0 = rng.integers(0, n, m).astype(np.int32)
    body1 = ((body0 + rng.integers(1, n, m)) % n).astype(np.int32)
    data = np.zeros((m, 7))
    for i in range(m):
        if kind[i] == 0:
            data[i, 0:6] = rng.uniform(-0.2, 0.2, 6)
            if rng.uniform() < 0.3:
                body1[i] = -1
        else:
            nrm = rng.normal(size=3)
            nrm /= np.linalg.norm(nrm)
            if i % 7 == 0:
                nrm = np.array([0.0, 0.0, -1.0])
            if i % 11 == 0:
                nrm = np.array([0.0, 0.0, 1.0])
            data[i, 0:3] = rng.uniform(-1, 1, 3)
            data[i, 3:6] = nrm
            data[i, 6] = rng.uniform(0, 0.01)
            if rng.uniform() < 0.3:
                body0[i] = -1
    return dict(p=p, R=R, v=rng.uniform(-1, 1, (n, 3)), w=rng.uniform(-1, 1, (n, 3)),
                mass=rng.uniform(0.5, 2, n), I_body=np.tile((np.eye(3) * 0.1).reshape(9), (n, 1)),
                kind=kind, body0=body0, body1=body1, data=data)


@pytest.mark.parametrize("case", ["chain8", "stack", "random"])
def test_assembly_bit_exact(ctx, case):
    """J blocks, error, bounds, row types and the ODE rhs equal the oracle's
    (joints.cc:3-35, contact.cc:14-117, ensembles.cc:569-570) bit for bit."""
    rng = np.random.default_rng(20)
    sc = {"chain8": lambda: scenes.chain(8), "stack": lambda: scenes.box_stack(3, 3, 3, jitter=1e-3),
          "random": lambda: random_contacts(rng, 30, 200)}[case]()
    if case != "random":
        sc["v"] = rng.uniform(-1, 1, sc["v"].shape); sc["w"] = rng.uniform(-1, 1, sc["w"].shape)
    dt = 1e-3 if case == "chain8" else 5e-3
    pr, Minv, f_ext = make_problem(ctx, sc)
    pr.assemble(dt, 0.2)
    J0, J1, is_eq, lo, hi, rhs, err = pr.blocks()
    oJ0, oJ1, ois_eq, olo, ohi, oerr = orc.assemble(sc["p"], sc["R"], sc["kind"], sc["body0"], sc["body1"], sc["data"])
    orhs = orc.ode_rhs(sc["v"], sc["w"], Minv, f_ext, sc["body0"], sc["body1"], oJ0, oJ1, oerr, dt, 0.2)
    assert np.array_equal(J0, oJ0) and np.array_equal(J1, oJ1)
    assert np.array_equal(is_eq, ois_eq) and np.array_equal(lo, olo) and np.array_equal(hi, ohi)
    assert np.array_equal(err, oerr)
    assert np.array_equal(rhs, orhs)
    pr.close()


@pytest.mark.parametrize("method", [capi.GAUSS_SEIDEL, capi.SOR])
def test_step_chain_and_stack(ctx, method):
    """assemble + solve + velocity update == oracle pipeline: lambda bit-exact,
    v_new = v + dt M^-1 (f + J^T lambda) to 1e-12 relative (the device adds the
    solver's accumulators instead of re-multiplying J^T lambda)."""
    for sc, dt, K in ((scenes.chain(8), 1e-3, 200), (scenes.box_stack(4, 4, 4), 5e-3, 50)):
        pr, Minv, f_ext = make_problem(ctx, sc)
        prm = capi.params(method=method, max_iters=K, tol=0.0, cfm=0.01)
        st = pr.step(dt, 0.2, prm, want_stats=True)
        lam, v6 = pr.lambda_(), pr.velocity()
        s, err = system_from_scene(sc)
        rhs = orc.ode_rhs(sc["v"], sc["w"], Minv, f_ext, s.body0, s.body1, s.J0, s.J1, err, dt, 0.2)
        xf, af, _, rf = orc.fast_iterate(s, rhs, 0.01, method, max_iters=K, tol=0.0)
        assert np.array_equal(lam, xf)
        v6o = orc.velocity_update(sc["v"], sc["w"], Minv, f_ext, s.body0, s.body1, s.J0, s.J1, xf, dt)
        assert np.abs(v6 - v6o).max() <= 1e-12 * max(1.0, np.abs(v6o).max())
        pr.close()


def test_c3_full_size_properties(ctx):
    """BASELINE C3 (4096 bodies, 16384 contacts, 100 sweeps): properties that do
    not need the oracle at full size, plus one oracle run (fast, ~0.1 s)."""
    sc = scenes.box_stack(16, 16, 16)
    assert sc["kind"].shape[0] == 16384
    pr, Minv, f_ext = make_problem(ctx, sc)
    prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=100, tol=0.0, cfm=0.01)
    st = pr.step(5e-3, 0.2, prm, want_stats=True)
    lam, acc = pr.lambda_(), pr.accumulators()
    J0, J1, is_eq, lo, hi, rhs, err = pr.blocks()
    assert st.status == capi.OK and st.n_islands == 256 and st.n_global == 0
    # bounds: friction box, normal impulse >= 0 (contact.cc:109-112)
    assert (lam >= lo).all() and (lam <= hi).all()
    assert (lam.reshape(-1, 3)[:, 2] >= 0).all()
    # accumulators are W J^T lambda (recomputed with numpy in fp64)
    g = np.zeros((4096, 6))
    L = lam.reshape(-1, 3)
    for side, J in ((sc["body0"], J0), (sc["body1"], J1)):
        ok = side >= 0
        np.add.at(g, side[ok], np.einsum("irc,ir->ic", J.reshape(-1, 3, 6)[ok], L[ok]))
    a_ref = np.einsum("brc,bc->br", Minv.reshape(-1, 6, 6), g)
    assert np.abs(acc - a_ref).max() <= 1e-9 * max(1.0, np.abs(a_ref).max())
    # determinism / idempotence: a second identical step gives identical bits
    pr.step(5e-3, 0.2, prm)
    assert np.array_equal(pr.lambda_(), lam)
    # the oracle at full size, bit for bit
    s = orc.Sys(Minv, sc["body0"], sc["body1"], J0, J1, is_eq, lo, hi)
    xf, af, _, rf = orc.fast_iterate(s, rhs, 0.01, orc.GAUSS_SEIDEL, max_iters=100, tol=0.0)
    assert np.array_equal(lam, xf) and np.array_equal(acc, af)
    assert abs(st.residual - rf) <= 1e-12 * max(1.0, rf)
    pr.close()


def test_batch_invariance(ctx):
    """Islands are independent: a pile solved inside a batch of piles gives the
    same bits as the pile solved alone (C4-style batching)."""
    rng = np.random.default_rng(30)
    piles = [scenes.box_stack(4, 4, 4, jitter=1e-3, seed=k, origin=(0.0, 10.0 * k)) for k in range(6)]
    prm = capi.params(method=capi.SOR, max_iters=50, tol=0.0, cfm=0.01)
    singles = []
    for sc in piles:
        pr, _, _ = make_problem(ctx, sc)
        pr.step(5e-3, 0.2, prm)
        singles.append(pr.lambda_())
        pr.close()
    pr, _, _ = make_problem(ctx, scenes.concat(piles))
    pr.step(5e-3, 0.2, prm)
    lam = pr.lambda_()
    pr.close()
    assert np.array_equal(lam, np.concatenate(singles))


def test_converged_c2_satisfies_reference_check(ctx):
    """C2 pile with cfm 0.1 run to the reference's tolerance: the residual
    metric of sparse_iterations.cc:51-69 evaluated by the LITERAL oracle on the
    GPU's lambda is <= 1e-9 (CheckMixedConstraintSolutions semantics)."""
    rng = np.random.default_rng(31)
    sc = scenes.box_stack(8, 8, 4)
    s, _ = system_from_scene(sc)
    rhs = rng.uniform(-1, 1, 3 * s.m)
    x, st = ctx.solve_blocks(s.Minv, s.body0, s.body1, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs,
                             capi.params(method=capi.SOR, max_iters=500, tol=1e-9, cfm=0.1, check_every=1))
    assert st.iterations < 500 and st.residual <= 1e-9
    # one column (island) of the pile through the O(m^2) literal residual
    col = np.nonzero(((s.body0 < 0) | (s.body0 % 64 == 0)) & (s.body1 % 64 == 0))[0]
    assert col.shape[0] == 16
    sub = orc.Sys(s.Minv, s.body0[col], s.body1[col], s.J0[col], s.J1[col],
                  s.is_eq.reshape(-1, 3)[col].reshape(-1), s.lo.reshape(-1, 3)[col].reshape(-1),
                  s.hi.reshape(-1, 3)[col].reshape(-1))
    r = orc.lit_residual(sub, rhs.reshape(-1, 3)[col].reshape(-1), x.reshape(-1, 3)[col].reshape(-1), 0.1)
    assert r <= 1e-9


def test_device_resident_chain_trajectory(ctx):
    """Chain(8), 20 steps entirely on the device (assemble + SOR to 1e-9 +
    velocity + StepPositions_ODE, ensembles.cc:390-427 with the sparse switch):
    state never leaves the GPU; compared after every 5 steps with the
    reference's live dense path (joints only, so both reference solvers agree)
    restated from oracle pieces."""
    from helpers import ode_step
    sc = scenes.chain(8)
    ref = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in sc.items()}
    pr, Minv, f_ext = make_problem(ctx, sc)
    prm = capi.params(method=capi.SOR, max_iters=5000, tol=1e-11, cfm=0.0)
    for step in range(1, 21):
        pr.step(1e-3, 0.2, prm)
        pr.advance(1e-3)
        ode_step(ref, 1e-3)
        if step % 5 == 0:
            pos, R, v, w = pr.state()
            assert np.abs(pos - ref["p"]).max() < 1e-9
            assert np.abs(R - ref["R"]).max() < 1e-9
            assert np.abs(v - ref["v"]).max() < 1e-6 and np.abs(w - ref["w"]).max() < 1e-6
    assert np.abs(pos[:, 2] - 2.0).max() > 1e-5      # the chain did swing
    pr.close()


def test_advance_matches_oracle_position_update(ctx):
    """One StepPositions_ODE with random velocities: p exact to rounding, R to
    1e-15 (device sin/cos vs libm)."""
    rng = np.random.default_rng(50)
    sc = scenes.box_stack(3, 3, 2)
    sc["v"] = rng.uniform(-1, 1, sc["v"].shape); sc["w"] = rng.uniform(-3, 3, sc["w"].shape)
    sc["w"][0] = 0.0                                   # zero angular velocity: identity rotation
    pr, Minv, f_ext = make_problem(ctx, sc)
    prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=10, tol=0.0, cfm=0.01)
    pr.step(5e-3, 0.2, prm)
    v6 = pr.velocity()
    pr.advance(5e-3)
    pos, R, v, w = pr.state()
    v6_old = np.concatenate([sc["v"], sc["w"]], axis=1)
    po, Ro = orc.position_update(sc["p"], sc["R"], v6_old, v6, 5e-3)
    assert np.abs(pos - po).max() <= 1e-15 and np.abs(R - Ro).max() <= 1e-15
    assert np.array_equal(v, v6[:, :3]) and np.array_equal(w, v6[:, 3:])
    pr.close()


def test_compact_mass_entry_equals_full_blocks(ctx):
    """egs_problem_set_mass (1/m + 3x3 inverse inertia per body, all that
    ConstructMassInertiaMatrixInverse stores) gives the step of the 6x6 blocks."""
    sc = scenes.chain(8)
    pr, Minv, f_ext = make_problem(ctx, sc)
    prm = capi.params(method=capi.SOR, max_iters=60, tol=0.0, cfm=0.0)
    pr.step(1e-3, 0.2, prm)
    v_full, lam_full = pr.velocity(), pr.lambda_()
    pr.close()
    blocks = Minv.reshape(-1, 6, 6)
    pr = capi.Problem(ctx, sc["p"].shape[0], sc["body0"], sc["body1"])
    pr.set_state(sc["p"], sc["R"], sc["v"], sc["w"], None, f_ext)
    pr.set_mass(blocks[:, 0, 0].copy(), blocks[:, 3:, 3:].reshape(-1, 9).copy())
    pr.set_constraints(sc["kind"], sc["data"])
    pr.step(1e-3, 0.2, prm)
    assert np.array_equal(pr.velocity(), v_full) and np.array_equal(pr.lambda_(), lam_full)
    pr.close()
