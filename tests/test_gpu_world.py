"""GPU (-m gpu): egs_world -- Ensemble::Step (ensembles.cc:390-427) resident on
the device: collide -> (re-plan on topology change) -> assemble -> solve ->
velocity -> positions, against the same loop built from oracle pieces."""
import numpy as np
import pytest

from eggshell_amd import capi, scenes
from oracle import oracle as orc
from test_gpu_fullstep import oracle_step_loop

pytestmark = pytest.mark.gpu


def make_world(ctx, p, R, v=None, w=None, I=0.1):
    n = p.shape[0]
    mass = np.ones(n); I_body = np.tile((np.eye(3) * I).reshape(9), (n, 1))
    v = np.zeros((n, 3)) if v is None else v
    w = np.zeros((n, 3)) if w is None else w
    Minv = orc.minv_blocks(R, mass, I_body)
    f_ext = orc.external_force(R, w, mass, I_body)
    wd = capi.World(ctx, n)
    wd.set_bodies(p, R, v, w, Minv, f_ext)
    return wd


def test_drop_three_boxes_device_resident(ctx):
    p0 = np.array([[0.0, 0.0, 0.2 + 0.35 * k] for k in range(3)])
    R0 = np.tile(np.eye(3).reshape(9), (3, 1))
    wd = make_world(ctx, p0, R0)
    prm = capi.params(method=capi.SOR, max_iters=500, tol=1e-9, cfm=0.01)
    seen = 0
    for _ in range(120):
        wd.step(0.005, 0.2, prm)
        seen += wd.info()["n_contacts"]
    pos, R, v, w = wd.bodies()
    po, v6o, seen_o = oracle_step_loop(p0, R0, 120, 0.005)
    assert seen == seen_o
    assert np.abs(pos - po).max() < 1e-7
    assert np.abs(np.concatenate([v, w], axis=1) - v6o).max() < 1e-5
    assert wd.info()["replans"] < 60          # the plan is rebuilt only when the contact topology changes
    wd.close()


def test_pile_contacts_follow_the_moving_state(ctx):
    """A C2 pile stepped on the device: the contact list the world uses in a step
    is exactly what the oracle's collision finds for the body state at the
    start of that step (order, indices and all bits)."""
    from test_gpu_collide import reference_contacts
    sc = scenes.box_stack(4, 4, 3)
    wd = make_world(ctx, sc["p"], sc["R"])
    prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=50, tol=0.0, cfm=0.01)
    wd.step(0.005, 0.2, prm)
    b0, b1, data = wd.contacts()
    assert np.array_equal(b0, sc["body0"]) and np.array_equal(b1, sc["body1"]) and np.array_equal(data, sc["data"])
    for _ in range(6):
        wd.step(0.005, 0.2, prm)
    pos, R, v, w = wd.bodies()
    wd.step(0.005, 0.2, prm)
    b0, b1, data = wd.contacts()
    r0, r1, rd = reference_contacts(pos, R)
    assert np.array_equal(b0, r0) and np.array_equal(b1, r1) and np.array_equal(data, rd)
    lam = wd.lambda_()
    assert lam.shape == (3 * len(b0),) and (lam.reshape(-1, 3)[:, 2] >= 0).all()
    assert np.abs(pos - sc["p"]).max() < 2e-2     # the pile stays put
    wd.close()


def test_chain_with_joints_no_contacts(ctx):
    """Chain(8) as a world: 8 ball joints, no contacts are ever found (the
    links do not touch); 20 steps equal the reference's dense path."""
    from helpers import ode_step
    sc = scenes.chain(8)
    ref = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in sc.items()}
    n = 8
    Minv = orc.minv_blocks(sc["R"], sc["mass"], sc["I_body"])
    f_ext = orc.external_force(sc["R"], sc["w"], sc["mass"], sc["I_body"])
    wd = capi.World(ctx, n)
    wd.set_bodies(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext)
    wd.set_joints(sc["body0"], sc["body1"], sc["data"])
    prm = capi.params(method=capi.SOR, max_iters=5000, tol=1e-11, cfm=0.0)
    for _ in range(20):
        wd.step(1e-3, 0.2, prm)
        ode_step(ref, 1e-3)
    pos, R, v, w = wd.bodies()
    assert wd.info() == dict(n_constraints=8, n_contacts=0, replans=wd.info()["replans"])
    assert np.abs(pos - ref["p"]).max() < 1e-9 and np.abs(R - ref["R"]).max() < 1e-9
    wd.close()


def test_free_fall_without_constraints(ctx):
    """No joints, no contacts: v_dot = M^-1 f (ensembles.cc:504-505)."""
    p0 = np.array([[0.0, 0.0, 5.0]])
    wd = make_world(ctx, p0, np.eye(3).reshape(1, 9))
    prm = capi.params(max_iters=10, tol=0.0)
    for _ in range(10):
        wd.step(0.01, 0.2, prm)
    pos, R, v, w = wd.bodies()
    assert abs(v[0, 2] - (-9.8 * 0.1)) < 1e-12
    assert abs(pos[0, 2] - (5.0 - 0.5 * 9.8 * 0.1 * 0.1)) < 1e-12   # midpoint rule is exact for constant g
    wd.close()


@pytest.mark.parametrize("precision", [capi.F64, capi.F32])
def test_world_step_equals_problem_step(ctx, precision):
    """One world step = collide + the device-resident problem step + advance, in either solve
    precision: same contact list, lambda, velocities and positions as the pieces called one by
    one (the pieces are each checked against the oracle elsewhere)."""
    sc = scenes.box_stack(4, 4, 3)
    n = sc["p"].shape[0]
    Minv = orc.minv_blocks(sc["R"], sc["mass"], sc["I_body"])
    f_ext = orc.external_force(sc["R"], sc["w"], sc["mass"], sc["I_body"])
    prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=40, tol=0.0, cfm=0.01)
    wd = capi.World(ctx, n, precision)
    wd.set_bodies(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext)
    pr = None
    pos, R, v, w = sc["p"], sc["R"], sc["v"], sc["w"]
    for step in range(3):
        wd.step(0.005, 0.2, prm)
        b0, b1, data = ctx.update_contacts(pos, R)
        wb0, wb1, wdata = wd.contacts()
        assert np.array_equal(wb0, b0) and np.array_equal(wb1, b1) and np.array_equal(wdata, data)
        pr = capi.Problem(ctx, n, b0, b1, precision)
        pr.set_state(pos, R, v, w, Minv, f_ext)
        pr.set_constraints(np.full(len(b0), capi.CONTACT_BOX, np.int32), data)
        pr.step(0.005, 0.2, prm)
        assert np.array_equal(wd.lambda_(), pr.lambda_()), step
        pr.advance(0.005)
        pos, R, v, w = pr.state()
        pr.close()
        wpos, wR, wv, ww = wd.bodies()
        assert np.array_equal(wpos, pos) and np.array_equal(wR, R) and np.array_equal(wv, v) and np.array_equal(ww, w), step
    wd.close()
