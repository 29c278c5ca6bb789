"""GPU (-m gpu): the whole Ensemble::Step loop through the C++ adapter --
UpdateContacts (device collision), StepVelocities_ODE with the sparse switch
(device assembly + projected SOR), StepPositions_ODE -- against the same loop
built from oracle pieces (ensembles.cc:390-427)."""
import numpy as np
import pytest

from oracle import oracle as orc
from test_gpu_adapter import demo_out  # noqa: F401  (fixture)
from test_gpu_collide import reference_contacts

pytestmark = pytest.mark.gpu


def oracle_step_loop(p, R, steps, dt, cfm=0.01):
    n = p.shape[0]
    v = np.zeros((n, 3)); w = np.zeros((n, 3))
    mass = np.ones(n); I_body = np.tile((np.eye(3) * 0.1).reshape(9), (n, 1))
    Minv = orc.minv_blocks(R, mass, I_body)            # frozen at Init (quirk Q5)
    f_ext = orc.external_force(R, w, mass, I_body)
    seen = 0
    for _ in range(steps):
        b0, b1, data = reference_contacts(p, R)
        seen += len(b0)
        v6_old = np.concatenate([v, w], axis=1)
        if len(b0) == 0:
            v6 = v6_old + dt * np.einsum("brc,bc->br", Minv.reshape(n, 6, 6), f_ext)
        else:
            kind = np.ones(len(b0), np.int32)
            J0, J1, is_eq, lo, hi, err = orc.assemble(p, R, kind, b0, b1, data)
            s = orc.Sys(Minv, b0, b1, J0, J1, is_eq, lo, hi)
            rhs = orc.ode_rhs(v, w, Minv, f_ext, b0, b1, J0, J1, err, dt, 0.2)
            lam, _, it, res = orc.fast_iterate(s, rhs, cfm, orc.SOR, max_iters=500, tol=1e-9)
            v6 = orc.velocity_update(v, w, Minv, f_ext, b0, b1, J0, J1, lam, dt)
        p, R = orc.position_update(p, R, v6_old, v6, dt)
        v, w = v6[:, :3].copy(), v6[:, 3:].copy()
    return p, np.concatenate([v, w], axis=1), seen


def test_drop_three_boxes(demo_out):  # noqa: F811
    p0 = np.array([[0.0, 0.0, 0.2 + 0.35 * k] for k in range(3)])
    R0 = np.tile(np.eye(3).reshape(9), (3, 1))
    p, v6, seen = oracle_step_loop(p0, R0, 120, 0.005)
    assert int(demo_out["drop_contacts"][0]) == seen and seen > 500
    assert np.abs(demo_out["drop_p"] - p.reshape(-1)).max() < 1e-7
    assert np.abs(demo_out["drop_v"] - v6.reshape(-1)).max() < 1e-5
    assert p[0, 2] < 0.16 and p[2, 2] < 0.78        # they landed and stacked
