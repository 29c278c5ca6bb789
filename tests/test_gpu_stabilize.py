"""GPU (-m gpu): Ensemble::InitStabilize / PostStabilize through the C++ adapter
(ensembles.cc:602-666) against the same loops built from oracle pieces, with
the reference's dense solve (J J^T) y = err done by numpy least squares."""
import numpy as np
import pytest

from eggshell_amd import scenes
from oracle import oracle as orc
from test_gpu_adapter import demo_out  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def dense_J(sc, J0, J1):
    n, m = sc["p"].shape[0], sc["kind"].shape[0]
    J = np.zeros((3 * m, 6 * n))
    for i in range(m):
        for b, blk in ((sc["body0"][i], J0[i]), (sc["body1"][i], J1[i])):
            if b >= 0:
                J[3 * i:3 * i + 3, 6 * b:6 * b + 6] = blk.reshape(3, 6)
    return J


def relaxation(sc, step_scale=0.2):
    """CalculateVelocityRelaxation, ensembles.cc:659-666."""
    J0, J1, is_eq, lo, hi, err = orc.assemble(sc["p"], sc["R"], sc["kind"], sc["body0"], sc["body1"], sc["data"])
    J = dense_J(sc, J0, J1)
    y = np.linalg.lstsq(J @ J.T, err, rcond=None)[0]
    return (-step_scale * (J.T @ y)).reshape(-1, 6), err


def explicit_euler(sc, v6, dt):
    """StepPositions_ExplicitEuler, ensembles.cc:553-561."""
    n = sc["p"].shape[0]
    sc["p"] = sc["p"] + dt * v6[:, :3]
    R = sc["R"].copy()
    for b in range(n):
        R[b] = (orc.w_to_R(v6[b, 3:], dt) @ R[b].reshape(3, 3)).reshape(9)
    sc["R"] = R


def test_init_and_post_stabilize(demo_out):  # noqa: F811
    sc = scenes.chain(4)
    for i in range(1, 4):
        sc["p"][i] += np.array([0.01 * i, -0.02 * i, 0.015 * i])
    steps = 0
    corr, err = relaxation(sc)
    while err @ err > 1e-9 and steps < 100:            # InitStabilize, ensembles.cc:602-622
        explicit_euler(sc, corr, 0.001 * 500)
        corr, err = relaxation(sc)
        steps += 1
    assert int(demo_out["stab_steps"][0]) == steps and 0 < steps < 100
    assert np.abs(demo_out["stab_p"] - sc["p"].reshape(-1)).max() < 1e-8
    assert demo_out["stab_err"] @ demo_out["stab_err"] <= 1e-9
    for i in range(1, 4):
        sc["p"][i] += np.array([-0.02, 0.01 * i, 0.0])
        sc["v"][i] = [0.1, 0.0, -0.2]
    steps = 0
    corr, err = relaxation(sc)
    while err @ err > 1e-9 and steps < 500:            # PostStabilize, ensembles.cc:624-646
        explicit_euler(sc, corr, 0.001 * 100)
        sc["v"] = sc["v"] + corr[:, :3]
        sc["w"] = sc["w"] + corr[:, 3:]
        corr, err = relaxation(sc)
        steps += 1
    assert int(demo_out["post_steps"][0]) == steps and steps > 0
    assert np.abs(demo_out["post_p"] - sc["p"].reshape(-1)).max() < 1e-7
    v6 = np.concatenate([sc["v"], sc["w"]], axis=1).reshape(-1)
    assert np.abs(demo_out["post_v"] - v6).max() < 1e-6
