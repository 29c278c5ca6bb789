"""GPU (-m gpu): contact generation on the device equals the
oracle's restated collision.cc + the reference's list order + its pruning,
BIT FOR BIT (positions, normals, depths, body indices, order)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from eggshell_amd import scenes
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def reference_contacts(p, R, side=0.3):
    """Ensemble::UpdateContacts (ensembles.cc:445-480) then the contact-vs-contact
    pruning of CheckAndCorrectEnsembleState (ensembles.cc:308-328) via the oracle."""
    n = p.shape[0]
    b0, b1, data = [], [], []
    for b in range(n):
        for c in orc.collide_box_ground(p[b], R[b]):
            b0.append(-1); b1.append(b); data.append(c)
    for i in range(n):
        for j in range(i + 1, n):
            if np.linalg.norm(p[i] - p[j]) > 0.53:
                continue
            cs, code = orc.collide_boxes(p[i], R[i], p[j], R[j])
            keep = []
            for a in range(len(cs)):
                if not any(np.linalg.norm(cs[b][:3] - cs[a][:3]) < 1e-6 for b in range(a)):
                    keep.append(cs[a])
            for c in keep:
                b0.append(i); b1.append(j); data.append(c)
    return np.array(b0, np.int32), np.array(b1, np.int32), np.array(data).reshape(-1, 7)


def check(ctx, p, R):
    g0, g1, gd = ctx.update_contacts(p, R)
    r0, r1, rd = reference_contacts(p, R)
    assert len(g0) == len(r0), (len(g0), len(r0))
    assert np.array_equal(g0, r0) and np.array_equal(g1, r1)
    assert np.array_equal(gd, rd)
    return len(g0)


def test_box_stacks_equal_generator_and_oracle(ctx):
    for shape, jitter in (((2, 2, 3), 0.0), ((4, 3, 5), 1e-3), ((8, 8, 4), 0.0)):
        sc = scenes.box_stack(*shape, jitter=jitter, seed=2)
        g0, g1, gd = ctx.update_contacts(sc["p"], sc["R"])
        assert np.array_equal(g0, sc["body0"]) and np.array_equal(g1, sc["body1"]) and np.array_equal(gd, sc["data"])
    sc = scenes.box_stack(2, 2, 3)
    assert check(ctx, sc["p"], sc["R"]) == 48


def test_c3_contact_set(ctx):
    sc = scenes.box_stack(16, 16, 16)
    g0, g1, gd = ctx.update_contacts(sc["p"], sc["R"])
    assert len(g0) == 16384
    assert np.array_equal(g0, sc["body0"]) and np.array_equal(g1, sc["body1"]) and np.array_equal(gd, sc["data"])


@pytest.mark.parametrize("seed", range(4))
def test_random_overlapping_boxes(ctx, seed):
    """Randomly rotated boxes dropped into a small volume: face-face, face-edge,
    face-corner and edge-edge contacts, partly below the ground plane."""
    rng = np.random.default_rng(seed)
    n = 60
    p = rng.uniform([-0.5, -0.5, 0.0], [0.5, 0.5, 0.6], (n, 3))
    R = Rotation.random(n, random_state=seed).as_matrix().reshape(n, 9)
    if seed == 0:                       # nearly aligned faces (aacount >= 2 branch)
        R[: n // 2] = Rotation.from_rotvec(rng.normal(size=(n // 2, 3)) * 0.02).as_matrix().reshape(-1, 9)
    m = check(ctx, p, R)
    assert m > 20


def test_brick_wall_is_one_island(ctx):
    """A running-bond wall: contacts from the device collider feed the solver;
    the whole wall is one island (cross-workgroup path) and the solve is
    bit-exact against the oracle."""
    from eggshell_amd import capi
    from helpers import system_from_scene
    sc = scenes.brick_wall(12, 10)
    check(ctx, sc["p"], sc["R"])
    b0, b1, data = ctx.update_contacts(sc["p"], sc["R"])
    sc.update(kind=np.full(len(b0), capi.CONTACT_BOX, np.int32), body0=b0, body1=b1, data=data)
    pl = capi.debug_plan(sc["p"].shape[0], b0, b1)
    assert pl["n_islands"] == 1
    s, err = system_from_scene(sc)
    rng = np.random.default_rng(3)
    rhs = rng.uniform(-1, 1, 3 * s.m)
    x, st = ctx.solve_blocks(s.Minv, s.body0, s.body1, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs,
                             capi.params(method=capi.GAUSS_SEIDEL, max_iters=20, tol=0.0, cfm=0.05))
    xf, _, _, _ = orc.fast_iterate(s, rhs, 0.05, orc.GAUSS_SEIDEL, max_iters=20, tol=0.0)
    assert st.n_global == s.m and np.array_equal(x, xf)


def test_no_bodies_and_no_contacts(ctx):
    g0, g1, gd = ctx.update_contacts(np.zeros((0, 3)), np.zeros((0, 9)))
    assert len(g0) == 0
    p = np.array([[0.0, 0.0, 5.0], [3.0, 0.0, 5.0]])
    g0, g1, gd = ctx.update_contacts(p, np.tile(np.eye(3).reshape(9), (2, 1)))
    assert len(g0) == 0
