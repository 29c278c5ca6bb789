"""GPU (-m gpu): the uniform-grid broad phase ("needs a broad
phase (uniform grid)") yields exactly the candidate pairs, hence exactly the
contact list (order included), of the all-pairs scan and of the oracle."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from eggshell_amd import scenes
from oracle import oracle as orc
from test_gpu_collide import reference_contacts

pytestmark = pytest.mark.gpu


def both(ctx, monkeypatch, p, R, side=None):
    out = {}
    for mode in ("pairs", "grid"):
        monkeypatch.setenv("EGS_BROADPHASE", mode)
        out[mode] = ctx.update_contacts(p, R) if side is None else ctx.update_contacts(p, R, side=side)
    a, b = out["pairs"], out["grid"]
    assert len(a[0]) == len(b[0])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    return b


@pytest.mark.parametrize("seed", range(3))
def test_grid_equals_all_pairs_and_oracle_on_random_boxes(ctx, monkeypatch, seed):
    """Rotated boxes scattered around the origin (negative cell coordinates,
    several bodies per cell, empty cells)."""
    rng = np.random.default_rng(100 + seed)
    n = 150
    p = rng.uniform([-1.2, -1.2, 0.0], [1.2, 1.2, 0.9], (n, 3))
    R = Rotation.random(n, random_state=seed).as_matrix().reshape(n, 9)
    g0, g1, gd = both(ctx, monkeypatch, p, R)
    r0, r1, rd = reference_contacts(p, R)
    assert np.array_equal(g0, r0) and np.array_equal(g1, r1) and np.array_equal(gd, rd)
    assert (g0 >= 0).sum() > 30          # body-body contacts present


def test_grid_on_piles_and_far_bodies(ctx, monkeypatch):
    sc = scenes.box_stack(8, 8, 4, jitter=1e-3, seed=5)
    g0, g1, gd = both(ctx, monkeypatch, sc["p"], sc["R"])
    assert np.array_equal(g0, sc["body0"]) and np.array_equal(g1, sc["body1"]) and np.array_equal(gd, sc["data"])
    # the same pile far from the origin, plus bodies beyond the cell-coordinate clamp
    p = sc["p"].copy()
    p[:, 0] += 12345.678
    p[:, 1] -= 9876.5
    far = np.array([[4.0e6, 0.0, 0.149], [4.0e6 + 0.29, 0.0, 0.149], [-5.0e6, 3.0e6, 0.149]])
    p = np.vstack([p, far])
    R = np.vstack([sc["R"], np.tile(np.eye(3).reshape(1, 9), (3, 1))])
    g0, g1, gd = both(ctx, monkeypatch, p, R)
    n = sc["p"].shape[0]
    assert ((g0 == n) & (g1 == n + 1)).sum() > 0       # the two far boxes touch each other
    assert np.array_equal(g0[(g1 < n)], sc["body0"][: (g1 < n).sum()])


def test_grid_is_the_default_for_large_ensembles(ctx, monkeypatch):
    """20 x 20 x 24 = 9600 bodies (above the 2048 switch-over): the default path
    equals the analytic generator's contact list."""
    monkeypatch.delenv("EGS_BROADPHASE", raising=False)
    sc = scenes.box_stack(20, 20, 24)
    g0, g1, gd = ctx.update_contacts(sc["p"], sc["R"])
    assert len(g0) == 4 * 9600
    assert np.array_equal(g0, sc["body0"]) and np.array_equal(g1, sc["body1"]) and np.array_equal(gd, sc["data"])
    monkeypatch.setenv("EGS_BROADPHASE", "pairs")
    h0, h1, hd = ctx.update_contacts(sc["p"], sc["R"])
    assert np.array_equal(g0, h0) and np.array_equal(g1, h1) and np.array_equal(gd, hd)


@pytest.mark.parametrize("mode", ["pairs", "grid"])
def test_edge_cases_fail_loudly_or_return_empty(ctx, monkeypatch, mode):
    """More than 64 overlapping partners for one body is a documented limit
    (EGS_ERR_INVALID, no hang, no truncated list); too small an output buffer
    likewise; one body far above the ground gives an empty list."""
    from eggshell_amd import capi
    monkeypatch.setenv("EGS_BROADPHASE", mode)
    eye = np.eye(3).reshape(1, 9)
    p = np.tile([[0.0, 0.0, 5.0]], (70, 1)) + np.arange(70)[:, None] * 1e-4      # 70 boxes in one spot
    with pytest.raises(capi.EgsError) as e:
        ctx.update_contacts(p, np.tile(eye, (70, 1)))
    assert e.value.status == capi.ERR_INVALID
    sc = scenes.box_stack(2, 2, 3)
    with pytest.raises(capi.EgsError) as e:
        ctx.update_contacts(sc["p"], sc["R"], max_contacts=10)
    assert e.value.status == capi.ERR_INVALID
    g0, g1, gd = ctx.update_contacts(np.array([[0.0, 0.0, 3.0]]), eye)
    assert len(g0) == 0 and gd.shape == (0, 7)
    g0, g1, gd = ctx.update_contacts(np.array([[0.0, 0.0, 0.149]]), eye)          # one box resting on the ground
    assert len(g0) == 4 and (g0 == -1).all() and (g1 == 0).all()
    g0, g1, gd = ctx.update_contacts(sc["p"], sc["R"])                            # the context still works
    assert len(g0) == 48


def test_more_than_64_partners_for_one_body(ctx):
    """A big plate under 100 small boxes: body 0 has 100 candidate partners, more than the 64 the capped
    candidate lists hold.  The reference (all pairs, ensembles.cc:462-477) has no such limit; the collider
    falls back to an uncapped count -> scan -> fill pass and the contact list stays the reference's."""
    n = 101
    p = np.zeros((n, 3)); side = np.tile([0.3, 0.3, 0.3], (n, 1))
    side[0] = [5.0, 5.0, 0.3]
    p[0] = [0.0, 0.0, 0.149]
    for k in range(100):
        p[1 + k] = [-2.0 + 0.42 * (k % 10), -2.0 + 0.42 * (k // 10), 0.448]
    R = np.tile(np.eye(3).reshape(9), (n, 1))
    g0, g1, gd = ctx.update_contacts(p, R, side)
    b0, b1, data = [], [], []
    for b in range(n):
        for c in orc.collide_box_ground(p[b], R[b], side[b]):
            b0.append(-1); b1.append(b); data.append(c)
    for i in range(n):
        for j in range(i + 1, n):
            cs, code = orc.collide_boxes(p[i], R[i], p[j], R[j], side[i], side[j])
            keep = [cs[a] for a in range(len(cs)) if not any(np.linalg.norm(cs[b][:3] - cs[a][:3]) < 1e-6 for b in range(a))]
            for c in keep:
                b0.append(i); b1.append(j); data.append(c)
    assert len(g0) == len(b0) == 4 + 400
    assert np.array_equal(g0, np.array(b0, np.int32)) and np.array_equal(g1, np.array(b1, np.int32))
    assert np.array_equal(gd, np.array(data).reshape(-1, 7))
