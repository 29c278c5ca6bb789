"""GPU (-m gpu): the joint-vs-contact check of CheckAndCorrectEnsembleState
(ensembles.cc:296-306): a contact within 1e-6 of a joint between the same two
bodies is dropped; joints to the world never prune (quirk Q4)."""
import numpy as np
import pytest

from eggshell_amd import scenes
from test_gpu_collide import reference_contacts

pytestmark = pytest.mark.gpu


def test_joint_on_a_contact_point_prunes_it(ctx):
    sc = scenes.box_stack(1, 1, 2)
    p, R = sc["p"], sc["R"]
    b0, b1, data = ctx.update_contacts(p, R)
    assert len(b0) == 8
    target = data[5, :3]                      # second contact of the (0, 1) pair
    # a ball joint between bodies 0 and 1 whose two anchor points both sit on `target`
    c0 = target - p[0]
    c1 = target - p[1]
    joints = (np.array([0, 0], np.int32), np.array([1, -1], np.int32),
              np.array([np.r_[c0, c1, 0.0], np.r_[0.0, 0.0, 0.0, data[0, :3], 0.0]]))   # 2nd: anchored to the world at a ground contact
    g0, g1, gd = ctx.update_contacts(p, R, joints=joints)
    assert len(g0) == 7
    keep = [k for k in range(8) if k != 5]
    assert np.array_equal(g0, b0[keep]) and np.array_equal(g1, b1[keep]) and np.array_equal(gd, data[keep])
    # a joint 1e-5 away does not prune
    joints2 = (joints[0][:1], joints[1][:1], np.array([np.r_[c0 + [1e-5, 0, 0], c1 + [1e-5, 0, 0], 0.0]]))
    g0, g1, gd = ctx.update_contacts(p, R, joints=joints2)
    assert len(g0) == 8
    # joints listed as (1, 0) prune too (the map key is the unordered pair, ensembles.cc:339-340)
    joints3 = (np.array([1], np.int32), np.array([0], np.int32), np.array([np.r_[c1, c0, 0.0]]))
    g0, g1, gd = ctx.update_contacts(p, R, joints=joints3)
    assert len(g0) == 7
