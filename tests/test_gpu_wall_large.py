"""GPU (-m gpu): a large connected pile (64x64 running-bond wall, one island of
~40k contacts) through the cross-workgroup path: ~4 million inter-workgroup
hand-offs, compared bit for bit with the oracle (catches any stale hand-off)."""
import numpy as np
import pytest

from eggshell_amd import capi, scenes
from helpers import system_from_scene
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("method", [capi.GAUSS_SEIDEL, capi.SOR])
def test_wall_64x64_bit_exact(ctx, method):
    sc = scenes.brick_wall(64, 64)
    b0, b1, data = ctx.update_contacts(sc["p"], sc["R"])
    sc.update(kind=np.full(len(b0), capi.CONTACT_BOX, np.int32), body0=b0, body1=b1, data=data)
    s, err = system_from_scene(sc)
    rng = np.random.default_rng(9)
    rhs = rng.uniform(-1, 1, 3 * s.m)
    for rep in range(2):
        x, st = ctx.solve_blocks(s.Minv, s.body0, s.body1, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs,
                                 capi.params(method=method, max_iters=100, tol=0.0, cfm=0.01))
        assert st.status == capi.OK and st.n_islands == 1 and st.n_global == s.m > 40000
        if rep == 0:
            xf, af, _, rf = orc.fast_iterate(s, rhs, 0.01, method, max_iters=100, tol=0.0)
        assert np.array_equal(x, xf)
