"""Debug aid: the 64x64 wall (one 40k-contact island) solved three times in a row through the
one-shot entry for K = 1, 2, 10, 100 sweeps, mismatches against the oracle printed per run.
Run with EGS_PATCH=0 to exercise the all-global kernel (this is how the stale-accumulator bug
of repeated solves was located)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from eggshell_amd import capi, scenes
from helpers import system_from_scene
from oracle import oracle as orc
ctx = capi.Context(0)
sc = scenes.brick_wall(64, 64)
b0, b1, data = ctx.update_contacts(sc["p"], sc["R"])
sc.update(kind=np.full(len(b0), capi.CONTACT_BOX, np.int32), body0=b0, body1=b1, data=data)
s, err = system_from_scene(sc)
rng = np.random.default_rng(9)
rhs = rng.uniform(-1, 1, 3 * s.m)
for method in (capi.GAUSS_SEIDEL, capi.SOR):
    for K in (1, 2, 10, 100):
        xf, af, _, rf = orc.fast_iterate(s, rhs, 0.01, method, max_iters=K, tol=0.0)
        for rep in range(3):
            x, st = ctx.solve_blocks(s.Minv, s.body0, s.body1, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs,
                                     capi.params(method=method, max_iters=K, tol=0.0, cfm=0.01))
            bad = np.flatnonzero(x != xf)
            print(method, K, rep, st.status, st.n_global, "mismatches", len(bad), bad[:6] // 3, np.abs(x - xf).max(), flush=True)
