import sys, time, numpy as np
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
from eggshell_amd import capi, scenes
from oracle import oracle as orc
from helpers import system_from_scene
ctx = capi.Context(0)
rng = np.random.default_rng(0)
for name, sc in (('chain8', scenes.chain(8)), ('stack2x2x3', scenes.box_stack(2,2,3)), ('stack8x8x4', scenes.box_stack(8,8,4))):
    s, err = system_from_scene(sc)
    rhs = rng.uniform(-1,1,3*s.m)
    for method in (1,2,0):
        for K in (0,1,7,50):
            prm = capi.params(method=method, max_iters=K, tol=0.0, cfm=0.01)
            t=time.time()
            x, st = ctx.solve_blocks(s.Minv, s.body0, s.body1, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs, prm)
            dt=time.time()-t
            xf, a, it, rf = orc.fast_iterate(s, rhs, 0.01, method, max_iters=K, tol=0.0)
            print(name, 'method',method,'K',K,'maxdiff', np.abs(x-xf).max(), 'bitexact', np.array_equal(x,xf), 'res', st.residual, rf, 'tiles', st.n_tiles, 'isl', st.n_islands, 'glob', st.n_global, '%.3fs'%dt, flush=True)
