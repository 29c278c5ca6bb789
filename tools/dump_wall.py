import sys, numpy as np
sys.path.insert(0, '.')
from eggshell_amd import capi, scenes
ctx = capi.Context(0)
for nx, nz in ((41, 40), (12, 10)):
    sc = scenes.brick_wall(nx, nz)
    b0, b1, data = ctx.update_contacts(sc["p"], sc["R"])
    np.savez("gpurun_out/wall_%dx%d.npz" % (nx, nz), body0=b0, body1=b1, data=data, p=sc["p"])
    print(nx, nz, len(b0))
