import sys, hashlib; sys.path.insert(0,'/root/repo')
import numpy as np
import bench
from eggshell_amd import capi, scenes
ctx = capi.Context(0)
nx, ny, nz, sweeps, prec, dt = bench.WORKLOADS["c3"]
sc = scenes.box_stack(nx, ny, nz, jitter=1e-3, seed=1)
Minv, f_ext = bench.host_mass_and_force(sc)
w = capi.World(ctx, sc["p"].shape[0])
w.set_bodies(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext)
prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=sweeps, tol=0.0, cfm=0.01)
seen = {}
seq = []
for k in range(120):
    w.step(dt, 0.2, prm)
    b0, b1, data = w.contacts()[:3]
    h = hashlib.md5(np.asarray(b0).tobytes() + np.asarray(b1).tobytes()).hexdigest()[:8]
    if h not in seen: seen[h] = len(seen)
    seq.append(seen[h])
print("distinct topologies in 120 steps:", len(seen))
print(seq)
print("replans", w.info()["replans"])
