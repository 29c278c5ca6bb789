"""Whole Ensemble::Step on the device (egs_world): collide + solve + integrate, C3 pile."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from eggshell_amd import capi, scenes
import bench
ctx = capi.Context(0)
for shape, K in (((8, 8, 4), 50), ((16, 16, 16), 100)):
    sc = scenes.box_stack(*shape)
    n = sc["p"].shape[0]
    Minv, f_ext = bench.host_mass_and_force(sc)
    wd = capi.World(ctx, n)
    wd.set_bodies(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext)
    prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=K, tol=0.0, cfm=0.01)
    for _ in range(3): wd.step(5e-3, 0.2, prm)
    ctx.synchronize(); r0 = wd.info()["replans"]
    t = time.perf_counter(); N = 30
    for _ in range(N): wd.step(5e-3, 0.2, prm)
    ctx.synchronize(); dt = (time.perf_counter() - t) / N
    info = wd.info()
    print(f"world {shape}: {dt*1e3:.3f} ms per full Step (collide+solve+integrate), contacts {info['n_contacts']}, re-plans in {N} steps: {info['replans']-r0}", flush=True)
    t = time.perf_counter()
    for _ in range(N): wd.step(5e-3, 0.2, prm, detect_contacts=False)
    ctx.synchronize(); dt2 = (time.perf_counter() - t) / N
    print(f"      without contact detection (frozen contact set): {dt2*1e3:.3f} ms per Step", flush=True)
    wd.close()
