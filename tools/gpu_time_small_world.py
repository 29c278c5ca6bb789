#!/usr/bin/env python3
"""Latency of Ensemble::Step() resident on the device for the reference's own sizes: a 4 x 4 x 4 box pile (Cairn(4))
and a smaller one, 50 GS sweeps, with the per-phase host times of egs_world_step (EGS_WORLD_TRACE=1)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from eggshell_amd import capi, scenes  # noqa: E402

ctx = capi.Context(0)
for nx, ny, nz in ((4, 4, 4), (2, 2, 2)):
    sc = scenes.box_stack(nx, ny, nz, jitter=1e-3, seed=1)
    Minv, f_ext = bench.host_mass_and_force(sc)
    n = sc["p"].shape[0]
    w = capi.World(ctx, n)
    w.set_bodies(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext)
    prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=50, tol=0.0, cfm=0.01)
    for _ in range(5):
        w.step(1e-3, 0.2, prm)
    r0 = w.info()["replans"]
    ctx.synchronize()
    t0 = time.perf_counter()
    steps = 200
    for _ in range(steps):
        w.step(1e-3, 0.2, prm)
    ctx.synchronize()
    el = time.perf_counter() - t0
    info = w.info()
    print("%dx%dx%d: %d bodies, %d contacts, %.3f ms per step, %d re-plans in %d steps" % (nx, ny, nz, n, info["n_contacts"], el / steps * 1e3, info["replans"] - r0, steps), flush=True)
    w.close()
ctx.close()
