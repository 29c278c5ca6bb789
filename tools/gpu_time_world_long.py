#!/usr/bin/env python3
"""Ensemble::Step() on the device for one C3 pile over a long run: steps 0-39 (the pile settles: every contact list is new)
and steps 40-239 (it rocks between a handful of contact lists) separately.  EGS_PLAN_CACHE=0 switches the parked
schedules off."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from eggshell_amd import capi, scenes  # noqa: E402

ctx = capi.Context(0)
nx, ny, nz, sweeps, prec, dt = bench.WORKLOADS["c3"]
sc = scenes.box_stack(nx, ny, nz, jitter=1e-3, seed=1)
Minv, f_ext = bench.host_mass_and_force(sc)
w = capi.World(ctx, sc["p"].shape[0])
w.set_bodies(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext)
prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=sweeps, tol=0.0, cfm=0.01)
for lo, hi in ((0, 40), (40, 240)):
    r0 = w.info()["replans"]
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(lo, hi):
        w.step(dt, 0.2, prm)
    ctx.synchronize()
    el = time.perf_counter() - t0
    print("steps %3d-%3d: %.3f ms per step, %d topology changes" % (lo, hi - 1, el / (hi - lo) * 1e3, w.info()["replans"] - r0), flush=True)
w.close()
ctx.close()
