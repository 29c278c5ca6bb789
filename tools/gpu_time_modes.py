"""Quad (4 lanes/constraint) vs tile (1 lane/constraint) schedule across batch sizes, C3."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from eggshell_amd import capi, scenes
sys.path.insert(0, R)
import bench
ctx = capi.Context(0)
for batch in (1, 2, 4, 8, 16):
    piles = [scenes.box_stack(16, 16, 16, origin=(0.0, 100.0 * b)) for b in range(batch)]
    sc = scenes.concat(piles) if batch > 1 else piles[0]
    Minv, f_ext = bench.host_mass_and_force(sc)
    for quad in ("1", "0"):
        os.environ["EGS_QUAD"] = quad
        pr = capi.Problem(ctx, sc["p"].shape[0], sc["body0"], sc["body1"])
        pr.set_state(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext); pr.set_constraints(sc["kind"], sc["data"])
        prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=100, tol=0.0, cfm=0.01)
        for _ in range(3): pr.step(5e-3, 0.2, prm)
        ctx.synchronize(); ctx.timer_start()
        for _ in range(20): pr.step(5e-3, 0.2, prm)
        ms = ctx.timer_stop() / 20
        print(f"batch {batch:2d} quad={quad}: {ms:.3f} ms/step  {batch*1000/ms:.0f} pile-steps/s", flush=True)
        pr.close()
