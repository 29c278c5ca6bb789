"""Lane order inside tiles: phase-major (default) vs island-major (EGS_LANE_ORDER=0)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from eggshell_amd import capi, scenes
import bench
ctx = capi.Context(0)
for batch in (1, 4, 16, 32):
    piles = [scenes.box_stack(16, 16, 16, origin=(0.0, 100.0 * b)) for b in range(batch)]
    sc = scenes.concat(piles) if batch > 1 else piles[0]
    Minv, f_ext = bench.host_mass_and_force(sc)
    for quad in ("auto", "0", "1"):
        if quad == "1" and batch > 16: continue
        for order in ("1", "0"):
            os.environ["EGS_LANE_ORDER"] = order
            if quad == "auto": os.environ.pop("EGS_QUAD", None)
            else: os.environ["EGS_QUAD"] = quad
            pr = capi.Problem(ctx, sc["p"].shape[0], sc["body0"], sc["body1"])
            pr.set_state(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext); pr.set_constraints(sc["kind"], sc["data"])
            prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=100, tol=0.0, cfm=0.01)
            for _ in range(3): pr.step(5e-3, 0.2, prm)
            ctx.synchronize(); ctx.timer_start()
            for _ in range(10): pr.step(5e-3, 0.2, prm)
            ms = ctx.timer_stop() / 10
            print(f"batch {batch:2d} quad={quad:4s} phase_order={order}: {ms:.3f} ms/step  {batch*1000/ms:.0f} pile-steps/s", flush=True)
            pr.close()
