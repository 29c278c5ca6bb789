"""4-lane (EGS_QUAD=1) vs 1-lane (EGS_QUAD=0) schedule for 1..6 C3 piles per launch: where the
automatic choice (m <= 32768 constraints -> 4-lane) should switch."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from eggshell_amd import capi, scenes
import bench
ctx = capi.Context(0)
for batch in (1, 2, 3, 4, 6):
    piles = [scenes.box_stack(16, 16, 16, origin=(0.0, 100.0 * b)) for b in range(batch)]
    sc = scenes.concat(piles) if batch > 1 else piles[0]
    Minv, f_ext = bench.host_mass_and_force(sc)
    out = []
    for quad in ("1", "0"):
        os.environ["EGS_QUAD"] = quad
        pr = capi.Problem(ctx, sc["p"].shape[0], sc["body0"], sc["body1"])
        pr.set_state(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext); pr.set_constraints(sc["kind"], sc["data"])
        prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=100, tol=0.0, cfm=0.01)
        for _ in range(3): pr.step(5e-3, 0.2, prm)
        ctx.synchronize(); ctx.timer_start()
        for _ in range(10): pr.step(5e-3, 0.2, prm)
        out.append(ctx.timer_stop() / 10)
        pr.close()
    print(f"batch {batch}: 4-lane {out[0]:.3f} ms, 1-lane {out[1]:.3f} ms per step", flush=True)
