"""Batched C3 throughput vs piles per launch (tile-count quantisation: a CU holds two
256-constraint tiles, three when the bodies are isotropic)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from eggshell_amd import capi, scenes
import bench
ctx = capi.Context(0)
for batch in (8, 12, 16, 20, 24, 32, 36, 48):
    piles = [scenes.box_stack(16, 16, 16, origin=(0.0, 100.0 * b)) for b in range(batch)]
    sc = scenes.concat(piles)
    Minv, f_ext = bench.host_mass_and_force(sc)
    pr = capi.Problem(ctx, sc["p"].shape[0], sc["body0"], sc["body1"])
    pr.set_state(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext); pr.set_constraints(sc["kind"], sc["data"])
    prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=100, tol=0.0, cfm=0.01)
    for _ in range(3): pr.step(5e-3, 0.2, prm)
    ctx.synchronize(); ctx.timer_start()
    for _ in range(10): pr.step(5e-3, 0.2, prm)
    ms = ctx.timer_stop() / 10
    st = pr.stats()
    print(f"batch {batch}: {ms:.3f} ms/step {batch*1000/ms:.0f} pile-steps/s (tiles {st.n_tiles})", flush=True)
    pr.close()
