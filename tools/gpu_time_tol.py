"""Tolerance-terminated solves (the reference's default: stop at 1e-9 or 500 sweeps, residual
checked every sweep): wall time per solve and per sweep."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from eggshell_amd import capi, scenes
import bench
ctx = capi.Context(0)
for name, sc, cfm in (("chain(8)", scenes.chain(8), 0.1), ("C2 pile", scenes.box_stack(8, 8, 4), 0.1),
                      ("C3 pile", scenes.box_stack(16, 16, 16), 0.1)):
    Minv, f_ext = bench.host_mass_and_force(sc)
    pr = capi.Problem(ctx, sc["p"].shape[0], sc["body0"], sc["body1"])
    pr.set_state(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext); pr.set_constraints(sc["kind"], sc["data"])
    for every in (1, 8):
        prm = capi.params(method=capi.SOR, max_iters=500, tol=1e-9, cfm=cfm, check_every=every)
        pr.step(1e-3, 0.2, prm)
        ctx.synchronize(); t = time.perf_counter(); N = 5
        for _ in range(N): st = pr.step(1e-3, 0.2, prm, want_stats=True)
        ctx.synchronize(); dt = (time.perf_counter() - t) / N
        print(f"{name}: check_every={every}: {st.iterations} sweeps, residual {st.residual:.2e}, {dt*1e3:.2f} ms/solve, "
              f"{dt*1e6/max(st.iterations,1):.1f} us/sweep", flush=True)
    pr.close()
