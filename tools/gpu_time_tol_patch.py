"""Tolerance-terminated solve of one connected pile (running-bond wall = one oversize island)
on the body-patch kernels: recorded chunks (default) -- 4-lane and 1-lane patches -- wall
time per solve and per sweep.  EGS_QUAD_PATCH is read when the schedule is built."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from eggshell_amd import capi, scenes
import bench
ctx = capi.Context(0)
for nx, nz in ((12, 10), (24, 20)):
    sc = scenes.brick_wall(nx, nz)
    b0, b1, data = ctx.update_contacts(sc["p"], sc["R"])
    sc.update(kind=np.full(len(b0), capi.CONTACT_BOX, np.int32), body0=b0, body1=b1, data=data)
    Minv, f_ext = bench.host_mass_and_force(sc)
    for qp in ("1", "0"):
        os.environ["EGS_QUAD_PATCH"] = qp
        pr = capi.Problem(ctx, sc["p"].shape[0], sc["body0"], sc["body1"])
        pr.set_state(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext); pr.set_constraints(sc["kind"], sc["data"])
        prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=200, tol=1e-9, cfm=0.01)
        pr.step(1e-3, 0.2, prm)
        ctx.synchronize(); t = time.perf_counter(); N = 3
        for _ in range(N): st = pr.step(1e-3, 0.2, prm, want_stats=True)
        ctx.synchronize(); dt = (time.perf_counter() - t) / N
        print(f"wall {nx}x{nz} ({len(b0)} contacts) EGS_QUAD_PATCH={qp}: {st.iterations} sweeps, residual {st.residual:.2e}, "
              f"{dt*1e3:.2f} ms/solve, {dt*1e6/max(st.iterations,1):.1f} us/sweep", flush=True)
        pr.close()
