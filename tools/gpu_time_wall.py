"""Connected pile (running-bond brick wall = ONE island): cross-workgroup path timing."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from eggshell_amd import capi, scenes
import bench
ctx = capi.Context(0)
for nx, nz, K in ((12, 10, 100), (32, 32, 100), (64, 64, 100)):
    sc = scenes.brick_wall(nx, nz)
    b0, b1, data = ctx.update_contacts(sc["p"], sc["R"])
    sc.update(kind=np.full(len(b0), 1, np.int32), body0=b0, body1=b1, data=data)
    n, m = sc["p"].shape[0], len(b0)
    Minv, f_ext = bench.host_mass_and_force(sc)
    t = time.perf_counter()
    pr = capi.Problem(ctx, n, b0, b1)
    tp = time.perf_counter() - t
    pr.set_state(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext); pr.set_constraints(sc["kind"], sc["data"])
    prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=K, tol=0.0, cfm=0.01)
    st = pr.step(5e-3, 0.2, prm, want_stats=True)
    ctx.synchronize(); ctx.timer_start()
    for _ in range(3): pr.step(5e-3, 0.2, prm)
    ms = ctx.timer_stop() / 3
    print(f"wall {nx}x{nz}: n={n} m={m} islands={st.n_islands} global={st.n_global} status={st.status}: {ms:.2f} ms/step ({m*K/ms/1e6:.3f} G contact-iters/s), plan {tp*1e3:.1f} ms, res {st.residual:.3g}", flush=True)
    pr.close()
