#!/usr/bin/env python3
"""Time the reference's stopping loop (tolerance-terminated solve, residual after every sweep) on one C3 / C2
pile (or `piles` of them in one problem): chunks of recorded sweeps on the device.
usage: gpu_time_tol.py [c3|c2] [cap=500] [piles=1]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from eggshell_amd import capi, scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 500
npiles = int(sys.argv[3]) if len(sys.argv) > 3 else 1
nx, ny, nz, sweeps, prec, dt = bench.WORKLOADS[wl]
ctx = capi.Context(0)
sc = scenes.box_stack(nx, ny, nz, jitter=1e-3, seed=1)
if npiles > 1:
    sc = scenes.concat([scenes.box_stack(nx, ny, nz, jitter=1e-3, seed=k + 1, origin=(0.0, 100.0 * k)) for k in range(npiles)])
pr, _ = bench.build_problem(ctx, sc, capi.F64)
pr.assemble(dt, 0.2)
prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=cap, tol=1e-9, cfm=0.01)
st = pr.solve(prm)
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); st = pr.solve(prm); best = min(best, time.perf_counter() - t0)
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("EGS_")}, "workload": wl, "piles": npiles, "ms_per_solve": best * 1e3,
                  "sweeps": st.iterations, "residual": st.residual, "schedule": st.schedule, "us_per_sweep": best * 1e6 / max(st.iterations, 1)}))
pr.close(); ctx.close()
