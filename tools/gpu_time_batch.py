#!/usr/bin/env python3
"""Time the batched C3 step for the schedule switches in the environment.
usage: gpu_time_batch.py [piles=24] [steps=20] [workload=c3]; prints one JSON line and checks pile 0 against the oracle."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from eggshell_amd import capi  # noqa: E402

piles = int(sys.argv[1]) if len(sys.argv) > 1 else 24
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
wl = sys.argv[3] if len(sys.argv) > 3 else "c3"
ctx = capi.Context(0)
seeds = bench.c4_shard_seeds(0, 1) if wl == "c4" else [b + 1 for b in range(piles)]
r = bench.run_piles(ctx, wl, seeds, capi.GAUSS_SEIDEL, steps, 3)
st = r["stats"]
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("EGS_")}, "piles": len(seeds), "ms_per_step": r["elapsed"] / steps * 1e3,
       "kernel_ms": r["roofline"]["kernel_ms"], "pile_steps_per_s": len(seeds) * steps / r["elapsed"], "schedule": st.schedule,
       "tile": st.tile_constraints, "residual": st.residual, "status": st.status}
if os.environ.get("EGS_CHECK", "1") != "0":     # bits of pile 0 against the sequential list-order solve
    from oracle import oracle as orc
    pr, sc = r["problem"], r["scene"]
    lam = pr.lambda_()
    J0, J1, is_eq, lo, hi, rhs, err = pr.blocks()
    m1 = r["m"] // len(seeds); n1 = r["n"] // len(seeds)
    Minv, _ = bench.host_mass_and_force(sc)
    s = orc.Sys(Minv[:n1], sc["body0"][:m1], sc["body1"][:m1], J0[:m1], J1[:m1], is_eq[:3 * m1], lo[:3 * m1], hi[:3 * m1])
    if r["prec"] == "f32":
        xf = orc.fast_iterate_f32(s, rhs[:3 * m1], 0.01, orc.GAUSS_SEIDEL, max_iters=r["sweeps"])[0]
        out["bit_exact_pile0"] = bool(np.array_equal(lam[:3 * m1].astype(np.float32), xf))
    else:
        xf = orc.fast_iterate(s, rhs[:3 * m1], 0.01, orc.GAUSS_SEIDEL, max_iters=r["sweeps"], tol=0.0)[0]
        out["bit_exact_pile0"] = bool(np.array_equal(lam[:3 * m1], xf))
print(json.dumps(out), flush=True)
r["problem"].close(); ctx.close()
