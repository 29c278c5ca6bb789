#!/usr/bin/env python3
"""Time the stand-alone product on N replicated C3 piles for the schedule switches in the environment
(EGS_MV_TILE=128/256).  usage: gpu_time_matvec.py [piles=64] [launches=30]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from eggshell_amd import capi, scenes  # noqa: E402

piles = int(sys.argv[1]) if len(sys.argv) > 1 else 64
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ctx = capi.Context(0)
one = scenes.box_stack(16, 16, 16, jitter=1e-3, seed=1)
sc = bench.replicate(one, piles)
pr, _ = bench.build_problem(ctx, sc, capi.F64)
pr.assemble(5e-3, 0.2)
pr.solve(capi.params(method=capi.GAUSS_SEIDEL, max_iters=1, tol=0.0, cfm=0.01), want_stats=False)
r = bench.matvec_measure(ctx, pr, sc["kind"].shape[0], sc["p"].shape[0], "f64", launches, 5)
print(json.dumps({"tile": os.environ.get("EGS_MV_TILE", "256"), "piles": piles, "kernel_ms": r["roofline"]["kernel_ms"],
                  "gbs": r["roofline"]["achieved"], "frac": r["roofline"]["frac"]}))
pr.close(); ctx.close()
