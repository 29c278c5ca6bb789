"""Contact generation timing (host in / host out) vs body count, for both broad
phases: "pairs" = one wavefront per body scanning all j > i, "grid" = hashed
uniform grid (default from 2048 bodies)."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from eggshell_amd import capi, scenes
ctx = capi.Context(0)
for batch in (1, 4, 16, 32):
    piles = [scenes.box_stack(16, 16, 16, origin=(0.0, 100.0 * b)) for b in range(batch)]
    sc = scenes.concat(piles) if batch > 1 else piles[0]
    for mode in ("pairs", "grid"):
        os.environ["EGS_BROADPHASE"] = mode
        ctx.update_contacts(sc["p"], sc["R"])
        t = time.perf_counter(); N = 5
        for _ in range(N): b0, b1, d = ctx.update_contacts(sc["p"], sc["R"])
        dt = (time.perf_counter() - t) / N
        print(f"n={sc['p'].shape[0]} {mode:5s} -> {len(b0)} contacts: {dt*1e3:.2f} ms per egs_update_contacts", flush=True)
