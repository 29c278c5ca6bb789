#!/usr/bin/env python3
"""Dependent-update latency of the solve kernels, measured on the real kernels: 64
contacts of ONE body with the world (one-sided updates) resp. between the same TWO bodies (two-sided)
are one island whose 64 constraints are totally ordered, so a K-sweep solve is a chain of 64 K
dependent updates, hand-offs included.  t_update = kernel time / (64 K); the smaller of the two is
the latency bound bench.py prices a launch against.  Prints one JSON object (merged into profiles/rNN/microbench.json)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eggshell_amd import capi  # noqa: E402


def chain_time(ctx, env, precision, two_sided, K=2000, m=64):
    for k in ("EGS_QUAD", "EGS_ISO", "EGS_TILE", "EGS_STEP"):
        os.environ.pop(k, None)
    os.environ.update(env)
    rng = np.random.default_rng(0)
    if two_sided:      # body-body contacts: both sides of every update
        n, body0, body1 = 2, np.zeros(m, np.int32), np.ones(m, np.int32)
        J0 = rng.uniform(-1, 1, (m, 18))
    else:              # body-world contacts: the world side is skipped
        n, body0, body1 = 1, np.full(m, -1, np.int32), np.zeros(m, np.int32)
        J0 = np.zeros((m, 18))
    Minv = np.tile(np.diag([1.0, 1.0, 1.0, 10.0, 10.0, 10.0]).reshape(1, 36), (n, 1))
    J1 = rng.uniform(-1, 1, (m, 18))
    is_eq = np.zeros(3 * m, np.uint8)
    lo = np.tile([-1.0, -1.0, 0.0], m); hi = np.tile([1.0, 1.0, np.inf], m)
    rhs = rng.uniform(-1, 1, 3 * m)
    pr = capi.Problem(ctx, n, body0, body1, precision)
    pr.set_blocks(Minv, J0, J1, is_eq, lo, hi, rhs)
    prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=K, tol=0.0, cfm=0.01)
    best = None
    for rep in range(4):
        ctx.kernel_time(reset=True)
        pr.solve(prm, want_stats=False)
        ms, launches = ctx.kernel_time(reset=True)
        best = ms if best is None else min(best, ms)
    st = pr.stats()
    pr.close()
    return best * 1e3 / (m * K), st.schedule, st.tile_constraints


def main():
    ctx = capi.Context(0)
    out = {}
    T, S = {"EGS_QUAD": "0", "EGS_STEP": "0"}, {"EGS_QUAD": "0", "EGS_STEP": "1"}   # ticket / static-timetable tile kernels
    for name, env, prec in (("quad_f64", {"EGS_QUAD": "1", "EGS_STEP": "0"}, capi.F64), ("stepq_f64", {"EGS_QUAD": "1", "EGS_STEP": "1"}, capi.F64),
                            ("stepq_f32", {"EGS_QUAD": "1", "EGS_STEP": "1"}, capi.F32), ("tile_reg_f64", dict(T, EGS_ISO="0"), capi.F64),
                            ("tile_iso_f64", dict(T, EGS_ISO="2"), capi.F64),
                            ("step_reg_f64", dict(S, EGS_ISO="0"), capi.F64), ("step_iso_f64", dict(S, EGS_ISO="2"), capi.F64),
                            ("quad_f32", {"EGS_QUAD": "1", "EGS_STEP": "0"}, capi.F32), ("tile_reg_f32", dict(T, EGS_ISO="0"), capi.F32),
                            ("tile_iso_f32", dict(T, EGS_ISO="2"), capi.F32),
                            ("step_reg_f32", dict(S, EGS_ISO="0"), capi.F32), ("step_iso_f32", dict(S, EGS_ISO="2"), capi.F32)):
        one, sched, tile = chain_time(ctx, env, prec, False)
        two, _, _ = chain_time(ctx, env, prec, True)
        out["chain_update_us_" + name] = min(one, two)          # the bound bench.py uses: the fastest dependent update measured
        out["chain_update_us_" + name + "_one_sided"] = one
        out["chain_update_us_" + name + "_two_sided"] = two
        out["chain_schedule_" + name] = [sched, tile]
    ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
