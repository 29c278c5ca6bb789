#!/usr/bin/env python3
"""One workload, a few launches, nothing else: the program rocprofv3 wraps for the PMC passes
(FETCH_SIZE / WRITE_SIZE / LDS / wait counters), so that a kernel name maps to ONE problem size.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f_tile -- python3 tools/pmc_case.py tile
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w_tile -- python3 tools/pmc_case.py tile
    python3 tools/pmc_summary.py gpurun_out/pmc_f_tile gpurun_out/pmc_w_tile profiles/r02/pmc_traffic.json 393216 tile_solve_kernel

cases: tile (24 C3 piles, the bench headline) | quad (1 C3 pile) | matvec (64 C3 piles, the product)
       | coupled (the 41 x 40 wall, one island) | c4 (1024 x 64-body ensembles fp32) | c2 (one 256-body pile)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from eggshell_amd import capi, scenes  # noqa: E402


def main():
    case = sys.argv[1]
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    ctx = capi.Context(0)
    if case in ("tile", "quad", "c4", "c2"):
        wl = "c4" if case == "c4" else ("c2" if case == "c2" else "c3")
        seeds = bench.c4_shard_seeds(0, 1) if case == "c4" else ([b + 1 for b in range(24)] if case == "tile" else [1])
        r = bench.run_piles(ctx, wl, seeds, capi.GAUSS_SEIDEL, launches, 1)
        print("contacts", r["m"], "kernel", bench.solve_kernel_name(r["stats"]))
        r["problem"].close()
    elif case == "matvec":
        one = scenes.box_stack(16, 16, 16, jitter=1e-3, seed=1)
        sc = bench.replicate(one, 64)
        pr, _ = bench.build_problem(ctx, sc, capi.F64)
        pr.assemble(5e-3, 0.2)
        pr.solve(capi.params(method=capi.GAUSS_SEIDEL, max_iters=1, tol=0.0, cfm=0.01), want_stats=False)
        for _ in range(launches + 1):
            pr.matvec(None, capi.MV_FULL, 0.01, 1.0, fetch=False)
        ctx.synchronize()
        print("contacts", sc["kind"].shape[0], "kernel matvec_tile_kernel")
        pr.close()
    elif case == "coupled":
        sc = scenes.brick_wall(41, 40)
        b0, b1, data = ctx.update_contacts(sc["p"], sc["R"])
        sc.update(kind=np.full(len(b0), capi.CONTACT_BOX, np.int32), body0=b0, body1=b1, data=data)
        pr, _ = bench.build_problem(ctx, sc, capi.F64)
        prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=100, tol=0.0, cfm=0.01)
        for _ in range(launches):
            pr.step(5e-3, 0.2, prm)
        st = pr.stats()
        print("contacts", len(b0), "kernel", bench.solve_kernel_name(st))
        pr.close()
    else:
        raise SystemExit("unknown case " + case)
    ctx.close()


if __name__ == "__main__":
    main()
