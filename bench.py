#!/usr/bin/env python3
"""bench.py -- constraint-solve steps/sec on MI355X (BASELINE.json metric).

A "step" is one whole pass of the hot path over one batch of synthetic input:
device-side Jacobian assembly (K1-K4) + projected Gauss-Seidel sweeps (K5-K8) +
velocity update (K9) for `--batch` independent C3 piles resident in HBM
(16x16x16 boxes = 4096 bodies, 16384 contacts with the friction box, 100
sweeps, fp64).  Inputs are uploaded before the timed region; nothing is copied
back inside it.  `value` = pile-steps per second summed over all GPUs.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
      --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: independent piles are sharded one process per GPU (weak scaling, no
data-path collective); RCCL carries one statistics reduction per run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from eggshell_amd import capi, scenes  # noqa: E402
from eggshell_amd import dist as egs_dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy rate 6290
BYTES_PER_CONTACT_SWEEP = {"f64": 768, "f32": 388}

WORKLOADS = {
    # name: (nx, ny, nz, sweeps, precision, dt)
    "c3": (16, 16, 16, 100, "f64", 5e-3),
    "c2": (8, 8, 4, 50, "f64", 5e-3),
    "c4": (4, 4, 4, 50, "f32", 5e-3),
}


def host_mass_and_force(sc):
    """M^-1 blocks and f_ext exactly as Ensemble::Init freezes them
    (ensembles.cc:202-222) for axis-aligned boxes with I = 0.1*I3, w = 0:
    M^-1 = diag(1/m I3, I^-1), f = (m g, 0).  Host-side input preparation."""
    n = sc["p"].shape[0]
    Minv = np.zeros((n, 6, 6))
    for k in range(3):
        Minv[:, k, k] = 1.0 / sc["mass"]
    Ig = np.einsum("nij,njk,nlk->nil", sc["R"].reshape(n, 3, 3), sc["I_body"].reshape(n, 3, 3),
                   sc["R"].reshape(n, 3, 3))
    Minv[:, 3:, 3:] = np.linalg.inv(Ig)
    f = np.zeros((n, 6))
    f[:, 2] = sc["mass"] * -9.8
    return Minv.reshape(n, 36), f


def measured_traffic_per_contact(kernel):
    """HBM bytes per contact per launch of the solve kernel, measured with
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950 FETCH_SIZE
    correction applied) on this same command; see profiles/*/pmc_traffic.json."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json"))):
        try:
            d = json.load(open(f))
            v = d["hbm_bytes_per_contact_per_launch"].get(kernel)
            if v:
                best = (float(v), os.path.relpath(f, ROOT))
        except Exception:
            pass
    return best


def cpu_baseline(workload, budget_s):
    """Single-thread CPU port (oracle/, the fast O(nnz) sequential PGS in list
    order + assembly + velocity update) on ONE pile of the same workload."""
    from oracle import oracle as orc
    nx, ny, nz, sweeps, prec, dt = WORKLOADS[workload]
    sc = scenes.box_stack(nx, ny, nz)
    Minv = orc.minv_blocks(sc["R"], sc["mass"], sc["I_body"])
    f_ext = orc.external_force(sc["R"], sc["w"], sc["mass"], sc["I_body"])
    done, t0 = 0, time.perf_counter()
    while True:
        J0, J1, is_eq, lo, hi, err = orc.assemble(sc["p"], sc["R"], sc["kind"], sc["body0"], sc["body1"], sc["data"])
        s = orc.Sys(Minv, sc["body0"], sc["body1"], J0, J1, is_eq, lo, hi)
        rhs = orc.ode_rhs(sc["v"], sc["w"], Minv, f_ext, s.body0, s.body1, J0, J1, err, dt, 0.2)
        if prec == "f32":
            x, a, _, _ = orc.fast_iterate_f32(s, rhs, 0.01, orc.GAUSS_SEIDEL, max_iters=sweeps)
        else:
            x, a, _, _ = orc.fast_iterate(s, rhs, 0.01, orc.GAUSS_SEIDEL, max_iters=sweeps, tol=0.0)
        orc.velocity_update(sc["v"], sc["w"], Minv, f_ext, s.body0, s.body1, J0, J1, x, dt)
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 1000:
            break
    return {"value": done / el, "unit": "pile-steps/s", "cores": 1, "kind": "port",
            "sample": "%d steps of one %dx%dx%d pile (%d contacts, GS %d sweeps, %s), oracle/ fast O(nnz) "
                      "port, gcc -O2, 1 thread, %.1f s" % (done, nx, ny, nz, s.m, sweeps, prec, el)}


def run_workload(ctx, workload, batch, method, steps, warmup, rank, torch, tdist, world):
    """Build `batch` piles, upload, warm up, time exactly `steps` steps.  Returns a dict."""
    nx, ny, nz, sweeps, prec, dt = WORKLOADS[workload]
    precision = capi.F32 if prec == "f32" else capi.F64
    # `batch` independent piles per GPU; seeds differ per rank and pile (C4 style
    # jitter of whole columns) so no two piles are identical.
    piles = [scenes.box_stack(nx, ny, nz, jitter=1e-3, seed=rank * batch + b + 1,
                              origin=(0.0, 100.0 * b)) for b in range(batch)]
    sc = scenes.concat(piles) if batch > 1 else piles[0]
    n, m = sc["p"].shape[0], sc["kind"].shape[0]
    Minv, f_ext = host_mass_and_force(sc)
    t_plan = time.perf_counter()
    pr = capi.Problem(ctx, n, sc["body0"], sc["body1"], precision)
    t_plan = time.perf_counter() - t_plan
    pr.set_state(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext)
    pr.set_constraints(sc["kind"], sc["data"])
    prm = capi.params(method=method, max_iters=sweeps, tol=0.0, cfm=0.01)
    for _ in range(warmup):
        pr.step(dt, 0.2, prm)
    ctx.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        tdist.barrier()
    ctx.kernel_time(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        pr.step(dt, 0.2, prm)
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tdist.barrier()
    ksum_ms, klaunches = ctx.kernel_time(reset=True)
    st = pr.stats()
    pr.close()
    kernel_ms = ksum_ms / max(klaunches, 1)
    alg_bytes = float(m) * sweeps * BYTES_PER_CONTACT_SWEEP[prec]   # per launch (one rank's batch)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    traffic = measured_traffic_per_contact("tile_solve_kernel")
    roof = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": (traffic[0] * m) if traffic else None,
        "traffic_source": ("%s: %.0f B per contact per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, "
                           "measured on the tile_solve_kernel of this bench command)" % (traffic[1], traffic[0])) if traffic else None,
        "kernel": "quad_solve_kernel (+cons_prepare)" if st.reserved == 1 else "tile_solve_kernel",
        "kernel_ms": kernel_ms, "launches": klaunches, "algorithmic_bytes_per_launch": alg_bytes,
        "note": "algorithmic bytes = contacts x sweeps x %d B (SURVEY 8d); J blocks and body accumulators stay in "
                "VGPRs/LDS across sweeps, so the achieved figure can exceed HBM peak" % BYTES_PER_CONTACT_SWEEP[prec],
    }
    return dict(elapsed=elapsed, n=n, m=m, sweeps=sweeps, prec=prec, dt=dt, stats=st, t_plan=t_plan, roofline=roof,
                shape=(nx, ny, nz))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=24, help="independent piles resident per GPU (24 x 1024 columns = two full rounds of 3 tiles per CU)")
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--method", default="gs", choices=["gs", "sor"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only to rehearse N>1 on a 1-GPU box")
    ap.add_argument("--share-device0", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--no-single", action="store_true", help="skip the extra single-pile (latency) measurement")
    args = ap.parse_args()

    rank, world, local = egs_dist.env_rank_world()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    import torch
    import torch.distributed as tdist
    dev = 0 if (world == 1 or args.share_device0) else local
    if world > 1:
        torch.cuda.set_device(dev)
        egs_dist.init_process_group(args.dist_backend)
    ctx = capi.Context(dev)       # raises without the HIP library / a GPU: no fallback
    method = capi.GAUSS_SEIDEL if args.method == "gs" else capi.SOR

    r = run_workload(ctx, args.workload, args.batch, method, args.steps, args.warmup, rank, torch, tdist, world)
    st, m, sweeps = r["stats"], r["m"], r["sweeps"]
    el, units, citers, resid, failed = egs_dist.reduce_stats(
        r["elapsed"], args.batch * args.steps, float(m) * sweeps * args.steps, st.residual, st.status != capi.OK,
        device=("cuda:%d" % dev) if (world > 1 and args.dist_backend == "nccl") else None)
    single = None
    if world == 1 and args.batch != 1 and not args.no_single:
        single = run_workload(ctx, args.workload, 1, method, args.steps, args.warmup, rank, torch, tdist, world)

    if rank == 0:
        nx, ny, nz = r["shape"]
        out = {
            "metric": "constraint_solve_steps_per_sec",
            "value": units / el,
            "unit": "pile-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": r["prec"],
            "data": "synthetic",
            "config": {
                "workload": "%s: %dx%dx%d box pile = %d bodies, %d contacts (friction box), projected %s %d sweeps, "
                            "%s; step = assemble + solve + velocity update" % (
                                args.workload.upper(), nx, ny, nz, nx * ny * nz, m // args.batch,
                                "Gauss-Seidel" if method == capi.GAUSS_SEIDEL else "SOR(1.5)", sweeps, r["prec"]),
                "piles_per_gpu": args.batch, "sweeps": sweeps, "cfm": 0.01, "dt": r["dt"], "erp": 0.2,
                "islands_per_gpu": st.n_islands, "tiles_per_gpu": st.n_tiles,
                "schedule": "host plan (islands->tiles, %.1f ms) built once per contact topology, outside the "
                            "timed region" % (r["t_plan"] * 1e3),
                "parallelism": "piles sharded x%d, no data-path collective" % world,
            },
            "contact_iters_per_sec": citers / el,
            "max_residual": resid,
            "failed": failed,
            "roofline": r["roofline"],
        }
        if single is not None:
            out["single_pile"] = {
                "value": args.steps / single["elapsed"], "unit": "pile-steps/s",
                "ms_per_step": single["elapsed"] / args.steps * 1e3,
                "contact_iters_per_sec": float(single["m"]) * single["sweeps"] * args.steps / single["elapsed"],
                "roofline": single["roofline"],
                "note": "the same workload with ONE pile on the GPU (latency-bound: the dependency chain of a pile)",
            }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_seconds)
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
