#!/usr/bin/env python3
"""bench.py -- constraint-solve steps/sec on MI355X (BASELINE.json metric).

A "step" is one whole pass of the hot path over one batch of synthetic input:
device-side Jacobian assembly (K1-K4) + projected Gauss-Seidel sweeps (K5-K8) +
velocity update (K9).  Inputs are uploaded before the timed region; nothing is
copied back inside it.

Headline (`value`), BASELINE config 3: `--batch` independent 16x16x16 box piles
(4096 bodies, 16384 contacts with the friction box, 100 sweeps, fp64) resident
per GPU, weak scaling (every rank its own piles).  `--workload c4` makes BASELINE
config 4 the headline instead: 1024 independent 64-body ensembles (fp32, 50
sweeps), SHARDED over the ranks (eggshell_amd.dist.shard_range), strong scaling.

On one GPU the same JSON line carries every other BASELINE configuration and
the kernels the north star names, each with the physical bound that applies:
  single_pile   one C3 pile (the dependency chain of a pile: latency)
  matvec        the stand-alone block-sparse J M^-1 J^T product on C3 x batch
                (the one kernel of the path that streams: HBM roofline)
  c1            the Chain(8) ensemble of ensembles.cc (the reference's CPU-runnable case): latency only
  c2            256-body pile, 50 sweeps
  c4            1024 x 64-body ensembles, fp32
  coupled       ONE island of ~16k contacts (a running-bond wall): what a
                genuinely coupled pile costs
  c5            dense direct LCP, N = 2048 (MFMA fp64), CPU dense oracle beside it
  cpu_baseline  the CPU port of the same step on the host cores

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \\
      --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...          (no launcher: bench.py starts its own N ranks)

Multi-GPU: one process per GPU, no data-path collective; RCCL carries one
statistics reduction per run.
"""
import argparse
import glob
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from eggshell_amd import capi, scenes  # noqa: E402
from eggshell_amd import dist as egs_dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy rate 6290
CLOCK_GHZ = 2.4                # max shader clock (MI355X_MICROARCH.md chip-level parameters)
N_CU = 256
BYTES_PER_CONTACT_SWEEP = {"f64": 768, "f32": 388}   # the re-stream-every-sweep accounting
FLOPS_PER_CONTACT_SWEEP = 230
# compulsory HBM bytes of ONE stand-alone product per constraint:
# J0 + J1 (36 REAL), x and y (3 + 3 REAL), the schedule entry (12 B lane + 4 B amortised slot
# tables); per body the 6x6 block of M^-1 (36 REAL) and its 20-byte slot.
MV_BYTES_PER_CONTACT = {"f64": 42 * 8 + 16, "f32": 42 * 4 + 16}
MV_BYTES_PER_BODY = {"f64": 36 * 8 + 20, "f32": 36 * 4 + 20}

WORKLOADS = {
    # name: (nx, ny, nz, sweeps, precision, dt)
    "c3": (16, 16, 16, 100, "f64", 5e-3),
    "c2": (8, 8, 4, 50, "f64", 5e-3),
    "c4": (4, 4, 4, 50, "f32", 5e-3),
}
C4_ENSEMBLES = 1024


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


def load_profile_json(name):
    """Newest profiles/r*/<name> (measured on the GPU box, committed): PMC traffic, microbenchmarks."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name))):
        try:
            best = (json.load(open(f)), os.path.relpath(f, ROOT))
        except Exception:
            pass
    return best


def measured_traffic_per_contact(kernel):
    """HBM bytes per contact per launch of `kernel`, measured with rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE (separate passes, gfx950 FETCH_SIZE x2 correction) on this bench command."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json"))):
        try:
            v = json.load(open(f))["hbm_bytes_per_contact_per_launch"].get(kernel)
            if v:
                best = (float(v), os.path.relpath(f, ROOT))
        except Exception:
            pass
    return best


def load_counters(case, kernel):
    """Newest profiles/r*/counters.json entry for `case` of tools/pmc_case.py whose kernel is `kernel`: SQ counters
    (mean per launch) and HBM bytes of exactly that workload, written by tools/profile_r3.sh + profile_summary.py."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "counters.json"))):
        try:
            e = json.load(open(f))["cases"].get(case)
            k = e["kernels"].get(kernel) if e else None
            if k and "valu_insts_per_launch" in k:
                best = (k, e.get("contacts_per_launch"), os.path.relpath(f, ROOT))
        except Exception:
            pass
    return best


# One dependent constraint update of the 1-lane sweep kernels, counted on the ISA of step_solve_kernel<double,256,1,true>
# (DESIGN.md section 5): ds_read -> 3 (row product) + 2 (pair sums) + 1 (cfm fma) + 1 (rhs - ..) -> row 0: fma, cmp,
# cndmask, cmp, cndmask, sub = 6 -> row 1: 1 + 6 -> row 2: 2 + 6 -> 1 (the accumulator fma that waits for dx2) -> ds_write
# -> barrier.  29 dependent fp64 VALU operations and two LDS trips.
DEPENDENT_VALU_OPS_PER_UPDATE = {"lane": 29, "quad": 24}


def valu_roofline(kernel, kernel_ms, m, sweeps, prec, case, lat):
    """The hardware-anchored bound of a sweep kernel: VALU issue.  achieved = VALU wave-instructions per launch (rocprofv3
    SQ_INSTS_VALU of this very workload, committed under profiles/) / kernel time; peak = CUs x 4 SIMDs x clock / issue
    cycles per wave-instruction (fp64: measured on this hardware with every SIMD kept busy, tools/microbench;
    fp32: 64 lanes over a 16-wide SIMD).  Beside it: lane utilisation (how full the issued wavefronts were), the
    ISA critical-path floor of one update, and the old latency model as `overlap_efficiency` (how well the tiles'
    passes overlap -- measured against the kernel's own chain latency, hence not a roofline)."""
    base = kernel.split("(")[0]
    mb = load_profile_json("microbench.json")
    clock_hz = (mb[0].get("memtime_ticks_per_us", 2400.0) if mb else 2400.0) * 1e6
    if prec == "f64":
        cyc = float(mb[0].get("fma_f64_issue_cycles_4_waves_per_simd", 6.76)) if mb else 6.76
        cyc_src = ("%s:fma_f64_issue_cycles_4_waves_per_simd (v_fma_f64 back to back on every SIMD)" % mb[1]) if mb else "6.76 (DESIGN.md section 5)"
        dep = float(mb[0].get("fma_f64_dependent_cycles", 6.0)) if mb else 6.0
    else:
        cyc, cyc_src, dep = 4.0, "64 lanes / 16 lanes per cycle per SIMD (architectural; fp32 not micro-benchmarked)", 4.0
    peak = N_CU * 4 * clock_hz / cyc            # VALU wave-instructions per second, whole chip
    c = load_counters(case, base) if case else None
    out = {"bound": "valu_issue", "unit": "G wave-instructions/s", "peak": peak / 1e9, "issue_cycles_per_instruction": cyc,
           "issue_cycles_source": cyc_src, "clock_mhz": clock_hz / 1e6, "kernel": kernel, "kernel_ms": kernel_ms}
    if c:
        k, contacts, src = c
        scale = float(m) / contacts if contacts else 1.0
        valu = k["valu_insts_per_launch"] * scale
        out.update({"achieved": valu / (kernel_ms * 1e-3) / 1e9, "frac": valu * cyc / (N_CU * 4 * kernel_ms * 1e-3 * clock_hz),
                    "valu_insts_per_launch": valu, "lane_utilisation": k.get("lane_utilisation"),
                    "valu_insts_per_update": valu / (float(m) * sweeps), "counters_source": "%s: case '%s' (%d contacts per launch%s)" % (
                        src, case, contacts or 0, "" if abs(scale - 1.0) < 1e-9 else ", scaled x%.3f to this launch" % scale),
                    "traffic": k.get("hbm_bytes_per_launch", 0.0) * scale if k.get("hbm_bytes_per_launch") else None,
                    "wave_cycles_waiting_frac": (k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"]) if k.get("SQ_WAVE_CYCLES") else None})
    else:
        out.update({"achieved": None, "frac": None, "traffic": None,
                    "note_counters": "no committed SQ counters for this workload (tools/profile_r3.sh %s)" % (case or "<case>")})
    ops = DEPENDENT_VALU_OPS_PER_UPDATE["quad" if "quad" in base else "lane"]
    lds = float(mb[0].get("ds_read_b32_dependent_cycles", 60.0)) if mb else 60.0
    out["t_update_floor_us"] = (ops * dep + 2.0 * lds) / clock_hz * 1e6
    out["t_update_floor_model"] = "%d dependent VALU operations x %.2f cycles + 2 LDS trips x %.0f cycles (ISA critical path of one update)" % (ops, dep, lds)
    if lat:
        out["overlap_efficiency"] = lat["frac"]
        out["t_update_measured_us"] = lat["t_update_min_us"]
    return out


def sweep_critical_path(body0, body1, sweeps):
    """Length, in dependent constraint updates, of the list-order sweep's critical path:
    depth of one sweep's dependency DAG (a constraint waits for the previous constraint of
    each of its bodies) + (sweeps - 1) x the largest per-body constraint count (a body's
    constraints run one after the other in every sweep).  One island's worth of topology."""
    n = int(max(body0.max(), body1.max())) + 1
    last = np.zeros(n + 1, np.int64)
    cnt = np.zeros(n + 1, np.int64)
    depth = 0
    for a, b in zip(body0.tolist(), body1.tolist()):
        lv = max(last[a] if a >= 0 else 0, last[b] if b >= 0 else 0)
        if a >= 0:
            last[a] = lv + 1; cnt[a] += 1
        if b >= 0:
            last[b] = lv + 1; cnt[b] += 1
        depth = max(depth, lv + 1)
    return int(depth + (sweeps - 1) * cnt.max()), int(depth), int(cnt.max())


def solve_kernel_name(st):
    if st.schedule & capi.SCHED_QUAD:
        return "step_quad_kernel" if st.schedule & capi.SCHED_STATIC else "quad_solve_kernel"
    if st.schedule & capi.SCHED_QUAD_PATCHES:
        return "quad_solve_kernel(patches)"
    if st.schedule & capi.SCHED_LANE_PATCHES:
        return "patch_solve_kernel"
    if st.schedule & capi.SCHED_ALL_GLOBAL:
        return "global_solve_kernel"
    if st.schedule & capi.SCHED_STATIC:
        return "step_solve_kernel"
    return "tile_solve_kernel"


def resident_tiles_per_cu(st, prec):
    """Workgroup tiles one CU keeps resident (register-limited)."""
    if st.schedule & capi.SCHED_QUAD:
        return {64: 4, 128: 2, 256: 1}.get(st.tile_constraints, 1)      # 98 VGPRs (104 allocated): 4 wavefronts per SIMD
    if st.schedule & (capi.SCHED_QUAD_PATCHES | capi.SCHED_ALL_GLOBAL):
        return 1
    if st.schedule & capi.SCHED_LANE_PATCHES:
        return 2
    if st.schedule & capi.SCHED_ISO:
        return 4 if prec == "f32" else 3
    if prec == "f32":
        return 3
    return {64: 8, 128: 4, 256: 2, 512: 1}.get(st.tile_constraints, 1)   # 232 VGPRs: 2 wavefronts per SIMD


def rooflines(kernel, kernel_ms, launches, m, sweeps, prec, st, crit, case=None):
    """The bounds of one solve launch.  `roofline` (bound = latency): the launch cannot end before
    rounds x critical-path updates x t_update_min -- rounds = how often the CUs' resident tiles
    turn over, t_update_min = ONE always-ready constraint update of this kernel's instruction
    sequence, measured alone on a CU (tools/microbench.hip).  `roofline_hbm`: HBM bytes actually
    moved (rocprofv3 counters) against the HBM peak.  Plus achieved fp64 FLOP/s and the SURVEY 8(d)
    re-stream-every-sweep figure as information."""
    upd = float(m) * sweeps
    achieved = upd / (kernel_ms * 1e-3)
    mb = load_profile_json("microbench.json")
    base = kernel.split("(")[0]
    # t_update_min: ONE dependent constraint update of this kernel, hand-off included, measured on the real
    # kernel with tools/gpu_time_chain.py (one body with 64 world contacts = a chain of 64 K totally ordered
    # updates); patches and the all-global kernel are held against the in-LDS figure of their lane layout
    fam = "quad" if base == "quad_solve_kernel" else ("tile_iso" if st.schedule & capi.SCHED_ISO else "tile_reg")
    if base == "step_solve_kernel":     # static timetable: one update + one workgroup barrier per time step
        fam = "step_iso" if st.schedule & capi.SCHED_ISO else "step_reg"
    if base == "step_quad_kernel":
        fam = "stepq"
    key = "chain_update_us_%s_%s" % (fam, prec)
    t_two = None
    if mb and key in mb[0]:
        t_upd, src = float(mb[0][key]), ("%s:%s (64 K totally ordered updates on the real kernel, hand-offs included; the faster of the "
                                         "one-sided and the two-sided chain)" % (mb[1], key))
        t_two = mb[0].get(key + "_two_sided")
    else:   # ~140 fp64 instructions of one wavefront at ~6.8 cycles each
        t_upd, src = (0.36 if fam in ("quad", "stepq") else 0.47), "DESIGN.md section 5 (no microbench.json)"
    tiles_cu = resident_tiles_per_cu(st, prec)
    n_tiles = max(st.n_tiles, 1)
    rounds = max(1, math.ceil(n_tiles / float(tiles_cu * N_CU)))
    crit_updates, model = crit[0], "t_kernel >= rounds x critical_path x t_update_min"
    if base in ("step_solve_kernel", "step_quad_kernel"):     # the timetable's own length: the x0 accumulation is a time slot of its own
        crit_updates = crit[1] + crit[2] * sweeps
        model = "t_kernel >= rounds x time_steps x t_step_min, time_steps = sweep_depth + per_body_period x sweeps"
    t_min_ms = rounds * crit_updates * t_upd * 1e-3
    lat = {
        "bound": "latency", "unit": "constraint-updates/s", "achieved": achieved, "peak": upd / (t_min_ms * 1e-3),
        "frac": t_min_ms / kernel_ms, "traffic": None,
        "kernel": kernel, "kernel_ms": kernel_ms, "launches": launches,
        "model": model,
        "rounds": rounds, "resident_tiles_per_cu": tiles_cu, "tiles": n_tiles,
        "critical_path_updates": crit_updates, "sweep_depth": crit[1], "per_body_period": crit[2],
        "t_update_min_us": t_upd, "t_update_min_source": src,
        "achieved_fp64_tflops" if prec == "f64" else "achieved_fp32_tflops": achieved * FLOPS_PER_CONTACT_SWEEP / 1e12,
        "algorithmic_restream_gbs": achieved * BYTES_PER_CONTACT_SWEEP[prec] / 1e9,
        "note": "algorithmic_restream_gbs = updates/s x %d B (SURVEY 8d: a design that re-streams the system every sweep); "
                "J blocks and accumulators stay in VGPRs/LDS across sweeps, so it is information, not a bound" % BYTES_PER_CONTACT_SWEEP[prec],
    }
    if t_two:   # information: the same bound priced with body-body updates only (a pile has 15 of them per body-world one)
        lat["t_update_two_sided_us"] = float(t_two)
        lat["frac_two_sided_updates"] = rounds * crit_updates * float(t_two) * 1e-3 / kernel_ms
    if st.schedule & (capi.SCHED_QUAD_PATCHES | capi.SCHED_LANE_PATCHES | capi.SCHED_ALL_GLOBAL):
        lat["note_patches"] = ("the bound counts every hand-off at the in-LDS latency; an island cut into patches also pays ~4 us of global "
                               "memory at each patch switch of a body's constraint list (2-4 per body and sweep), DESIGN.md section 4")
    # counters are per workload (tools/pmc_case.py): "kernel(case)" if that case was measured, else the kernel's main case
    tr = (measured_traffic_per_contact("%s(%s)" % (kernel, case)) if case else None) or measured_traffic_per_contact(kernel) \
        or (measured_traffic_per_contact(base) if kernel == base else None)
    hbm = None
    if tr:
        traffic = tr[0] * m
        gbs = traffic / (kernel_ms * 1e-3) / 1e9
        hbm = {"bound": "hbm", "unit": "GB/s", "achieved": gbs, "peak": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS,
               "traffic": traffic, "traffic_source": "%s: %.0f B per contact per launch of %s (rocprofv3 --pmc FETCH_SIZE x2 + "
               "WRITE_SIZE, separate passes)" % (tr[1], tr[0], base)}
        lat["traffic"] = traffic
    return lat, hbm


def hbm_roofline_from_counters(kernel, kernel_ms, m, case):
    c = load_counters(case, kernel.split("(")[0])
    if not c or not c[0].get("hbm_bytes_per_launch"):
        return None
    k, contacts, src = c
    traffic = k["hbm_bytes_per_launch"] * (float(m) / contacts if contacts else 1.0)
    gbs = traffic / (kernel_ms * 1e-3) / 1e9
    return {"bound": "hbm", "unit": "GB/s", "achieved": gbs, "peak": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": "%s: case '%s', %.0f B per contact per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
                              % (src, case, traffic / m)}


def build_problem(ctx, sc, precision):
    n = sc["p"].shape[0]
    Minv, f_ext = host_mass_and_force(sc)
    t_plan = time.perf_counter()
    pr = capi.Problem(ctx, n, sc["body0"], sc["body1"], precision)
    t_plan = time.perf_counter() - t_plan
    pr.set_state(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext)
    pr.set_constraints(sc["kind"], sc["data"])
    return pr, t_plan


def time_region(ctx, fn, steps, warmup, torch=None, tdist=None, world=1):
    """W untimed calls, then exactly `steps` calls between barrier + synchronize on both sides."""
    for _ in range(warmup):
        fn()
    ctx.synchronize()
    if torch is not None:
        torch.cuda.synchronize()
    if world > 1:
        tdist.barrier()
    ctx.kernel_time(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    ctx.synchronize()
    if torch is not None:
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tdist.barrier()
    ksum_ms, klaunches = ctx.kernel_time(reset=True)
    return elapsed, ksum_ms / max(klaunches, 1), klaunches


def run_piles(ctx, workload, seeds, method, steps, warmup, torch=None, tdist=None, world=1):
    """`len(seeds)` piles of `workload` in one problem; time exactly `steps` steps."""
    nx, ny, nz, sweeps, prec, dt = WORKLOADS[workload]
    piles = [scenes.box_stack(nx, ny, nz, jitter=1e-3, seed=int(sd), origin=(0.0, 100.0 * k)) for k, sd in enumerate(seeds)]
    sc = scenes.concat(piles) if len(piles) > 1 else piles[0]
    m = sc["kind"].shape[0]
    pr, t_plan = build_problem(ctx, sc, capi.F32 if prec == "f32" else capi.F64)
    prm = capi.params(method=method, max_iters=sweeps, tol=0.0, cfm=0.01)
    elapsed, kernel_ms, launches = time_region(ctx, lambda: pr.step(dt, 0.2, prm), steps, warmup, torch, tdist, world)
    st = pr.stats()          # the stall flag is sticky: this covers every step of the run
    # a pile is nx*ny independent columns: the critical path is one column's
    col = nx * ny
    one = piles[0]
    keep = (np.where(one["body0"] >= 0, one["body0"], one["body1"]) % col) == 0
    crit = sweep_critical_path(np.where(one["body0"][keep] >= 0, one["body0"][keep] // col, -1),
                               np.where(one["body1"][keep] >= 0, one["body1"][keep] // col, -1), sweeps)
    kname = solve_kernel_name(st)
    lat, hbm = rooflines(kname, kernel_ms, launches, m, sweeps, prec, st, crit, case="c4" if workload == "c4" else None)
    # which committed counter set belongs to this launch (tools/pmc_case.py): the headline batch, one pile, C2, C4
    case = {"c4": "c4", "c2": "c2"}.get(workload, "quad" if len(piles) == 1 else "tile")
    valu = valu_roofline(kname, kernel_ms, m, sweeps, prec, case, lat)
    hbm = hbm_roofline_from_counters(kname, kernel_ms, m, case) or hbm
    return dict(elapsed=elapsed, n=sc["p"].shape[0], m=m, sweeps=sweeps, prec=prec, dt=dt, stats=st, t_plan=t_plan,
                roofline=valu, roofline_hbm=hbm, latency_model=lat, shape=(nx, ny, nz), problem=pr, scene=sc, piles=len(piles))


def leg_from_run(r, steps, unit="pile-steps/s", note=None):
    out = {"value": r["piles"] * steps / r["elapsed"], "unit": unit, "ms_per_step": r["elapsed"] / steps * 1e3,
           "contact_iters_per_sec": float(r["m"]) * r["sweeps"] * steps / r["elapsed"],
           "contacts": r["m"], "sweeps": r["sweeps"], "dtype": r["prec"], "islands": r["stats"].n_islands,
           "failed": r["stats"].status != capi.OK, "max_residual": r["stats"].residual,
           "roofline": r["roofline"], "roofline_hbm": r["roofline_hbm"], "latency_model": r["latency_model"]}
    if note:
        out["note"] = note
    return out


def replicate(sc, times):
    """`times` copies of a scene side by side (body indices offset; numpy only, so a million contacts
    take a moment instead of a minute of Python loops)."""
    n = sc["p"].shape[0]
    out = {}
    for k in ("p", "R", "v", "w", "mass", "I_body", "kind", "data"):
        out[k] = np.concatenate([sc[k]] * times)
    shift = np.repeat(np.arange(times, dtype=np.int64) * 1000.0, n)
    out["p"] = out["p"].copy(); out["p"][:, 1] += shift
    out["data"] = out["data"].copy(); out["data"][:, 1] += np.repeat(np.arange(times) * 1000.0, sc["kind"].shape[0])
    off = np.repeat(np.arange(times, dtype=np.int64) * n, sc["kind"].shape[0])
    b0, b1 = np.tile(sc["body0"], times).astype(np.int64), np.tile(sc["body1"], times).astype(np.int64)
    out["body0"] = np.where(b0 >= 0, b0 + off, -1).astype(np.int32)
    out["body1"] = np.where(b1 >= 0, b1 + off, -1).astype(np.int32)
    return out


def matvec_measure(ctx, pr, m, n, prec, steps, warmup):
    elapsed, kernel_ms, launches = time_region(ctx, lambda: pr.matvec(None, capi.MV_FULL, 0.01, 1.0, fetch=False), steps, warmup)
    alg = float(m) * MV_BYTES_PER_CONTACT[prec] + float(n) * MV_BYTES_PER_BODY[prec]
    gbs = alg / (kernel_ms * 1e-3) / 1e9
    tr = measured_traffic_per_contact("matvec_tile_kernel")
    return {
        "value": steps / elapsed, "unit": "products/s", "ms_per_product": elapsed / steps * 1e3, "contacts": m, "bodies": n,
        "contacts_per_sec": float(m) * steps / elapsed,
        "roofline": {"bound": "hbm", "unit": "GB/s", "achieved": gbs, "peak": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS,
                     "traffic": (tr[0] * m) if tr else None,
                     "traffic_source": ("%s: %.0f B per contact per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)" % (tr[1], tr[0])) if tr else None,
                     "kernel": "matvec_tile_kernel", "kernel_ms": kernel_ms, "launches": launches,
                     "algorithmic_bytes_per_launch": alg}}


def matvec_leg(ctx, r, steps, warmup, big_piles=64):
    """The stand-alone product y = (J M^-1 J^T + cfm I) lambda (sparse::CalculateSparseJMJtX): the kernel of the
    path that streams, hence the HBM roofline.  Measured on `big_piles` C3 piles (1 M contacts, ~450 MB per
    product: beyond the 256 MiB Infinity Cache, so every launch really reads HBM) and, for comparison, on the
    resident problem of the main leg (170 MB: re-read from the Infinity Cache launch after launch)."""
    prec = r["prec"]
    one = scenes.box_stack(*r["shape"], jitter=1e-3, seed=1)
    sc = replicate(one, big_piles)
    pr, _ = build_problem(ctx, sc, capi.F32 if prec == "f32" else capi.F64)
    pr.assemble(r["dt"], 0.2)
    pr.solve(capi.params(method=capi.GAUSS_SEIDEL, max_iters=1, tol=0.0, cfm=0.01), want_stats=False)   # a non-trivial lambda to multiply
    out = matvec_measure(ctx, pr, sc["kind"].shape[0], sc["p"].shape[0], prec, steps, warmup)
    pr.close()
    out["workload"] = "%d C3 piles in one problem (identical copies), full product, lambda of one GS sweep" % big_piles
    out["roofline"]["note"] = ("algorithmic bytes = compulsory traffic of one product: %d B per contact (J0, J1, x, y, schedule entry) + "
                               "%d B per body (M^-1 block, slot): every byte read or written once; SURVEY 8(d)'s 768 B figure also "
                               "counts the body sums, which stay in LDS here" % (MV_BYTES_PER_CONTACT[prec], MV_BYTES_PER_BODY[prec]))
    small = matvec_measure(ctx, r["problem"], r["m"], r["n"], prec, steps, warmup)
    out["infinity_cache_resident"] = {"contacts": r["m"], "ms_per_product": small["ms_per_product"],
                                      "achieved_gbs": small["roofline"]["achieved"], "kernel_ms": small["roofline"]["kernel_ms"],
                                      "note": "the main leg's problem (%.0f MB per product < 256 MiB): served by the Infinity Cache, "
                                              "not a statement about HBM" % (small["roofline"]["algorithmic_bytes_per_launch"] / 1e6)}
    return out


def c4_shard_seeds(rank, world):
    """BASELINE config 4: ensembles 0..1023, seeds = the global ensemble index, contiguous shards."""
    begin, end = egs_dist.shard_range(C4_ENSEMBLES, rank, world)
    return list(range(begin, end))


def coupled_leg(ctx, method, steps, warmup, cpu_seconds):
    """ONE island: a 41 x 40 running-bond wall (every brick rests on two), ~16k contacts, GS 100 sweeps fp64."""
    from oracle import oracle as orc
    sc = scenes.brick_wall(41, 40)
    b0, b1, data = ctx.update_contacts(sc["p"], sc["R"])
    sc.update(kind=np.full(len(b0), capi.CONTACT_BOX, np.int32), body0=b0, body1=b1, data=data)
    m, sweeps, dt = len(b0), 100, 5e-3
    pr, t_plan = build_problem(ctx, sc, capi.F64)
    prm = capi.params(method=method, max_iters=sweeps, tol=0.0, cfm=0.01)
    elapsed, kernel_ms, launches = time_region(ctx, lambda: pr.step(dt, 0.2, prm), steps, warmup)
    st = pr.stats()
    crit = sweep_critical_path(sc["body0"], sc["body1"], sweeps)
    kname = solve_kernel_name(st)
    lat, hbm = rooflines(kname, kernel_ms, launches, m, sweeps, "f64", st, crit)
    valu = valu_roofline(kname, kernel_ms, m, sweeps, "f64", "coupled", lat)
    hbm = hbm_roofline_from_counters(kname, kernel_ms, m, "coupled") or hbm
    compulsory = float(m) * (36 + 3 + 3 + 3 + 3 + 3) * 8 + float(sc["p"].shape[0]) * (36 + 6) * 8     # J, rhs, bounds, lambda, w in/out; M^-1 block, accumulator
    if hbm:
        hbm["compulsory_bytes"] = compulsory
        hbm["traffic_over_compulsory"] = hbm["traffic"] / compulsory
    out = {"value": steps / elapsed, "unit": "pile-steps/s", "ms_per_step": elapsed / steps * 1e3, "bodies": sc["p"].shape[0],
           "contacts": m, "islands": st.n_islands, "sweeps": sweeps, "contact_iters_per_sec": float(m) * sweeps * steps / elapsed,
           "failed": st.status != capi.OK, "roofline": valu, "roofline_hbm": hbm, "latency_model": lat,
           "workload": "41 x 40 running-bond brick wall on the ground, contacts from the device collider: ONE island"}
    if cpu_seconds > 0:
        Minv = orc.minv_blocks(sc["R"], sc["mass"], sc["I_body"])
        f_ext = orc.external_force(sc["R"], sc["w"], sc["mass"], sc["I_body"])
        with orc.timing_build() as flags:
            done, t0 = 0, time.perf_counter()
            while True:
                J0, J1, is_eq, lo, hi, err = orc.assemble(sc["p"], sc["R"], sc["kind"], sc["body0"], sc["body1"], sc["data"])
                s = orc.Sys(Minv, sc["body0"], sc["body1"], J0, J1, is_eq, lo, hi)
                rhs = orc.ode_rhs(sc["v"], sc["w"], Minv, f_ext, s.body0, s.body1, J0, J1, err, dt, 0.2)
                x, a, _, _ = orc.fast_iterate(s, rhs, 0.01, method, max_iters=sweeps, tol=0.0)
                orc.velocity_update(sc["v"], sc["w"], Minv, f_ext, s.body0, s.body1, J0, J1, x, dt)
                done += 1
                el = time.perf_counter() - t0
                if el >= cpu_seconds or done >= 200:
                    break
        out["cpu_baseline"] = {"value": done / el, "unit": "pile-steps/s", "cores": 1, "kind": "port",
                               "sample": "%d steps of the same wall, oracle/ fast O(nnz) port, 1 thread, %.1f s; build: %s" % (done, el, flags)}
        out["gpu_over_cpu_1_core"] = out["value"] / out["cpu_baseline"]["value"]
    pr.close()
    return out


def c5_problem(N, seed=0):
    """SURVEY 8(d) C5 recipe: A = M^T M, M ~ U(-1,1) (GenerateSPDMatrix, utils.cc:203-215; + 1e-3 I if
    ill-conditioned), b ~ U(-1,1), C ~ Bernoulli(1/2), lo = 0, hi = inf."""
    rng = np.random.default_rng(seed)
    M = rng.uniform(-1, 1, (N, N))
    A = M.T @ M
    if N <= 512 and not np.linalg.cond(A) < 1e7:
        A = A + 1e-3 * np.eye(N)
    elif N > 512:
        A = A + 1e-3 * np.eye(N)     # the condition check itself (an SVD of 2048^2) is not part of the timed path
    b = rng.uniform(-1, 1, N)
    C = (rng.uniform(size=N) < 0.5).astype(np.uint8)
    return A, b, C, np.zeros(N), np.full(N, np.inf)


def c5_leg(ctx, cpu_seconds):
    """BASELINE config 5: dense direct LCP (Lcp::MixedConstraintsSolver semantics), N = 2048, fp64 MFMA."""
    from oracle import oracle as orc
    out = {}
    for N, mode, reps in ((2048, 2, 3), (512, 2, 3), (512, 0, 1), (256, 0, 1)):
        A, b, C, lo, hi = c5_problem(N)
        ok, x, w, piv = ctx.mixed_constraints_solve(A, b, C, lo, hi, use_bounds=mode)    # warm-up (code objects, buffers)
        ms = float("inf")
        for _ in range(reps):      # best of `reps`: each call allocates, uploads pageable memory, solves, downloads
            t0 = time.perf_counter()
            ok, x, w, piv = ctx.mixed_constraints_solve(A, b, C, lo, hi, use_bounds=mode)
            ms = min(ms, (time.perf_counter() - t0) * 1e3)
        ne = int(C.sum()); ni = N - ne
        flops = ne ** 3 / 3.0 + 2.0 * ne * ne * ni + ni * ni * ne       # the Schur stage
        resid = float(np.abs(A @ x - b - w).max())
        key = "N%d_%s" % (N, "block_pivoting" if mode == 2 else "reference_rule")
        out[key] = {"ms_per_solve": ms, "ok": bool(ok), "pivots": int(piv), "kkt_residual": resid,
                    "schur_gflop": flops / 1e9, "achieved_tflops_schur_only": flops / (ms * 1e-3) / 1e12,
                    "includes": "pageable upload of A (%.1f MB) + solve + download" % (A.nbytes / 1e6)}
    # toolkit/lcp.cc's incremental-factor solvers (one wavefront, everything in LDS, n <= 96): latency only
    A, b, C, lo, hi = c5_problem(96)
    lo96 = np.where(C != 0, -np.inf, np.minimum(lo, 0.0)); hi96 = np.where(C != 0, np.inf, np.maximum(hi, 0.5))
    for name, fn in (("N96_box_dantzig_incremental", ctx.box_lcp_dantzig), ("N96_box_murty_linear_reducer", ctx.box_lcp_murty)):
        fn(np.tril(A), b, lo96, hi96)
        t0 = time.perf_counter()
        ok, x, w, Ap, perm, piv = fn(np.tril(A), b, lo96, hi96)
        ofn = orc.tk_box_dantzig if "dantzig" in name else orc.tk_box_murty
        tc = time.perf_counter()
        ofn(np.tril(A), b, lo96, hi96)
        tc = time.perf_counter() - tc
        out[name] = {"ms_per_solve": (time.perf_counter() - t0 - tc) * 1e3, "ok": bool(ok), "pivots": int(piv),
                     "cpu_port_ms_per_solve": tc * 1e3,
                     "kkt_residual": float(np.abs(A @ x - b - w).max()),
                     "includes": "upload + one single-wavefront launch + download (toolkit/lcp.cc:213-619 semantics, A permuted in place)"}
    out["note"] = ("reference_rule = Murty single-index principal pivoting as lcp.cc:157-274 (cap min(1000, 2^n) pivots: it "
                   "cannot finish N >= 1024 mixed problems, in the reference as here); block_pivoting = same solution, tens of "
                   "factorisations, starting from the set the diagonal suggests; pivots that move <= 64 indexes solve a bordered "
                   "system on the previous factor.  Latency-bound at these sizes: one launch of ~23 us per 64-column panel (the "
                   "diagonal tile's column chain); fp64 MFMA peak is not in the local guides, so TFLOP/s is reported, not a fraction")
    if cpu_seconds > 0:
        cpu = {}
        for N in (256, 512):
            A, b, C, lo, hi = c5_problem(N)
            t0 = time.perf_counter()
            ok, x, w, piv = orc.mixed_constraints(A, b, C, lo, hi, 0)
            cpu["N%d" % N] = {"ms_per_solve": (time.perf_counter() - t0) * 1e3, "ok": bool(ok), "pivots": int(piv)}
        out["cpu_baseline"] = {"kind": "port", "cores": 1, "unit": "ms/solve", "sample": "oracle/lcp_dense.c (restated "
                               "MixedConstraintsSolver + Murty, lcp.cc:141-336), one solve each", **cpu}
    return out


def lcp_batch_leg(ctx, cpu_seconds, count=2048, n=24):
    """toolkit/lcp.cc's incremental-factor box LCP as a BATCH: `count` independent problems of n rows (8 contacts x 3
    with the friction box of contact.cc:103-113) in one launch, one workgroup per problem (egs_box_lcp_batch) -- what a
    batch of ensembles hands lcp::SolveLCP; the CPU figure is the oracle's restatement of the same algorithm."""
    from oracle import oracle as orc
    rng = np.random.default_rng(5)
    As, bs, los, his = [], [], [], []
    for k in range(count):
        M = rng.uniform(-1, 1, (n, n))
        As.append(np.tril(M @ M.T + 0.1 * np.eye(n))); bs.append(rng.uniform(-1, 1, n))
        los.append(np.tile([-1.0, -1.0, 0.0], n // 3)); his.append(np.tile([1.0, 1.0, np.inf], n // 3))
    out = {"problems": count, "rows": n, "includes": "upload of the %d matrices + one launch + download (host arrays in, host arrays out)" % count}
    ns = np.full(count, n, np.int32)
    Ap_, bp_, lop_, hip_ = np.concatenate([a.reshape(-1) for a in As]), np.concatenate(bs), np.concatenate(los), np.concatenate(his)
    for name, alg in (("box_dantzig", 1), ("box_murty", 0)):
        ctx.box_lcp_batch(alg, As[:8], bs[:8], los[:8], his[:8])
        best = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            ok, x, w, Ap, perm, piv = ctx.box_lcp_batch_packed(alg, ns, Ap_, bp_, lop_, hip_)
            best = min(best, time.perf_counter() - t0)
        out[name] = {"problems_per_sec": count / best, "ms_per_batch": best * 1e3, "all_ok": bool(all(ok)), "pivots_mean": float(np.mean(piv))}
        if cpu_seconds > 0:
            fn = orc.tk_box_dantzig if alg == 1 else orc.tk_box_murty
            with orc.timing_build() as flags:
                done, t0 = 0, time.perf_counter()
                while done < count and time.perf_counter() - t0 < min(2.0, cpu_seconds):
                    fn(As[done], bs[done], los[done], his[done])
                    done += 1
                el = time.perf_counter() - t0
            out[name]["cpu_baseline"] = {"value": done / el, "unit": "problems/s", "cores": 1, "kind": "port",
                                         "sample": "%d of the same problems, oracle/lcp_toolkit.c, 1 thread, %.2f s (ctypes call overhead included); build: %s" % (done, el, flags)}
            out[name]["gpu_over_cpu_1_core"] = out[name]["problems_per_sec"] / (done / el)
    return out


def world_leg(ctx, steps=30):
    """Ensemble::Step() resident on the device (egs_world: contact generation with the
    uniform-grid broad phase -> re-plan only when the contact topology changed -> assemble -> 100 GS sweeps ->
    velocity -> position integration), one C3 pile under gravity, body state never leaves the GPU."""
    nx, ny, nz, sweeps, prec, dt = WORKLOADS["c3"]
    sc = scenes.box_stack(nx, ny, nz, jitter=1e-3, seed=1)
    Minv, f_ext = host_mass_and_force(sc)
    n = sc["p"].shape[0]
    w = capi.World(ctx, n)
    w.set_bodies(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext)
    prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=sweeps, tol=0.0, cfm=0.01)
    for _ in range(3):
        w.step(dt, 0.2, prm)
    r0 = w.info()["replans"]
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        w.step(dt, 0.2, prm)
    ctx.synchronize()
    el = time.perf_counter() - t0
    info = w.info()
    st = w.step(dt, 0.2, prm, want_stats=True)
    w.close()
    return {"value": steps / el, "unit": "world-steps/s", "ms_per_step": el / steps * 1e3, "bodies": n, "contacts": info["n_contacts"],
            "replans_in_timed_steps": info["replans"] - r0, "sweeps": sweeps, "failed": st.status != capi.OK,
            "note": "collide + (re-plan on change) + assemble + solve + velocity + positions per step, host sees 8 B per contact of "
                    "topology per step; the solve alone is the `single_pile` leg"}


def stopping_loop_leg(ctx):
    """The reference's own iteration loop (sparse_iterations.cc:204-221: residual after every sweep, stop at
    err <= 1e-9 or after 500 sweeps) on C3: these piles do not converge in 500 sweeps, so this is the cost of
    500 sweeps WITH the per-sweep stopping test, which the library evaluates from recorded snapshots."""
    nx, ny, nz, sweeps, prec, dt = WORKLOADS["c3"]
    out = {}
    for piles in (1, 24):
        sc = scenes.concat([scenes.box_stack(nx, ny, nz, jitter=1e-3, seed=k + 1, origin=(0.0, 100.0 * k)) for k in range(piles)]) \
            if piles > 1 else scenes.box_stack(nx, ny, nz, jitter=1e-3, seed=1)
        pr, _ = build_problem(ctx, sc, capi.F64)
        pr.assemble(dt, 0.2)
        prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=500, tol=1e-9, cfm=0.01)
        st = pr.solve(prm)
        best = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            st = pr.solve(prm)
            best = min(best, time.perf_counter() - t0)
        out["c3_x%d" % piles] = {"ms_per_solve": best * 1e3, "sweeps": int(st.iterations), "residual": float(st.residual),
                                 "us_per_sweep": best * 1e6 / max(int(st.iterations), 1)}
        pr.close()
    out["note"] = "tol 1e-9, cap 500, residual evaluated after every sweep (from snapshots: one host synchronisation per launch of 64, 128, then 256 recorded sweeps)"
    return out


def c1_leg(ctx, cpu_seconds):
    """BASELINE config 1: the single Chain ensemble of ensembles.cc (8 bodies, 7 ball joints + the anchor),
    fp64 -- the reference's own CPU-runnable case.  Through the GPU library: the iterative path with the
    reference's constants (SOR, omega 1.5, tol 1e-9, at most 500 sweeps; x0 = rhs) and one step through the
    dense path (ComputeVDot: dense J M^-1 J^T, condition estimate, MixedConstraintsSolver).  Latency only:
    24 rows cannot fill a GPU."""
    from oracle import oracle as orc
    sc = scenes.chain(8)
    pr, _ = build_problem(ctx, sc, capi.F64)
    prm = capi.params(method=capi.SOR, max_iters=500, tol=1e-9, cfm=0.1)      # cfm 0.1 as the reference's tests (sparse_iterations.cc:305)
    pr.assemble(1e-3, 0.2)
    st = pr.solve(prm)
    reps, t0 = 20, time.perf_counter()
    for _ in range(reps):
        st = pr.solve(prm)
    ms_iter = (time.perf_counter() - t0) / reps * 1e3
    ok, piv = pr.step_dense(1e-3, 0.2, 0.0)
    t0 = time.perf_counter()
    for _ in range(reps):
        ok, piv = pr.step_dense(1e-3, 0.2, 0.0)
    ms_dense = (time.perf_counter() - t0) / reps * 1e3
    pr.close()
    out = {"workload": "Chain(8, anchor (0,0,2)): 8 bodies, 8 ball joints (7 + the anchor), 24 rows, dt = 1e-3",
           "iterative": {"ms_per_solve": ms_iter, "sweeps": st.iterations, "residual": st.residual, "method": "SOR(1.5), tol 1e-9, cfm 0.1",
                         "includes": "the reference's stopping loop (residual after every sweep), host-synchronous"},
           "dense": {"ms_per_step": ms_dense, "ok": bool(ok), "pivots": int(piv),
                     "includes": "assemble + dense system + MixedConstraintsSolver + velocity update, host-synchronous"}}
    if cpu_seconds > 0:
        J0, J1, is_eq, lo, hi, err = orc.assemble(sc["p"], sc["R"], sc["kind"], sc["body0"], sc["body1"], sc["data"])
        Minv = orc.minv_blocks(sc["R"], sc["mass"], sc["I_body"])
        f_ext = orc.external_force(sc["R"], sc["w"], sc["mass"], sc["I_body"])
        s = orc.Sys(Minv, sc["body0"], sc["body1"], J0, J1, is_eq, lo, hi)
        rhs = orc.ode_rhs(sc["v"], sc["w"], Minv, f_ext, s.body0, s.body1, J0, J1, err, 1e-3, 0.2)
        t0 = time.perf_counter()
        for _ in range(50):
            x, it, res = orc.lit_iterate(s, rhs, 0.1, orc.SOR)
        out["cpu_baseline"] = {"kind": "port", "cores": 1, "unit": "ms/solve", "literal_ms_per_solve": (time.perf_counter() - t0) / 50 * 1e3,
                               "sweeps": it, "sample": "50 solves, oracle/sparse_literal.c (the reference's O(m^2) pair loops), 1 thread"}
        t0 = time.perf_counter()
        for _ in range(200):
            x, a, it2, res = orc.fast_iterate(s, rhs, 0.1, orc.SOR)
        out["cpu_baseline"]["fast_ms_per_solve"] = (time.perf_counter() - t0) / 200 * 1e3
    return out


def cpu_baseline(workload, budget_s):
    """Single-thread CPU port (oracle/, the fast O(nnz) sequential PGS in list order + assembly +
    velocity update) on ONE pile of the same workload.  Timed on the oracle sources built with the REFERENCE's
    flags (liboracle_refflags.so: -O2 -DNDEBUG, GCC's default contraction); the parity build (-ffp-contract=off,
    the checker) is timed beside it for 2 s."""
    from oracle import oracle as orc
    nx, ny, nz, sweeps, prec, dt = WORKLOADS[workload]
    sc = scenes.box_stack(nx, ny, nz)

    def run(budget):
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
            if el >= budget or done >= 1000:
                return done, el, s.m
    with orc.timing_build() as flags:
        done, el, m = run(budget_s)
    done_p, el_p, _ = run(min(2.0, budget_s))
    unit = "ensemble-steps/s" if workload == "c4" else "pile-steps/s"
    return {"value": done / el, "unit": unit, "cores": 1, "kind": "port",
            "sample": "%d steps of one %dx%dx%d pile (%d contacts, GS %d sweeps, %s), oracle/ fast O(nnz) "
                      "port, 1 thread, %.1f s; build: %s" % (done, nx, ny, nz, m, sweeps, prec, el, flags),
            "parity_build_value": done_p / el_p, "parity_build_flags": orc.BUILD_FLAGS["parity"]}


def cpu_baseline_threads(workload, budget_s, threads):
    """The same port on `threads` host threads at once, one pile each (the oracle's C functions run
    outside the GIL): a MEASURED multi-core figure for the cores this process may use."""
    import threading
    from oracle import oracle as orc
    nx, ny, nz, sweeps, prec, dt = WORKLOADS[workload]
    counts, t_start = [0] * threads, time.perf_counter()

    def worker(k):
        sc = scenes.box_stack(nx, ny, nz, seed=k + 1)
        Minv = orc.minv_blocks(sc["R"], sc["mass"], sc["I_body"])
        f_ext = orc.external_force(sc["R"], sc["w"], sc["mass"], sc["I_body"])
        while time.perf_counter() - t_start < budget_s:
            J0, J1, is_eq, lo, hi, err = orc.assemble(sc["p"], sc["R"], sc["kind"], sc["body0"], sc["body1"], sc["data"])
            s = orc.Sys(Minv, sc["body0"], sc["body1"], J0, J1, is_eq, lo, hi)
            rhs = orc.ode_rhs(sc["v"], sc["w"], Minv, f_ext, s.body0, s.body1, J0, J1, err, dt, 0.2)
            if prec == "f32":
                x = orc.fast_iterate_f32(s, rhs, 0.01, orc.GAUSS_SEIDEL, max_iters=sweeps)[0]
            else:
                x = orc.fast_iterate(s, rhs, 0.01, orc.GAUSS_SEIDEL, max_iters=sweeps, tol=0.0)[0]
            orc.velocity_update(sc["v"], sc["w"], Minv, f_ext, s.body0, s.body1, J0, J1, x, dt)
            counts[k] += 1

    with orc.timing_build():          # the reference-flag build, as cpu_baseline()
        t_start = time.perf_counter()
        ts = [threading.Thread(target=worker, args=(k,)) for k in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    el = time.perf_counter() - t_start
    return {"value": sum(counts) / el, "unit": "ensemble-steps/s" if workload == "c4" else "pile-steps/s", "cores": threads,
            "kind": "port", "sample": "%d steps on %d threads, one %dx%dx%d pile each, %.1f s (measured)" % (sum(counts), threads, nx, ny, nz, el)}


def cpu_literal_baseline(budget_s):
    """SURVEY 8(d) baseline (i): the LITERAL O(m^2) restatement -- the reference's real cost, without its
    virtual calls and mallocs -- in full at C1 and C2, one sweep at C3 (extrapolated to 100)."""
    from oracle import oracle as orc
    out = {"kind": "port", "cores": 1, "unit": "s/solve"}
    t_all = time.perf_counter()

    def lit(sc, sweeps, cfm):
        Minv = orc.minv_blocks(sc["R"], sc["mass"], sc["I_body"])
        J0, J1, is_eq, lo, hi, err = orc.assemble(sc["p"], sc["R"], sc["kind"], sc["body0"], sc["body1"], sc["data"])
        s = orc.Sys(Minv, sc["body0"], sc["body1"], J0, J1, is_eq, lo, hi)
        rhs = np.random.default_rng(0).uniform(-1, 1, 3 * s.m)
        t0 = time.perf_counter()
        orc.lit_iterate(s, rhs, cfm, orc.GAUSS_SEIDEL, max_iters=sweeps, tol=0.0)
        return time.perf_counter() - t0, s.m
    t, m = lit(scenes.chain(8), 500, 0.0)
    out["C1_chain8_500_sweeps"] = t
    t, m = lit(scenes.box_stack(8, 8, 4), 50, 0.01)
    out["C2_50_sweeps"] = t
    if budget_s >= 10:
        t, m = lit(scenes.box_stack(16, 16, 16), 1, 0.01)
        out["C3_one_sweep"] = t
        out["C3_100_sweeps_extrapolated"] = 100.0 * t
    out["sample"] = "oracle/sparse_literal.c (sparse_iterations_utils.cc pair loops), 1 thread, %.1f s in total" % (time.perf_counter() - t_all)
    return out


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: start N ranks of this very command, one per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run sets them), BEFORE any
    GPU call in this process -- children are started with subprocess, never exec'd from a process that touched the
    GPU.  Rank 0 prints the one JSON line on the inherited stdout; any rank failing ends the others and the exit
    code is non-zero."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in alive:          # one rank failed: the others would wait for it at the barrier for ever
                    q.terminate()
        time.sleep(0.05)
    return rc


def dry_run(args, rank, world):
    """The N > 1 plumbing without a GPU: rendezvous, the shard of this rank (the same functions the real run uses),
    one statistics reduction, rank 0 prints the line.  No solve, no timing claim."""
    import torch.distributed as tdist
    if world > 1:
        egs_dist.init_process_group(args.dist_backend)
    if args.workload == "c4":
        seeds, scaling, unit = c4_shard_seeds(rank, world), "strong", "ensemble-steps/s"
    else:
        seeds, scaling, unit = [rank * args.batch + b + 1 for b in range(args.batch)], "weak", "pile-steps/s"
    el, units, citers, resid, failed = egs_dist.reduce_stats(1.0 + 0.25 * rank, len(seeds) * args.steps, 0.0, 0.0, False)
    if world > 1:
        tdist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "constraint_solve_steps_per_sec", "dry_run": True, "value": None, "unit": unit, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "scaling": scaling, "units_reduced": units,
                          "elapsed_max_s": el, "rank0_seeds": [seeds[0], seeds[-1]]}), flush=True)
    if world > 1:
        tdist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=24, help="C3: independent piles resident per GPU (24 x 1024 columns = two full rounds of 3 tiles per CU)")
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "c4"], help="the headline workload (`value`)")
    ap.add_argument("--method", default="gs", choices=["gs", "sor"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip every CPU leg)")
    ap.add_argument("--legs", default="all", help="'all', 'none' or a comma list of single_pile,matvec,c1,c2,c4,coupled,c5,lcp_batch,literal,stopping_loop,world (1 GPU only)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only to rehearse N>1 on a 1-GPU box")
    ap.add_argument("--share-device0", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous / sharding / statistics reduction only, no GPU "
                    "work (CPU test of the N > 1 plumbing; the line says dry_run)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher (nothing has touched the GPU yet)
        raise SystemExit(self_launch(args.gpus))
    rank, world, local = egs_dist.env_rank_world()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or unset "
                         "WORLD_SIZE and let bench.py start its own ranks" % (args.gpus, world, args.gpus))
    if args.dry_run:
        return dry_run(args, rank, world)
    import torch
    import torch.distributed as tdist
    dev = 0 if (world == 1 or args.share_device0) else local
    if world > 1:
        torch.cuda.set_device(dev)
        egs_dist.init_process_group(args.dist_backend)
    ctx = capi.Context(dev)       # raises without the HIP library / a GPU: no fallback
    method = capi.GAUSS_SEIDEL if args.method == "gs" else capi.SOR
    legs = set() if (args.legs == "none" or world > 1) else \
        ({"single_pile", "matvec", "c1", "c2", "c4", "coupled", "c5", "lcp_batch", "literal", "stopping_loop", "world"} if args.legs == "all" else set(args.legs.split(",")))

    if args.workload == "c4":     # BASELINE config 4: 1024 ensembles sharded over the ranks
        seeds, scaling, unit = c4_shard_seeds(rank, world), "strong", "ensemble-steps/s"
    else:                         # every rank its own piles; seeds differ per rank and pile
        seeds, scaling, unit = [rank * args.batch + b + 1 for b in range(args.batch)], "weak", "pile-steps/s"
    r = run_piles(ctx, args.workload, seeds, method, args.steps, args.warmup, torch, tdist, world)
    st, m, sweeps = r["stats"], r["m"], r["sweeps"]
    el, units, citers, resid, failed = egs_dist.reduce_stats(
        r["elapsed"], len(seeds) * args.steps, float(m) * sweeps * args.steps, st.residual, st.status != capi.OK,
        device=("cuda:%d" % dev) if (world > 1 and args.dist_backend == "nccl") else None)

    out = None
    if rank == 0:
        nx, ny, nz = r["shape"]
        out = {
            "metric": "constraint_solve_steps_per_sec",
            "value": units / el,
            "unit": unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": r["prec"],
            "data": "synthetic",
            "config": {
                "workload": "%s: %dx%dx%d box pile = %d bodies, %d contacts (friction box), projected %s %d sweeps, "
                            "%s; step = assemble + solve + velocity update" % (
                                args.workload.upper(), nx, ny, nz, nx * ny * nz, m // len(seeds),
                                "Gauss-Seidel" if method == capi.GAUSS_SEIDEL else "SOR(1.5)", sweeps, r["prec"]),
                "piles_per_gpu": len(seeds), "piles_total": units // args.steps, "sweeps": sweeps, "cfm": 0.01, "dt": r["dt"], "erp": 0.2,
                "islands_per_gpu": st.n_islands, "tiles_per_gpu": st.n_tiles,
                "note": "a BASELINE pile is %d independent columns (SURVEY 8d generator: lateral gap 1e-2); the `coupled` leg is ONE island" % (nx * ny),
                "schedule": "host plan (islands->tiles, %.1f ms) built once per contact topology, outside the "
                            "timed region" % (r["t_plan"] * 1e3),
                "parallelism": ("ensembles 0..%d sharded over %d ranks (shard_range), no data-path collective" % (C4_ENSEMBLES - 1, world))
                if args.workload == "c4" else ("piles sharded x%d, no data-path collective" % world),
            },
            "contact_iters_per_sec": citers / el,
            "max_residual": resid,
            "failed": failed,
            "roofline": r["roofline"],
            "roofline_hbm": r["roofline_hbm"],
            "latency_model": r["latency_model"],
        }
    if rank == 0 and world == 1:
        extra = {}
        if "matvec" in legs:
            extra["matvec"] = matvec_leg(ctx, r, args.steps, args.warmup)
    r["problem"].close()
    if rank == 0 and world == 1:
        if "single_pile" in legs and len(seeds) != 1:
            s1 = run_piles(ctx, args.workload, [1], method, args.steps, args.warmup)
            extra["single_pile"] = leg_from_run(s1, args.steps, unit, "the headline workload with ONE pile on the GPU: the "
                                                "dependency chain of a pile (latency); the >= 10x north-star target reads against this")
            s1["problem"].close()
        if "c1" in legs:
            extra["c1"] = c1_leg(ctx, args.cpu_seconds)
        if "c2" in legs and args.workload != "c2":
            s2 = run_piles(ctx, "c2", [1], method, args.steps, args.warmup)
            extra["c2"] = leg_from_run(s2, args.steps, "pile-steps/s", "BASELINE config 2: 256-body pile, 1024 contacts, 50 sweeps fp64")
            s2["problem"].close()
            if args.cpu_seconds > 0:
                extra["c2"]["cpu_baseline"] = cpu_baseline("c2", min(3.0, args.cpu_seconds))
        if "c4" in legs and args.workload != "c4":
            s4 = run_piles(ctx, "c4", c4_shard_seeds(0, 1), method, args.steps, args.warmup)
            extra["c4"] = leg_from_run(s4, args.steps, "ensemble-steps/s", "BASELINE config 4 on ONE GPU: 1024 independent 64-body "
                                       "ensembles, fp32, 50 sweeps, one launch (`--workload c4 --gpus N` shards them)")
            s4["problem"].close()
            if args.cpu_seconds > 0:
                extra["c4"]["cpu_baseline"] = cpu_baseline("c4", min(3.0, args.cpu_seconds))
        if "coupled" in legs:
            extra["coupled"] = coupled_leg(ctx, method, max(5, args.steps // 2), 2, min(5.0, args.cpu_seconds))
        if "c5" in legs:
            extra["c5"] = c5_leg(ctx, args.cpu_seconds)
        if "lcp_batch" in legs:
            extra["lcp_batch"] = lcp_batch_leg(ctx, args.cpu_seconds)
        if "stopping_loop" in legs:
            extra["stopping_loop"] = stopping_loop_leg(ctx)
        if "world" in legs:
            extra["world_step"] = world_leg(ctx)
        out.update(extra)
        if args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_seconds)
            cores = os.cpu_count()
            out["cpu_baseline"]["host_cores_available"] = cores
            out["cpu_baseline_all_cores"] = {
                "value": out["cpu_baseline"]["value"] * cores, "unit": out["cpu_baseline"]["unit"], "cores": cores, "kind": "port",
                "sample": "IDEAL: the 1-thread figure x %d host cores (independent piles, no memory-bandwidth loss assumed); "
                          "not measured" % cores}
            out["gpu_over_cpu"] = {"batched_vs_1_core": out["value"] / out["cpu_baseline"]["value"],
                                   "batched_vs_all_cores_ideal": out["value"] / (out["cpu_baseline"]["value"] * cores)}
            try:
                usable = len(os.sched_getaffinity(0))
            except AttributeError:
                usable = cores
            nthr = max(1, min(16, usable))
            if nthr > 1:     # measured on the cores this process may use (a GPU box gives 16 of the host's)
                out["cpu_baseline_threads"] = cpu_baseline_threads(args.workload, min(6.0, args.cpu_seconds), nthr)
                out["gpu_over_cpu"]["batched_vs_%d_threads_measured" % nthr] = out["value"] / out["cpu_baseline_threads"]["value"]
            if "single_pile" in out:
                out["gpu_over_cpu"]["single_pile_vs_1_core"] = out["single_pile"]["value"] / out["cpu_baseline"]["value"]
            if "literal" in legs:
                out["cpu_baseline_literal"] = cpu_literal_baseline(args.cpu_seconds)
    if rank == 0:
        # every leg's fractions in one small object right behind the headline figures, so that a truncated record of
        # this line still carries them (each is recomputable from profiles/: see roofline.counters_source)
        def fr(leg, key="roofline"):
            o = out if leg is None else out.get(leg)
            o = o.get(key) if isinstance(o, dict) else None
            return None if not isinstance(o, dict) or o.get("frac") is None else round(float(o["frac"]), 4)
        summary = {"headline_valu_issue": fr(None), "headline_hbm": fr(None, "roofline_hbm"),
                   "headline_lane_utilisation": (out.get("roofline") or {}).get("lane_utilisation")}
        for leg in ("single_pile", "c2", "c4", "coupled"):
            if leg in out:
                summary[leg + "_valu_issue"] = fr(leg)
                summary[leg + "_hbm"] = fr(leg, "roofline_hbm")
        if "matvec" in out:
            summary["matvec_hbm"] = fr("matvec")
        head = {}
        for k, v in out.items():
            head[k] = v
            if k == "ms_per_step":
                head["roofline_summary"] = summary
        print(json.dumps(head), flush=True)
    ctx.close()
    if world > 1:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
