#!/usr/bin/env python3
"""Where does a coupled island's sweep period go?  Runs the brick wall with EGS_TRACE_UPDATES=1 (every update of the
4-lane patch kernel stamped with the 100 MHz wall clock), then walks the CRITICAL CHAIN of the sweep pipeline backwards
from the last update: an update (constraint c, sweep s) waits for the previous constraint of each of its two bodies
(list order, cyclically: the last constraint of sweep s - 1 for the first of sweep s); the one that finished LATER is
the binding edge.  Every edge is classed -- same patch or across patches, same wavefront or not -- and its latency
(stamp difference) accumulated.  usage: tools/trace_patches.py [nx nz sweeps]; writes gpurun_out/trace_<nx>x<nz>.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["EGS_TRACE_UPDATES"] = "1"
import bench  # noqa: E402
from eggshell_amd import capi, scenes  # noqa: E402

nx, nz, K = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (12, 10, 100)
ctx = capi.Context(0)
sc = scenes.brick_wall(nx, nz)
b0, b1, data = ctx.update_contacts(sc["p"], sc["R"])
sc.update(kind=np.full(len(b0), 1, np.int32), body0=b0, body1=b1, data=data)
n, m = sc["p"].shape[0], len(b0)
Minv, f_ext = bench.host_mass_and_force(sc)
pr = capi.Problem(ctx, n, b0, b1)
pr.set_state(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext)
pr.set_constraints(sc["kind"], sc["data"])
prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=K, tol=0.0, cfm=0.01)
pr.step(5e-3, 0.2, prm)
pr.step(5e-3, 0.2, prm)
ctx.synchronize()
T = pr.debug_trace().astype(np.int64)          # [K][m] ticks of 10 ns
npat, cp, cl, r0, r1 = capi.debug_plan_patches(n, b0, b1)
assert T.shape == (K, m) and (T > 0).all(), "no trace: is the island on 4-lane patches?"
t_first = T.min()
T = (T - t_first) * 0.01                       # us
# list-order neighbours per body
prev = np.full((m, 2), -1, np.int64)           # predecessor constraint on side 0 / 1 (cyclic)
first_of = np.zeros((m, 2), bool)
lists = [[] for _ in range(n)]
for c in range(m):
    for side, b in enumerate((b0[c], b1[c])):
        if b >= 0:
            lists[b].append((c, side))
for b in range(n):
    L = lists[b]
    for k, (c, side) in enumerate(L):
        prev[c, side] = L[k - 1][0]
        first_of[c, side] = k == 0
wave = cl // 16                                 # 16 constraints (64 lanes) per wavefront of the 4-lane kernel
# walk back from the last stamp
s, c = np.unravel_index(np.argmax(T), T.shape)
classes = {}
chain = []
while True:
    best = None
    for side in range(2):
        pc = prev[c, side]
        if pc < 0:
            continue
        ps = s - 1 if first_of[c, side] else s
        if ps < 0:
            continue
        if best is None or T[ps, pc] > best[0]:
            best = (T[ps, pc], ps, pc)
    if best is None:
        break
    lat = T[s, c] - best[0]
    same_patch = cp[c] == cp[best[2]]
    grp = "poller" if ((r0[c] | r1[c]) & 1) else ("releaser" if ((r0[c] | r1[c]) & 2) else "interior")
    key = (("same patch, same wavefront" if same_patch and wave[c] == wave[best[2]] else "same patch, other wavefront") if same_patch else "ACROSS patches") + " -> " + grp
    e = classes.setdefault(key, {"edges": 0, "us": 0.0})
    e["edges"] += 1; e["us"] += float(lat)
    chain.append((int(s), int(c), float(lat), key))
    s, c = best[1], best[2]
total = float(T.max())
out = {"wall": "%dx%d" % (nx, nz), "contacts": m, "patches": int(npat), "sweeps": K, "span_us": total, "us_per_sweep": total / K,
       "critical_chain": {k: {"edges": v["edges"], "us": v["us"], "us_per_edge": v["us"] / v["edges"], "share": v["us"] / total} for k, v in classes.items()},
       "chain_edges": len(chain), "edges_per_sweep": len(chain) / K}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "trace_%dx%d.json" % (nx, nz)), "w"), indent=1)
print(json.dumps(out, indent=1))
pr.close(); ctx.close()
