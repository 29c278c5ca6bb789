#!/usr/bin/env python3
"""Randomised soak of egs_box_lcp_schur (lcp::SolveLCP_BoxSchur): sizes 2..1500, random share of unbounded rows (also
none and all), both inner algorithms, true box bounds incl. one-sided ones.  Held against the KKT conditions of the box
LCP  w = A x - b,  lo <= x <= hi,  w >= 0 at lo, w <= 0 at hi, w = 0 inside, w = 0 on unbounded rows.
usage: schur_fuzz.py [cases=100] [seed0=0]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eggshell_amd import capi  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = capi.Context(0)
INF = np.inf
bad = 0
for k in range(cases):
    rng = np.random.default_rng(seed0 + k)
    n = int(rng.choice([2, 5, 24, 63, 64, 65, 96, 97, 130, 200, 300, 500, 700, 1024, 1025, 1100, 1300, 1500]))
    M = rng.uniform(-1, 1, (n, n))
    A = M @ M.T + (10.0 ** rng.uniform(-2, 0)) * np.eye(n)
    b = rng.uniform(-1, 1, n)
    unb = rng.uniform(size=n) < rng.choice([0.0, 0.1, 0.5, 0.9, 1.0])
    lo = np.where(rng.uniform(size=n) < 0.3, -INF, -rng.uniform(0.0, 0.5, n))
    hi = np.where(rng.uniform(size=n) < 0.3, INF, rng.uniform(0.0, 0.5, n))
    lo[unb] = -INF; hi[unb] = INF
    alg = int(rng.integers(0, 2))
    ok, x, w, Ap, perm, nub, piv = ctx.box_lcp_schur(np.tril(A), b, lo, hi, algorithm=alg, reference_quirks=False)
    msg = []
    if not ok:
        msg.append("not solved")
    else:
        scale = max(1.0, np.abs(A @ x).max())
        tol = 1e-7 * scale
        if np.abs(A @ x - b - w).max() > tol:
            msg.append("A x - b - w = %.2e" % np.abs(A @ x - b - w).max())
        if (x < lo - tol).any() or (x > hi + tol).any():
            msg.append("x outside its box")
        inside = (x > lo + tol) & (x < hi - tol)
        if np.abs(w[inside]).max(initial=0.0) > 10 * tol:
            msg.append("w inside the box %.2e" % np.abs(w[inside]).max())
        at_lo = np.isfinite(lo) & (np.abs(x - lo) <= tol) & ~inside
        at_hi = np.isfinite(hi) & (np.abs(x - hi) <= tol) & ~inside
        if (w[at_lo & ~at_hi] < -10 * tol).any() or (w[at_hi & ~at_lo] > 10 * tol).any():
            msg.append("w sign at a bound")
    if msg:
        bad += 1
    print("case %3d n %4d alg %d unbounded %4d (nub %4d) pivots %5d %s" % (seed0 + k, n, alg, int(unb.sum()), nub, piv, "; ".join(msg) if msg else "ok"), flush=True)
print("%d cases, %d bad" % (cases, bad))
sys.exit(1 if bad else 0)
