#!/usr/bin/env python3
"""Randomised soak of the dense mixed LCP (egs_mixed_constraints_solve): sizes 3..1400, all four modes (reference rule /
block rule, with and without box bounds), random equality fractions, degenerate right-hand sides.  Every answer is held
against the KKT conditions; up to N = 260 also against the CPU oracle's reference rule (same unique solution).
usage: dense_fuzz.py [cases=120] [seed0=0]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eggshell_amd import capi  # noqa: E402
from oracle import oracle as orc  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = capi.Context(0)
INF = np.inf
bad = 0
for k in range(cases):
    rng = np.random.default_rng(seed0 + k)
    N = int(rng.choice([3, 7, 20, 63, 64, 65, 100, 128, 130, 200, 260, 300, 450, 513, 700, 1023, 1024, 1100, 1400]))
    mode = int(rng.integers(0, 4))
    if mode < 2 and N > 300:
        mode += 2                      # the reference's single-index rule needs hundreds of pivots there
    M = rng.uniform(-1, 1, (N, N))
    A = M.T @ M + (10.0 ** rng.uniform(-3, 0)) * np.eye(N)
    b = rng.uniform(-1, 1, N) * (10.0 ** rng.uniform(-1, 1))
    if rng.uniform() < 0.2:
        b[rng.uniform(size=N) < 0.3] = 0.0      # degenerate rows
    Ceq = (rng.uniform(size=N) < rng.choice([0.0, 0.2, 0.5, 0.9, 1.0])).astype(np.uint8)
    if mode & 1:
        lo = np.where(rng.uniform(size=N) < 0.5, -rng.uniform(0.05, 0.5, N), 0.0)
        hi = np.where(rng.uniform(size=N) < 0.5, rng.uniform(0.05, 0.5, N), INF)
    else:
        lo, hi = np.zeros(N), np.full(N, INF)
    ok, x, w, piv = ctx.mixed_constraints_solve(A, b, Ceq, lo, hi, use_bounds=mode)
    eq = Ceq.astype(bool)
    msg = []
    if not ok:
        if mode >= 2 or N <= 9:
            msg.append("not solved")
    else:
        scale = max(1.0, np.abs(A @ x).max())
        if np.abs(A @ x - b - w).max() > 1e-7 * scale:
            msg.append("A x - b - w = %.2e" % np.abs(A @ x - b - w).max())
        if w[eq].any():
            msg.append("w on equality rows")
        xi, wi, li, hi_i = x[~eq], w[~eq], lo[~eq], hi[~eq]
        tol = 1e-7 * scale
        if (xi < li - tol).any() or (xi > hi_i + tol).any():
            msg.append("x outside its box")
        inside = (xi > li + tol) & (xi < hi_i - tol)
        if np.abs(wi[inside]).max(initial=0.0) > 1e-6 * scale:
            msg.append("w inside the box")
        if (wi[np.abs(xi - li) <= tol] < -1e-6 * scale).any() or (wi[np.abs(xi - hi_i) <= tol] > 1e-6 * scale).any():
            msg.append("w sign at a bound")
        if N <= 260:
            oko, xo, wo, _ = orc.mixed_constraints(A, b, Ceq, lo, hi, use_bounds=mode & 1)
            if oko and np.abs(x - xo).max() > 1e-7 * max(1.0, np.abs(xo).max()):
                msg.append("differs from the oracle by %.2e" % np.abs(x - xo).max())
    if msg:
        bad += 1
    print("case %3d N %4d mode %d eq %4d pivots %4d %s" % (seed0 + k, N, mode, int(eq.sum()), piv, "; ".join(msg) if msg else "ok"), flush=True)
print("%d cases, %d bad" % (cases, bad))
sys.exit(1 if bad else 0)
