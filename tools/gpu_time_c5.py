#!/usr/bin/env python3
"""Time the dense mixed LCP (BASELINE config 5 recipe of bench.py) at N = 2048 / 512: host in, host out."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from eggshell_amd import capi  # noqa: E402

ctx = capi.Context(0)
cases = ((2048, 2), (512, 2), (512, 0)) if len(sys.argv) < 2 else ((int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 2),)
for N, mode in cases:
    A, b, C, lo, hi = bench.c5_problem(N)
    ctx.mixed_constraints_solve(A, b, C, lo, hi, use_bounds=mode)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); ok, x, w, piv = ctx.mixed_constraints_solve(A, b, C, lo, hi, use_bounds=mode); best = min(best, time.perf_counter() - t0)
    print(N, mode, round(best * 1e3, 3), ok, piv, float(np.abs(A @ x - b - w).max()))
