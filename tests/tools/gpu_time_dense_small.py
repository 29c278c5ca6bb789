"""Dense LCP at the reference's own sizes (a few dozen rows: Cairn(4) has 3 rows per contact):
GPU wall time per call vs the CPU oracle."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from eggshell_amd import capi
from oracle import oracle as orc
ctx = capi.Context(0)
for N in (12, 24, 48, 96, 112, 113, 160):
    rng = np.random.default_rng(N)
    M = rng.uniform(-1, 1, (N, N)); A = M.T @ M + 0.1 * np.eye(N); b = rng.uniform(-1, 1, N)
    Ceq = np.zeros(N, np.uint8); lo, hi = np.zeros(N), np.full(N, np.inf)
    ctx.mixed_constraints_solve(A, b, Ceq, lo, hi)
    t = time.perf_counter(); n = 5
    for _ in range(n): ok, x, w, piv = ctx.mixed_constraints_solve(A, b, Ceq, lo, hi)
    tg = (time.perf_counter() - t) / n
    t = time.perf_counter()
    for _ in range(n): oko, xo, wo, pivo = orc.murty(A, b)
    tc = (time.perf_counter() - t) / n
    print(f"N={N}: pivots {piv} (oracle {pivo}), GPU {tg*1e3:.3f} ms ({tg*1e6/max(piv,1):.1f} us/pivot), CPU oracle {tc*1e3:.3f} ms, max|dx| {np.abs(x-xo).max():.1e}", flush=True)
