"""Config C5 probe: dense mixed LCP (Lcp::MixedConstraintsSolver semantics), N = 256..2048,
A = M^T M + 1e-3 I, b ~ U(-1,1), C ~ Bernoulli(1/2), seed 0.
Prints GPU wall time (upload + Schur + Murty + download), pivots, and the CPU oracle
time where it finishes in reasonable time."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from eggshell_amd import capi
from oracle import oracle as orc

ctx = capi.Context(0)
for N in (256, 512, 1024, 2048):
    rng = np.random.default_rng(0)
    M = rng.uniform(-1, 1, (N, N))
    A = M.T @ M + 1e-3 * np.eye(N)
    b = rng.uniform(-1, 1, N)
    Ceq = rng.integers(0, 2, N).astype(np.uint8)
    lo, hi = np.zeros(N), np.full(N, np.inf)
    ctx.mixed_constraints_solve(A[:64, :64].copy(), b[:64], Ceq[:64], lo[:64], hi[:64])  # warm up
    t = time.perf_counter()
    ok, x, w, piv = ctx.mixed_constraints_solve(A, b, Ceq, lo, hi)
    tg = time.perf_counter() - t
    res = np.linalg.norm(A @ x - b - w) if ok else float('nan')
    line = f"N={N} ne={int(Ceq.sum())} ni={int(N - Ceq.sum())}: GPU ok={ok} pivots={piv} {tg*1e3:.1f} ms ({tg/max(piv,1)*1e3:.3f} ms/pivot) |Ax-b-w|={res:.2e}"
    if N <= 512:
        t = time.perf_counter()
        oko, xo, wo, pivo = orc.mixed_constraints(A, b, Ceq, lo, hi)
        tc = time.perf_counter() - t
        line += f" | CPU oracle ok={oko} pivots={pivo} {tc*1e3:.0f} ms, speedup {tc/tg:.1f}x, max|dx|={np.abs(x-xo).max():.1e}"
    print(line, flush=True)
    t = time.perf_counter()
    ok2, x2, w2, piv2 = ctx.mixed_constraints_solve(A, b, Ceq, lo, hi, use_bounds=2)
    tb = time.perf_counter() - t
    res2 = np.linalg.norm(A @ x2 - b - w2) if ok2 else float('nan')
    extra = f" max|x-x_ref_rule|={np.abs(x2-x).max():.1e}" if ok else ""
    print(f"      block principal pivoting: ok={ok2} factorisations={piv2} {tb*1e3:.1f} ms |Ax-b-w|={res2:.2e}{extra}", flush=True)
