"""PCIe-inclusive rate of entry 1 (host buffers in, host lambda out): the one-shot
egs_solve_blocks on C3, including plan + allocation + upload + solve + download.
Never bench.py's `value`; quoted in DESIGN.md."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from eggshell_amd import capi, scenes
import bench
ctx = capi.Context(0)
sc = scenes.box_stack(16, 16, 16)
Minv, f_ext = bench.host_mass_and_force(sc)
_pr = capi.Problem(ctx, sc['p'].shape[0], sc['body0'], sc['body1'])
_pr.set_state(sc['p'], sc['R'], sc['v'], sc['w'], Minv, f_ext); _pr.set_constraints(sc['kind'], sc['data'])
_pr.assemble(5e-3, 0.2)                 # the flat system (J blocks, rhs, bounds) from the device assembly
class S: pass
s = S(); s.Minv = Minv; s.body0 = sc['body0']; s.body1 = sc['body1']; s.n = sc['p'].shape[0]
s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs, _ = _pr.blocks(); _pr.close()
prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=100, tol=0.0, cfm=0.01)
for _ in range(3): ctx.solve_blocks(s.Minv, s.body0, s.body1, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs, prm)
t = time.perf_counter(); N = 20
for _ in range(N): x, st = ctx.solve_blocks(s.Minv, s.body0, s.body1, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs, prm)
dt = (time.perf_counter() - t) / N
print(f"one-shot egs_solve_blocks C3: {dt*1e3:.2f} ms per call = {1/dt:.0f} calls/s (H2D 6.2 MB+solve+D2H; schedule reused from the previous call)")
pr = capi.Problem(ctx, s.n, s.body0, s.body1)
t = time.perf_counter()
for _ in range(N):
    pr.set_blocks(s.Minv, s.J0, s.J1, s.is_eq, s.lo, s.hi, rhs); pr.solve(prm, want_stats=False); x = pr.lambda_()
dt = (time.perf_counter() - t) / N
print(f"resident problem, re-upload blocks + solve + download: {dt*1e3:.2f} ms per call = {1/dt:.0f} calls/s")
b0, b1, d = ctx.update_contacts(sc['p'], sc['R'])
t = time.perf_counter()
for _ in range(N): ctx.update_contacts(sc['p'], sc['R'])
dt = (time.perf_counter() - t) / N
print(f"egs_update_contacts C3 (4096 bodies -> {len(b0)} contacts, host in/out): {dt*1e3:.2f} ms per call")
