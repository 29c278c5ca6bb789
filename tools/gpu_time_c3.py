"""Quick timing probe: C3 pile (16x16x16), step = assemble + K sweeps + velocity."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from eggshell_amd import capi, scenes
import bench

ctx = capi.Context(0)
def run(nx, ny, nz, K, batch, method=1, prec=capi.F64, steps=20):
    piles = [scenes.box_stack(nx, ny, nz, origin=(0.0, 100.0 * b)) for b in range(batch)]
    sc = scenes.concat(piles) if batch > 1 else piles[0]
    n = sc['p'].shape[0]
    Minv, f_ext = bench.host_mass_and_force(sc)
    t = time.time()
    pr = capi.Problem(ctx, n, sc['body0'], sc['body1'], prec)
    t_plan = time.time() - t
    pr.set_state(sc['p'], sc['R'], sc['v'], sc['w'], Minv, f_ext)
    pr.set_constraints(sc['kind'], sc['data'])
    prm = capi.params(method=method, max_iters=K, tol=0.0, cfm=0.01)
    for _ in range(3): pr.step(5e-3, 0.2, prm)
    ctx.synchronize(); ctx.kernel_time(reset=True)
    ctx.timer_start()
    for _ in range(steps): pr.step(5e-3, 0.2, prm)
    ms = ctx.timer_stop()
    ksum, kn = ctx.kernel_time()
    st = pr.stats()
    m = pr.m
    per = ms / steps
    print(f'{nx}x{ny}x{nz} K={K} batch={batch} m={m} tiles={st.n_tiles}: {per:.3f} ms/step, solve-kernel {ksum/kn:.3f} ms, '
          f'{batch*1000/per:.1f} pile-solves/s, {m*K/(per*1e-3)/1e9:.3f} G contact-iters/s, alg {m*K*768/(ksum/kn*1e-3)/1e12:.3f} TB/s, plan {t_plan*1e3:.1f} ms, res {st.residual:.4g}', flush=True)
    pr.close()

run(8, 8, 4, 50, 1)
run(16, 16, 16, 100, 1)
run(16, 16, 16, 100, 8)
run(16, 16, 16, 100, 32)
run(16, 16, 16, 100, 1, method=2)
run(4, 4, 4, 50, 1024, prec=capi.F32)
