"""CPU baseline (i) of BASELINE.md: the LITERAL O(m^2)-per-pass restatement of the
reference's matrix-free GS (oracle/sparse_literal.c) -- the reference's real cost --
timed on C1 and C2 in full and on C3 for ONE sweep (extrapolated, flagged as such),
beside baseline (ii), the fast O(nnz) port."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from eggshell_amd import scenes
from helpers import system_from_scene, ode_rhs_from_scene
from oracle import oracle as orc

def run(name, sc, dt, K, literal_sweeps):
    s, err = system_from_scene(sc)
    rhs, _ = ode_rhs_from_scene(sc, s, err, dt)
    t = time.perf_counter(); orc.fast_iterate(s, rhs, 0.01, orc.GAUSS_SEIDEL, max_iters=K, tol=0.0); tf = time.perf_counter() - t
    t = time.perf_counter(); orc.lit_iterate(s, rhs, 0.01, orc.GAUSS_SEIDEL, max_iters=literal_sweeps, tol=0.0); tl = time.perf_counter() - t
    per_sweep = tl / max(literal_sweeps, 1)
    # a literal "iteration" = get_Nx + Mx_solver + residual = 3 O(m^2) passes, plus the initial residual
    full = per_sweep * K
    flag = "" if literal_sweeps == K else f" (EXTRAPOLATED from {literal_sweeps} sweep(s))"
    print(f"{name}: m={s.m}  fast O(nnz) {tf*1e3:.2f} ms/solve ({1/tf:.1f} solves/s) | literal O(m^2) {full:.2f} s/solve{flag} ({per_sweep*1e3:.1f} ms/sweep) | ratio {full/tf:.0f}x", flush=True)

run("C1 Chain(8), 50 sweeps", scenes.chain(8), 1e-3, 50, 50)
run("C2 8x8x4 pile, 50 sweeps", scenes.box_stack(8, 8, 4), 5e-3, 50, 50)
run("C3 16x16x16 pile, 100 sweeps", scenes.box_stack(16, 16, 16), 5e-3, 100, 1)
