"""Soak: random box scenes, device contact list (both broad phases) vs the oracle's restated
collision.cc, bit for bit.  python tests/tools/soak_collide.py [scenes]"""
import os, sys
import numpy as np
from scipy.spatial.transform import Rotation
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from eggshell_amd import capi
from test_gpu_collide import reference_contacts

ctx = capi.Context(0)
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
total = 0
for seed in range(scenes):
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.integers(2, 90))
    ext = float(rng.uniform(0.3, 1.5))
    p = rng.uniform([-ext, -ext, -0.05], [ext, ext, 0.9], (n, 3))
    Rm = Rotation.random(n, random_state=seed).as_matrix().reshape(n, 9)
    if seed % 3 == 0:   # nearly aligned boxes: the axis-aligned branches of collision.cc
        k = n // 2
        Rm[:k] = Rotation.from_rotvec(rng.normal(size=(k, 3)) * 0.01).as_matrix().reshape(-1, 9)
    r0, r1, rd = reference_contacts(p, Rm)
    for mode in ("pairs", "grid"):
        os.environ["EGS_BROADPHASE"] = mode
        g0, g1, gd = ctx.update_contacts(p, Rm)
        assert len(g0) == len(r0) and np.array_equal(g0, r0) and np.array_equal(g1, r1) and np.array_equal(gd, rd), (seed, mode)
    total += len(r0)
print(f"{scenes} scenes, {total} contacts: device == oracle in both broad phases")
