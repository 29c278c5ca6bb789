"""Shared test helpers: build flat systems from scenes with the CPU oracle,
and an INDEPENDENT numpy dense formulation used to cross-check the oracle."""
import numpy as np

from oracle import oracle as orc


def system_from_scene(sc):
    """Oracle-assembled flat system (what sparse::*Iteration consumes)."""
    J0, J1, is_eq, lo, hi, err = orc.assemble(sc["p"], sc["R"], sc["kind"], sc["body0"], sc["body1"], sc["data"])
    Minv = orc.minv_blocks(sc["R"], sc["mass"], sc["I_body"])
    return orc.Sys(Minv, sc["body0"], sc["body1"], J0, J1, is_eq, lo, hi), err


def ode_rhs_from_scene(sc, s, err, dt, erp=0.2):
    f_ext = orc.external_force(sc["R"], sc["w"], sc["mass"], sc["I_body"])
    rhs = orc.ode_rhs(sc["v"], sc["w"], s.Minv, f_ext, s.body0, s.body1, s.J0, s.J1, err, dt, erp)
    return rhs, f_ext


def dense_numpy(s, eps=0.0):
    """J W J^T + eps I built with plain numpy matmul (independent of the oracle's
    block code): ensembles.cc:510."""
    n, m = s.n, s.m
    J = np.zeros((3 * m, 6 * n))
    W = np.zeros((6 * n, 6 * n))
    for b in range(n):
        W[6 * b:6 * b + 6, 6 * b:6 * b + 6] = s.Minv[b].reshape(6, 6)
    for i in range(m):
        for bb, JJ in ((s.body0[i], s.J0[i]), (s.body1[i], s.J1[i])):
            if bb >= 0:
                J[3 * i:3 * i + 3, 6 * bb:6 * bb + 6] = JJ.reshape(3, 6)
    return J @ W @ J.T + eps * np.eye(3 * m), J, W


def numpy_pgs(A, b, is_eq, lo, hi, method, omega, iters):
    """Scalar projected Jacobi/GS/backward-SOR on a dense matrix (Appendix A of
    SURVEY.md / sparse_iterations.cc:72-144), independent of the oracle."""
    R = b.shape[0]
    x = b.copy()
    k = 1.0 / omega
    for _ in range(iters):
        if method == 0:
            xn = x.copy()
            for r in range(R):
                t = (b[r] - (A[r] @ x - A[r, r] * x[r])) / A[r, r]
                xn[r] = t if is_eq[r] else min(max(t, lo[r]), hi[r])
            x = xn
        elif method == 1:
            for r in range(R):
                t = (b[r] - (A[r] @ x - A[r, r] * x[r])) / A[r, r]
                x[r] = t if is_eq[r] else min(max(t, lo[r]), hi[r])
        else:
            for r in range(R - 1, -1, -1):
                t = (b[r] - (A[r] @ x - A[r, r] * x[r]) - (1 - k) * A[r, r] * x[r]) / (k * A[r, r])
                x[r] = t if is_eq[r] else min(max(t, lo[r]), hi[r])
    return x


def check_mixed_solution(A, b, x, is_eq, lo, hi, tol=1e-9):
    """sparse_iterations.cc:308-353 CheckMixedConstraintSolutions."""
    w = A @ x - b
    eq = is_eq.astype(bool)
    if np.linalg.norm(w[eq]) >= tol:
        return False
    for r in np.nonzero(~eq)[0]:
        if lo[r] < x[r] < hi[r]:
            if not abs(w[r]) < tol:
                return False
        elif x[r] == lo[r]:
            if not w[r] > -tol:
                return False
        elif x[r] == hi[r]:
            if not w[r] < tol:
                return False
        else:
            return False
    return True


def ode_step(sc, dt, erp=0.2):
    """One Ensemble::Step(dt, OPEN_DYNAMICS_ENGINE) of a contact-free ensemble
    through the reference's LIVE dense path (ensembles.cc:390-427, 498-538,
    563-591), built from oracle pieces.  M^-1 and f_ext are frozen at Init
    (quirk Q5): pass them in sc['Minv0'], sc['f_ext0'] (created on first use)."""
    if "Minv0" not in sc:
        sc["Minv0"] = orc.minv_blocks(sc["R"], sc["mass"], sc["I_body"])
        sc["f_ext0"] = orc.external_force(sc["R"], sc["w"], sc["mass"], sc["I_body"])
    J0, J1, is_eq, lo, hi, err = orc.assemble(sc["p"], sc["R"], sc["kind"], sc["body0"], sc["body1"], sc["data"])
    s = orc.Sys(sc["Minv0"], sc["body0"], sc["body1"], J0, J1, is_eq, lo, hi)
    rhs = orc.ode_rhs(sc["v"], sc["w"], s.Minv, sc["f_ext0"], s.body0, s.body1, J0, J1, err, dt, erp)
    A = orc.dense_JMJt(s, 0.0)
    if not np.linalg.cond(A) < 1e7:           # ensembles.cc:513-521
        A = A + 0.01 * np.eye(A.shape[0])
    ok, lam, w, _ = orc.mixed_constraints(A, rhs, is_eq, lo, hi)
    assert ok
    v6 = orc.velocity_update(sc["v"], sc["w"], s.Minv, sc["f_ext0"], s.body0, s.body1, J0, J1, lam, dt)
    v6_old = np.concatenate([sc["v"], sc["w"]], axis=1)
    p, R = orc.position_update(sc["p"], sc["R"], v6_old, v6, dt)
    sc["p"], sc["R"] = p, R
    sc["v"], sc["w"] = v6[:, 0:3].copy(), v6[:, 3:6].copy()
    return lam


def random_system(rng, n, m, world_frac=0.2, eq_frac=0.4, connected=False):
    """A random flat system: random topology, random J blocks, SPD 6x6 M^-1
    blocks, mixed equality / box rows (some infinite bounds)."""
    body0 = np.zeros(m, np.int32); body1 = np.zeros(m, np.int32)
    for i in range(m):
        if connected and i < n - 1:
            a, b = i, i + 1
        else:
            a = int(rng.integers(0, n)); b = int(rng.integers(0, n - 1))
            if b >= a:
                b += 1
        r = rng.uniform()
        if r < world_frac / 2:
            a = -1
        elif r < world_frac:
            b = -1
        body0[i], body1[i] = a, b
    J0 = rng.uniform(-1, 1, (m, 18)); J1 = rng.uniform(-1, 1, (m, 18))
    J0[body0 < 0] = 0.0; J1[body1 < 0] = 0.0
    Minv = np.zeros((n, 36))
    for b in range(n):
        M = rng.uniform(-1, 1, (6, 6))
        Minv[b] = (M @ M.T + 0.5 * np.eye(6)).reshape(36)
    is_eq = (rng.uniform(size=3 * m) < eq_frac).astype(np.uint8)
    lo = -rng.uniform(0.1, 2.0, 3 * m); hi = rng.uniform(0.1, 2.0, 3 * m)
    inf_hi = rng.uniform(size=3 * m) < 0.3
    hi[inf_hi] = np.inf
    lo[rng.uniform(size=3 * m) < 0.1] = -np.inf
    lo[is_eq == 1] = 0.0; hi[is_eq == 1] = 0.0
    rhs = rng.uniform(-1, 1, 3 * m)
    return orc.Sys(Minv, body0, body1, J0, J1, is_eq, lo, hi), rhs


def grouped_system(rng, s, rhs, rep=4):
    """Every constraint of `s` `rep` times in a row with fresh J blocks, bounds and right-hand sides:
    the shape of a box face's contact points (the 4-lane plan treats aligned groups of four as runs)."""
    if rep == "ragged":      # 1..4 contact points per pair, mostly 4: what a collider leaves after pruning
        rep = rng.choice([1, 2, 3, 4, 4, 4, 4, 4, 4, 5, 8], size=s.body0.shape[0])
    body0, body1 = np.repeat(s.body0, rep), np.repeat(s.body1, rep)
    m = body0.shape[0]
    J0 = rng.uniform(-1, 1, (m, 18)); J1 = rng.uniform(-1, 1, (m, 18))
    J0[body0 < 0] = 0.0; J1[body1 < 0] = 0.0
    is_eq = np.repeat(s.is_eq.reshape(-1, 3), rep, axis=0).reshape(-1)
    lo = np.repeat(s.lo.reshape(-1, 3), rep, axis=0).reshape(-1)
    hi = np.repeat(s.hi.reshape(-1, 3), rep, axis=0).reshape(-1)
    return orc.Sys(s.Minv, body0, body1, J0, J1, is_eq, lo, hi), rng.uniform(-1, 1, 3 * m)
