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
