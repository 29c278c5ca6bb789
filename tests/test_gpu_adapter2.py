"""GPU (-m gpu): what round 2 added to the reference-shaped C++ API (eggshell_amd/host), driven by
`adapter_demo --round2` and compared with the oracle on the state the demo prints:
  * sparse::CalculateSparse{JMJtX,Lx,Ux,LxUx,Dx,UxDx,LxDx} (sparse_iterations_utils.cc:427-695) on the
    reference's own scenario -- Chain(4) at t = 0 and after steps (:938-1052) -- and a box pile, against
    the LITERAL O(m^2) restatement at the reference's tolerance 1e-9;
  * Ensemble::Step through the dense solver path (ComputeVDot, ensembles.cc:498-538);
  * Cairn (ensembles.cc:708-728): InitStabilize + 20 Step(5e-3) against the loops built from oracle pieces;
  * CheckAndCorrectEnsembleState: joint-vs-contact pruning identical on both Step paths, joint-vs-joint
    conflict refused (ensembles.cc:280-306);
  * the lcp::SolveLCP contract (toolkit/lcp.h:104-174, toolkit/lcp.cc:627-785)."""
import os
import subprocess

import numpy as np
import pytest

from eggshell_amd import capi, scenes
from helpers import ode_step, system_from_scene
from oracle import oracle as orc
from test_gpu_collide import reference_contacts
from test_gpu_stabilize import dense_J, explicit_euler

pytestmark = pytest.mark.gpu
DEMO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eggshell_amd", "host", "adapter_demo")


@pytest.fixture(scope="module")
def out():
    if not os.path.exists(DEMO):
        pytest.fail("adapter_demo is not built: run __graft_entry__.build()")
    txt = subprocess.run([DEMO, "--round2"], check=True, capture_output=True, text=True, timeout=600).stdout
    res = {}
    for line in txt.splitlines():
        k, *v = line.split()
        res[k] = np.array([float(t) for t in v])
    return res


def test_products_keep_the_reference_names_and_values(out):
    for rep in (0, 1):
        sc = scenes.chain(4)
        sc["p"] = out["prod%d_p" % rep].reshape(-1, 3)
        sc["R"] = out["prod%d_R" % rep].reshape(-1, 9)
        J0, J1, is_eq, lo, hi, err = orc.assemble(sc["p"], sc["R"], sc["kind"], sc["body0"], sc["body1"], sc["data"])
        Minv0 = orc.minv_blocks(scenes.chain(4)["R"], sc["mass"], sc["I_body"])     # M_inverse() is frozen at Init (Q5)
        s = orc.Sys(Minv0, sc["body0"], sc["body1"], J0, J1, is_eq, lo, hi)
        x = out["prod%d_x" % rep]
        Lx, Ux, Dx = orc.lit_Lx(s, x), orc.lit_Ux(s, x), orc.lit_Dx(s, x, 0.01, 1.0 / 1.5)
        ref = {"JMJtX": orc.lit_JMJtX(s, x, 0.01), "Lx": Lx, "Ux": Ux, "LxUx": Lx + Ux, "Dx": Dx, "UxDx": Ux + Dx, "LxDx": Lx + Dx}
        for name, want in ref.items():
            assert np.linalg.norm(out["prod%d_%s" % (rep, name)] - want) < 1e-9, (rep, name)
    if True:
        sc = scenes.box_stack(2, 2, 2)
        s, _ = system_from_scene(sc)
        x = out["prodpile_x"]
        assert x.shape[0] == 3 * s.m
        assert np.linalg.norm(out["prodpile_JMJtX"] - orc.lit_JMJtX(s, x, 0.01)) < 1e-9
        assert np.linalg.norm(out["prodpile_LxDx"] - (orc.lit_Lx(s, x) + orc.lit_Dx(s, x, 0.01, 1.0 / 1.5))) < 1e-9
        assert np.linalg.norm(out["prodpile_Ux"] - orc.lit_Ux(s, x)) < 1e-9


def test_chain_steps_through_the_dense_solver(out):
    sc = scenes.chain(8)
    for _ in range(3):
        lam = ode_step(sc, 1e-3)
    assert np.abs(out["dense3_p"] - sc["p"].reshape(-1)).max() < 1e-9
    assert np.abs(out["dense3_R"] - sc["R"].reshape(-1)).max() < 1e-9
    v6 = np.concatenate([sc["v"], sc["w"]], axis=1).reshape(-1)
    assert np.abs(out["dense3_v"] - v6).max() < 1e-9
    assert np.abs(out["dense3_lambda"] - lam).max() < 1e-9 * max(1.0, np.abs(lam).max())
    assert 1.0 <= out["dense3_cond"][0] < 1e7


def _relax(p, R, b0, b1, data, step_scale=0.2):
    kind = np.ones(len(b0), np.int32)
    J0, J1, is_eq, lo, hi, err = orc.assemble(p, R, kind, b0, b1, data)
    sc = dict(p=p, kind=kind, body0=b0, body1=b1)
    J = dense_J(sc, J0, J1)
    y = np.linalg.lstsq(J @ J.T, err, rcond=None)[0]
    return (-step_scale * (J.T @ y)).reshape(-1, 6), err


def _penetration(p, R):
    b0, b1, data = reference_contacts(p, R)
    return float((data[:, 6] ** 2).sum()) if len(b0) else 0.0


def _cairn_steps(p_init, R_init, p, R, v, w, steps, dt):
    """`steps` x Ensemble::Step(dt) from (p, R, v, w) with M^-1 / f_ext frozen at Init (quirk Q5), i.e.
    computed from the state the Cairn was constructed in -- the loop of ensembles.cc:390-427 from oracle pieces."""
    n = p.shape[0]
    mass = np.ones(n); I_body = np.tile((np.eye(3) * 0.1).reshape(9), (n, 1))
    Minv = orc.minv_blocks(R_init, mass, I_body)
    f_ext = orc.external_force(R_init, w, mass, I_body)
    seen = 0
    for _ in range(steps):
        b0, b1, data = reference_contacts(p, R)
        seen += len(b0)
        v6_old = np.concatenate([v, w], axis=1)
        if len(b0) == 0:
            v6 = v6_old + dt * np.einsum("brc,bc->br", Minv.reshape(n, 6, 6), f_ext)
        else:
            kind = np.ones(len(b0), np.int32)
            J0, J1, is_eq, lo, hi, err = orc.assemble(p, R, kind, b0, b1, data)
            s = orc.Sys(Minv, b0, b1, J0, J1, is_eq, lo, hi)
            rhs = orc.ode_rhs(v, w, Minv, f_ext, b0, b1, J0, J1, err, dt, 0.2)
            lam, _, it, res = orc.fast_iterate(s, rhs, 0.01, orc.SOR, max_iters=500, tol=1e-9)
            v6 = orc.velocity_update(v, w, Minv, f_ext, b0, b1, J0, J1, lam, dt)
        p, R = orc.position_update(p, R, v6_old, v6, dt)
        v, w = v6[:, :3].copy(), v6[:, 3:].copy()
    return p, R, np.concatenate([v, w], axis=1), seen


@pytest.mark.parametrize("tag0,tag1,tag2", [("cairn0", "cairn1", "cairn2"), ("tall0", "tall1", "tall2")])
def test_cairn_stabilise_and_drop(out, tag0, tag1, tag2):
    p0, R0 = out[tag0 + "_p"].reshape(-1, 3), out[tag0 + "_R"].reshape(-1, 9)
    v60 = out[tag0 + "_v"].reshape(-1, 6)
    assert p0.shape[0] == 4 and (np.abs(p0[:, :2]) <= 0.2).all() and (p0[:, 2] >= 1.0).all()
    for b in range(4):                                                     # RandomRotation: orthonormal, det +1
        Rb = R0[b].reshape(3, 3)
        assert np.abs(Rb @ Rb.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(Rb) - 1) < 1e-12
    assert (np.abs(v60) <= 1.0).all() and np.abs(v60).max() > 0.1          # max_init_v_ = max_init_w_ = 1
    p1, R1 = p0, R0
    if tag1:
        # InitStabilize (ensembles.cc:602-622) relaxes the interpenetrating rocks apart.  The reference solves
        # (J J^T) y = err by LDLT on a matrix that redundant contacts make singular, so its own trajectory is not
        # defined there; what is checked is the contract: it ends on err^2 <= 1e-9 or after 100 steps, velocities
        # untouched, and the penetration went down.
        p1, R1 = out[tag1 + "_p"].reshape(-1, 3), out[tag1 + "_R"].reshape(-1, 9)
        steps = int(out[tag0[:-1] + "_stab_steps"][0])
        before, after = _penetration(p0, R0), _penetration(p1, R1)
        if before > 1e-9:
            assert 1 <= steps <= 100 and after < before
            assert after <= 1e-9 or steps == 100
        else:
            assert steps == 0 and np.array_equal(p1, p0)
        assert np.array_equal(out[tag1 + "_v"], out[tag0 + "_v"])
    p2, R2, v62, seen = _cairn_steps(p0, R0, p1.copy(), R1.copy(), v60[:, :3].copy(), v60[:, 3:].copy(), 20, 0.005)
    assert int(out[tag0[:-1] + "_contacts"][0]) == seen
    assert np.abs(out[tag2 + "_p"] - p2.reshape(-1)).max() < 1e-6
    assert np.abs(out[tag2 + "_R"] - R2.reshape(-1)).max() < 1e-6
    assert np.abs(out[tag2 + "_v"] - v62.reshape(-1)).max() < 1e-5


def test_constraint_pair_checks(out):
    free_all, free_pair = out["jc_contacts_free"].astype(int)
    assert free_all == 8 and free_pair == 4                     # 4 ground + 4 box-box contacts
    for tag in ("explicit", "device"):                          # the contact on the joint is dropped on both Step paths
        allc, pair = out["jc_contacts_" + tag].astype(int)
        assert pair == free_pair - 1 and allc == free_all - 1
    assert out["jc_lambda_explicit"].shape == out["jc_lambda_device"].shape == (3 * (1 + 7),)
    assert np.abs(out["jc_lambda_explicit"] - out["jc_lambda_device"]).max() < 1e-6 * max(1.0, np.abs(out["jc_lambda_device"]).max())
    assert int(out["jj_conflict_status"][0]) == capi.ERR_INVALID     # the reference Panics (ensembles.cc:283-288)
    assert int(out["jj_apart_ok"][0]) == 1


def test_solvelcp_contract(out):
    n = 12
    M = np.array([[((i * 7 + j * 13) % 17 - 8) / 9.0 for j in range(n)] for i in range(n)])
    A = M.T @ M + np.eye(n)
    b = np.array([((i * 5) % 7 - 3) * 1.5 for i in range(n)])
    lo = np.full(n, -0.25); hi = np.full(n, 0.5)
    lo[[3, 8]] = -np.inf; hi[[3, 8]] = np.inf
    lo[5] = -np.inf

    def check_box(x, w, lo, hi, unb):
        assert np.linalg.norm(A @ x - b - w) < 1e-9
        assert np.abs(w[unb]).max(initial=0) == 0.0
        box = ~unb
        assert (x[box] >= lo[box]).all() and (x[box] <= hi[box]).all()
        inside = box & (x > lo) & (x < hi)
        assert np.abs(w[inside]).max(initial=0) < 1e-9
        assert (w[box & (x == lo)] >= -1e-9).all() and (w[box & (x == hi)] <= 1e-9).all()

    # sparse::{GaussSeidel,SOR}Iteration on an explicit matrix through the reference's signatures
    Ad = A + 6.0 * np.eye(n)
    lo0, hi0 = np.full(n, -0.25), np.full(n, 0.5)
    xg, itg, _ = orc.dense_iterate(Ad, b, orc.GAUSS_SEIDEL)
    assert np.abs(out["dense_gs_x"] - xg).max() < 1e-12 and abs(int(out["dense_gs_iters"][0]) - itg) <= 1
    assert np.linalg.norm(Ad @ out["dense_gs_x"] - b) < 1e-9 and np.linalg.norm(Ad @ out["dense_sor_x"] - b) < 1e-9
    Cm = (np.arange(n) % 3 == 0)
    xm, _, _ = orc.dense_iterate(Ad, b, orc.GAUSS_SEIDEL, Cm, lo0, hi0)
    assert np.abs(out["dense_gs_mixed_x"] - xm).max() < 1e-12
    # default settings: rows 3 and 8 are unbounded (eliminated first), row 5 keeps its finite hi
    assert int(out["lcp_default_ok"][0]) == 1
    unb = np.zeros(n, bool); unb[[3, 8]] = True
    check_box(out["lcp_default_x"], out["lcp_default_w"], lo, hi, unb)
    assert out["lcp_default_x"][5] <= 0.5
    # A was permuted in place as BoxSchur's partition does: swap(0, 8), swap(1, 3) -> unbounded rows first
    perm = list(range(n)); perm[0], perm[8] = perm[8], perm[0]; perm[1], perm[3] = perm[3], perm[1]
    # (the lower triangle only: toolkit/lcp.cc:171-195 never touches the upper one)
    Ad = out["lcp_default_A"].reshape(n, n)
    assert np.abs(np.tril(Ad) - np.tril(A[np.ix_(perm, perm)])).max() < 1e-12
    assert np.abs(np.triu(Ad, 1) - np.triu(A, 1)).max() < 1e-12
    assert np.abs(np.tril(Ad) - np.tril(A)).max() > 0.1                      # ... and that is not the caller's order
    # quirk Q6 reproduced on request: row 5 (lo = -inf, hi finite) is classed unbounded, its hi never looked at
    assert int(out["lcp_q6_ok"][0]) == 1
    unb6 = unb.copy(); unb6[5] = True
    hi6 = hi.copy(); hi6[5] = np.inf
    check_box(out["lcp_q6_x"], out["lcp_q6_w"], lo, hi6, unb6)
    # without the Schur complement the same box problem has the same (unique) solution
    assert int(out["lcp_noschur_ok"][0]) == 1 and np.abs(out["lcp_noschur_x"] - out["lcp_default_x"]).max() < 1e-9
    # box_lcp = false: lo = 0, hi = inf
    assert int(out["lcp_nobox_ok"][0]) == 1
    x, w = out["lcp_nobox_x"], out["lcp_nobox_w"]
    assert np.linalg.norm(A @ x - b - w) < 1e-9 and (x >= 0).all() and (w >= -1e-9).all() and abs(x @ w) < 1e-8
    # algorithm = COTTLE_DANTZIG, schur_complement = false: SolveLCP_BoxDantzig with the incremental factor; the
    # oracle's restatement (oracle/lcp_toolkit.c) takes the same steps and leaves the same permuted lower triangle
    lod = np.array([0.0 if i % 4 == 1 else -0.25 for i in range(n)]); hid = np.array([np.inf if i % 5 == 2 else 0.5 for i in range(n)])
    okd, pivd = out["lcp_dantzig_ok"].astype(int)
    oko, xo, wo, Ao, permo, pivo = orc.tk_box_dantzig(A, b, lod, hid)
    assert okd == 1 and oko and pivd == pivo
    assert np.abs(out["lcp_dantzig_x"] - xo).max() < 1e-12 and np.abs(out["lcp_dantzig_w"] - wo).max() < 1e-12
    Ad = out["lcp_dantzig_A"].reshape(n, n)
    # (the demo builds A with its own summation order: equal to the numpy A to rounding, hence tolerances here;
    #  tests/test_gpu_dantzig.py compares bit for bit through the C ABI)
    assert np.abs(np.tril(Ad) - np.tril(Ao)).max() < 1e-12 and np.abs(np.tril(Ad) - np.tril(A[np.ix_(permo, permo)])).max() < 1e-12
    assert np.abs(np.triu(Ad, 1) - np.triu(A, 1)).max() < 1e-12     # the upper triangle is neither read nor written
    xd, wd = out["lcp_dantzig_x"], out["lcp_dantzig_w"]
    assert np.linalg.norm(A @ xd - b - wd) < 1e-9 and (xd >= lod).all() and (xd <= hid).all()
    assert int(out["lcp_refused"][0]) == capi.ERR_INVALID          # Schur complement without box_lcp
    ok, piv = out["lcp_big_ok"].astype(int)
    assert ok == 1 and piv > 1
    okc, pivc = out["lcp_capped_ok"].astype(int)
    assert okc == 0 and pivc <= 1                                  # max_iterations = 1: gives up, returns false
    # the reference's own test shape: a lower triangle with a sentinel above it, default Settings (SolveLCP_BoxSchur);
    # x, w and the permuted matrix against the oracle's restatement of toolkit/lcp.cc:627-747
    for pas, m in ((0, 20), (1, 300), (2, 20)):
        i, j = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
        Mm = ((i * 31 + j * 17 + (i * j) % 11) % 23 - 11) / 11.0
        Af = Mm.T @ Mm + 0.25 * np.eye(m)
        k = np.arange(m)
        bm = ((k * 13) % 11 - 5) * 0.3
        bounded = (k % 2 == 1) | (k % 7 == 0)
        big = np.finfo(np.float64).max
        lm = np.where(bounded, -0.05 * (1 + k % 3), -big); hm = np.where(bounded, 0.04 * (1 + k % 4), big)
        okl, pivl = out["lcp_lower%d_ok" % pas].astype(int)
        xl, wl, Al = out["lcp_lower%d_x" % pas], out["lcp_lower%d_w" % pas], out["lcp_lower%d_A" % pas].reshape(m, m)
        oko, xo, wo, Ao, permo, nubo, ito = orc.tk_box_schur(np.tril(Af), bm, lm, hm, algorithm=1 if pas == 2 else 0, q6=False)
        assert okl == 1 and oko and nubo == m - bounded.sum()
        assert np.abs(xl - xo).max() < 1e-9 and np.abs(wl - wo).max() < 1e-9
        assert np.linalg.norm(Af @ xl - bm - wl) < 1e-6 and (xl >= lm).all() and (xl <= hm).all()      # toolkit/lcp.cc:1168-1172
        assert np.all(wl[~bounded] == 0)
        assert np.all(Al[np.triu_indices(m, 1)] == 555.0)          # the upper triangle was neither read nor written
        assert np.abs(np.tril(Al) - np.tril(Ao)).max() < 1e-11     # (demo and numpy build A in different summation orders)
        assert np.abs(np.tril(Al) - np.tril(Af[np.ix_(permo, permo)])).max() < 1e-11
