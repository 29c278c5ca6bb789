"""CPU: the oracle reproduces the committed golden vectors bit for bit (guards
the oracle against drift), and the literal O(m^2) algorithm agrees with them."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as orc

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
TAGS = ((0, "jacobi"), (1, "gs"), (2, "sor"))


def load_system(g):
    return orc.Sys(g["Minv"], g["body0"], g["body1"], g["J0"], g["J1"], g["is_eq"], g["lo"], g["hi"])


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_golden(path):
    g = np.load(path)
    s = load_system(g)
    cfm = float(g["cfm"])
    if "kind" in g.files:
        J0, J1, is_eq, lo, hi, err = orc.assemble(g["p"], g["R"], g["kind"], g["body0"], g["body1"], g["data"])
        assert np.array_equal(J0, g["J0"]) and np.array_equal(J1, g["J1"]) and np.array_equal(err, g["err"])
        rhs = orc.ode_rhs(g["v"], g["w"], g["Minv"], g["f_ext"], g["body0"], g["body1"], J0, J1, err,
                          float(g["dt"]), float(g["erp"]))
        assert np.array_equal(rhs, g["rhs"])
    for method, tag in TAGS:
        for K in (1, 10, 50):
            x, a, it, res = orc.fast_iterate(s, g["rhs"], cfm, method, max_iters=K, tol=0.0)
            assert np.array_equal(x, g["x_%s_%d" % (tag, K)])
            assert np.array_equal(a, g["a_%s_%d" % (tag, K)])
        if "xlit_%s_10" % tag in g.files:
            xl = g["xlit_%s_10" % tag]
            xf = g["x_%s_10" % tag]
            ok = np.isfinite(xf)
            assert np.abs(xl[ok] - xf[ok]).max() <= 1e-9 * max(1.0, np.abs(xf[ok]).max())


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_matvec_twin_and_literal_match_golden(path):
    """The matrix-free products of x = rhs: the O(nnz) twin reproduces the committed vectors bit for bit and
    the literal O(m^2) restatement of CalculateSparse{Lx,Ux,Dx,JMJtX} agrees with them to the reference's 1e-9."""
    g = np.load(path)
    s = load_system(g)
    cfm, scale, x = float(g["cfm"]), 1.0 / 1.5, g["rhs"]
    for tag, parts in (("full", 8), ("L", 1), ("U", 2), ("D", 4), ("LU", 3), ("UD", 6), ("LD", 5)):
        assert np.array_equal(orc.fast_matvec(s, x, parts, cfm, scale), g["mv_" + tag])
    assert np.linalg.norm(g["mv_full"] - g["Ax_rhs"]) < 1e-9
    Lx, Ux, Dx = orc.lit_Lx(s, x), orc.lit_Ux(s, x), orc.lit_Dx(s, x, cfm, scale)
    assert np.array_equal(Lx, g["mvlit_L"]) and np.array_equal(Ux, g["mvlit_U"]) and np.array_equal(Dx, g["mvlit_D"])
    for tag, want in (("L", Lx), ("U", Ux), ("D", Dx), ("LU", Lx + Ux), ("UD", Ux + Dx), ("LD", Lx + Dx)):
        assert np.linalg.norm(g["mv_" + tag] - want) < 1e-9
