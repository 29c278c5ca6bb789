"""GPU (-m gpu): lcp::SolveLCP of toolkit/lcp.h:172-174 (the adjacent solver the
north star names) through the adapter: box LCP with unbounded rows as
equalities; checked by its own complementarity diagram (toolkit/lcp.h:116-133)
and against the oracle's corrected box Murty."""
import numpy as np
import pytest

from oracle import oracle as orc
from test_gpu_adapter import demo_out  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def test_solvelcp_box_with_unbounded_rows(demo_out):  # noqa: F811
    n = 12
    M = np.array([[((i * 7 + j * 13) % 17 - 8) / 9.0 for j in range(n)] for i in range(n)])
    A = M.T @ M + np.eye(n)
    b = np.array([((i * 5) % 7 - 3) * 1.5 for i in range(n)])
    lo = np.full(n, -0.25); hi = np.full(n, 0.5)
    lo[[3, 8]] = -np.inf; hi[[3, 8]] = np.inf
    assert int(demo_out["solvelcp_ok"][0]) == 1
    x, w = demo_out["solvelcp_x"], demo_out["solvelcp_w"]
    assert np.linalg.norm(A @ x - b - w) < 1e-9
    unb = np.isinf(lo)
    assert np.abs(w[unb]).max() == 0.0                      # unbounded rows are equalities
    box = ~unb
    assert (x[box] >= lo[box]).all() and (x[box] <= hi[box]).all()
    inside = box & (x > lo) & (x < hi)
    assert np.abs(w[inside]).max(initial=0) < 1e-9
    assert (w[box & (x == lo)] >= -1e-9).all() and (w[box & (x == hi)] <= 1e-9).all()
    ok, xo, wo, _ = orc.mixed_constraints(A, b, unb.astype(np.uint8), np.where(unb, 0, lo), np.where(unb, 0, hi), use_bounds=1)
    assert ok and np.abs(x - xo).max() < 1e-9 and np.abs(w - wo).max() < 1e-9
