"""CPU: the C-ABI library loads and exports every symbol include/eggshell_amd.h
declares; host-side schedule logic (islands, tiles, tickets); loud failure
without a GPU.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

from eggshell_amd import capi, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "eggshell_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(egs_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported():
    lib = capi.load()
    names = _declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), "missing export: " + name
    assert sorted(capi.EXPORTS) == names


def test_no_oracle_in_product():
    """The product path must not reference the oracle (or any CPU fallback)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "eggshell_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in src and "import oracle" not in src and "from oracle" not in src, f
                assert "egs_oracle.h" not in src, f


def test_context_fails_loudly_without_gpu():
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    with pytest.raises(capi.EgsError) as e:
        capi.Context(0)
    assert e.value.status == capi.ERR_NO_DEVICE


def test_default_params_match_reference_constants():
    p = capi.SolveParams()
    capi.load().egs_default_params(ctypes.byref(p))
    assert (p.method, p.max_iters, p.check_every) == (capi.GAUSS_SEIDEL, 500, 1)   # sparse_iterations.cc:19
    assert (p.omega, p.cfm, p.tol) == (1.5, 0.0, 1e-9)                            # :15, constants.h:5


def _reference_tickets(n, body0, body1):
    cnt = np.zeros(n, int)
    pos0 = np.zeros(len(body0), int); pos1 = np.zeros(len(body0), int)
    for i, (a, b) in enumerate(zip(body0, body1)):
        if a >= 0:
            pos0[i] = cnt[a]; cnt[a] += 1
        if b >= 0:
            pos1[i] = cnt[b]; cnt[b] += 1
    return pos0, pos1, cnt


def test_plan_box_stack():
    sc = scenes.box_stack(4, 4, 4)
    n = sc["p"].shape[0]
    pl = capi.debug_plan(n, sc["body0"], sc["body1"])
    assert pl["n_islands"] == 16 and pl["n_global"] == 0
    assert pl["n_tiles"] == 1          # 16 columns x 16 constraints = 256 = one tile
    pos0, pos1, cnt = _reference_tickets(n, sc["body0"], sc["body1"])
    has0, has1 = sc["body0"] >= 0, sc["body1"] >= 0
    assert np.array_equal(pl["pos0"][has0], pos0[has0]) and np.array_equal(pl["pos1"][has1], pos1[has1])
    assert np.array_equal(pl["cnt0"][has0], cnt[sc["body0"][has0]])
    assert np.array_equal(pl["cnt1"][has1], cnt[sc["body1"][has1]])
    # every body's constraints sit in one tile
    for b in range(n):
        tiles = set(pl["cons_tile"][(sc["body0"] == b) | (sc["body1"] == b)])
        assert len(tiles) == 1


def test_plan_islands_never_split_and_oversize_goes_global():
    rng = np.random.default_rng(0)
    # 40 small chains of random length + one chain of 400 links (> tile)
    body0, body1, off = [], [], 0
    for L in list(rng.integers(2, 30, 40)) + [400]:
        for i in range(L - 1):
            body0.append(off + i); body1.append(off + i + 1)
        body0.append(off); body1.append(-1)
        off += L
    order = rng.permutation(len(body0))      # shuffle the list order
    body0 = np.array(body0, np.int32)[order]; body1 = np.array(body1, np.int32)[order]
    pl = capi.debug_plan(off, body0, body1)
    assert pl["n_islands"] == 41
    assert pl["n_global"] == 400
    assert (pl["cons_tile"] == -1).sum() == 400
    # island closure: constraints sharing a body share a tile
    for i in range(len(body0)):
        for j in range(i + 1, min(i + 50, len(body0))):
            if {body0[i], body1[i]} & {body0[j], body1[j]} - {-1}:
                assert pl["cons_tile"][i] == pl["cons_tile"][j]
    # tile capacity
    tiles, counts = np.unique(pl["cons_tile"][pl["cons_tile"] >= 0], return_counts=True)
    assert counts.max() <= 256 and len(tiles) == pl["n_tiles"]
    pos0, pos1, cnt = _reference_tickets(off, body0, body1)
    has0, has1 = body0 >= 0, body1 >= 0
    assert np.array_equal(pl["pos0"][has0], pos0[has0]) and np.array_equal(pl["pos1"][has1], pos1[has1])


def test_plan_rejects_bad_indices():
    with pytest.raises(capi.EgsError):
        capi.debug_plan(3, [0, 5], [1, 2])
    pl = capi.debug_plan(0, [], [])
    assert pl["n_islands"] == 0 and pl["n_tiles"] == 0
