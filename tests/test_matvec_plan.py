"""CPU: the schedule of the stand-alone mat-vec (egs_debug_matvec_plan, host only):
every constraint in exactly one tile, lanes ascending in the list, whole small
islands per tile (no shared body), oversize islands cut with a boundary list."""
import numpy as np
import pytest

from eggshell_amd import capi, scenes
from helpers import random_system


def shared_bodies(n, b0, b1, tile):
    first = np.full(n, -1)
    shared = np.zeros(n, bool)
    for i in range(len(b0)):
        for b in (b0[i], b1[i]):
            if b < 0:
                continue
            if first[b] < 0:
                first[b] = tile[i]
            elif first[b] != tile[i]:
                shared[b] = True
    return shared


@pytest.mark.parametrize("block", [128, 256])
def test_box_stack_tiles_are_closed(block):
    sc = scenes.box_stack(6, 5, 4)
    n, m = sc["p"].shape[0], sc["kind"].shape[0]
    pl = capi.debug_matvec_plan(n, sc["body0"], sc["body1"], block)
    assert pl["n_islands"] == 30 and pl["n_shared_bodies"] == 0 and pl["n_boundary"] == 0
    assert (pl["cons_tile"] >= 0).all() and (pl["cons_lane"] < block).all()
    # one (tile, lane) per constraint, ascending list index inside a tile
    keys = pl["cons_tile"].astype(np.int64) * block + pl["cons_lane"]
    assert np.unique(keys).size == m
    for t in range(pl["n_tiles"]):
        idx = np.nonzero(pl["cons_tile"] == t)[0]
        assert (np.diff(pl["cons_lane"][idx]) > 0).all()
    assert not shared_bodies(n, sc["body0"], sc["body1"], pl["cons_tile"]).any()
    per_tile = block // 16   # 16 contacts per column of 4
    assert pl["n_tiles"] == -(-30 // per_tile)


def test_random_topologies_partition_and_boundary():
    rng = np.random.default_rng(5)
    for n, m, connected in [(40, 700, True), (300, 500, False), (7, 900, True), (50, 0, False)]:
        s, _ = random_system(rng, n, m, connected=connected)
        pl = capi.debug_matvec_plan(n, s.body0, s.body1, 256)
        if m == 0:
            assert pl["n_tiles"] == 0
            continue
        keys = pl["cons_tile"].astype(np.int64) * 256 + pl["cons_lane"]
        assert np.unique(keys).size == m and (pl["cons_tile"] >= 0).all()
        sh = shared_bodies(n, s.body0, s.body1, pl["cons_tile"])
        assert sh.sum() == pl["n_shared_bodies"]
        on_shared = ((s.body0 >= 0) & sh[np.maximum(s.body0, 0)]) | ((s.body1 >= 0) & sh[np.maximum(s.body1, 0)])
        assert on_shared.sum() == pl["n_boundary"]
        if connected:
            assert pl["n_shared_bodies"] > 0   # one island larger than a tile must be cut


def test_bad_indices_are_rejected():
    with pytest.raises(capi.EgsError):
        capi.debug_matvec_plan(3, [0, 5], [1, 2], 256)
    with pytest.raises(capi.EgsError):
        capi.debug_matvec_plan(3, [0, 1], [1, 1], 256)   # the same body on both sides
    with pytest.raises(capi.EgsError):
        capi.debug_matvec_plan(3, [0], [1], 100)
