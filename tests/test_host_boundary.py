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


def _modelled_conflicts(lane, tile, slot, body):
    """Extra LDS cycles if every lane is active: per 32-lane ticket pass the largest number of
    DIFFERENT slots on one bank (slot mod 32) minus one, per 16-lane ds_read_b128 pass likewise
    with slot mod 16 (passes {0-3,12-15,20-27} / {4-11,16-19,28-31} of each half-wave)."""
    tick = acc = 0
    has = body >= 0
    for t in np.unique(tile[has]):
        sel = has & (tile == t)
        for h in np.unique(lane[sel] // 32):
            hs = sel & (lane // 32 == h)
            li = lane[hs] % 32
            first = (li < 4) | ((li >= 12) & (li < 16)) | ((li >= 20) & (li < 28))
            by_bank = {}
            for s in np.unique(slot[hs]):
                by_bank[s % 32] = by_bank.get(s % 32, 0) + 1
            tick += max(by_bank.values()) - 1
            for grp in (first, ~first):
                by_bank = {}
                for s in np.unique(slot[hs][grp]):
                    by_bank[s % 16] = by_bank.get(s % 16, 0) + 1
                acc += (max(by_bank.values()) - 1) if by_bank else 0
    return tick, acc


@pytest.mark.parametrize("scene", ["pile", "random"])
def test_plan_slots_are_consistent_and_bank_aware(scene, monkeypatch):
    """LDS slot numbers: one slot per body inside its tile, none shared, slot 0 = the world,
    at most 64 numbers beyond a dense numbering; and the bank-aware numbering removes the
    modelled conflicts of first-use numbering (EGS_SLOT_BANKS=0) on the box pile."""
    if scene == "pile":
        sc = scenes.box_stack(8, 8, 8)
        n, body0, body1 = sc["p"].shape[0], sc["body0"], sc["body1"]
    else:
        rng = np.random.default_rng(5)
        n, m = 3000, 9000
        body0 = rng.integers(0, n, m).astype(np.int32)
        body1 = (body0 // 10 * 10 + rng.integers(0, 10, m)).astype(np.int32)   # islands of <= 10 bodies
        body1[rng.random(m) < 0.2] = -1
    results = {}
    for banks in ("1", "0"):
        monkeypatch.setenv("EGS_SLOT_BANKS", banks)
        pl = capi.debug_plan(n, body0, body1)
        sl = capi.debug_plan_slots(n, body0, body1)
        tile = pl["cons_tile"]
        assert (tile >= 0).all()
        for side_body, side_slot in ((body0, sl["slot0"]), (body1, sl["slot1"])):
            assert (side_slot[side_body < 0] == 0).all() and (side_slot[side_body >= 0] >= 1).all()
            assert (side_slot < sl["tile_nslots"]).all()
        # body -> (tile, slot) is a function, (tile, slot) -> body is one too
        body_all = np.concatenate([body0, body1]); slot_all = np.concatenate([sl["slot0"], sl["slot1"]])
        tile_all = np.concatenate([tile, tile]); keep = body_all >= 0
        pairs = np.unique(np.stack([body_all[keep], tile_all[keep], slot_all[keep]], 1), axis=0)
        assert len(np.unique(pairs[:, 0])) == len(pairs)
        assert len(np.unique(pairs[:, 1:], axis=0)) == len(pairs)
        # density: nslots <= bodies of the tile + 1 (world) + 64
        for t in np.unique(tile):
            nb = (pairs[:, 1] == t).sum()
            ns = sl["tile_nslots"][tile == t][0]
            assert nb + 1 <= ns <= nb + 1 + 64
            if banks == "0":
                assert ns == nb + 1
        # lanes: a permutation inside each tile
        for t in np.unique(tile):
            ln = sl["lane"][tile == t]
            assert len(np.unique(ln)) == len(ln) and ln.max() < 256
        t0 = _modelled_conflicts(sl["lane"], tile, sl["slot0"], body0)
        t1 = _modelled_conflicts(sl["lane"], tile, sl["slot1"], body1)
        results[banks] = (t0[0] + t1[0], t0[1] + t1[1])
    assert results["1"][0] <= results["0"][0] and results["1"][1] <= results["0"][1]
    if scene == "pile":
        assert results["0"][0] > 0 and results["0"][1] > 0
        assert results["1"] == (0, 0)


def test_oversize_schedule_chooser_follows_the_occupancy():
    """Patches wait on each other: their count is capped by (resident workgroups per CU, as the
    occupancy query reports for the kernel) x CUs, and the island falls through to the next
    schedule when it does not fit (faked occupancies; the library feeds the runtime's)."""
    ch = capi.debug_choose_oversize_schedule
    assert ch(157, 1, 2) == 0            # the 64x64 wall: 157 patches, one 1024-thread patch per CU
    assert ch(256, 1, 2) == 0 and ch(257, 1, 2) == 1 and ch(512, 1, 2) == 1 and ch(513, 1, 2) == 2
    assert ch(157, 0, 2) == 1            # the 4-lane kernel no longer fits a CU: 1-lane patches
    assert ch(300, 1, 1) == 2 and ch(256, 0, 1) == 1   # 1-lane kernel at one workgroup per CU
    assert ch(157, 0, 0) == 2            # nothing resident: the all-global kernel
    assert ch(157, 4, 8) == 0 and ch(600, 4, 8) == 2   # never above the measured one / two per CU
    assert ch(100, 1, 2, cu_count=64) == 1 and ch(130, 1, 2, cu_count=64) == 2   # fewer CUs (partitioned device)
    assert ch(10, 1, 2, patches=False) == 2 and ch(10, 1, 2, quad_patches=False) == 1
    assert ch(0, 1, 2) == 2


def _chunk_places(body0, body1):
    """The plan's chunks (plan.h): consecutive constraints on the same two bodies, at most four per chunk.
    Returns each constraint's place in its chunk and the number of chunks."""
    place = np.zeros(body0.shape[0], np.int64)
    chunks = 0
    for i in range(body0.shape[0]):
        if i > 0 and body0[i] == body0[i - 1] and body1[i] == body1[i - 1] and place[i - 1] < 3:
            place[i] = place[i - 1] + 1
        else:
            chunks += 1
    return place, chunks


def _timetable_is_list_order(body0, body1, tt, sweeps=3):
    """Replay the timetable of step_solve.hip on the host: at time step level + period * s the
    constraint runs sweep s.  Per body, the (sweep, list index) pairs must come in exactly the
    order of the sequential list-order sweeps, with no two of them in one time step."""
    m = body0.shape[0]
    on_tile = tt["level"] >= 0
    grp = 4 if tt["runs"] else 1     # runs: a time step holds the (up to four) updates of a chunk, in list order
    place = _chunk_places(body0, body1)[0] if tt["runs"] else np.zeros(m, np.int64)
    events = {}
    for c in np.nonzero(on_tile)[0]:
        for s in range(sweeps):
            step = int(tt["level"][c]) + int(tt["period"][c]) * s
            assert step < int(tt["depth"][c]) + int(tt["period"][c]) * (sweeps - 1)    # inside the kernel's loop bound
            t = step * grp + int(place[c])
            for b in {int(body0[c]), int(body1[c])} - {-1}:
                events.setdefault(b, []).append((t, s, int(c)))
    for b, ev in events.items():
        ev.sort()
        times = [e[0] for e in ev]
        assert len(set(times)) == len(times), "two updates of body %d in one time step" % b
        assert [(e[1], e[2]) for e in ev] == sorted((e[1], e[2]) for e in ev), "body %d out of list order" % b
    return int(on_tile.sum())


def test_static_timetable_reproduces_list_order(monkeypatch):
    # regular columns: period = per-body count (8), depth = 64
    sc = scenes.box_stack(4, 4, 16, jitter=1e-3, seed=3)
    tt = capi.debug_plan_timetable(sc["p"].shape[0], sc["body0"], sc["body1"], 256)
    assert _timetable_is_list_order(sc["body0"], sc["body1"], tt) == sc["body0"].shape[0]
    assert not tt["runs"] and tt["period"].max() == 8 and tt["depth"].max() == 64
    # the 4-lane plan (tile size 0 = its automatic choice) groups the four contact points of a box face
    tt = capi.debug_plan_timetable(sc["p"].shape[0], sc["body0"], sc["body1"], 0)
    assert _timetable_is_list_order(sc["body0"], sc["body1"], tt) == sc["body0"].shape[0]
    assert tt["runs"] and tt["period"].max() == 2 and tt["depth"].max() == 16
    for c in range(0, sc["body0"].shape[0], 4):                                      # ... adjacent in their tile
        assert len(set(tt["level"][c:c + 4])) == 1
    # ragged random graphs, world sides, repeated pairs, every tile size
    rng = np.random.default_rng(11)
    for trial in range(30):
        n = int(rng.integers(1, 40)); m = int(rng.integers(1, 300))
        b1 = rng.integers(0, n, m).astype(np.int32)
        b0 = np.where(rng.random(m) < 0.3, -1, rng.integers(0, n, m)).astype(np.int32)
        b0 = np.where(b0 == b1, -1, b0).astype(np.int32)
        for rep in (1, 4, "ragged"):     # 4: every constraint four times in a row, as a box face's contact points -> runs
            if rep == "ragged":            # 1..4 contact points per pair, as a collider leaves them
                reps = rng.choice([1, 2, 3, 4, 4, 4, 4, 4], size=m)
                c0, c1 = np.repeat(b0, reps), np.repeat(b1, reps)
            else:
                c0, c1 = np.repeat(b0, rep), np.repeat(b1, rep)
            for tile in (0, 64, 128, 256, 512):
                # EGS_RUNS=2: chunks wherever the padding allows (the default also wants the busiest body's chunk count
                # well below its constraint count, which random graphs rarely offer)
                monkeypatch.setenv("EGS_RUNS", "2" if trial % 2 else "1")
                tt = capi.debug_plan_timetable(n, c0, c1, tile)
                chunks = _chunk_places(c0, c1)[1]
                if tt["runs"]:
                    assert tile == 0 and 4 * chunks <= (5 * c0.shape[0]) // 4
                elif tile == 0 and trial % 2 and 4 * chunks <= (5 * c0.shape[0]) // 4:
                    # refused although the padding is small: only because it would push an island past a workgroup
                    plain = capi.debug_plan(n, c0, c1, 256)
                    assert plain["n_global"] == 0 or c0.shape[0] > 256
                _timetable_is_list_order(c0, c1, tt)
                on = tt["level"] >= 0
                assert np.all(tt["period"][on] <= tt["depth"][on]) and np.all(tt["period"][on] >= 1)
