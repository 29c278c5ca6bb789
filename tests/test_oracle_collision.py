"""CPU: pin the oracle's contact generation (oracle/collision.c) with the reference's
OWN tests of it, restated property by property (eggshell/collision.cc:527-808):
LineClosestApproach, ClipPolygonByHalfSpace, IntersectBoxAndRectangle and
CollideBoxes against the slow-but-sure 15-axis separation test, the 99 % / 101 %
depth checks, and the code-dependent contact checks.  Fewer instances than the
reference's 100 000 so the CPU suite stays fast; seeds fixed."""
import numpy as np
from scipy.spatial.transform import Rotation

from oracle import oracle as orc


def axes(R):
    """Box axes = columns of the (row-major 3x3) rotation."""
    R = np.asarray(R).reshape(3, 3)
    return [R[:, 0], R[:, 1], R[:, 2]]


def separated_by_axis(c1, R1, h1, c2, R2, h2, axis):   # collision.cc:443-452
    span1 = sum(h1[k] * abs(axis @ axes(R1)[k]) for k in range(3))
    span2 = sum(h2[k] * abs(axis @ axes(R2)[k]) for k in range(3))
    return abs(axis @ c1 - axis @ c2) > span1 + span2


def boxes_separated(c1, R1, h1, c2, R2, h2):            # collision.cc:457-473
    a1, a2 = axes(R1), axes(R2)
    cand = a1 + a2 + [np.cross(u, v) for u in a1 for v in a2]
    return any(separated_by_axis(c1, R1, h1, c2, R2, h2, ax) for ax in cand)


def face_pseudo_distance(c, R, h, p):                   # collision.cc:488-492
    q = np.asarray(R).reshape(3, 3).T @ (p - c)
    return (np.abs(q) / h).max() - 1


def random_box(rng, axis1=None):                        # SetRandomBox, collision.cc:500-523
    c = rng.uniform(-1, 1, 3) * 0.5
    a0 = rng.uniform(-1, 1, 3) if axis1 is None else np.array(axis1, float)
    a0 /= np.linalg.norm(a0)
    a1 = rng.uniform(-1, 1, 3)
    a1 -= (a0 @ a1) * a0
    a1 /= np.linalg.norm(a1)
    R = np.stack([a0, a1, np.cross(a0, a1)], axis=1)    # columns
    return c, R, rng.uniform(0.05, 1.0, 3)


def test_line_closest_approach():
    """collision.cc:527-548."""
    rng = np.random.default_rng(0)
    for _ in range(500):
        pa, pb = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
        ua, ub = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
        ua /= np.linalg.norm(ua); ub /= np.linalg.norm(ub)
        alpha, beta = orc.line_closest_approach(pa, ua, pb, ub)
        delta = (pa + alpha * ua) - (pb + beta * ub)
        assert abs(ua @ delta) < 1e-9 and abs(ub @ delta) < 1e-9
        n0 = np.linalg.norm(delta)
        for fa, fb in ((1.01, 1), (0.99, 1), (1, 1.01), (1, 0.99)):
            assert np.linalg.norm((pa + alpha * fa * ua) - (pb + beta * fb * ub)) > n0


def test_clip_polygon_by_half_space():
    """collision.cc:579-631 incl. the three degenerate squares."""
    rng = np.random.default_rng(1)
    square = np.array([[0, 0], [0, 1], [1, 1], [1, 0]], float)
    for i in range(2000):
        if i <= 2:
            poly, normal, d = square, np.array([0.0, 1.0]), -(i - 1) * 1e-12
        else:
            poly = rng.uniform(-0.5, 0.5, (int(rng.integers(3, 8)), 2))
            normal, d = rng.uniform(-1, 1, 2), rng.uniform(-1, 1)
        new = orc.clip_polygon(poly, normal, d)
        hit = bool(((poly @ normal + d) > 0).any())      # PolygonIntersectsHalfspace
        assert (len(new) > 0) == hit
        if len(new):
            assert 3 <= len(new) <= 2 * len(poly)
            assert (new @ normal + d >= -1e-12).all()
            assert (np.linalg.norm(new - np.roll(new, -1, axis=0), axis=1) > 1e-12).all()
        if i <= 2:
            assert len(new) == 4
            if i < 2:
                assert np.array_equal(new, poly)


def test_intersect_box_and_rectangle():
    """collision.cc:633-681: emptiness agrees with the 15-axis test; polygon points lie in
    the box, on its boundary or at rectangle corners."""
    rng = np.random.default_rng(2)
    for i in range(3000):
        bc, bR, bh = random_box(rng)
        rc, rR, rh = random_box(rng, bR[:, 0] if (i & 1) == 0 else None)
        if (i & 2) == 0:
            rR = np.stack([rR[:, 1], rR[:, 0], -rR[:, 2]], axis=1)
        if (i & 15) == 0:
            i1 = int(rng.integers(0, 3)); i2 = int(rng.integers(0, 2)); i2 += (i2 == i1)
            rR = np.stack([bR[:, i1], bR[:, i2], np.cross(bR[:, i1], bR[:, i2])], axis=1)
        rh = rh.copy(); rh[2] = 0.0
        poly = orc.box_rectangle(bc, bR.reshape(9), bh, rc, rR.reshape(9), rh[:2])
        assert (len(poly) == 0) == boxes_separated(bc, bR, bh, rc, rR, rh)
        assert len(poly) == 0 or len(poly) >= 3
        for pt in poly:
            q1 = rc + rR @ np.array([pt[0], pt[1], 0.0])
            q2 = bR.T @ (q1 - bc)
            assert (np.abs(q2) < bh + 1e-9).all()
            if not (np.abs(q2) > bh - 1e-9).any():
                assert abs(abs(pt[0]) - rh[0]) < 1e-9 and abs(abs(pt[1]) - rh[1]) < 1e-9


def test_collide_boxes_properties():
    """collision.cc:683-808."""
    rng = np.random.default_rng(3)
    seen = set()
    for it in range(4000):
        c1, c2 = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
        h1, h2 = np.abs(rng.uniform(-1, 1, 3)) + 1e-3, np.abs(rng.uniform(-1, 1, 3)) + 1e-3
        R1 = Rotation.from_quat(rng.uniform(-0.5, 0.5, 4)).as_matrix()
        R2 = Rotation.from_quat(rng.uniform(-0.5, 0.5, 4)).as_matrix()
        aligned = False
        if it % 5 == 0:
            i = it // 5
            c2, R2, h2 = random_box(rng, R1[:, 0])
            if (i & 3) == 1:
                R2 = np.stack([R2[:, 1], R2[:, 0], -R2[:, 2]], axis=1)
            elif (i & 3) == 2:
                i1 = int(rng.integers(0, 3)); i2 = int(rng.integers(0, 2)); i2 += (i2 == i1)
                R2 = np.stack([R1[:, i1], R1[:, i2], np.cross(R1[:, i1], R1[:, i2])], axis=1)
                aligned = True
        s1, s2 = 2 * h1, 2 * h2
        sep1 = boxes_separated(c1, R1, h1, c2, R2, h2)
        contacts, code, axis, depth = orc.collide_boxes_info(c1, R1.reshape(9), c2, R2.reshape(9), s1, s2)
        sep2 = len(contacts) == 0
        assert sep1 == sep2 and sep2 == (code == 0)
        if sep2:
            continue
        seen.add(code)
        assert abs(np.linalg.norm(axis) - 1) < 1e-9 and depth >= -1e-9
        # 99 % of the depth along the axis keeps them colliding, 101 % separates them
        moved = c1 - 0.99 * depth * axis
        dummy, _ = orc.collide_boxes(moved, R1.reshape(9), c2, R2.reshape(9), s1, s2)
        assert len(dummy) > 0
        assert (dummy[:, 3:6] @ axis > 0).all()
        moved = moved - 0.02 * depth * axis
        assert len(orc.collide_boxes(moved, R1.reshape(9), c2, R2.reshape(9), s1, s2)[0]) == 0
        assert (contacts[:, 6] >= -1e-9).all()
        assert (np.abs(np.linalg.norm(contacts[:, 3:6], axis=1) - 1) < 1e-9).all()
        if 1 <= code <= 3:        # contacts on a face of box 2; pushed by the depth onto box 1
            for c in contacts:
                assert abs(face_pseudo_distance(c2, R2, h2, c[:3])) < 1e-9
                assert abs(face_pseudo_distance(c1, R1, h1, c[:3] + c[3:6] * c[6])) < 1e-9
        elif 4 <= code <= 6:
            for c in contacts:
                assert abs(face_pseudo_distance(c1, R1, h1, c[:3])) < 1e-9
                assert abs(face_pseudo_distance(c2, R2, h2, c[:3] - c[3:6] * c[6])) < 1e-9
        elif 7 <= code <= 15:     # one edge-edge contact, its normal is the separating axis
            assert len(contacts) == 1 and np.array_equal(contacts[0, 3:6], axis)
        else:
            assert code == 16 and len(contacts) == 1 and np.array_equal(contacts[0, :3], c2)
        if aligned and 1 <= code <= 6:
            assert len(contacts) == 4          # a contact rectangle
    assert {c for c in seen if c <= 6} and {c for c in seen if 7 <= c <= 15}   # face and edge cases both hit
