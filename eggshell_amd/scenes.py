"""Synthetic ensembles for tests and benchmarks (numpy, host side).

These mirror the reference's generators and what its collision code would emit
for them; tests/ check that claim against the CPU oracle's restated
collision.cc bit for bit.  Nothing here runs on the GPU.

  * chain(n, anchor)       -- Chain::Chain, ensembles.cc:668-707.
  * box_stack(nx, ny, nz)  -- the BASELINE.md piles: axis-aligned boxes of side
    0.3 (body.h:91), mass 1, I = 0.1*I3 (Cairn convention, ensembles.cc:719),
    stacked in an nx x ny grid of columns, nz layers, bodies indexed bottom-up
    layer by layer; each layer sunk `sink` into the one below; lateral gap
    `gap` so side faces never touch.  Contacts are listed in the order
    Ensemble::UpdateContacts produces (ensembles.cc:445-480): all ground
    contacts by body, then body pairs i<j.
"""
import math

import numpy as np

JOINT_BALL, CONTACT_BOX = 0, 1
SIDE = 0.3


def _quat_to_R(w, x, y, z):
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([1.0 - (tyy + tzz), txy - twz, txz + twy,
                     txy + twz, 1.0 - (txx + tzz), tyz - twx,
                     txz - twy, tyz + twx, 1.0 - (txx + tyy)])


def chain(num_links, anchor=(0.0, 0.0, 2.0)):
    """ensembles.cc:668-707 + body.h:25-34 / body.cc:19-36."""
    n = num_links
    az, ax = 0.95531661812451, math.pi / 4
    qz_w, qz_z = math.cos(az / 2), math.sin(az / 2)
    qx_w, qx_x = math.cos(ax / 2), math.sin(ax / 2)
    Rm = _quat_to_R(qz_w * qx_w, qz_w * qx_x, qz_z * qx_x, qz_z * qx_w)
    p = np.zeros((n, 3))
    for i in range(n):
        p[i] = [math.sqrt(3.0) * 0.3 * i + anchor[0], 0 + anchor[1], 0 + anchor[2]]
    R = np.tile(Rm, (n, 1))
    inertia = 1.0 / 12 * (SIDE * SIDE + SIDE * SIDE)
    I_body = np.tile(np.diag([inertia] * 3).reshape(9), (n, 1))
    kind = np.full(n, JOINT_BALL, np.int32)
    body0 = np.zeros(n, np.int32)
    body1 = np.zeros(n, np.int32)
    data = np.zeros((n, 7))
    for i in range(n - 1):
        body0[i], body1[i] = i, i + 1
        data[i, 0:3] = [0.15, -0.15, 0.15]
        data[i, 3:6] = [-0.15, 0.15, -0.15]
    body0[n - 1], body1[n - 1] = 0, -1
    data[n - 1, 3:6] = p[0]
    return dict(p=p, R=R, v=np.zeros((n, 3)), w=np.zeros((n, 3)), mass=np.ones(n),
                I_body=I_body, kind=kind, body0=body0, body1=body1, data=data)


def box_stack(nx, ny, nz, sink=1e-3, gap=1e-2, jitter=0.0, seed=0, origin=(0.0, 0.0)):
    """nx*ny columns of nz axis-aligned boxes; returns bodies + contact list.

    jitter: columns are displaced in x,y by U(-jitter, jitter) (splitmix64),
    whole columns only, so the stacking geometry stays exact.
    """
    h = SIDE / 2
    n = nx * ny * nz
    p = np.zeros((n, 3))
    pitch = SIDE + gap
    state = np.uint64(seed * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)

    def splitmix():
        nonlocal state
        with np.errstate(over="ignore"):
            state = np.uint64(state + np.uint64(0x9E3779B97F4A7C15))
            z = state
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        return float(z >> np.uint64(11)) / float(1 << 53)

    col_xy = np.zeros((nx * ny, 2))
    for iy in range(ny):
        for ix in range(nx):
            dx = (2 * splitmix() - 1) * jitter if jitter else 0.0
            dy = (2 * splitmix() - 1) * jitter if jitter else 0.0
            col_xy[iy * nx + ix] = [origin[0] + ix * pitch + dx, origin[1] + iy * pitch + dy]
    ncol = nx * ny
    for k in range(nz):
        for c in range(ncol):
            b = k * ncol + c
            p[b, 0:2] = col_xy[c]
            p[b, 2] = (h - sink) + k * (SIDE - sink)
    R = np.tile(np.eye(3).reshape(9), (n, 1))
    I_body = np.tile((np.eye(3) * 0.1).reshape(9), (n, 1))

    kind, body0, body1, data = [], [], [], []
    # ground contacts: collision.cc:408-436, vertex loop x,y,z with z = -1 only
    for b in range(ncol):
        for sx in (-1, 1):
            for sy in (-1, 1):
                vx = (p[b, 0] + 1.0 * SIDE * 0.5 * sx) + 0.0 * SIDE * 0.5 * sy + 0.0 * SIDE * 0.5 * -1
                vy = (p[b, 1] + 0.0 * SIDE * 0.5 * sx) + 1.0 * SIDE * 0.5 * sy + 0.0 * SIDE * 0.5 * -1
                vz = (p[b, 2] + 0.0 * SIDE * 0.5 * sx) + 0.0 * SIDE * 0.5 * sy + 1.0 * SIDE * 0.5 * -1
                assert vz < 0
                kind.append(CONTACT_BOX); body0.append(-1); body1.append(b)
                data.append([vx, vy, vz, 0.0, 0.0, 1.0, -vz])
    # box-box: pairs (i, j=i+ncol), reference box = lower (code 3), incident
    # face = bottom of the upper box, polygon order (-,-),(-,+),(+,+),(+,-).
    for k in range(nz - 1):
        for c in range(ncol):
            i, j = k * ncol + c, (k + 1) * ncol + c
            bc = p[j].copy()
            bc[2] = bc[2] + (-1.0) * h          # B.center += Bface_normal * half
            ztop = p[i, 2] + 1.0 * h            # AfaceCenter.z
            Ad = -((0.0 * (p[i, 0] + 0.0 * h) + 0.0 * (p[i, 1] + 0.0 * h)) + 1.0 * ztop)
            for (px, py) in ((-h, -h), (-h, h), (h, h), (h, -h)):
                pos = [(bc[0] + 1.0 * px) + 0.0 * py, (bc[1] + 0.0 * px) + 1.0 * py,
                       (bc[2] + 0.0 * px) + 0.0 * py]
                depth = -(((0.0 * pos[0] + 0.0 * pos[1]) + 1.0 * pos[2]) + Ad)
                kind.append(CONTACT_BOX); body0.append(i); body1.append(j)
                data.append([pos[0], pos[1], pos[2], 0.0, 0.0, 1.0, depth])
    return dict(p=p, R=R, v=np.zeros((n, 3)), w=np.zeros((n, 3)), mass=np.ones(n),
                I_body=I_body, kind=np.array(kind, np.int32),
                body0=np.array(body0, np.int32), body1=np.array(body1, np.int32),
                data=np.array(data, np.float64).reshape(-1, 7))


def brick_wall(nx, nz, sink=1e-3):
    """A connected pile: running-bond wall, nx bricks per course, nz courses;
    odd courses are shifted by half a brick so every brick rests on two below
    (one big island).  Contacts come from the caller's collision routine; this
    returns bodies only."""
    h = SIDE / 2
    bodies = []
    for k in range(nz):
        cnt = nx if k % 2 == 0 else nx - 1
        off = 0.0 if k % 2 == 0 else h
        for i in range(cnt):
            bodies.append([off + i * SIDE * 1.0, 0.0, (h - sink) + k * (SIDE - sink)])
    p = np.array(bodies)
    n = p.shape[0]
    return dict(p=p, R=np.tile(np.eye(3).reshape(9), (n, 1)), v=np.zeros((n, 3)),
                w=np.zeros((n, 3)), mass=np.ones(n),
                I_body=np.tile((np.eye(3) * 0.1).reshape(9), (n, 1)))


def concat(scenes):
    """Batch independent ensembles into one system (body indices offset)."""
    out = {}
    off = 0
    b0s, b1s = [], []
    for s in scenes:
        n = s["p"].shape[0]
        b0s.append(np.where(s["body0"] >= 0, s["body0"] + off, -1))
        b1s.append(np.where(s["body1"] >= 0, s["body1"] + off, -1))
        off += n
    for k in ("p", "R", "v", "w", "mass", "I_body", "kind", "data"):
        out[k] = np.concatenate([s[k] for s in scenes])
    out["body0"] = np.concatenate(b0s).astype(np.int32)
    out["body1"] = np.concatenate(b1s).astype(np.int32)
    return out
