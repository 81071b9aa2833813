"""Seeded synthetic worlds and scans for the LIO hot path.

The reference ships no data and there is no network, so every test and the benchmark run on
these generators.  They are INPUT synthesis only: no algorithm of the hot path lives here.

World  : urban canyon -- ground plane z~0 (+-2 cm), two facade planes per 40 m block, 5 %
         volumetric clutter; one point per `voxel` cell (mirrors the 1-point-per-voxel insert rule
         of the reference map, include/ikd-Tree/ikd_Tree.cpp:422-449), continuous jitter so that
         no two squared distances tie.
Scans  : KITTI-64-like spinning LiDAR (64 beams, -24.8..+2.0 deg, 1563+ azimuth steps) or a
         Livox-HAP-like narrow field of view (120 x 25 deg rosette), ray-cast analytically
         against the same planes, range noise N(0, 2 cm), blind-zone and max-range drop,
         resampled to exactly n_q points.  Extrinsics follow config/velodyne64_kitti_dataset.yaml
         :32-33 and config/hap_livox.yaml:49-52.
"""
from __future__ import annotations

import dataclasses
import math
import os

import numpy as np

SEED = 20240926

# config/velodyne64_kitti_dataset.yaml:32-33
KITTI_EXTRINSIC_T = np.array([-8.086759e-01, 3.195559e-01, -7.997231e-01])
KITTI_EXTRINSIC_R = np.array(
    [9.999976e-01, 7.553071e-04, -2.035826e-03,
     -7.854027e-04, 9.998898e-01, -1.482298e-02,
     2.024406e-03, 1.482454e-02, 9.998881e-01]).reshape(3, 3)
# config/hap_livox.yaml:49-52
HAP_EXTRINSIC_T = np.array([0.05512, 0.02226, -0.0297])
HAP_EXTRINSIC_R = np.eye(3)


@dataclasses.dataclass
class World:
    points: np.ndarray        # [n,3] float32 map points (shuffled)
    voxel: float
    ground_half: float        # ground square is [-ground_half, ground_half]^2
    walls: np.ndarray         # [w,6] float64: axis(0=x-plane,1=y-plane), c, a, b, h, unused
    seed: int


def _quat_from_rot(Rm: np.ndarray) -> np.ndarray:
    """Rotation matrix -> unit quaternion (x, y, z, w)."""
    t = np.trace(Rm)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        w = 0.25 * s
        x = (Rm[2, 1] - Rm[1, 2]) / s
        y = (Rm[0, 2] - Rm[2, 0]) / s
        z = (Rm[1, 0] - Rm[0, 1]) / s
    else:
        i = int(np.argmax(np.diag(Rm)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(1.0 + Rm[i, i] - Rm[j, j] - Rm[k, k]) * 2
        q = np.zeros(3)
        q[i] = 0.25 * s
        q[j] = (Rm[j, i] + Rm[i, j]) / s
        q[k] = (Rm[k, i] + Rm[i, k]) / s
        w = (Rm[k, j] - Rm[j, k]) / s
        x, y, z = q
    q = np.array([x, y, z, w])
    return q / np.linalg.norm(q)


def quat_to_rot(q: np.ndarray) -> np.ndarray:
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rot_from_rpy(roll: float, pitch: float, yaw: float) -> np.ndarray:
    cr, sr = math.cos(roll), math.sin(roll)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cy, sy = math.cos(yaw), math.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def make_world(n_map: int, voxel: float = 0.2, seed: int = SEED, ground_frac: float = 0.80,
               facade_frac: float = 0.15) -> World:
    """Exactly n_map float32 points, at most one per `voxel` cell."""
    # LIO_SYNTH_CACHE=<dir>: several scripts of one GPU-box visit share the generated world (5 M points take ~20 s of numpy)
    cache = os.environ.get("LIO_SYNTH_CACHE")
    cfile = os.path.join(cache, "world_%d_%g_%d_%g_%g.npz" % (n_map, voxel, seed, ground_frac, facade_frac)) if cache else None
    if cfile and os.path.exists(cfile):
        z = np.load(cfile)
        return World(points=z["points"], voxel=voxel, ground_half=float(z["ground_half"]), walls=z["walls"], seed=seed)
    rng = np.random.default_rng(seed)
    n_ground = int(round(ground_frac * n_map))
    side_cells = max(4, int(math.ceil(math.sqrt(n_ground))))
    half = 0.5 * side_cells * voxel

    # ground: one point per (i, j) voxel column, jittered inside the voxel, z = +-2 cm
    ij = np.stack(np.meshgrid(np.arange(side_cells), np.arange(side_cells), indexing="ij"),
                  -1).reshape(-1, 2)
    gx = (ij[:, 0] + 0.5 + rng.uniform(-0.42, 0.42, len(ij))) * voxel - half
    gy = (ij[:, 1] + 0.5 + rng.uniform(-0.42, 0.42, len(ij))) * voxel - half
    gz = rng.uniform(-0.02, 0.02, len(ij))
    ground = np.stack([gx, gy, gz], 1)

    # facades: per 40 m block one x-plane wall and one y-plane wall, 30 m long, height h
    block = 40.0
    nb = max(1, int(math.floor(2 * half / block)))
    n_fac = int(round(facade_frac * n_map))
    wall_len = min(30.0, 0.75 * 2 * half / nb)
    h = max(2 * voxel, n_fac * voxel * voxel / (2 * nb * nb * wall_len))
    walls = []
    fac_pts = []
    for bi in range(nb):
        for bj in range(nb):
            x0 = -half + bi * (2 * half / nb)
            y0 = -half + bj * (2 * half / nb)
            cell = 2 * half / nb
            # x-plane wall: x = cx, y in [a, b]
            cx = x0 + 0.27 * cell + 0.0137
            a, b = y0 + 0.12 * cell, y0 + 0.12 * cell + wall_len
            walls.append((0, cx, a, b, h, 0.0))
            nu, nv = int(wall_len / voxel), max(1, int(h / voxel))
            uv = np.stack(np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij"), -1).reshape(-1, 2)
            u = a + (uv[:, 0] + 0.5 + rng.uniform(-0.42, 0.42, len(uv))) * voxel
            v = 0.03 + (uv[:, 1] + 0.5 + rng.uniform(-0.42, 0.42, len(uv))) * voxel
            n = rng.uniform(-0.02, 0.02, len(uv))
            fac_pts.append(np.stack([cx + n, u, v], 1))
            # y-plane wall: y = cy, x in [a2, b2]
            cy = y0 + 0.31 * cell + 0.0171
            a2, b2 = x0 + 0.45 * cell, x0 + 0.45 * cell + wall_len
            walls.append((1, cy, a2, b2, h, 0.0))
            u = a2 + (uv[:, 0] + 0.5 + rng.uniform(-0.42, 0.42, len(uv))) * voxel
            v = 0.03 + (uv[:, 1] + 0.5 + rng.uniform(-0.42, 0.42, len(uv))) * voxel
            n = rng.uniform(-0.02, 0.02, len(uv))
            fac_pts.append(np.stack([u, cy + n, v], 1))
    facade = np.concatenate(fac_pts, 0) if fac_pts else np.zeros((0, 3))

    def dedupe(p):
        p32 = p.astype(np.float32)
        key = np.floor(p32.astype(np.float64) / voxel).astype(np.int64)
        k = (key[:, 0] + (1 << 20)) << 42 | (key[:, 1] + (1 << 20)) << 21 | (key[:, 2] + (1 << 20))
        _, first = np.unique(k, return_index=True)
        return p32[np.sort(first)]

    pts = dedupe(np.concatenate([ground, facade], 0))
    if len(pts) > n_map:
        # drop surplus uniformly (keeps the surface structure)
        keep = np.sort(rng.choice(len(pts), n_map, replace=False))
        pts = pts[keep]
    # clutter tops the set up to exactly n_map
    while len(pts) < n_map:
        need = n_map - len(pts)
        c = np.stack([rng.uniform(-half, half, need + 64),
                      rng.uniform(-half, half, need + 64),
                      rng.uniform(0.15, 3.0, need + 64)], 1)
        pts = dedupe(np.concatenate([pts.astype(np.float64), c], 0))[:n_map]
    perm = rng.permutation(len(pts))
    w = World(points=np.ascontiguousarray(pts[perm]), voxel=voxel, ground_half=half,
              walls=np.array(walls, dtype=np.float64).reshape(-1, 6), seed=seed)
    if cfile:
        os.makedirs(cache, exist_ok=True)
        np.savez(cfile + ".tmp.npz", points=w.points, ground_half=w.ground_half, walls=w.walls)
        os.replace(cfile + ".tmp.npz", cfile)
    return w


def _raycast(world: World, origin: np.ndarray, dirs: np.ndarray, max_range: float) -> np.ndarray:
    """Nearest hit range for rays origin + t * dirs against the ground and the walls."""
    n = len(dirs)
    t_best = np.full(n, np.inf)
    # ground z = 0
    with np.errstate(divide="ignore", invalid="ignore"):
        t = -origin[2] / dirs[:, 2]
    hit = (dirs[:, 2] < 0) & (t > 0)
    px = origin[0] + t * dirs[:, 0]
    py = origin[1] + t * dirs[:, 1]
    hit &= (np.abs(px) <= world.ground_half) & (np.abs(py) <= world.ground_half)
    t_best = np.where(hit, np.minimum(t_best, t), t_best)
    # walls (only those that can be in range)
    for axis, c, a, b, h, _ in world.walls:
        ax = int(axis)
        o_ax, o_other = origin[ax], origin[1 - ax]
        if abs(c - o_ax) > max_range:
            continue
        mid = 0.5 * (a + b)
        if abs(mid - o_other) > max_range + 0.5 * (b - a):
            continue
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (c - o_ax) / dirs[:, ax]
        u = o_other + t * dirs[:, 1 - ax]
        v = origin[2] + t * dirs[:, 2]
        hit = (t > 0) & np.isfinite(t) & (u >= a) & (u <= b) & (v >= 0.03) & (v <= 0.03 + h)
        t_best = np.where(hit, np.minimum(t_best, t), t_best)
    return t_best


@dataclasses.dataclass
class Scan:
    body: np.ndarray          # [n_q,3] float32 points in the LiDAR frame
    pos: np.ndarray           # true IMU position in world (3)
    rot: np.ndarray           # true IMU orientation quaternion (x,y,z,w)
    off_R: np.ndarray         # LiDAR->IMU rotation quaternion (x,y,z,w)
    off_T: np.ndarray         # LiDAR->IMU translation (3)
    extrinsic_est_en: bool


def make_scan(world: World, n_q: int, kind: str = "kitti64", seed: int = SEED + 1,
              pos=(3.1, -2.2, 0.0), rpy=(0.004, -0.006, 0.52), blind: float | None = None,
              max_range: float = 100.0, range_sigma: float = 0.02) -> Scan:
    """Ray-cast scan with exactly n_q returns, expressed in the LiDAR frame."""
    rng = np.random.default_rng(seed)
    if kind == "kitti64":
        off_T, off_R, ext_en = KITTI_EXTRINSIC_T, KITTI_EXTRINSIC_R, False
        blind = 1.0 if blind is None else blind       # yaml:22
        sensor_h = 1.73
    elif kind == "hap":
        off_T, off_R, ext_en = HAP_EXTRINSIC_T, HAP_EXTRINSIC_R, True
        blind = 0.5 if blind is None else blind
        sensor_h = 1.2
    elif kind == "wide":     # config 5: 360 x 40 deg
        off_T, off_R, ext_en = KITTI_EXTRINSIC_T, KITTI_EXTRINSIC_R, False
        blind = 1.0 if blind is None else blind
        sensor_h = 1.73
    else:
        raise ValueError(kind)
    R_wi = rot_from_rpy(*rpy)
    # place the IMU so that the LiDAR origin sits sensor_h above the ground
    p_wi = np.array(pos, dtype=np.float64)
    lidar_in_world = R_wi @ off_T + p_wi
    p_wi[2] += sensor_h - lidar_in_world[2]
    origin = R_wi @ off_T + p_wi
    R_wl = R_wi @ off_R

    body_parts = []
    have = 0
    attempt = 0
    while have < n_q:
        attempt += 1
        if kind == "kitti64":
            n_az = 1563 * (1 + attempt)
            el = np.deg2rad(np.linspace(-24.8, 2.0, 64))
            az = np.linspace(0, 2 * np.pi, n_az, endpoint=False) + rng.uniform(0, 1e-3)
            E, A = np.meshgrid(el, az, indexing="ij")
            E = E.ravel() + rng.normal(0, 2e-4, E.size)
            A = A.ravel()
        elif kind == "hap":
            n_r = int(n_q * 1.6 * attempt)
            # non-repetitive rosette inside 120 x 25 deg
            tt = rng.uniform(0, 1, n_r)
            A = np.deg2rad(60.0) * np.sin(2 * np.pi * 7.13 * tt + rng.uniform(0, 6.28)) \
                * rng.uniform(0.2, 1.0, n_r)
            E = np.deg2rad(12.5) * np.cos(2 * np.pi * 11.7 * tt) * rng.uniform(0.2, 1.0, n_r) \
                - np.deg2rad(6.0)
        else:
            n_r = int(n_q * 1.5 * attempt)
            A = rng.uniform(0, 2 * np.pi, n_r)
            E = np.deg2rad(rng.uniform(-35.0, 5.0, n_r))
        d_l = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], 1)
        d_w = d_l @ R_wl.T
        t = _raycast(world, origin, d_w, max_range)
        t = t + rng.normal(0, range_sigma, len(t))
        ok = np.isfinite(t) & (t > blind) & (t < max_range)
        pb = (d_l[ok] * t[ok, None])
        body_parts.append(pb)
        have += len(pb)
        if attempt > 8:
            raise RuntimeError("world too small for the requested scan size")
    body = np.concatenate(body_parts, 0)
    sel = np.sort(rng.choice(len(body), n_q, replace=False))
    body = body[sel].astype(np.float32)
    return Scan(body=np.ascontiguousarray(body), pos=p_wi, rot=_quat_from_rot(R_wi),
                off_R=_quat_from_rot(off_R), off_T=off_T.copy(), extrinsic_est_en=ext_en)


def perturb_state(scan: Scan, dpos: float = 0.10, drot_deg: float = 0.5, seed: int = SEED + 2):
    """Ground truth (+) a fixed-size perturbation, so that several ESKF iterations are exercised."""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=3)
    d = d / np.linalg.norm(d) * dpos
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(drot_deg)
    dq = np.concatenate([axis * math.sin(ang / 2), [math.cos(ang / 2)]])
    x, y, z, w = scan.rot
    dx, dy, dz, dw = dq
    # q * dq  (Hamilton, xyzw storage)
    rot = np.array([
        w * dx + x * dw + y * dz - z * dy,
        w * dy - x * dz + y * dw + z * dx,
        w * dz + x * dy - y * dx + z * dw,
        w * dw - x * dx - y * dy - z * dz])
    return scan.pos + d, rot / np.linalg.norm(rot)


def state_vector(pos, rot, off_R, off_T) -> np.ndarray:
    """The 14-double pose block handed across the C ABI: pos3, rot xyzw, offR xyzw, offT3."""
    return np.concatenate([pos, rot, off_R, off_T]).astype(np.float64)


def init_cov() -> np.ndarray:
    """Post-IMU_init covariance (src/IMU_Processing.hpp:224-230)."""
    P = np.eye(23)
    for i in range(6, 12):
        P[i, i] = 1e-5
    for i in range(15, 18):
        P[i, i] = 1e-4
    for i in range(18, 21):
        P[i, i] = 1e-3
    P[21, 21] = P[22, 22] = 1e-5
    return P
