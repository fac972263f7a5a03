"""Seeded synthetic LiDAR inputs for the parity tests and bench.py (no dataset ships with this image).

Behavioural template: the reference's scene generator (src/SlamCore/experimental/synthetic.cxx — planes,
lines, spheres sampled per frame along an interpolated trajectory; unseeded there, seeded here) and the
6-plane box of its integration test (test/integration/testint_utils.h:39-96). Configs follow SURVEY.md
section 8d: B = HDL-64E over a procedural street, C = HDL-32E with jittery motion, D = Ouster-128-style
dense scan. Everything is ray-cast, so points lie exactly on the primitives before noise is added, and
every point's pose is the slerp/lerp interpolation of its frame's begin/end pose — i.e. the scans are
exactly representable by the continuous-time model the solver fits.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import se3


# --------------------------------------------------------------------------------------------------
# Scene = planes (optionally bounded), axis-aligned boxes, vertical cylinders, spheres
# --------------------------------------------------------------------------------------------------
@dataclass
class Scene:
    planes: np.ndarray      # (P, 4): n.x + d = 0 ; plus bounds in plane_bounds
    plane_bounds: np.ndarray  # (P, 6): xmin,xmax,ymin,ymax,zmin,zmax of the valid hit region
    boxes: np.ndarray       # (B, 6): xmin,ymin,zmin,xmax,ymax,zmax
    cylinders: np.ndarray   # (C, 5): cx, cy, radius, zmin, zmax  (vertical axis)
    spheres: np.ndarray     # (S, 4): cx, cy, cz, radius

    def raycast(self, origins: np.ndarray, dirs: np.ndarray, max_range: float) -> np.ndarray:
        """Nearest positive hit distance per ray (np.inf when nothing within max_range)."""
        n = len(origins)
        best = np.full(n, np.inf)
        eps = 1e-6
        for (a, b, c, d), bd in zip(self.planes, self.plane_bounds):
            denom = dirs @ np.array([a, b, c])
            num = -(origins @ np.array([a, b, c]) + d)
            with np.errstate(divide="ignore", invalid="ignore"):
                t = num / denom
                hit = origins + t[:, None] * dirs
            ok = (np.abs(denom) > 1e-12) & (t > eps) & (t < best)
            ok &= (hit[:, 0] >= bd[0]) & (hit[:, 0] <= bd[1]) & (hit[:, 1] >= bd[2]) & (hit[:, 1] <= bd[3]) \
                  & (hit[:, 2] >= bd[4]) & (hit[:, 2] <= bd[5])
            best = np.where(ok, t, best)
        if len(self.boxes):
            # prune boxes that no ray of this sweep can reach, then slab-test per axis on (rays x boxes) tiles
            omin, omax = origins.min(axis=0) - max_range, origins.max(axis=0) + max_range
            keep = np.all(self.boxes[:, 3:6] >= omin, axis=1) & np.all(self.boxes[:, 0:3] <= omax, axis=1)
            bxs = self.boxes[keep]
            chunk = max(1, int(3_000_000 // max(1, len(bxs))))
            with np.errstate(divide="ignore", invalid="ignore"):
                inv_all = 1.0 / dirs
                for s0 in range(0, n, chunk if len(bxs) else n):
                    if not len(bxs):
                        break
                    tmin = None
                    tmax = None
                    for ax in range(3):
                        o = origins[s0:s0 + chunk, ax, None]
                        inv = inv_all[s0:s0 + chunk, ax, None]
                        t1 = (bxs[None, :, ax] - o) * inv
                        t2 = (bxs[None, :, 3 + ax] - o) * inv
                        lo_, hi_ = np.fmin(t1, t2), np.fmax(t1, t2)
                        tmin = lo_ if tmin is None else np.fmax(tmin, lo_)
                        tmax = hi_ if tmax is None else np.fmin(tmax, hi_)
                    tmin[(tmax < tmin) | (tmin <= eps)] = np.inf
                    best[s0:s0 + chunk] = np.minimum(best[s0:s0 + chunk], tmin.min(axis=1))
        for cx, cy, r, z0, z1 in self.cylinders:
            ox, oy = origins[:, 0] - cx, origins[:, 1] - cy
            dx, dy = dirs[:, 0], dirs[:, 1]
            A = dx * dx + dy * dy
            B = 2 * (ox * dx + oy * dy)
            Cc = ox * ox + oy * oy - r * r
            disc = B * B - 4 * A * Cc
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (-B - np.sqrt(np.maximum(disc, 0))) / (2 * A)
            z = origins[:, 2] + t * dirs[:, 2]
            ok = (disc > 0) & (A > 1e-12) & (t > eps) & (t < best) & (z >= z0) & (z <= z1)
            best = np.where(ok, t, best)
        for cx, cy, cz, r in self.spheres:
            oc = origins - np.array([cx, cy, cz])
            B = 2 * np.einsum("ij,ij->i", oc, dirs)
            Cc = np.einsum("ij,ij->i", oc, oc) - r * r
            disc = B * B - 4 * Cc
            t = (-B - np.sqrt(np.maximum(disc, 0))) / 2
            ok = (disc > 0) & (t > eps) & (t < best)
            best = np.where(ok, t, best)
        best[best > max_range] = np.inf
        return best


def raycast_torch(scene: Scene, origins: np.ndarray, dirs: np.ndarray, max_range: float, device=None, budget_elems: int = 40_000_000) -> np.ndarray:
    """Scene.raycast with the (rays x primitives) tests done as broadcast torch float64 operations — on the GPU when there is one.
    For the dense config-D sweep (2.1 M rays against ~8 k primitives: minutes in the per-primitive NumPy loop, seconds here). Same
    hit rule per primitive as Scene.raycast; the results agree to rounding."""
    import torch
    dev = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    f64 = torch.float64
    n = len(origins)
    eps = 1e-6
    planes = torch.as_tensor(scene.planes, dtype=f64, device=dev)
    pb = torch.as_tensor(scene.plane_bounds, dtype=f64, device=dev)
    boxes = torch.as_tensor(scene.boxes.reshape(-1, 6), dtype=f64, device=dev)
    cyl = torch.as_tensor(scene.cylinders.reshape(-1, 5), dtype=f64, device=dev)
    sph = torch.as_tensor(scene.spheres.reshape(-1, 4), dtype=f64, device=dev)
    nprim = max(1, len(planes), len(boxes), len(cyl), len(sph))
    chunk = max(1024, budget_elems // nprim)
    inf = torch.tensor(float("inf"), dtype=f64, device=dev)
    out = np.empty(n)
    for s0 in range(0, n, chunk):
        o = torch.as_tensor(origins[s0:s0 + chunk], dtype=f64, device=dev)
        d = torch.as_tensor(dirs[s0:s0 + chunk], dtype=f64, device=dev)
        best = torch.full((len(o),), float("inf"), dtype=f64, device=dev)
        if len(planes):
            denom = d @ planes[:, :3].T                                   # (R, P)
            t = -(o @ planes[:, :3].T + planes[:, 3]) / denom
            hit = o[:, None, :] + t[:, :, None] * d[:, None, :]
            ok = (denom.abs() > 1e-12) & (t > eps)
            for a in range(3):
                ok &= (hit[:, :, a] >= pb[:, 2 * a]) & (hit[:, :, a] <= pb[:, 2 * a + 1])
            best = torch.minimum(best, torch.where(ok, t, inf).min(dim=1).values)
        if len(boxes):
            inv = 1.0 / d
            tmin = torch.full((len(o), len(boxes)), -float("inf"), dtype=f64, device=dev)
            tmax = torch.full((len(o), len(boxes)), float("inf"), dtype=f64, device=dev)
            for a in range(3):
                t1 = (boxes[None, :, a] - o[:, a, None]) * inv[:, a, None]
                t2 = (boxes[None, :, 3 + a] - o[:, a, None]) * inv[:, a, None]
                tmin = torch.fmax(tmin, torch.fmin(t1, t2))
                tmax = torch.fmin(tmax, torch.fmax(t1, t2))
            tmin = torch.where((tmax < tmin) | (tmin <= eps), inf, tmin)
            best = torch.minimum(best, tmin.min(dim=1).values)
        if len(cyl):
            ox, oy = o[:, 0, None] - cyl[None, :, 0], o[:, 1, None] - cyl[None, :, 1]
            dx, dy = d[:, 0, None], d[:, 1, None]
            A = dx * dx + dy * dy
            B = 2 * (ox * dx + oy * dy)
            Cc = ox * ox + oy * oy - cyl[None, :, 2] ** 2
            disc = B * B - 4 * A * Cc
            t = (-B - torch.sqrt(torch.clamp(disc, min=0))) / (2 * A)
            z = o[:, 2, None] + t * d[:, 2, None]
            ok = (disc > 0) & (A > 1e-12) & (t > eps) & (z >= cyl[None, :, 3]) & (z <= cyl[None, :, 4])
            best = torch.minimum(best, torch.where(ok, t, inf).min(dim=1).values)
        if len(sph):
            oc = o[:, None, :] - sph[None, :, :3]
            B = 2 * (oc * d[:, None, :]).sum(-1)
            Cc = (oc * oc).sum(-1) - sph[None, :, 3] ** 2
            disc = B * B - 4 * Cc
            t = (-B - torch.sqrt(torch.clamp(disc, min=0))) / 2
            ok = (disc > 0) & (t > eps)
            best = torch.minimum(best, torch.where(ok, t, inf).min(dim=1).values)
        best = torch.where(best > max_range, inf, best)
        out[s0:s0 + chunk] = best.cpu().numpy()
    return out


def _inf_bounds():
    return [-np.inf, np.inf, -np.inf, np.inf, -np.inf, np.inf]


def street_scene(length: float = 400.0, seed: int = 1, half_width=(10.0, 12.0), building_height: float = 12.0) -> Scene:
    """Procedural street along +x: ground plane z=0, rows of box 'buildings' with staggered
    fronts and alleys between them (so the along-track direction is observable, as in a real street), parked box
    'cars', 0.4 m-diameter poles, and two far backstop planes behind the buildings."""
    rng = np.random.default_rng(seed)
    back = 18.0
    planes = [[0, 0, 1, 0.0], [0, 1, 0, -(half_width[0] + back)], [0, 1, 0, half_width[1] + back]]
    bounds = [_inf_bounds(),
              [-np.inf, np.inf, -np.inf, np.inf, 0.0, building_height],
              [-np.inf, np.inf, -np.inf, np.inf, 0.0, building_height]]
    boxes = []
    for side, hw in ((1, half_width[0]), (-1, half_width[1])):
        x = -60.0
        while x < length + 60.0:
            w = rng.uniform(6.0, 18.0)
            front = hw + rng.uniform(0.0, 3.0)
            depth = rng.uniform(8.0, 14.0)
            h = rng.uniform(5.0, building_height)
            if side > 0:
                boxes.append([x, front, 0.0, x + w, front + depth, h])
            else:
                boxes.append([x, -front - depth, 0.0, x + w, -front, h])
            # a porch / bay window on some fronts
            if rng.random() < 0.5:
                bw, bd = rng.uniform(1.5, 4.0), rng.uniform(0.5, 1.5)
                bx = x + rng.uniform(0.5, max(0.6, w - bw - 0.5))
                if side > 0:
                    boxes.append([bx, front - bd, 0.0, bx + bw, front, rng.uniform(2.5, h)])
                else:
                    boxes.append([bx, -front, 0.0, bx + bw, -front + bd, rng.uniform(2.5, h)])
            x += w + (rng.uniform(1.5, 5.0) if rng.random() < 0.6 else 0.0)
    x = -40.0
    while x < length + 40.0:       # parked cars
        side = 1 if rng.random() < 0.5 else -1
        y = side * rng.uniform(3.0, 5.5)
        boxes.append([x, y - 0.9, 0.0, x + rng.uniform(3.8, 4.8), y + 0.9, rng.uniform(1.3, 1.7)])
        x += rng.uniform(6.0, 15.0)
    cyl = []
    x = -45.0
    while x < length + 45.0:       # poles
        for side in (1, -1):
            cyl.append([x + rng.uniform(-1, 1), side * rng.uniform(6.5, 8.0), 0.2, 0.0, rng.uniform(4.0, 8.0)])
        x += 15.0
    return Scene(np.array(planes, float), np.array(bounds, float), np.array(boxes, float), np.array(cyl, float),
                 np.zeros((0, 4)))



def suburb_scene(seed: int = 7, x_range=(-140.0, 190.0), half_extent: float = 125.0, n_buildings: int = 150, n_trees: int = 1700) -> Scene:
    """An open residential block around a road along +x: ground, scattered box buildings, box cars, poles and many trees (sphere crown on
    a cylinder trunk). Unlike the street canyon of config B's small workload, a 64-beam sweep sees structure in every direction out to
    its 100 m range, and the volumetric clutter fills voxels in 3-D — the regime in which a steady-state driving-profile map (0.8 m
    voxels, 100 m eviction radius) reaches the ~3 x 10^5 voxels DESIGN.md section 2 estimates for a KITTI scene."""
    rng = np.random.default_rng(seed)
    planes = [[0, 0, 1, 0.0]]
    bounds = [_inf_bounds()]
    boxes, cyl, sph = [], [], []
    x0, x1 = x_range
    for _ in range(n_buildings):
        w, d, h = rng.uniform(8, 22), rng.uniform(8, 18), rng.uniform(4, 14)
        cx = rng.uniform(x0, x1)
        cy = rng.choice([-1, 1]) * rng.uniform(9.0 + d / 2, half_extent)
        boxes.append([cx - w / 2, cy - d / 2, 0.0, cx + w / 2, cy + d / 2, h])
    x = x0
    while x < x1:                   # parked cars
        side = 1 if rng.random() < 0.5 else -1
        y = side * rng.uniform(3.0, 5.5)
        boxes.append([x, y - 0.9, 0.0, x + rng.uniform(3.8, 4.8), y + 0.9, rng.uniform(1.3, 1.7)])
        x += rng.uniform(6.0, 15.0)
    x = x0
    while x < x1:                   # poles
        for side in (1, -1):
            cyl.append([x + rng.uniform(-1, 1), side * rng.uniform(6.5, 8.0), 0.2, 0.0, rng.uniform(4.0, 8.0)])
        x += 15.0
    for _ in range(n_trees):
        cx = rng.uniform(x0, x1)
        cy = rng.choice([-1, 1]) * rng.uniform(7.5, half_extent)
        r = rng.uniform(1.5, 3.6)
        hz = rng.uniform(2.5, 7.0) + r
        cyl.append([cx, cy, rng.uniform(0.12, 0.3), 0.0, hz - 0.6 * r])
        sph.append([cx, cy, hz, r])
    return Scene(np.array(planes, float), np.array(bounds, float), np.array(boxes, float), np.array(cyl, float), np.array(sph, float))


def sample_scene_surfaces(scene: Scene, center, radius: float = 100.0, density: float = 70.0, noise: float = 0.02, seed: int = 0) -> np.ndarray:
    """World points spread uniformly over the scene's surfaces within `radius` of `center` (density in points / m^2, Gaussian range-like
    noise): what a local map has accumulated after a few hundred sweeps from many viewpoints, without ray-casting those sweeps. The
    points then go through the map's own insert rule (minimum distance, capacity per voxel)."""
    rng = np.random.default_rng(seed)
    c = np.asarray(center, float)
    out = []
    # ground: the disc of the first plane (z = const)
    n = int(np.pi * radius * radius * density)
    rr, th = radius * np.sqrt(rng.random(n)), rng.uniform(0, 2 * np.pi, n)
    z0 = -scene.planes[0, 3] / scene.planes[0, 2]
    out.append(np.stack([c[0] + rr * np.cos(th), c[1] + rr * np.sin(th), np.full(n, z0)], 1))
    for bx in scene.boxes:
        lo, hi = bx[:3], bx[3:]
        if np.linalg.norm(np.clip(c, lo, hi) - c) > radius:
            continue
        w, d, h = hi - lo
        for axis, area in ((0, d * h), (1, w * h)):                 # the two pairs of vertical faces
            for val in (lo[axis], hi[axis]):
                m = int(area * density)
                p = lo + rng.random((m, 3)) * (hi - lo)
                p[:, axis] = val
                out.append(p)
        m = int(w * d * density)                                    # roof
        p = lo + rng.random((m, 3)) * (hi - lo)
        p[:, 2] = hi[2]
        out.append(p)
    for cx, cy, r, za, zb in scene.cylinders:
        if np.hypot(cx - c[0], cy - c[1]) > radius + r:
            continue
        m = int(2 * np.pi * r * (zb - za) * density)
        th = rng.uniform(0, 2 * np.pi, m)
        out.append(np.stack([cx + r * np.cos(th), cy + r * np.sin(th), rng.uniform(za, zb, m)], 1))
    for cx, cy, cz, r in scene.spheres:
        if np.hypot(cx - c[0], cy - c[1]) > radius + r:
            continue
        m = int(4 * np.pi * r * r * density)
        v = rng.normal(size=(m, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        out.append(np.array([cx, cy, cz]) + r * v)
    pts = np.concatenate(out)
    pts = pts[np.linalg.norm(pts - c, axis=1) <= radius]
    if noise > 0:
        pts = pts + rng.normal(0.0, noise, pts.shape)
    return pts[rng.permutation(len(pts))]


def box_scene(half: float = 10.0, n_spheres: int = 4, seed: int = 20240901) -> Scene:
    """Closed 6-plane box (reference test/integration/testint_utils.h:39-96) with a few spheres and pillars inside
    (courtyard-like)."""
    rng = np.random.default_rng(seed)
    planes, bounds = [], []
    for axis in range(3):
        for sgn in (-1.0, 1.0):
            n = [0.0, 0.0, 0.0]
            n[axis] = 1.0
            planes.append(n + [-sgn * half])
            bounds.append(_inf_bounds())
    spheres = [[*rng.uniform(-0.6 * half, 0.6 * half, 3), rng.uniform(0.5, 1.5)] for _ in range(n_spheres)]
    boxes = []
    for _ in range(6):
        c = rng.uniform(-0.8 * half, 0.8 * half, 3)
        s = rng.uniform(0.5, 2.0, 3)
        if np.all(np.abs(c) - s > 1.5):   # keep the sensor corridor near the origin free
            boxes.append([*(c - s), *(c + s)])
    return Scene(np.array(planes, float), np.array(bounds, float), np.array(boxes, float).reshape(-1, 6),
                 np.zeros((0, 5)), np.array(spheres, float).reshape(-1, 4))


# --------------------------------------------------------------------------------------------------
# Sensors: unit directions in the sensor frame + relative firing time in [0, 1)
# --------------------------------------------------------------------------------------------------
def lidar_pattern(kind: str = "hdl64", azimuth_steps: int | None = None, sweeps: int = 1, azimuth_offset: float = 0.0):
    """Beam directions (sensor frame, x forward) and relative firing time of one sweep. `azimuth_offset`: azimuth of the sweep's first column;
    0 = the sweep starts and ends looking straight ahead (rounds 1-4), pi = it starts and ends at the REAR, as a KITTI Velodyne scan does."""
    if kind == "hdl64":          # HDL-64E: +2 .. -24.8 deg, ~0.17 deg azimuth -> ~133 k returns
        elev = np.radians(np.linspace(2.0, -24.8, 64))
        az_steps = azimuth_steps or 2083
    elif kind == "hdl32":        # HDL-32E: +10.67 .. -30.67 deg
        elev = np.radians(np.linspace(10.67, -30.67, 32))
        az_steps = azimuth_steps or 1800
    elif kind == "os128":        # Ouster-128 style: +-22.5 deg, 2048 columns, accumulated sub-sweeps
        elev = np.radians(np.linspace(22.5, -22.5, 128))
        az_steps = azimuth_steps or 2048
    else:
        raise ValueError(kind)
    cols = az_steps * sweeps
    # sub-sweeps are offset by a fraction of a column so accumulated scans are denser, not duplicated
    az = (np.arange(cols) % az_steps + (np.arange(cols) // az_steps) / max(sweeps, 1)) * (2 * np.pi / az_steps) + azimuth_offset
    rel_t = np.arange(cols) / cols
    ce, se_ = np.cos(elev), np.sin(elev)
    dirs = np.stack([np.outer(np.cos(az), ce), np.outer(np.sin(az), ce), np.outer(np.ones_like(az), se_)], axis=-1)
    return dirs.reshape(-1, 3), np.repeat(rel_t, len(elev))


# --------------------------------------------------------------------------------------------------
# Trajectory: frame k spans [k*dt, (k+1)*dt]; end pose of frame k == begin pose of frame k+1
# --------------------------------------------------------------------------------------------------
def driving_trajectory(num_frames: int, dt: float = 0.1, speed: float = 10.0, yaw_rate: float = 0.1,
                       height: float = 1.73, jitter: float = 0.0, seed: int = 0, start_x: float = 0.0, ramp_frames: int = 0,
                       centered: bool = False):
    """Knot poses (num_frames+1, 7) of a constant-speed, constant-yaw-rate vehicle; optional roll/pitch jitter
    (config C). Yaw oscillates so the vehicle stays inside the street. `ramp_frames` > 0: the vehicle pulls away from rest, its speed
    rising linearly to `speed` over that many frames (a recording that starts with the car standing, as the KITTI drives do: an
    odometry that starts from the identity — Odometry::InitializeMotion, odometry.cpp:276-300 — has no velocity to extrapolate yet).
    `centered`: the yaw is A sin(2 pi k / 40) in closed form, zero mean, so the vehicle weaves +-0.4 m about the centre line however long the
    drive. The default accumulates yaw_rate dt cos(2 pi k / 40), whose running sum oscillates about +0.5 terms: a mean heading of 0.005 rad,
    0.5 m of lateral drift per 100 m — short workloads never notice, but a 450-frame drive ends up in the row of parked cars at y = 2-4 m
    (found in round 5: this, not the street's geometry, is what lost config E's three longest sequences between frames 260 and 350)."""
    rng = np.random.default_rng(seed)
    poses = np.zeros((num_frames + 1, 7))
    x, y, yaw = start_x, 0.0, 0.0
    for k in range(num_frames + 1):
        roll, pitch = (rng.normal(0, jitter, 2) if jitter > 0 else (0.0, 0.0))
        q = se3.quat_mul(se3.quat_from_rotvec([0, 0, yaw]),
                         se3.quat_mul(se3.quat_from_rotvec([0, pitch, 0]), se3.quat_from_rotvec([roll, 0, 0])))
        poses[k, 0:4] = se3.quat_normalize(q)
        poses[k, 4:7] = [x, y, height]
        v = speed * min(1.0, (k + 0.5) / ramp_frames) if ramp_frames > 0 else speed
        x += v * dt * np.cos(yaw)
        y += v * dt * np.sin(yaw)
        if centered:
            yaw = yaw_rate * dt * (40.0 / (2 * np.pi)) * np.sin(2 * np.pi * (k + 1) / 40.0)
        else:
            yaw += yaw_rate * dt * np.cos(2 * np.pi * k / 40.0)
    return poses


def frame_pose14(knots: np.ndarray, k: int) -> np.ndarray:
    return np.concatenate([knots[k], knots[k + 1]])


@dataclass
class Scan:
    raw: np.ndarray          # (N, 3) sensor-frame points
    t: np.ndarray            # (N,) absolute timestamps
    world_gt: np.ndarray     # (N, 3) ground-truth world points
    pose_gt: np.ndarray      # (14,) begin|end ground-truth pose
    t_begin_end: np.ndarray  # (2,)


def generate_scan(scene: Scene, dirs: np.ndarray, rel_t: np.ndarray, pose14: np.ndarray, t_begin: float,
                  t_end: float, max_range: float = 100.0, min_range: float = 1.0, noise: float = 0.0,
                  seed: int = 0, use_torch: bool = False) -> Scan:
    """Ray-cast one sweep with the sensor moving continuously from the begin to the end pose (use_torch: raycast_torch)."""
    rng = np.random.default_rng(seed)
    alpha = rel_t
    q = se3.quat_normalize(se3.quat_slerp(pose14[0:4], pose14[7:11], alpha))
    origin = (1 - alpha)[:, None] * pose14[4:7] + alpha[:, None] * pose14[11:14]
    wdirs = se3.quat_rotate(q, dirs)
    rng_hit = raycast_torch(scene, origin, wdirs, max_range) if use_torch else scene.raycast(origin, wdirs, max_range)
    ok = np.isfinite(rng_hit) & (rng_hit > min_range)
    r = rng_hit[ok]
    if noise > 0:
        r = r + rng.normal(0.0, noise, len(r))
    raw = dirs[ok] * r[:, None]
    t = t_begin + alpha[ok] * (t_end - t_begin)
    t = np.clip(t, t_begin, t_end)
    world = se3.quat_rotate(q[ok], raw) + origin[ok]
    return Scan(raw, t, world, np.asarray(pose14, float).copy(), np.array([t_begin, t_end]))


def grid_sample_indices(points: np.ndarray, voxel_size: float) -> np.ndarray:
    """Indices kept by the reference's sub_sample_frame (src/ct_icp/ct_icp.cpp:65-83): first point of every
    voxel, voxel = static_cast<short>(p / size) per axis. Output sorted by first occurrence (the reference's
    robin_map iteration order is unspecified)."""
    v = np.trunc(np.asarray(points, float) / voxel_size).astype(np.int64)
    v = ((v + 32768) % 65536) - 32768      # static_cast<short>
    key = (v[:, 0] + 32768) * 65536 * 65536 + (v[:, 1] + 32768) * 65536 + (v[:, 2] + 32768)
    _, first = np.unique(key, return_index=True)
    return np.sort(first)


def perturb_pose(pose14: np.ndarray, rot: float, trans: float, seed: int = 0) -> np.ndarray:
    """Initial guess = ground truth composed with a small random left rotation / translation per end."""
    rng = np.random.default_rng(seed)
    out = np.asarray(pose14, float).copy()
    for off in (0, 7):
        rv = rng.normal(size=3)
        rv *= rot / np.linalg.norm(rv)
        out[off:off + 4] = se3.quat_normalize(se3.quat_mul(se3.quat_from_rotvec(rv), out[off:off + 4]))
        tv = rng.normal(size=3)
        out[off + 4:off + 7] += tv * trans / np.linalg.norm(tv)
    return out
