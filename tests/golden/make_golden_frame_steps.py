"""Generates tests/golden/frame_steps_small.npz — golden vectors for the steps either side of the registration path (SURVEY.md
section 8f rows 1-3), produced by plain-Python / NumPy restatements of the reference, not by the C oracle and not by the product:

* sub_sample_frame (src/ct_icp/ct_icp.cpp:65-83): first point per voxel of static_cast<short>(raw / size), in first-insertion order,
* AdaptiveSamplePointsInGrid (include/ct_icp/algorithm/sampling.h:55-110): dict per range band, first k indices per voxel, the
  `size() > max` stop, order band -> voxel (z, y, x) -> index,
* the undistortion loop (src/ct_icp/odometry.cpp:461-486): InterpolatePose(t) * raw through oracle/numpy_check.py (slerp + lerp),
* InsertPointInVoxelMap + RemoveElementsFarFromLocation (include/ct_icp/map.h:261-293,305-322) on a dict of lists.

    python tests/golden/make_golden_frame_steps.py        # rewrites frame_steps_small.npz (deterministic)
"""
import bisect
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import numpy_check as npc             # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
BANDS = ((0.5, 0.1), (2.0, 0.2), (4.0, 0.4), (8.0, 0.8), (16.0, 1.6), (200.0, -1.0))   # sampling.h:18-25


def grid_sampling(pts, size):
    seen, out = set(), []
    for i, p in enumerate(pts):
        v = tuple(int(np.int16(int(c / size))) for c in p)          # static_cast<short>
        if v not in seen:
            seen.add(v)
            out.append(i)
    return np.array(out, dtype=np.uint32)


def adaptive_sampling(pts, bands, k, max_points):
    dist = [b[0] for b in bands]
    maps = [dict() for _ in bands]
    for i, p in enumerate(pts):
        d = float(np.sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]))
        lw = bisect.bisect_left(dist, d)
        if not (dist[0] <= d < dist[-1]) or lw == 0:
            continue
        size = bands[lw - 1][1]
        lst = maps[lw - 1].setdefault(tuple(int(c / size) for c in p), [])
        if len(lst) < k:
            lst.append(i)
    out, limit = [], (max_points if max_points > 0 else 2 ** 31)
    for m in maps:
        for vox in sorted(m, key=lambda v: (v[2], v[1], v[0])):
            for i in m[vox]:
                if len(out) > limit:
                    break
                out.append(i)
    return np.array(out, dtype=np.uint32)


class DictMap:
    """One resolution of MultipleResolutionVoxelMap: voxel -> list of points, map.h:261-293 / 305-322."""

    def __init__(self, resolution, min_dist, max_points):
        self.res, self.min_d2, self.cap, self.vox = resolution, min_dist * min_dist, max_points, {}

    def insert(self, pts):
        kept = np.zeros(len(pts), dtype=bool)
        for i, p in enumerate(pts):
            v = tuple(int(c / self.res) for c in p)
            cell = self.vox.get(v)
            if cell is None:
                self.vox[v] = [p.copy()]
                kept[i] = True
            elif len(cell) < self.cap and min(float(np.sum((q - p) ** 2)) for q in cell) > self.min_d2:
                cell.append(p.copy())
                kept[i] = True
        return kept

    def remove_far(self, loc, distance):
        for v in [v for v, cell in self.vox.items() if float(np.sum((cell[0] - loc) ** 2)) > distance * distance]:
            del self.vox[v]

    def points(self):
        return np.array([p for cell in self.vox.values() for p in cell]).reshape(-1, 3)


def main():
    rng = np.random.default_rng(20240924)
    # a sensor-frame scan: points on a ground disc, two walls and scattered clutter, 4 000 points, ranges 0.2 .. 60 m
    n = 4000
    ang, rad = rng.uniform(0, 2 * np.pi, n), 60.0 * rng.uniform(0.003, 1.0, n) ** 1.7
    raw = np.stack([rad * np.cos(ang), rad * np.sin(ang), rng.normal(-1.7, 0.03, n)], 1)
    wall = rng.random(n) < 0.3
    raw[wall, 1] = np.where(rng.random(wall.sum()) < 0.5, 9.0, -7.5) + rng.normal(0, 0.02, wall.sum())
    raw[wall, 2] = rng.uniform(-1.7, 4.0, wall.sum())
    raw[:6] = [[0.5, 0, 0], [0.3, 0.4, 0.0], [0.1, 0.05, 0.02], [150.0, 140.0, 3.0], [-0.79, 0.81, -0.01], [0.8, -0.8, 0.0]]
    t = np.sort(rng.uniform(0.3, 0.4, n))
    tbe = np.array([0.3, 0.4])
    pose = np.array([0.01, -0.02, 0.03, 1.0, 12.0, -3.0, 0.4, -0.015, 0.01, 0.06, 1.0, 13.1, -2.6, 0.45])
    pose[0:4] /= np.linalg.norm(pose[0:4])
    pose[7:11] /= np.linalg.norm(pose[7:11])
    out = dict(raw=raw, t=t, tbe=tbe, pose=pose)
    for size in (0.5, 1.5):
        out[f"grid_{size}"] = grid_sampling(raw, size)
    out["adaptive_default"] = adaptive_sampling(raw, BANDS, 1, -1)
    out["adaptive_k2_max300"] = adaptive_sampling(raw, BANDS, 2, 300)
    world = npc.ct_transform(pose, tbe, t, raw)
    out["world"] = world
    # map update: two batches into a {0.8 m, 0.1 m, 6 points} level, eviction in between
    m = DictMap(0.8, 0.1, 6)
    out["map_params"] = np.array([0.8, 0.1, 6.0])
    out["insert_kept_1"] = m.insert(world[:2500])
    loc = pose[11:14]
    out["remove_loc"], out["remove_distance"] = loc, np.array(35.0)
    m.remove_far(loc, 35.0)
    out["points_after_remove"] = np.array(sorted(map(tuple, m.points())))
    out["insert_kept_2"] = m.insert(world[2500:])
    out["points_final"] = np.array(sorted(map(tuple, m.points())))
    np.savez_compressed(os.path.join(HERE, "frame_steps_small.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
