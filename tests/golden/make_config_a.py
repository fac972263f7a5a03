"""Generates tests/golden/config_a.npz — BASELINE.json configs[0]: the reference's own CPU-runnable synthetic case
(config/synthetic_ct_icp_config.yaml -> config/synthetic/courtyard.yaml: 40 triangles, 9 lines, 4 spheres, 11 key poses).

Runs only where /root/reference exists (it parses the reference's scene FILE; no reference code is used). The sampling follows
the semantics of the reference's generator as SURVEY.md section 8d reads them (src/SlamCore/experimental/synthetic.cxx): per
primitive `num_points_per_primitives` random points (triangle: barycentric weights (1 + U(-1,1))^1.5 normalised, :94-103; line:
|U(-1,1)| weights, :142-146; sphere: uniform direction), a random time inside the frame per point, raw = pose(alpha)^-1 world,
kept if |raw| < max_lidar_distance (:293-327); key poses one second apart, frames at `sample_frequency` (:370-418). The reference
is unseeded (rand()); this script is seeded, and the fixture it writes is what the tests use.

    python tests/golden/make_config_a.py        # rewrites config_a.npz (deterministic)
"""
import os
import sys

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ct_icp_amd import se3      # noqa: E402

SCENE = "/root/reference/config/synthetic/courtyard.yaml"
SEED = 20240901
FRAMES = [(40 + f) for f in range(5)]          # frames 40..44: between key poses 1 and 2 (translation + rotation)


def pose_at(keys, t):
    i = min(int(np.floor(t)), len(keys) - 2)
    a = t - i
    q = se3.quat_normalize(se3.quat_slerp(keys[i][0:4], keys[i + 1][0:4], np.array(a)))
    return np.concatenate([q, (1 - a) * keys[i][4:7] + a * keys[i + 1][4:7]])


def main():
    cfg = yaml.safe_load(open(SCENE))
    acq = cfg["acquisition"]
    freq, max_d, npp = float(cfg["sample_frequency"]), float(cfg["max_lidar_distance"]), int(cfg["num_points_per_primitives"])
    keys = np.array([np.concatenate([se3.quat_normalize(np.array(p["quaternion"], float)), np.array(p["translation"], float)])
                     for p in acq["poses"]])
    tris = np.array(acq["triangles"], float)
    lines = np.array(acq["lines"], float)
    spheres = [(float(s["radius"]), np.array(s["center"], float)) for s in acq["spheres"]]
    rng = np.random.default_rng(SEED)
    out = dict(key_poses=keys, sample_frequency=freq, max_lidar_distance=max_d)
    counts, raws, ts, poses, tbes = [], [], [], [], []
    for f in FRAMES:
        t0, t1 = f / freq, (f + 1) / freq
        pb, pe = pose_at(keys, t0), pose_at(keys, t1)
        world = []
        for tri in tris:
            c = (1.0 + rng.uniform(-1, 1, (npp, 3))) ** 1.5
            c /= c.sum(1, keepdims=True)
            world.append(c @ tri)
        for ln in lines:
            c = np.abs(rng.uniform(-1, 1, (npp, 2)))
            c /= c.sum(1, keepdims=True)
            world.append(c @ ln)
        for rad, cen in spheres:
            d = rng.normal(size=(npp, 3))
            world.append(cen + rad * d / np.linalg.norm(d, axis=1, keepdims=True))
        world = np.concatenate(world)
        alpha = rng.uniform(size=len(world))
        t = t0 * (1 - alpha) + alpha * t1
        q = se3.quat_normalize(se3.quat_slerp(pb[0:4], pe[0:4], alpha))
        tr = (1 - alpha)[:, None] * pb[4:7] + alpha[:, None] * pe[4:7]
        raw = se3.quat_rotate(se3.quat_conj(q), world - tr)
        keep = np.linalg.norm(raw, axis=1) < max_d
        raws.append(raw[keep]); ts.append(t[keep]); counts.append(int(keep.sum()))
        poses.append(np.concatenate([pb, pe])); tbes.append([t0, t1])
    out.update(raw=np.concatenate(raws), t=np.concatenate(ts), counts=np.array(counts), pose_gt=np.array(poses), tbe=np.array(tbes))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_a.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "points per frame", counts, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
