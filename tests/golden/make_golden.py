"""Generates tests/golden/gn_small.npz — golden vectors for the GN path produced by the INDEPENDENT NumPy/SciPy
re-derivation (oracle/numpy_check.py), not by the C oracle and not by the product. The reference itself cannot be
run here, so these vectors pin the oracle against a second derivation of the same cited lines.

    python tests/golden/make_golden.py        # rewrites gn_small.npz (deterministic)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ct_icp_amd import se3, synthetic as syn      # noqa: E402  (scene generator + host SE(3) helpers only)
from oracle import numpy_check as npc             # noqa: E402


def np_insert(points, resolution, min_dist, max_pts):
    """Independent restatement of InsertPointInVoxelMap (include/ct_icp/map.h:261-293) with Python containers."""
    vox = {}
    kept = np.zeros(len(points), dtype=bool)
    for i, p in enumerate(points):
        key = tuple(np.trunc(p / resolution).astype(np.int64))
        blk = vox.get(key)
        if blk is None:
            vox[key] = [p]
            kept[i] = True
        elif len(blk) < max_pts:
            d2 = min(float(np.sum((q - p) ** 2)) for q in blk)
            if d2 > min_dist * min_dist:
                blk.append(p)
                kept[i] = True
    return vox, kept


def main():
    resolution, min_dist, max_pts, radius = 0.5, 0.05, 20, 0.8
    k, min_nb, max_dist = 20, 20, 0.3
    scene = syn.box_scene(6.0, n_spheres=2, seed=7)
    el = np.radians(np.linspace(-70, 70, 36))
    az = np.linspace(0, 2 * np.pi, 150, endpoint=False)
    dirs = np.stack([np.outer(np.cos(az), np.cos(el)), np.outer(np.sin(az), np.cos(el)),
                     np.outer(np.ones_like(az), np.sin(el))], -1).reshape(-1, 3)
    rel_t = np.repeat(np.arange(len(az)) / len(az), len(el))
    knots = np.zeros((5, 7))
    for j in range(5):
        knots[j, :4] = se3.quat_from_rotvec(np.array([0.03 * j, -0.02 * j, 0.06 * j]))
        knots[j, 4:] = [0.15 * j, 0.08 * j, -0.03 * j]
    insert_points = []
    for j in range(3):
        sc = syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, j), 0.1 * j, 0.1 * (j + 1), max_range=30.0,
                               min_range=0.3, noise=0.01, seed=50 + j)
        insert_points.append(sc.world_gt)
    insert_points = np.concatenate(insert_points)
    vox, kept = np_insert(insert_points, resolution, min_dist, max_pts)
    map_points = np.concatenate([np.array(v) for v in vox.values()])
    sc = syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, 3), 0.3, 0.4, max_range=30.0, min_range=0.3,
                           noise=0.01, seed=99)
    sel = syn.grid_sample_indices(sc.raw, 0.7)[:400]
    raw, t = sc.raw[sel], sc.t[sel]
    pose0 = syn.perturb_pose(sc.pose_gt, 0.01, 0.04, seed=3)
    world0 = npc.ct_transform(pose0, sc.t_begin_end, t, raw)
    nb = int(np.ceil(radius / resolution))
    A, b, n_used, info = npc.gn_accumulate(map_points, resolution, nb, radius, raw, world0, t, pose0, sc.t_begin_end,
                                           k=k, min_nb=min_nb, max_dist=max_dist)
    prior = (0.001, 0.001, knots[2, 4:7], knots[3, 4:7])
    pose1, x = npc.gn_solve_update(A, b, n_used, pose0, prior)
    world1 = npc.ct_transform(pose1, sc.t_begin_end, t, raw)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gn_small.npz")
    np.savez_compressed(out, resolution=resolution, min_dist=min_dist, max_pts=max_pts, radius=radius, k=k, min_nb=min_nb,
                        max_dist=max_dist, insert_points=insert_points, insert_kept=kept, map_points=map_points,
                        raw=raw, t=t, world0=world0, pose0=pose0, tbe=sc.t_begin_end, pose_gt=sc.pose_gt,
                        A=A, b=b, n_used=n_used, n_neighbors=info["n_neighbors"], normal=info["normal"], a2d=info["a2d"],
                        farthest=info["farthest"], used=info["used"], prior_beta=np.array(prior[:2]),
                        prior_prev_b=prior[2], prior_prev_e=prior[3], pose1=pose1, x=x, world1=world1)
    print("wrote", out, "map points", len(map_points), "keypoints", len(t), "n_used", n_used)


if __name__ == "__main__":
    main()
