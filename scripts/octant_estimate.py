"""How many map points would a bounded search still stream if the points of a voxel block were kept in octant order (DESIGN.md section 0,
"the lever that is left")? NumPy only, workload B2's inputs (bench.make_inputs_large): for a sample of keypoints, the k-th neighbour
distance by brute force over the 27-voxel sweep, then the points of (a) every voxel whose box the ball reaches — what the kernel streams
today, in whole 16-point chunks — and (b) every OCTANT (half-size box) the ball reaches, octants of a voxel contiguous, in 16-point chunks.
Bounds looked at: the k-th distance itself (a perfectly bounded search), 1.25 x the surface estimate the first search guesses
(ctgn_api.hip, launch_accumulate). The map is rebuilt only where the sample needs it, with the map's own rule (first 30 points per
0.8 m voxel, 0.1 m apart, insertion order)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
from ct_icp_amd import se3                              # noqa: E402

RES, CAP, MIN_D, K, RADIUS = 0.8, 30, 0.1, 20, 0.75


def vox(p):
    return np.floor(p / RES).astype(np.int64)


def key(v):
    return (v[..., 0] + (1 << 20)) << 42 | (v[..., 1] + (1 << 20)) << 21 | (v[..., 2] + (1 << 20))


def main(n_sample=1500, seed=0):
    t0 = time.time()
    inp = bench.make_inputs_large(0)
    world = se3.ct_transform(inp["pose_gt"], inp["tbe"], inp["t"], inp["raw"])
    rng = np.random.default_rng(seed)
    kp = world[rng.choice(len(world), n_sample, replace=False)]
    offs = np.array([(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)], dtype=np.int64)
    need = np.unique(key(vox(kp)[:, None, :] + offs[None]).ravel())
    mp = inp["map_points"]
    mk = key(vox(mp))
    sel = np.flatnonzero(np.isin(mk, need))
    order = sel[np.argsort(mk[sel], kind="stable")]                      # insertion order kept inside a voxel
    ks = mk[order]
    starts = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    ends = np.r_[starts[1:], len(ks)]
    blocks = {}
    for s, e in zip(starts, ends):
        pts = mp[order[s:e]]
        kept = []
        for p in pts:
            if len(kept) == CAP:
                break
            if not kept or np.min(np.sum((np.array(kept) - p) ** 2, axis=1)) >= MIN_D * MIN_D:
                kept.append(p)
        blocks[int(ks[s])] = np.array(kept)
    ppv = np.mean([len(b) for b in blocks.values()])
    print(f"# {len(blocks)} voxels rebuilt, {ppv:.1f} points per voxel, {time.time() - t0:.0f} s", file=sys.stderr)
    guess = 1.25 * RES * np.sqrt(K / (np.pi * ppv))
    rows = {"kth": [], "guess": []}
    for p in kp:
        v = vox(p)
        nb = [(v + o, blocks.get(int(key(v + o)))) for o in offs]
        nb = [(c, b) for c, b in nb if b is not None and len(b)]
        if not nb:
            continue
        allp = np.concatenate([b for _, b in nb])
        d = np.sqrt(np.sum((allp - p) ** 2, axis=1))
        d = np.sort(d[d <= RADIUS])
        if len(d) < K:
            continue
        for name, bound in (("kth", d[K - 1]), ("guess", max(guess, 0.0))):
            if name == "guess" and d[K - 1] > guess:
                continue                                             # the guess fails for this keypoint: it is searched again on the radius
            now = oct_ = now16 = oct16 = 0
            for c, b in nb:
                lo = c * RES
                gap = np.maximum(np.maximum(lo - p, p - (lo + RES)), 0.0)
                if np.sum(gap * gap) > bound * bound:
                    continue
                now += len(b)
                now16 += -(-len(b) // 16) * 16
                o = np.minimum(((b - lo) / (RES / 2)).astype(np.int64), 1)
                oid = o[:, 0] * 4 + o[:, 1] * 2 + o[:, 2]
                got = 0
                for q in range(8):
                    olo = lo + np.array([(q >> 2) & 1, (q >> 1) & 1, q & 1]) * (RES / 2)
                    g = np.maximum(np.maximum(olo - p, p - (olo + RES / 2)), 0.0)
                    if np.sum(g * g) <= bound * bound:
                        got += int(np.count_nonzero(oid == q))
                oct_ += got
                oct16 += -(-got // 16) * 16
            rows[name].append((now, oct_, now16, oct16))
    for name, r in rows.items():
        a = np.array(r, dtype=np.float64)
        m = a.mean(axis=0)
        print(f"{name:5s} bound: {len(a)} keypoints | voxel granularity {m[0]:.1f} points ({m[2]:.1f} lanes in 16-point chunks) | "
              f"octant granularity {m[1]:.1f} points ({m[3]:.1f} lanes) | ratio {m[1] / m[0]:.2f} ({m[3] / m[2]:.2f} in chunks)")


if __name__ == "__main__":
    main()
