#!/usr/bin/env python3
"""Which keypoints does the pool check fail, and why? One fresh solve of a bench workload, one iteration per call; after each the carried
search state (ctgn_debug_pool_state: completeness radius r, k-th distance d_k, record count word) and the world points are read back, and
the check of the NEXT iteration is predicted on the host from them: a keypoint passes iff d_k' < r - moved (d_k' unknown: bracketed by
d_k -/+ moved). Prints, per iteration, how many keypoints have a pool, their margin r - d_k against the distance they then move, and the
range / neighbour-count profile of the ones that cannot pass. Measurement script.   usage: pool_probe.py [B2|D]"""
import argparse, sys, os, json
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ct_icp_amd as cia
from ct_icp_amd import se3, synthetic as syn, _lib as L

wl = sys.argv[1] if len(sys.argv) > 1 else "B2"
args = argparse.Namespace(map_frames=20, d_sweeps=8, d_radius=100.0, local_rank=0)
W = bench.build_workload(wl, 0, 1, args, cia, syn, se3)
s = cia.GnSolver(W["gm"])
s.set_rewind(True)
s.set_keypoints(W["raw"], W["world0"], W["t"])
n = len(W["t"])
ipf = W["ipf"]
o = cia.CTICPOptions(solver=cia.GN, num_iters_icp=ipf, min_number_neighbors=W["min_nb"], threshold_orientation_norm=0.0, debug_print=False)
s.rewind(); s.gn_begin(W["pose0"], W["inp"]["tbe"], o, W["mm"])
rng = np.linalg.norm(np.asarray(W["raw"], float).reshape(-1, 3), axis=1)


def state():
    kth = np.zeros(2 * n, np.float32); cnt = np.zeros(n, np.uint32)
    L.check(s._h, L.lib().ctgn_debug_pool_state(s._h, kth.ctypes.data_as(C.POINTER(C.c_float)), cnt.ctypes.data_as(C.POINTER(C.c_uint32)), n))
    return kth.reshape(-1, 2).astype(float), cnt, s.world_points().copy()


prev = None
for it in range(ipf):
    s.gn_iterate(1)
    kth, cnt, world = state()
    r, dk = kth[:, 0], kth[:, 1]
    nn, m = cnt & 63, (cnt >> 8) & 63
    rec = dict(workload=wl, iteration=it, keypoints=n, with_pool=int((r > 0).sum()), pool_size_mean=float(m[r > 0].mean()) if (r > 0).any() else 0,
               fewer_than_k=int((nn < 20).sum()))
    if prev is not None:
        pr, pdk, pworld, pnn = prev
        moved = np.linalg.norm(world - pworld, axis=1)          # what the keypoints moved between the two searches: the check of THIS iteration saw it
        had = pr > 0
        margin = pr - pdk
        sure_pass = had & (pdk + moved < pr - moved)            # even if the k-th neighbour receded by `moved`
        sure_fail = (~had) | (pdk - moved >= pr - moved)        # even if it came closer by `moved` (i.e. margin <= 0)
        rec.update(checked=int(had.sum()), no_pool=int((~had).sum()), moved_cm_p50_p90=[round(float(np.percentile(moved, q)) * 100, 2) for q in (50, 90)],
                   margin_cm_p10_p50=[round(float(np.percentile(margin[had], q)) * 100, 2) for q in (10, 50)] if had.any() else None,
                   sure_pass=int(sure_pass.sum()), sure_fail=int(sure_fail.sum()), undecided=int(n - sure_pass.sum() - sure_fail.sum()),
                   no_pool_fewer_than_k=int(((~had) & (pnn < 20)).sum()),
                   no_pool_range_m_p50=round(float(np.median(rng[~had])), 1) if (~had).any() else None,
                   thin_margin=int((had & (margin < 2 * moved)).sum()),
                   thin_margin_range_m_p50=round(float(np.median(rng[had & (margin < 2 * moved)])), 1) if (had & (margin < 2 * moved)).any() else None,
                   range_m_p50_all=round(float(np.median(rng)), 1))
    print(json.dumps(rec))
    prev = (r, dk, world, nn)
s.gn_end()
