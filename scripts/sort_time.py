#!/usr/bin/env python3
"""Device time of the home-voxel ordering (hand-written radix sort + permute) on workload D's keypoints. Measurement script."""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, ct_icp_amd as cia
from ct_icp_amd import se3, synthetic as syn
args = argparse.Namespace(map_frames=20, d_sweeps=int(sys.argv[1]) if len(sys.argv) > 1 else 8, d_radius=40.0, local_rank=0)
W = bench.build_workload("D", 0, 1, args, cia, syn, se3)
s = cia.GnSolver(W["gm"]); s.set_rewind(True)
s.set_keypoints(W["raw"], W["world0"], W["t"])
o = cia.CTICPOptions(solver=cia.GN, num_iters_icp=1, threshold_orientation_norm=0.0, debug_print=False)
for order in (0, 1):
    s.set_ordering(order)
    s.set_keypoints(W["raw"], W["world0"], W["t"])
    ts = []
    for _ in range(6):
        s.rewind(); torch.cuda.synchronize(); t0 = time.perf_counter()
        s.gn_begin(W["pose0"], W["inp"]["tbe"], o, W["mm"]); s.gn_iterate(1); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3); s.gn_end()
    print("n", len(W["t"]), "ordering", order, "first iteration of a solve (ms):", [round(t, 3) for t in ts])
