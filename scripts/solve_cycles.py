#!/usr/bin/env python3
"""Shader clocks of the phases of k_reduce_solve on the B1 frame (debug_print = 2 makes ctgn_gn_end print them). Measurement script."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ct_icp_amd as cia
from ct_icp_amd import se3, synthetic as syn
inp = bench.make_inputs(0, 20)
gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(0.8, 0.1, 30)], default_radius=0.75))
gm.InsertPointCloud(inp["map_points"]); gm.Sync()
raw, t = inp["raw"], inp["t"]
if len(sys.argv) > 1 and sys.argv[1] == "B1":
    sel = syn.grid_sample_indices(raw, 0.5); sel = sel[syn.grid_sample_indices(raw[sel], 1.5)]; raw, t = raw[sel], t[sel]
pose0 = syn.perturb_pose(inp["pose_gt"], 0.003, 0.03, seed=4)
world0 = se3.ct_transform(pose0, inp["tbe"], t, raw)
s = cia.GnSolver(gm)
s.set_keypoints(raw, world0, t)
for _ in range(3):
    s.solve(pose0, inp["tbe"], cia.CTICPOptions(solver=cia.GN, num_iters_icp=30, threshold_orientation_norm=0.0, debug_print=2))
