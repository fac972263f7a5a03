#!/usr/bin/env python3
"""A/B of the small-frame Register call (bench.measure_frames_per_sec: host WPoint3D buffer in, pose + world points out) under tuning
switches. Usage: scripts/register_time.py [workload] key=v0,v1 [key=v0,v1 ...]   (measurement script)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, ct_icp_amd as cia
from ct_icp_amd import se3, synthetic as syn, _lib as L

wl = sys.argv[1] if len(sys.argv) > 1 and "=" not in sys.argv[1] else "B1"
switches = [a for a in sys.argv[1:] if "=" in a] or ["host_threads=3"]
args = argparse.Namespace(map_frames=20, d_sweeps=8, d_radius=100.0, local_rank=0)
W = bench.build_workload(wl, 0, 1, args, cia, syn, se3)
probe = cia.GnSolver(W["gm"])
for sw in switches:
    key, vals = sw.split("=")
    for rep in range(2):
        for v in vals.split(","):
            L.lib().ctgn_set_tuning(key.encode(), float(v))
            c0 = probe.path_counters()
            r = bench.measure_frames_per_sec(cia, W["gm"], W["inp"], syn, se3, W["mm"], reps=80)
            c1 = probe.path_counters()
            print(f"{wl} {key}={v}: Register {r['ms_per_frame']:.4f} ms ({r['keypoints']} keypoints, {r['gn_iterations']} iterations; "
                  f"pre-summed residual launches {c1[0] - c0[0]})", flush=True)
    L.lib().ctgn_set_tuning(key.encode(), -1.0)
