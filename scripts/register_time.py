#!/usr/bin/env python3
"""A/B of the small-frame Register call (bench.measure_frames_per_sec: host WPoint3D buffer in, pose + world points out) under tuning
switches. Usage: scripts/register_time.py [--robust] [workload] key=v0,v1 [key=v0,v1 ...]   (measurement script)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, ct_icp_amd as cia
from ct_icp_amd import se3, synthetic as syn, _lib as L

robust = "--robust" in sys.argv            # the CERES-profile route (bench.measure_robust_frames_per_sec) instead of GN
argv = [a for a in sys.argv[1:] if a != "--robust"]
wl = argv[0] if argv and "=" not in argv[0] else "B1"
switches = [a for a in argv if "=" in a] or ["host_threads=3"]
args = argparse.Namespace(map_frames=20, d_sweeps=8, d_radius=100.0, local_rank=0)
W = bench.build_workload(wl, 0, 1, args, cia, syn, se3)
probe = cia.GnSolver(W["gm"])
for sw in switches:
    key, vals = sw.split("=")
    for rep in range(2):
        for v in vals.split(","):
            L.lib().ctgn_set_tuning(key.encode(), float(v))
            c0 = probe.path_counters()
            r = (bench.measure_robust_frames_per_sec(cia, W["gm"], W["inp"], syn, se3, W["mm"], reps=40) if robust else
                 bench.measure_frames_per_sec(cia, W["gm"], W["inp"], syn, se3, W["mm"], reps=80))
            c1 = probe.path_counters()
            print(f"{wl} {key}={v}: Register{' (robust route)' if robust else ''} {r['ms_per_frame']:.4f} ms ({r['keypoints']} keypoints, {r.get('gn_iterations', r.get('icp_iterations'))} iterations; "
                  f"pre-summed residual launches {c1[0] - c0[0]})", flush=True)
    L.lib().ctgn_set_tuning(key.encode(), -1.0)
