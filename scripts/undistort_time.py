#!/usr/bin/env python3
"""Stage-call undistortion of a full scan through host views (ctgn_transform_points): time per call on a warm handle with the output
buffer reused, and with a fresh output array per call (its page faults). The CPU loop on the same inputs is timed by bench.py
(cpu_baseline.frame_stages). Measurement script."""
import argparse, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, ct_icp_amd as cia
from ct_icp_amd import se3, synthetic as syn
args = argparse.Namespace(map_frames=2, d_sweeps=8, d_radius=100.0, local_rank=0)
W = bench.build_workload("B2", 0, 1, args, cia, syn, se3)
raw, t, pose, tbe = W["raw"], W["t"], W["pose0"], W["inp"]["tbe"]
gm = W["gm"]
out = np.zeros_like(raw)
for _ in range(3):
    cia.transform_points(gm, raw, t, pose, tbe, out=out)
ts = []
for _ in range(20):
    t0 = time.perf_counter(); cia.transform_points(gm, raw, t, pose, tbe, out=out); ts.append((time.perf_counter() - t0) * 1e3)
fresh = []
for _ in range(5):
    t0 = time.perf_counter(); cia.transform_points(gm, raw, t, pose, tbe); fresh.append((time.perf_counter() - t0) * 1e3)
want = se3.ct_transform(pose, tbe, t, raw)
print("points %d  stage call: median %.3f ms (min %.3f), with a fresh output array %.3f ms;  max |diff| to the NumPy transform %.2e" %
      (len(t), np.median(ts), min(ts), np.median(fresh), np.abs(out - want).max()))
