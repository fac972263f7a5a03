"""Timing of the robust-loss (CERES-profile) route on the GPU: whole CT_ICP_Registration::Register calls
(DESIGN.md section 9). The CPU comparison lives in bench.py's cpu_baseline leg.

  python scripts/robust_bench.py [--reps 20] [--full]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
import ct_icp_amd as cia  # noqa: E402
from ct_icp_amd import se3, synthetic as syn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--full", action="store_true", help="also time the whole 0.5 m-subsampled sweep without a cap")
    ap.add_argument("--no-cpu", action="store_true", help="accepted for compatibility; this script is GPU-only")
    args = ap.parse_args()
    inp = bench.make_inputs(0, 20)
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(0.8, 0.1, 30)], default_radius=0.75))
    gm.InsertPointCloud(inp["map_points"])
    mm = cia.PreviousFrameMotionModel()
    prev = np.concatenate([[0, 0, 0, 1.0], inp["prev_b"], [0, 0, 0, 1.0], inp["prev_e"]])
    mm.previous_frame = cia.TrajectoryFrame.from_pose14(prev, 0.0, 0.0)
    raw, t = inp["raw"], inp["t"]
    sel05 = syn.grid_sample_indices(raw, 0.5)
    sel15 = sel05[syn.grid_sample_indices(raw[sel05], 1.5)]
    pose0 = syn.perturb_pose(inp["pose_gt"], 0.003, 0.03, seed=4)
    # driving profile: config/odometry/driving_config.yaml:52-89
    cases = [("driving profile (1.5 m keypoints, cap 900, 5 x 5)", sel15,
              dict(num_iters_icp=5, ls_max_num_iters=5, max_num_residuals=900, loss_function="CAUCHY", ls_sigma=0.1))]
    if args.full:
        cases.append(("whole sweep (0.5 m grid, no cap, 5 x 5)", sel05,
                      dict(num_iters_icp=5, ls_max_num_iters=5, loss_function="CAUCHY", ls_sigma=0.1)))
    for name, sel, kw in cases:
        o = cia.CTICPOptions(solver=cia.CERES, debug_print=False, **kw)
        kps = np.zeros(len(sel), dtype=cia.WPOINT3D_DTYPE)
        kps["raw_point"], kps["t"] = raw[sel], t[sel]
        reg = cia.CT_ICP_Registration(o)
        times = []
        for _ in range(args.reps):
            frame = cia.TrajectoryFrame.from_pose14(pose0, *inp["tbe"])
            t0 = time.perf_counter()
            summ = reg.Register(gm, kps, frame, mm)
            times.append(time.perf_counter() - t0)
        med = float(np.median(times[2:]))
        s = cia.GnSolver(gm)
        s._n = len(sel)
        rep = s.robust_report()
        tr, rot = se3.pose_error(frame.pose14(), inp["pose_gt"])
        print(f"[gpu] {name}: {len(sel)} keypoints, {summ.num_residuals_used} residuals, {summ.num_iters} ICP iterations, "
              f"{rep['ls_iterations']} LM iterations ({rep['ls_accepted']} accepted): {med * 1e3:.3f} ms/frame "
              f"({1 / med:.1f} frames/s); error vs ground truth {tr * 100:.2f} cm / {np.degrees(rot):.4f} deg")
        print("      step-kernel clocks (stage-in+reduce | control | scale | solve | candidate):", rep["step_cycles"][:5])


if __name__ == "__main__":
    main()
