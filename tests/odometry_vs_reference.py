"""One config-E sequence through three loops, frame by frame, with where each departs from ground truth and from the others:

  ref-cpu   the REFERENCE'S OWN ct_icp::Odometry::RegisterFrame (oracle/_ref/libctgn_ref_odometry.so: src/ct_icp/odometry.cpp compiled where it
            lies) on its own MULTI_RESOLUTION_VOXEL_HASHMAP — AssessRegistration, insertion policy, the init regime, everything
  ref-gpu   the same Odometry object code on GPU_VOXEL_HASHMAP (integration/gpu_map.h + the solver arms, libctgn.so underneath)
  ref-gpu-armed   odometry.cpp compiled with the four arms of integration/odometry_gpu_arm.h (oracle/_ref/libctgn_ref_odometry_armed.so):
            InitializeFrame, TryRegister, the undistortion loops and the map half of UpdateMap on the device as well
  ctgn      ct_icp_amd.sequence_runner: one ctgn_frame call per frame (the product's own loop, what bench.py's config_e times)

The sequences are SURVEY.md 8d's config E (config-B generator, seeds 10-20, KITTI lengths / scale). All three loops start from the identity
(Odometry::InitializeMotion, odometry.cpp:276-300), so the vehicle pulls away from rest (`--ramp` frames) as a KITTI drive does.

  python tests/odometry_vs_reference.py --sequence 0 --frames 300 --solver GN --impl ref-cpu,ref-gpu,ctgn --out profiles/r05_config_e_seq0_vs_reference.json

Test infrastructure: lives under tests/ because it drives oracle/_ref (only tests/, smoke() and bench.py's cpu_baseline leg may); not collected by pytest."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ct_icp_amd import se3, synthetic as syn  # noqa: E402
from ct_icp_amd.sequence_runner import KITTI_LENGTHS  # noqa: E402


def make_scans(sid: int, frames: int, ramp: int, azimuth_steps=None, scene_kind: str = "street", log=None, centered: bool = True,
               azimuth_offset: float = 0.0):
    seed = 10 + sid
    scene = syn.config_e_scene(frames, seed) if hasattr(syn, "config_e_scene") and scene_kind == "config_e" else \
        syn.street_scene(max(300.0, frames * 1.2 + 60.0), seed=seed)
    dirs, rel_t = syn.lidar_pattern("hdl64", azimuth_steps=azimuth_steps, azimuth_offset=azimuth_offset)
    knots = syn.driving_trajectory(frames + 1, seed=seed, start_x=20.0, ramp_frames=ramp, centered=centered)
    scans = []
    t0 = time.perf_counter()
    for j in range(frames):
        sc = syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, j), 0.1 * j, 0.1 * (j + 1), noise=0.02, seed=1000 * seed + j, use_torch=True)
        scans.append((sc.raw, sc.t, (0.1 * j, 0.1 * (j + 1))))
        if log and j % 50 == 49:
            log(f"  generated {j + 1}/{frames} scans ({time.perf_counter() - t0:.0f} s)")
    return scans, knots


def relative_truth(knots, j):
    """Ground-truth begin|end pose of frame j expressed in the frame of the sensor at the start of frame 0 (the world of a loop that starts
    from the identity)."""
    g0 = syn.frame_pose14(knots, 0)
    q0c = se3.quat_conj(g0[0:4])
    g = syn.frame_pose14(knots, j)
    out = np.zeros(14)
    for o in (0, 7):
        out[o:o + 4] = se3.quat_normalize(se3.quat_mul(q0c, g[o:o + 4]))
        out[o + 4:o + 7] = se3.quat_rotate(q0c, g[o + 4:o + 7] - g0[4:7])
    return out


def run_reference(scans, map_kind, solver, extra):
    from oracle import ref_odometry as ro
    kind = {"cpu": ro.CPU_MAP, "gpu": ro.GPU_MAP, "gpu-armed": ro.GPU_MAP_ARMED, "gpu-armed-device-shuffle": ro.GPU_MAP_ARMED_DEVICE_SHUFFLE}[map_kind]
    od = ro.RefOdometry(kind, solver=ro.GN if solver == "GN" else ro.CERES, **extra)
    poses, rec = [], dict(success=[], residuals=[], keypoints=[], points_added=[], ms=[], attempts=[], sampled=[], phases={k: [] for k in ro.PHASES},
                     arm={k: [] for k in ro.ARM_PHASES})
    for raw, t, _ in scans:
        r = od.register_frame(raw, t)
        poses.append(r["pose"])
        rec["success"].append(bool(r["success"]))
        rec["residuals"].append(int(r["number_of_residuals"]))
        rec["keypoints"].append(int(r["sample_size"]))
        rec["points_added"].append(bool(r["points_added"]))
        rec["attempts"].append(int(r["number_of_attempts"]))
        rec["sampled"].append(int(r["num_corrected"]))
        rec["ms"].append(float(r["milliseconds"]))
        for k, v in r["phase_ms"].items():
            rec["phases"][k].append(float(v))
        for k, v in r["arm_ms"].items():
            rec["arm"][k].append(float(v))
    return np.array(poses), rec


def run_ctgn(scans, solver, reference_regime: bool):
    import ct_icp_amd as cia
    from ct_icp_amd import sequence_runner as sr
    # from the identity, with the reference's start-up regime and its default motion model on both routes (odometry.h:138-139); the frame's
    # pose interval is the span of its timestamps, as Odometry::RegisterFrame takes it (odometry.cpp:199-205)
    kw = dict(solver=cia.GN if solver == "GN" else cia.CERES, voxel_size=0.5, sample_voxel_size=1.5, max_distance=100.0, init_poses=None, init_frames=1,
              use_motion_model=True)
    scans = [(raw, t, (float(t.min()), float(t.max()))) for raw, t, _ in scans]
    r = sr.run_sequence(scans, **kw)
    return r["poses"], dict(success=[bool(v) for v in r["success"]], keypoints=[int(v) for v in r["keypoints"]],
                            ms=[1e3 * r["seconds"] / max(1, r["frames"])] * r["frames"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sequence", type=int, default=0)
    ap.add_argument("--scale", type=int, default=10)
    ap.add_argument("--frames", type=int, default=300, help="0 = the whole sequence")
    ap.add_argument("--ramp", type=int, default=20)
    ap.add_argument("--azimuth-steps", type=int, default=0, help="0 = the full HDL-64E pattern (133 k rays)")
    ap.add_argument("--solver", default="GN")
    ap.add_argument("--impl", default="ref-cpu")
    ap.add_argument("--scene", default="config_e")
    ap.add_argument("--set", action="append", default=[], help="key=value handed to the reference's OdometryOptions (repeatable)")
    ap.add_argument("--uncentered", action="store_true", help="the trajectory of rounds 1-4 (mean heading 0.005 rad: drifts into the parked cars)")
    ap.add_argument("--front-cut", action="store_true", help="the sweep starts / ends looking straight ahead (rounds 1-4) instead of at the rear (KITTI)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    log = lambda m: print(m, file=sys.stderr, flush=True)
    length = max(3, int(round(KITTI_LENGTHS[args.sequence] / args.scale)))
    frames = length if args.frames <= 0 else min(args.frames, length)
    extra = {k: float(v) for k, v in (kv.split("=") for kv in args.set)}
    log(f"sequence {args.sequence} (seed {10 + args.sequence}): {frames} of {length} frames")
    scans, knots = make_scans(args.sequence, frames, args.ramp, args.azimuth_steps or None, args.scene, log, centered=not args.uncentered,
                             azimuth_offset=0.0 if args.front_cut else np.pi)
    truth = np.array([relative_truth(knots, j) for j in range(frames)])
    result = dict(sequence=args.sequence, seed=10 + args.sequence, frames=frames, length=length, ramp_frames=args.ramp, scene=args.scene, centered=not args.uncentered, sweep_cut="front" if args.front_cut else "rear",
                  points_per_frame=float(np.mean([len(s[1]) for s in scans])), options=extra, runs={})
    poses = {}
    for solver in args.solver.split(","):
        for impl in args.impl.split(","):
            t0 = time.perf_counter()
            if impl in ("ref-cpu", "ref-gpu", "ref-gpu-armed", "ref-gpu-armed-device-shuffle"):
                p, rec = run_reference(scans, impl[4:], solver, extra)
            else:
                p, rec = run_ctgn(scans, solver, impl == "ctgn-ref-regime")
            seconds = time.perf_counter() - t0
            err = np.array([se3.pose_error(p[j], truth[j]) for j in range(frames)])
            # a loop that starts from the identity fixes its world frame with its first two (rigidly registered) scans: the frame of the
            # moving sensor somewhere inside them. Against ground truth that is a gauge offset (a fraction of a degree of heading = metres
            # after a few hundred metres), not drift: also report the error after the best rigid alignment of the end positions (Kabsch).
            P, Q = p[:, 11:14], truth[:, 11:14]
            Pc, Qc = P - P.mean(0), Q - Q.mean(0)
            U, _, Vt = np.linalg.svd(Pc.T @ Qc)
            Rk = (U @ np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))]) @ Vt).T
            aligned = np.linalg.norm((Rk @ Pc.T).T - Qc, axis=1)
            fails = [j for j, ok in enumerate(rec["success"]) if not ok]
            key = f"{impl}:{solver}"
            poses[key] = p
            lost = next((j for j in range(frames) if err[j, 0] > 2.0), None)
            result["runs"][key] = dict(seconds=seconds, failures=len(fails), first_failure=fails[0] if fails else None, failed_frames=fails[:50],
                                       err_tr_max=float(err[:, 0].max()), err_tr_final=float(err[-1, 0]), err_rot_max=float(err[:, 1].max()),
                                       err_tr_max_after_rigid_alignment=float(aligned.max()), err_tr_rms_after_rigid_alignment=float(np.sqrt((aligned ** 2).mean())),
                                       first_frame_beyond_2m=lost, err_tr_every_10=[round(float(e), 4) for e in err[::10, 0]],
                                       keypoints_mean=float(np.mean(rec["keypoints"][2:])) if frames > 2 else 0.0,
                                       ms_per_frame_mean=float(np.mean(rec["ms"])), ms_per_frame_median=float(np.median(rec["ms"])),
                                       ms_per_frame_mean_after_startup=float(np.mean(rec["ms"][25:])) if frames > 30 else None)
            if "phases" in rec:
                # where the reference's RegisterFrame spends its call, from its own logged_values (steady state: past the 20-frame start-up
                # regime): compute_frame_info + InitializeMotion | InitializeFrame | TryRegister | the undistortion loops | UpdateMap; the
                # rest of `total` is LogSummary and the summary's copies
                lo = 25 if frames > 30 else 0
                table = {k: round(float(np.mean(v[lo:])), 4) for k, v in rec["phases"].items()}
                table["unaccounted"] = round(table["total"] - sum(table[k] for k in table if k != "total"), 4)
                result["runs"][key]["host_time_table_ms"] = table
                if impl.startswith("ref-gpu-armed"):             # the arms' own marks inside those phases (integration/odometry_gpu_arm.h)
                    result["runs"][key]["arm_time_table_ms"] = {k: round(float(np.mean(v[lo:])), 4) for k, v in rec["arm"].items()}
                result["runs"][key]["sampled_frame_points_mean"] = float(np.mean(rec["sampled"][lo:]))
                result["runs"][key]["attempts_max"] = int(max(rec["attempts"]))
            log(f"{key}: {seconds:.1f} s, failures {len(fails)} (first {fails[0] if fails else None}), max |dt| {err[:, 0].max():.3f} m, "
                f"final {err[-1, 0]:.3f} m, beyond 2 m at frame {lost}; after rigid alignment max {aligned.max():.3f} m")
    keys = list(poses)
    result["between_runs"] = {}
    for i in range(len(keys)):
        for k in range(i + 1, len(keys)):
            d = np.abs(poses[keys[i]] - poses[keys[k]]).max(axis=1)
            first = next((j for j in range(frames) if d[j] > 1e-6), None)
            result["between_runs"][f"{keys[i]} vs {keys[k]}"] = dict(max_abs_pose_difference=float(d.max()), first_frame_above_1e_6=first,
                                                                      every_10=[float(f"{v:.2e}") for v in d[::10]])
    line = json.dumps(result)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")
    print(line)


if __name__ == "__main__":
    main()
