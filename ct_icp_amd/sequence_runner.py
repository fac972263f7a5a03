"""Sequence runner over the frame pipeline — the data-parallel side of the reference's `run_odometry` / odometry_runner loop
(reference command/odometry_runner.cpp, src/ct_icp/odometry.cpp:386-501): one sequence = one GpuVoxelMap + one FramePipeline,
every frame is ONE ctgn_frame call (sampling -> keypoints -> registration -> undistortion -> evict + insert, scan resident on the
device). SURVEY.md section 8d config E ("KITTI full 11-sequence batch, one sequence per GPU, zero communication"): `deal_sequences`
hands every rank the sequences rank::world, longest first; ranks never talk during the run, rank 0 gathers the per-sequence results
at the end (any backend; the tests use gloo).

Host logic kept from the reference: the constant-velocity initial guess (odometry.cpp:276-330) and the first-frames regime (frame 0
is inserted as it is; `init_frames` frames use the initial voxel size / iteration count, odometry.cpp:340-342, 552-556).
"""
from __future__ import annotations

import time

import numpy as np

from . import se3
from .map import GpuVoxelMap, GpuVoxelMapOptions, ResolutionParam
from .registration import FramePipeline
from .types import CERES, GN, CTICPOptions, PreviousFrameMotionModel, TrajectoryFrame

# relative lengths of the 11 KITTI odometry sequences (reference src/ct_icp/dataset.cpp:49-50)
KITTI_LENGTHS = (4540, 1100, 4660, 800, 270, 2760, 1100, 1100, 4070, 1590, 1200)


def deal_sequences(lengths, world_size: int):
    """Longest first, round-robin: rank r gets order[r::world_size]. Returns (order, [list of sequence ids per rank])."""
    order = [int(i) for i in np.argsort(-np.asarray(lengths, dtype=np.int64), kind="stable")]
    return order, [order[r::world_size] for r in range(world_size)]


def constant_velocity_guess(prev_pose14: np.ndarray) -> np.ndarray:
    """begin = previous end, end = previous end advanced by the previous frame's motion (odometry.cpp:276-330)."""
    pb, pe = prev_pose14[0:7], prev_pose14[7:14]
    rel_q = se3.quat_mul(pe[0:4], se3.quat_conj(pb[0:4]))
    return np.concatenate([pe, se3.quat_normalize(se3.quat_mul(rel_q, pe[0:4])), pe[4:7] + (pe[4:7] - pb[4:7])])


def run_sequence(scans, device: int = 0, solver=GN, voxel_size: float = 0.5, sample_voxel_size: float = 1.5,
                 max_distance: float = 100.0, init_poses=None, init_frames: int = 1, options: CTICPOptions = None,
                 use_motion_model: bool = None, resolutions=((0.8, 0.1, 30),), default_radius: float = 0.75, frame_period: float = 0.1,
                 orders=None):
    """scans: iterable of (raw (N, 3), t (N,), (t_begin, t_end)). The first `init_frames` frames enter the map with `init_poses[j]`
    (ground truth / identity) and no registration; every later frame is registered from the constant-velocity guess and inserted if
    the registration succeeded. Returns dict(poses (F, 14), success (F,), seconds, frames, keypoints, sampled, map_points)."""
    gm = GpuVoxelMap(GpuVoxelMapOptions(resolutions=[ResolutionParam(*r) for r in resolutions], default_radius=default_radius,
                                        device=device, device_updates=True))
    fp = FramePipeline(gm, frame_voxel_size=voxel_size, sample_voxel_size=sample_voxel_size)
    if options is None:
        if solver == GN:                  # driving profile with the solver forced to GN
            options = CTICPOptions(solver=GN, num_iters_icp=5, threshold_orientation_norm=1e-4, debug_print=False)
        else:                             # config/odometry/driving_config.yaml:52-89
            options = CTICPOptions(solver=CERES, num_iters_icp=5, ls_max_num_iters=5, max_num_residuals=900, loss_function="CAUCHY",
                                   ls_sigma=0.1, debug_print=False)
    if use_motion_model is None:
        use_motion_model = options.solver == CERES
    no_registration = CTICPOptions(solver=GN, num_iters_icp=0, debug_print=False)
    mm = PreviousFrameMotionModel()
    poses, success, n_kp, n_sampled = [], [], [], []
    prev = None
    t_start = time.perf_counter()
    for j, (raw, t, tbe) in enumerate(scans):
        order = None if orders is None else orders[j]
        if j < init_frames:
            pose0 = np.asarray(init_poses[j], dtype=np.float64) if init_poses is not None else se3.identity_pose14()
            r = fp.frame(raw, t, pose0, tbe, no_registration, max_distance, order=order, want_all=False)
        else:
            guess = constant_velocity_guess(prev)
            mm.previous_frame = TrajectoryFrame.from_pose14(prev, tbe[0] - frame_period, tbe[0])
            r = fp.frame(raw, t, guess, tbe, options, max_distance, motion_model=mm if use_motion_model else None, order=order,
                         want_all=False)
        poses.append(r["pose"])
        success.append(bool(r["summary"].success))
        n_kp.append(len(r["keypoint_indices"]))
        n_sampled.append(len(r["sampled_indices"]))
        prev = r["pose"] if r["summary"].success or prev is None else constant_velocity_guess(prev)
    seconds = time.perf_counter() - t_start
    return dict(poses=np.array(poses), success=np.array(success), seconds=seconds, frames=len(poses), keypoints=np.array(n_kp),
                sampled=np.array(n_sampled), map_points=int(gm.NumPoints()))


def run_batch(sequences: dict, lengths, rank: int = 0, world_size: int = 1, group=None, **kw):
    """Config E: this rank runs its share of `sequences` ({id: scans}) on its own GPU, nothing is exchanged while it runs; the
    per-sequence results are gathered on every rank afterwards (torch.distributed, any backend) and the aggregate frames/s is
    frames / the slowest rank's wall time. `sequences` needs only this rank's ids."""
    order, shares = deal_sequences(lengths, world_size)
    results = []
    for sid in shares[rank]:
        r = run_sequence(sequences[sid], **kw)
        r["sequence"] = sid
        results.append(r)
    if world_size > 1:
        import torch.distributed as dist
        gathered = [None] * world_size
        dist.all_gather_object(gathered, results, group=group)
    else:
        gathered = [results]
    per_rank_seconds = [sum(r["seconds"] for r in g) for g in gathered]
    frames = sum(r["frames"] for g in gathered for r in g)
    wall = max(per_rank_seconds) if per_rank_seconds else 0.0
    return dict(frames=frames, wall_seconds=wall, frames_per_sec=frames / wall if wall > 0 else 0.0, shares=shares,
                per_rank_seconds=per_rank_seconds, results=[r for g in gathered for r in g])
