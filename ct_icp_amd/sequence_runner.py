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


def _se3_advance(a7: np.ndarray, b7: np.ndarray) -> np.ndarray:
    """The SE(3) product a * b^-1 * a of two poses (quat xyzw | translation): `a` advanced once more by the motion b -> a."""
    rel_q = se3.quat_mul(a7[0:4], se3.quat_conj(b7[0:4]))                       # R_a R_b^T
    q = se3.quat_normalize(se3.quat_mul(rel_q, a7[0:4]))
    t = a7[4:7] + se3.quat_rotate(se3.quat_normalize(rel_q), a7[4:7] - b7[4:7])    # R_a R_b^T (t_a - t_b) + t_a
    return np.concatenate([q, t])


def constant_velocity_guess(prev_pose14: np.ndarray, prev_prev_pose14: np.ndarray | None = None, third_frame: bool = False) -> np.ndarray:
    """Odometry::InitializeMotion with INIT_CONSTANT_VELOCITY and CONTINUOUS motion compensation (odometry.cpp:293-325):
    end = T_end(k-1) T_end(k-2)^-1 T_end(k-1) (a full SE(3) product: the previous translation increment is ROTATED by the relative
    rotation); begin = T_end(k-1) for the second registered frame (:294-300), T_begin(k-1) T_begin(k-2)^-1 T_begin(k-1) afterwards
    (:311-316); the third frame (index 2, `third_frame`) starts at T_end(1) and ends at T_end(1) T_end(0)^-1 T_end(1) (:296-300).
    Without the frame before the previous one, the previous frame's own begin pose stands in for T_end(k-2)."""
    pb, pe = prev_pose14[0:7], prev_pose14[7:14]
    if prev_prev_pose14 is None:
        return np.concatenate([pe, _se3_advance(pe, pb)])
    ppb, ppe = prev_prev_pose14[0:7], prev_prev_pose14[7:14]
    if third_frame:                       # odometry.cpp:296-300: begin = T_end(1), end = T_end(1) T_end(0)^-1 T_end(1)
        return np.concatenate([pe, _se3_advance(pe, ppe)])
    return np.concatenate([_se3_advance(pb, ppb), _se3_advance(pe, ppe)])


def run_sequence(scans, device: int = 0, solver=GN, voxel_size: float = 0.5, sample_voxel_size: float = 1.5,
                 max_distance: float = 100.0, init_poses=None, init_frames: int = 1, options: CTICPOptions = None,
                 use_motion_model: bool = None, resolutions=((0.8, 0.1, 30),), default_radius: float = 0.75, frame_period: float = 0.1,
                 orders=None, init_num_frames: int = 20, init_voxel_size: float = 0.2, init_sample_voxel_size: float = 1.0, init_num_iters: int = 15,
                 shuffle_seed: int = 5489):
    """scans: iterable of (raw (N, 3), t (N,), (t_begin, t_end)). The first `init_frames` frames enter the map with `init_poses[j]`
    (ground truth / identity) and no registration; every later frame is registered from the constant-velocity guess and inserted if
    the registration succeeded. Without `init_poses` the sequence starts from the identity like the reference's Odometry, and the first
    `init_num_frames` frames run its start-up regime (odometry.cpp:340-342, 533-534, 552-556: a 0.2 m frame grid and a 1.0 m keypoint grid
    instead of 0.5 / 1.5 m, at least 15 ICP iterations) — one frame sampled at 0.5 m leaves 2-4 points per 0.8 m map voxel, too few for any
    20-point neighbourhood. The reference shuffles every scan before it samples it (odometry.cpp:349: WHICH point of a voxel survives is a
    random choice, not the first in firing order); here the shuffle is made on the device (ctgn_frame_options::shuffle_seed, a keyed
    permutation per frame: `shuffle_seed` + the frame number; 0 = scan order, as rounds 2-5 ran) unless the caller passes `orders`.
    Returns dict(poses (F, 14), success (F,), seconds, frames, keypoints, sampled, map_points)."""
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
    prev, prev2 = None, None
    t_start = time.perf_counter()
    import copy
    startup_options = copy.copy(options)
    startup_options.num_iters_icp = max(options.num_iters_icp, init_num_iters)
    for j, (raw, t, tbe) in enumerate(scans):
        order = None if orders is None else orders[j]
        seed = 0 if (order is not None or not shuffle_seed) else (int(shuffle_seed) * 0x9E3779B97F4A7C15 + j + 1) & 0xFFFFFFFFFFFFFFFF
        startup = init_poses is None and j < init_num_frames
        fp.frame_voxel_size = init_voxel_size if startup else voxel_size
        fp.sample_voxel_size = init_sample_voxel_size if startup else sample_voxel_size
        # the first two registered frames carry the frame's end timestamp on every point: "no elastic ICP for first frame because no
        # initialization of ego-motion" (odometry.cpp:354-359) — unless the caller supplies their begin / end poses (a bootstrap from
        # ground truth, which the reference does not have): those frames are then inserted undistorted with the poses given
        override = float(tbe[1]) if (j <= 1 and init_poses is None) else None
        if j < init_frames:
            pose0 = np.asarray(init_poses[j], dtype=np.float64) if init_poses is not None else se3.identity_pose14()
            r = fp.frame(raw, t, pose0, tbe, no_registration, max_distance, order=order, override_timestamp=override, want_all=False, shuffle_seed=seed)
        else:
            # odometry.cpp:293-300: the frame right after the bootstrap starts at the previous end pose; later ones extrapolate both ends
            # (frame index 2 uses T_end(0), :296-300; a frame right after a ground-truth bootstrap keeps starting at the previous end pose)
            use_prev2 = j >= 2 and (init_poses is None or j >= init_frames + 1)
            guess = constant_velocity_guess(prev, prev2 if use_prev2 else None, third_frame=(j == 2))
            mm.previous_frame = TrajectoryFrame.from_pose14(prev, tbe[0] - frame_period, tbe[0])
            r = fp.frame(raw, t, guess, tbe, startup_options if startup else options, max_distance, motion_model=mm if use_motion_model else None, order=order,
                         override_timestamp=override, want_all=False, shuffle_seed=seed)
        poses.append(r["pose"])
        success.append(bool(r["summary"].success))
        n_kp.append(len(r["keypoint_indices"]))
        n_sampled.append(len(r["sampled_indices"]))
        new = r["pose"] if r["summary"].success or prev is None else constant_velocity_guess(prev, prev2)
        prev2, prev = prev, new
    seconds = time.perf_counter() - t_start
    return dict(poses=np.array(poses), success=np.array(success), seconds=seconds, frames=len(poses), keypoints=np.array(n_kp),
                sampled=np.array(n_sampled), map_points=int(gm.NumPoints()))


def run_batch(sequences: dict, lengths, rank: int = 0, world_size: int = 1, group=None, **kw):
    """Config E: this rank runs its share of `sequences` ({id: scans}) on its own GPU, nothing is exchanged while it runs; the
    per-sequence results are gathered on every rank afterwards (torch.distributed, any backend) and the aggregate frames/s is
    frames / the slowest rank's wall time. `sequences` needs only this rank's ids."""
    order, shares = deal_sequences(lengths, world_size)
    if "device" not in kw:                  # one GPU per rank (config E); ranks beyond the visible devices share them
        try:
            import torch
            kw["device"] = rank % max(1, torch.cuda.device_count())
        except ImportError:
            kw["device"] = 0
    results = []
    t_job = time.perf_counter()
    for sid in shares[rank]:
        r = run_sequence(sequences[sid], **kw)
        r["device"] = kw["device"]
        r["sequence"] = sid
        results.append(r)
    job_seconds = time.perf_counter() - t_job              # this rank's wall time over its whole share (maps built, frames run)
    for r in results:
        r["rank_job_seconds"] = job_seconds
    if world_size > 1:
        import torch.distributed as dist
        gathered = [None] * world_size
        dist.all_gather_object(gathered, results, group=group)
    else:
        gathered = [results]
    per_rank_seconds = [max(r["rank_job_seconds"] for r in g) if g else 0.0 for g in gathered]
    frames = sum(r["frames"] for g in gathered for r in g)
    wall = max(per_rank_seconds) if per_rank_seconds else 0.0
    return dict(frames=frames, wall_seconds=wall, frames_per_sec=frames / wall if wall > 0 else 0.0, shares=shares,
                per_rank_seconds=per_rank_seconds, results=[r for g in gathered for r in g])
