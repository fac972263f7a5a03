"""Host-side mirror of the reference's API types on the registration path (same names, same meaning):

  slam::Pose / TPose<double>   include/SlamCore/types.h:161-274
  ct_icp::TrajectoryFrame      include/ct_icp/types.h:31-61
  ct_icp::CTICPOptions         include/ct_icp/ct_icp.h:56-153   (every field kept; GN reads six of them, CERES sixteen)
  ct_icp::ICPSummary           include/ct_icp/ct_icp.h:155-169
  slam::WPoint3D               include/SlamCore/types.h:35-60    (64-byte record, numpy structured dtype)
  PreviousFrameMotionModel     include/ct_icp/motion_model.h:35-80 (the part GN reads)
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import se3

# WPoint3D: raw xyz f64 @0, t f64 @24, world xyz f64 @32, index_frame @56
WPOINT3D_DTYPE = np.dtype({"names": ["raw_point", "t", "world_point", "index_frame"],
                           "formats": [("<f8", 3), "<f8", ("<f8", 3), "<u4"],
                           "offsets": [0, 24, 32, 56], "itemsize": 64})

GN, CERES, ROBUST = 0, 1, 2      # ct_icp::CT_ICP_SOLVER (ct_icp.h:35-39)


@dataclass
class Pose:
    """slam::Pose: quat (x, y, z, w), tr, dest_timestamp, frame ids."""
    quat: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, 0.0, 1.0]))
    tr: np.ndarray = field(default_factory=lambda: np.zeros(3))
    dest_timestamp: float = -1.0
    ref_timestamp: float = 0.0
    dest_frame_id: int = -1
    ref_frame_id: int = 0

    def params(self) -> np.ndarray:
        return np.concatenate([np.asarray(self.quat, float), np.asarray(self.tr, float)])

    def GetAlphaTimestamp(self, t, other: "Pose"):
        return se3.alpha_timestamp(t, self.dest_timestamp, other.dest_timestamp)

    def InterpolatePose(self, other: "Pose", t: float) -> "Pose":
        if not (self.dest_timestamp <= t <= other.dest_timestamp):
            raise ValueError("The timestamp cannot be interpolated between the two poses")     # types.h:456
        a = float(self.GetAlphaTimestamp(t, other))
        q = se3.quat_slerp(self.quat, other.quat, np.array(a))
        return Pose(q, (1 - a) * np.asarray(self.tr) + a * np.asarray(other.tr), t, self.ref_timestamp,
                    self.dest_frame_id, self.ref_frame_id)

    def __mul__(self, point):
        return se3.quat_rotate(se3.quat_normalize(self.quat), np.asarray(point, float)) + self.tr


@dataclass
class TrajectoryFrame:
    begin_pose: Pose = field(default_factory=Pose)
    end_pose: Pose = field(default_factory=Pose)

    def BeginTr(self): return self.begin_pose.tr
    def EndTr(self): return self.end_pose.tr
    def BeginQuat(self): return self.begin_pose.quat
    def EndQuat(self): return self.end_pose.quat

    def pose14(self) -> np.ndarray:
        return np.concatenate([self.begin_pose.params(), self.end_pose.params()])

    def set_pose14(self, p):
        p = np.asarray(p, float)
        self.begin_pose.quat, self.begin_pose.tr = p[0:4].copy(), p[4:7].copy()
        self.end_pose.quat, self.end_pose.tr = p[7:11].copy(), p[11:14].copy()

    @staticmethod
    def from_pose14(p, t_begin, t_end) -> "TrajectoryFrame":
        f = TrajectoryFrame(Pose(dest_timestamp=t_begin), Pose(dest_timestamp=t_end))
        f.set_pose14(p)
        return f


@dataclass
class CTICPOptions:
    """ct_icp::CTICPOptions with the reference's defaults (ct_icp.h:56-153)."""
    num_iters_icp: int = 5
    parametrization: str = "CONTINUOUS_TIME"
    distance: str = "POINT_TO_PLANE"
    solver: int = CERES                      # the reference's default; GN and CERES are served by this package
    max_num_residuals: int = -1
    min_num_residuals: int = 100
    weighting_scheme: str = "ALL"
    weight_alpha: float = 0.9
    weight_neighborhood: float = 0.1
    power_planarity: float = 2.0
    max_number_neighbors: int = 20
    min_number_neighbors: int = 20
    threshold_voxel_occupancy: int = 1
    estimate_normal_from_neighborhood: bool = True
    num_closest_neighbors: int = 1
    threshold_orientation_norm: float = 0.0001
    threshold_translation_norm: float = 0.001
    point_to_plane_with_distortion: bool = True
    loss_function: str = "CAUCHY"
    ls_max_num_iters: int = 1
    ls_num_threads: int = 16
    ls_sigma: float = 0.1
    ls_tolerant_min_threshold: float = 0.05
    max_dist_to_plane_ct_icp: float = 0.3
    debug_print: bool = True


@dataclass
class ICPSummary:
    success: bool = False
    num_residuals_used: int = 0
    num_iters: int = 0
    error_log: str = ""
    duration_total: float = 0.0
    duration_init: float = 0.0
    avg_duration_iter: float = 0.0
    avg_duration_neighborhood: float = 0.0
    avg_duration_solve: float = 0.0
    last_step_norm: float = 0.0


@dataclass
class PreviousFrameMotionModel:
    """The part of ct_icp::PreviousFrameMotionModel the solvers read: GN the first two betas and the previous
    translations (ct_icp.cpp:888-908), CERES all four and the previous end orientation (motion_model.cpp:12-61)."""
    beta_location_consistency: float = 0.001
    beta_constant_velocity: float = 0.001
    beta_small_velocity: float = 0.0
    beta_orientation_consistency: float = 0.0
    previous_frame: TrajectoryFrame = field(default_factory=TrajectoryFrame)

    def UpdateState(self, optimized_frame: TrajectoryFrame, frame_index: int = 0):
        self.previous_frame = TrajectoryFrame.from_pose14(optimized_frame.pose14(),
                                                          optimized_frame.begin_pose.dest_timestamp,
                                                          optimized_frame.end_pose.dest_timestamp)

    def PreviousFrame(self) -> TrajectoryFrame:
        return self.previous_frame


@dataclass
class AdaptiveGridSamplingOptions:
    """ct_icp::AdaptiveGridSamplingOptions (reference include/ct_icp/algorithm/sampling.h:13-26), same names and defaults."""
    num_points_per_voxel: int = 1
    max_num_points: int = -1
    distance_voxel_size: list = field(default_factory=lambda: [(0.5, 0.1), (2.0, 0.2), (4.0, 0.4), (8.0, 0.8), (16.0, 1.6),
                                                               (200.0, -1.0)])
