"""CT_ICP_Registration — host-side mirror of ct_icp::CT_ICP_Registration for `solver: GN` and `solver: CERES`
(reference include/ct_icp/ct_icp.h:171-215, src/ct_icp/ct_icp.cpp:998-1053). `Register` keeps the reference's
argument order and in-place semantics: the TrajectoryFrame poses and the keypoints' world points are updated,
an ICPSummary is returned. The work is done by libctgn (HIP, gfx950); nothing here computes on the CPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .map import GpuVoxelMap
from .types import CERES, GN, AdaptiveGridSamplingOptions, CTICPOptions, ICPSummary, PreviousFrameMotionModel, TrajectoryFrame, WPOINT3D_DTYPE


def _view(arr: np.ndarray, offset: int = 0) -> L.View:
    dt = L.CTGN_F64 if arr.dtype == np.float64 else L.CTGN_F32
    return L.View(arr.ctypes.data + offset, arr.strides[0], dt, 0)


def _c_options(o: CTICPOptions) -> L.Options:
    return L.Options(o.num_iters_icp, o.min_number_neighbors, o.max_number_neighbors, int(o.debug_print),
                     o.max_dist_to_plane_ct_icp, o.threshold_orientation_norm)


def _c_prior(motion_model) -> L.MotionPrior | None:
    # ct_icp.cpp:885-889: only a PreviousFrameMotionModel contributes
    if motion_model is None or not isinstance(motion_model, PreviousFrameMotionModel):
        return None
    p = L.MotionPrior()
    p.beta_location_consistency = motion_model.beta_location_consistency
    p.beta_constant_velocity = motion_model.beta_constant_velocity
    pf = motion_model.PreviousFrame()
    for i in range(3):
        p.previous_begin_tr[i] = float(pf.BeginTr()[i])
        p.previous_end_tr[i] = float(pf.EndTr()[i])
    return p


def _c_robust_options(o: CTICPOptions) -> L.RobustOptions:
    if o.parametrization != "CONTINUOUS_TIME" or o.distance != "POINT_TO_PLANE":
        raise RuntimeError("the CERES route serves parametrization CONTINUOUS_TIME with distance POINT_TO_PLANE only")
    return L.RobustOptions(o.num_iters_icp, o.min_number_neighbors, o.max_number_neighbors, int(o.debug_print),
                           o.max_num_residuals, L.LOSS[o.loss_function], o.ls_max_num_iters, o.num_closest_neighbors,
                           o.weight_alpha, o.weight_neighborhood, o.power_planarity, o.max_dist_to_plane_ct_icp, o.ls_sigma,
                           o.ls_tolerant_min_threshold, o.threshold_orientation_norm, o.threshold_translation_norm)


def _c_robust_prior(motion_model, continuous_time=True) -> L.RobustPrior | None:
    # ct_icp.cpp:608-610: `_previous_frame && parametrization == CONTINUOUS_TIME`
    if motion_model is None or not continuous_time or not isinstance(motion_model, PreviousFrameMotionModel):
        return None
    p = L.RobustPrior()
    p.beta_location_consistency = motion_model.beta_location_consistency
    p.beta_constant_velocity = motion_model.beta_constant_velocity
    p.beta_small_velocity = motion_model.beta_small_velocity
    p.beta_orientation_consistency = motion_model.beta_orientation_consistency
    pf = motion_model.PreviousFrame()
    for i in range(3):
        p.previous_begin_tr[i] = float(pf.BeginTr()[i])
        p.previous_end_tr[i] = float(pf.EndTr()[i])
    for i in range(4):
        p.previous_end_quat[i] = float(pf.EndQuat()[i])
    return p


def _summary(s: L.Summary) -> ICPSummary:
    return ICPSummary(success=bool(s.success), num_residuals_used=s.num_residuals_used, num_iters=s.num_iters,
                      error_log=s.error_log.decode(), duration_total=s.duration_total_ms, duration_init=s.duration_init_ms,
                      avg_duration_iter=s.avg_duration_iter_ms, avg_duration_neighborhood=s.avg_duration_neighborhood_ms,
                      avg_duration_solve=s.avg_duration_solve_ms, last_step_norm=s.last_step_norm)          # milliseconds, as ICPSummary


class CT_ICP_Registration:
    def __init__(self, options: CTICPOptions | None = None):
        self._options = options or CTICPOptions(solver=GN)
        self._buf = np.zeros(16)                                   # pose (14) | t_begin, t_end
        dp = C.POINTER(C.c_double)
        self._pose_ptr = self._buf.ctypes.data_as(dp)
        self._tbe_ptr = C.cast(self._buf.ctypes.data + 14 * 8, dp)

    def Options(self) -> CTICPOptions:
        return self._options

    def Register(self, voxel_map: GpuVoxelMap, keypoints: np.ndarray, trajectory_frame: TrajectoryFrame,
                 motion_model=None, strategy=None) -> ICPSummary:
        """keypoints: structured array of WPOINT3D_DTYPE (the vector<slam::WPoint3D> overload, ct_icp.cpp:1026-1037).
        `strategy` is accepted and ignored, as DoRegisterGaussNewton ignores it."""
        if self._options.solver not in (GN, CERES):
            raise RuntimeError("Unsupported Solver Type")        # ct_icp.cpp:1022 — the ROBUST arm does not live here
        if not isinstance(voxel_map, GpuVoxelMap):
            raise TypeError("Register needs a GpuVoxelMap (the GPU path has no other map backend)")
        if keypoints.dtype != WPOINT3D_DTYPE:
            raise TypeError("keypoints must be a WPOINT3D_DTYPE structured array")
        h = voxel_map.handle
        n = len(keypoints)
        # (the wrapper itself is on the latency path of a 0.16 ms call: one .ctypes lookup, a reused 16-double buffer for
        # pose | timestamps)
        base = keypoints.ctypes.data
        stride = keypoints.strides[0] if n else 64
        raw = L.View(base, stride, L.CTGN_F64, 0)
        ts = L.View(base + 24, stride, L.CTGN_F64, 0)
        b, e = trajectory_frame.begin_pose, trajectory_frame.end_pose
        buf = self._buf
        buf[0:4], buf[4:7], buf[7:11], buf[11:14] = b.quat, b.tr, e.quat, e.tr
        buf[14], buf[15] = b.dest_timestamp, e.dest_timestamp
        pose, tbe = self._pose_ptr, self._tbe_ptr
        s = L.Summary()
        if self._options.solver == GN:
            opts, prior, fn = _c_options(self._options), _c_prior(motion_model), L.lib().ctgn_register
        else:
            opts, prior, fn = _c_robust_options(self._options), _c_robust_prior(motion_model), L.lib().ctgn_register_robust
        st = fn(h, raw, base + 32, stride, L.CTGN_F64, ts, n, pose, tbe, C.byref(opts),
                C.byref(prior) if prior is not None else None, C.byref(s))
        if st != L.OK:
            L.check(h, st)
        b.quat, b.tr, e.quat, e.tr = buf[0:4].copy(), buf[4:7].copy(), buf[7:11].copy(), buf[11:14].copy()
        return _summary(s)


def pinned_array(shape, dtype=np.float64) -> np.ndarray:
    """A page-locked host array (the runtime's pinned allocator behind a NumPy view). FramePipeline calls whose scan (N x 3 float64
    rows), timestamps (float64) and `all_world_out` live in such arrays are not staged: the DMA engine reads and writes them where
    they lie (ctgn_frame_register). Needs a GPU runtime."""
    import torch
    buf = torch.empty(tuple(np.atleast_1d(shape)), dtype=getattr(torch, np.dtype(dtype).name), pin_memory=True)
    return buf.numpy()          # the view keeps the tensor (and its allocation) alive


def transform_points(voxel_map: GpuVoxelMap, raw, t, pose14, t_begin_end, out=None):
    """Full-scan continuous-time undistortion on the GPU (reference src/ct_icp/odometry.cpp:461-486): world[i] =
    begin.InterpolatePose(end, t[i]) * raw[i]. numpy in -> numpy out; torch CUDA tensors in -> torch CUDA tensor out (no host hop).
    `out` (numpy, N x 3 float64, C-contiguous): write there instead of into a fresh array — the reference undistorts in place, and a
    fresh 3 MB array costs its page faults on every call."""
    if L.is_device_tensor(raw):
        import torch
        if out is None:
            out = torch.empty((len(raw), 3), dtype=torch.float64, device=raw.device)
        elif not (L.is_device_tensor(out) and out.dtype == torch.float64 and tuple(out.shape) == (len(raw), 3) and out.is_contiguous()
                  and out.device == raw.device):
            raise ValueError("transform_points: `out` for device inputs must be a contiguous N x 3 float64 tensor on the same device")
        pose = np.ascontiguousarray(pose14, dtype=np.float64)
        tbe = np.ascontiguousarray(t_begin_end, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        h = voxel_map.handle
        L.check(h, L.lib().ctgn_transform_points(h, L.tensor_view(raw), L.tensor_view(t, 1), len(raw), pose.ctypes.data_as(dp),
                                                tbe.ctypes.data_as(dp), out.data_ptr(), 24, L.CTGN_F64))
        return out
    raw = np.ascontiguousarray(raw, dtype=np.float64).reshape(-1, 3)
    t = np.ascontiguousarray(t, dtype=np.float64).ravel()
    pose = np.ascontiguousarray(pose14, dtype=np.float64)
    tbe = np.ascontiguousarray(t_begin_end, dtype=np.float64)
    if out is None:
        out = np.empty_like(raw)
    elif not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.shape == raw.shape and out.flags.c_contiguous):
        raise ValueError("out must be a C-contiguous float64 array of the shape of the points")
    dp = C.POINTER(C.c_double)
    h = voxel_map.handle
    L.check(h, L.lib().ctgn_transform_points(h, L.View(raw.ctypes.data, 24, L.CTGN_F64, 0), L.View(t.ctypes.data, 8, L.CTGN_F64, 0),
                                            len(t), pose.ctypes.data_as(dp), tbe.ctypes.data_as(dp), out.ctypes.data, 24,
                                            L.CTGN_F64))
    return out


def grid_sampling(voxel_map: GpuVoxelMap, points, voxel_size: float) -> np.ndarray:
    """ct_icp::grid_sampling / sub_sample_frame on the GPU (reference src/ct_icp/ct_icp.cpp:65-101): indices of the first
    point of every voxel, ascending (= the order in which the reference's loop first meets the voxels)."""
    h = voxel_map.handle
    if L.is_device_tensor(points):                       # device in, device out
        import torch
        out_t = torch.zeros(len(points), dtype=torch.int32, device=points.device)
        cnt = C.c_size_t()
        L.check(h, L.lib().ctgn_grid_sampling(h, L.tensor_view(points), len(points), float(voxel_size),
                                             C.cast(out_t.data_ptr(), C.POINTER(C.c_uint32)), C.byref(cnt)))
        return out_t[:cnt.value].clone()
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    out = np.zeros(len(pts), dtype=np.uint32)
    cnt = C.c_size_t()
    L.check(h, L.lib().ctgn_grid_sampling(h, L.View(pts.ctypes.data, 24, L.CTGN_F64, 0), len(pts), float(voxel_size),
                                         out.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(cnt)))
    return out[:cnt.value].copy()


def _c_adaptive_options(options: AdaptiveGridSamplingOptions) -> "L.AdaptiveSamplingOptions":
    pairs = list(options.distance_voxel_size)
    if len(pairs) > 16:
        raise ValueError("at most 16 distance / voxel-size pairs")
    c = L.AdaptiveSamplingOptions()
    c.num_points_per_voxel, c.max_num_points, c.num_bands = int(options.num_points_per_voxel), int(options.max_num_points), len(pairs)
    for j, (d, v) in enumerate(pairs):
        c.distance[j], c.voxel_size[j] = float(d), float(v)
    return c


def AdaptiveSamplePointsInGrid(voxel_map: GpuVoxelMap, points, options: AdaptiveGridSamplingOptions = None) -> np.ndarray:
    """ct_icp::AdaptiveSamplePointsInGrid on the GPU (reference include/ct_icp/algorithm/sampling.h:55-110; `sampling: ADAPTIVE`
    of Odometry::TryRegister, src/ct_icp/odometry.cpp:539-545): indices kept by the range-banded grid, ordered by band, voxel,
    index. `points` is an (n, 3) array of SENSOR-frame points, or a torch CUDA tensor (device in, device out)."""
    h = voxel_map.handle
    c = _c_adaptive_options(options or AdaptiveGridSamplingOptions())
    cnt = C.c_size_t()
    if L.is_device_tensor(points):
        import torch
        out_t = torch.zeros(max(len(points), 1), dtype=torch.int32, device=points.device)
        L.check(h, L.lib().ctgn_adaptive_sampling(h, L.tensor_view(points), len(points), C.byref(c),
                                                 C.cast(out_t.data_ptr(), C.POINTER(C.c_uint32)), C.byref(cnt)))
        return out_t[:cnt.value].clone()
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    out = np.zeros(max(len(pts), 1), dtype=np.uint32)
    L.check(h, L.lib().ctgn_adaptive_sampling(h, L.View(pts.ctypes.data, 24, L.CTGN_F64, 0), len(pts), C.byref(c),
                                             out.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(cnt)))
    return out[:cnt.value].copy()


class FramePipeline:
    """One frame of the odometry loop with the scan resident on the device (ctgn_frame_register / ctgn_frame_update_map): what
    Odometry::DoRegister does around the registration (reference src/ct_icp/odometry.cpp:333-382, 455-501, 526-590, 936-952) —
    sub_sample_frame, grid_sampling, Register, both undistortions, RemoveElementsFarFromLocation + InsertPointCloud — with one
    upload, one small read-back in the middle and one copy back. The map must be device-maintained (update_mode 1)."""

    def __init__(self, voxel_map: GpuVoxelMap, frame_voxel_size=0.5, sample_voxel_size=1.5, max_num_keypoints=-1):
        self.map = voxel_map
        self._h = voxel_map.handle
        self.frame_voxel_size, self.sample_voxel_size, self.max_num_keypoints = frame_voxel_size, sample_voxel_size, max_num_keypoints

    def _call(self, fn, raw, t, pose14, t_begin_end, options, motion_model, order, override_timestamp, want_all, want_sampled, extra,
              all_world_out=None, shuffle_seed=0):
        raw = np.ascontiguousarray(raw, dtype=np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(t, dtype=np.float64).ravel()
        n = len(raw)
        assert len(t) == n
        pose = np.ascontiguousarray(pose14, dtype=np.float64).copy()
        tbe = np.ascontiguousarray(t_begin_end, dtype=np.float64)
        fo = L.FrameOptions(float(self.frame_voxel_size), float(self.sample_voxel_size), int(self.max_num_keypoints),
                            0 if override_timestamp is None else 1, 0.0 if override_timestamp is None else float(override_timestamp),
                            int(shuffle_seed))
        out = L.FrameOutputs()
        res = {}
        if want_all:
            # `all_world_out` (N x 3 float64, C-contiguous): write the undistorted scan there — the reference undistorts in place
            # (odometry.cpp:461-486), and a fresh 3 MB array costs its page faults on every frame
            if all_world_out is not None:
                if not (isinstance(all_world_out, np.ndarray) and all_world_out.dtype == np.float64 and all_world_out.shape == (n, 3)
                        and all_world_out.flags.c_contiguous):
                    raise ValueError("all_world_out must be a C-contiguous N x 3 float64 array")
                res["all_world"] = all_world_out
            else:
                res["all_world"] = np.zeros((n, 3))
            out.all_world_base, out.all_world_stride_bytes, out.all_world_dtype = res["all_world"].ctypes.data, 24, L.CTGN_F64
        if getattr(self, "_idx_cap", 0) < n:               # index staging, reused across frames (results are copied out below)
            self._idx = np.zeros((2, max(n, 1)), dtype=np.uint32)
            self._idx_cap = n
        sampled_idx, kp_idx = self._idx[0], self._idx[1]
        out.sampled_indices, out.keypoint_indices = sampled_idx.ctypes.data, kp_idx.ctypes.data
        if want_sampled:
            sw = np.zeros((n, 3))
            out.sampled_world_base, out.sampled_world_stride_bytes, out.sampled_world_dtype = sw.ctypes.data, 24, L.CTGN_F64
        robust = options.solver == CERES
        s = L.Summary()
        dp = C.POINTER(C.c_double)
        if order is not None:
            order = np.ascontiguousarray(order, dtype=np.uint32)
            assert len(order) == n
        c_opts = c_prior = c_ropts = c_rprior = None
        if robust:
            c_ropts, c_rprior = _c_robust_options(options), _c_robust_prior(motion_model)
        else:
            c_opts, c_prior = _c_options(options), _c_prior(motion_model)
        st = fn(self._h, L.View(raw.ctypes.data, 24, L.CTGN_F64, 0), L.View(t.ctypes.data, 8, L.CTGN_F64, 0), n,
                order.ctypes.data if order is not None else None, C.byref(fo), pose.ctypes.data_as(dp), tbe.ctypes.data_as(dp),
                C.byref(c_opts) if c_opts is not None else None, C.byref(c_prior) if c_prior is not None else None,
                C.byref(c_ropts) if c_ropts is not None else None, C.byref(c_rprior) if c_rprior is not None else None,
                *extra, C.byref(out), C.byref(s))
        L.check(self._h, st)
        self._last_n1 = int(out.num_sampled)
        res.update(pose=pose, summary=_summary(s), sampled_indices=sampled_idx[:out.num_sampled].copy(),
                   keypoint_indices=kp_idx[:out.num_keypoints].copy())
        if want_sampled:
            res["sampled_world"] = sw[:out.num_sampled].copy()
        return res

    def register(self, raw, t, pose14, t_begin_end, options: CTICPOptions, motion_model=None, order=None, override_timestamp=None,
                 want_all=True, want_sampled=True, all_world_out=None, shuffle_seed=0) -> dict:
        """Sampling -> keypoints -> registration -> undistortion. Returns pose (14), summary, sampled_indices, keypoint_indices and
        (as asked) all_world / sampled_world. shuffle_seed != 0 (and no `order`): the scan is shuffled on the device first
        (ctgn_frame_options::shuffle_seed)."""
        return self._call(L.lib().ctgn_frame_register, raw, t, pose14, t_begin_end, options, motion_model, order, override_timestamp,
                          want_all, want_sampled, (), all_world_out, shuffle_seed)

    # ---- the same stages one by one (ctgn_frame_begin / _try_register / _undistort): the calls integration/odometry_gpu_arm.h makes from
    # the reference's InitializeFrame, TryRegister and undistortion loops
    def stage(self, raw, t, pose14, t_begin_end, override_timestamp=None) -> None:
        """ctgn_frame_stage: upload the scan ahead of begin(None, None, ...) — for a caller that computes its `order` meanwhile."""
        raw = np.ascontiguousarray(raw, dtype=np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(t, dtype=np.float64).ravel()
        pose = np.ascontiguousarray(pose14, dtype=np.float64)
        tbe = np.ascontiguousarray(t_begin_end, dtype=np.float64)
        fo = L.FrameOptions(float(self.frame_voxel_size), float(self.sample_voxel_size), int(self.max_num_keypoints),
                            0 if override_timestamp is None else 1, 0.0 if override_timestamp is None else float(override_timestamp), 0)
        dp = C.POINTER(C.c_double)
        L.check(self._h, L.lib().ctgn_frame_stage(self._h, L.View(raw.ctypes.data, 24, L.CTGN_F64, 0), L.View(t.ctypes.data, 8, L.CTGN_F64, 0), len(raw),
                                                 C.byref(fo), pose.ctypes.data_as(dp), tbe.ctypes.data_as(dp)))
        self._staged_n = len(raw)

    def begin(self, raw, t, pose14, t_begin_end, order=None, override_timestamp=None, want_world=False, shuffle_seed=0) -> dict:
        """InitializeFrame (odometry.cpp:333-382): stage + upload + sub_sample_frame (+ the keypoint sampler at sample_voxel_size).
        Returns sampled_indices, num_keypoints and — want_world — the sampled frame under pose14 (the initial estimate).
        raw = t = None: the scan stage() uploaded."""
        if raw is None:
            n = self._staged_n
            raw = np.zeros((0, 3))
            t = np.zeros(0)
            prestaged = True
        else:
            prestaged = False
            raw = np.ascontiguousarray(raw, dtype=np.float64).reshape(-1, 3)
            t = np.ascontiguousarray(t, dtype=np.float64).ravel()
            n = len(raw)
        pose = np.ascontiguousarray(pose14, dtype=np.float64)
        tbe = np.ascontiguousarray(t_begin_end, dtype=np.float64)
        fo = L.FrameOptions(float(self.frame_voxel_size), float(self.sample_voxel_size), int(self.max_num_keypoints),
                            0 if override_timestamp is None else 1, 0.0 if override_timestamp is None else float(override_timestamp),
                            int(shuffle_seed))
        out = L.FrameOutputs()
        idx = np.zeros(max(n, 1), dtype=np.uint32)
        out.sampled_indices = idx.ctypes.data
        if want_world:
            sw = np.zeros((max(n, 1), 3))
            out.sampled_world_base, out.sampled_world_stride_bytes, out.sampled_world_dtype = sw.ctypes.data, 24, L.CTGN_F64
        if order is not None:
            order = np.ascontiguousarray(order, dtype=np.uint32)
            assert len(order) == n
        dp = C.POINTER(C.c_double)
        L.check(self._h, L.lib().ctgn_frame_begin(self._h, L.View(None if prestaged else raw.ctypes.data, 24, L.CTGN_F64, 0),
                                                 L.View(None if prestaged else t.ctypes.data, 8, L.CTGN_F64, 0), n,
                                                 order.ctypes.data if order is not None else None, C.byref(fo), pose.ctypes.data_as(dp),
                                                 tbe.ctypes.data_as(dp), C.byref(out)))
        self._last_n, self._last_n1 = n, int(out.num_sampled)
        res = dict(sampled_indices=idx[:out.num_sampled].copy(), num_keypoints=int(out.num_keypoints))
        if want_world:
            res["sampled_world"] = sw[:out.num_sampled].copy()
        return res

    def try_register(self, pose14, t_begin_end, options: CTICPOptions, motion_model=None, sample_voxel_size=None, max_num_keypoints=None) -> dict:
        """TryRegister (odometry.cpp:525-601) on the sampled frame begin() left on the device; may be called again with other options /
        another sample_voxel_size (the robust retry loop). Returns pose, summary, keypoint_indices, keypoint_world."""
        pose = np.ascontiguousarray(pose14, dtype=np.float64).copy()
        tbe = np.ascontiguousarray(t_begin_end, dtype=np.float64)
        fo = L.FrameOptions(float(self.frame_voxel_size), float(self.sample_voxel_size if sample_voxel_size is None else sample_voxel_size),
                            int(self.max_num_keypoints if max_num_keypoints is None else max_num_keypoints), 0, 0.0, 0)
        n1 = max(1, getattr(self, "_last_n1", 0))
        out = L.FrameOutputs()
        kp_idx = np.zeros(n1, dtype=np.uint32)
        kp_world = np.zeros((n1, 3))
        out.keypoint_indices = kp_idx.ctypes.data
        out.keypoint_world_base, out.keypoint_world_stride_bytes, out.keypoint_world_dtype = kp_world.ctypes.data, 24, L.CTGN_F64
        s = L.Summary()
        dp = C.POINTER(C.c_double)
        c_opts = c_prior = c_ropts = c_rprior = None
        if options.solver == CERES:
            c_ropts, c_rprior = _c_robust_options(options), _c_robust_prior(motion_model)
        else:
            c_opts, c_prior = _c_options(options), _c_prior(motion_model)
        L.check(self._h, L.lib().ctgn_frame_try_register(
            self._h, C.byref(fo), pose.ctypes.data_as(dp), tbe.ctypes.data_as(dp), C.byref(c_opts) if c_opts is not None else None,
            C.byref(c_prior) if c_prior is not None else None, C.byref(c_ropts) if c_ropts is not None else None,
            C.byref(c_rprior) if c_rprior is not None else None, C.byref(out), C.byref(s)))
        m = int(out.num_keypoints)
        return dict(pose=pose, summary=_summary(s), keypoint_indices=kp_idx[:m].copy(), keypoint_world=kp_world[:m].copy())

    def undistort(self, pose14, t_begin_end, want_all=True) -> dict:
        """The two undistortion loops (odometry.cpp:461-486) with the poses the caller settled on; leaves the undistorted sampled frame
        on the device for update_map(). Returns sampled_world and (want_all) all_world."""
        pose = np.ascontiguousarray(pose14, dtype=np.float64)
        tbe = np.ascontiguousarray(t_begin_end, dtype=np.float64)
        n, n1 = getattr(self, "_last_n", 0), getattr(self, "_last_n1", 0)
        out = L.FrameOutputs()
        res = {}
        if want_all:
            res["all_world"] = np.zeros((n, 3))
            out.all_world_base, out.all_world_stride_bytes, out.all_world_dtype = res["all_world"].ctypes.data, 24, L.CTGN_F64
        sw = np.zeros((max(n1, 1), 3))
        out.sampled_world_base, out.sampled_world_stride_bytes, out.sampled_world_dtype = sw.ctypes.data, 24, L.CTGN_F64
        dp = C.POINTER(C.c_double)
        L.check(self._h, L.lib().ctgn_frame_undistort(self._h, pose.ctypes.data_as(dp), tbe.ctypes.data_as(dp), C.byref(out)))
        res["sampled_world"] = sw[:n1]
        return res

    def update_map(self, location, max_distance: float, add_points: bool = True) -> np.ndarray | None:
        """UpdateMap for the frame register() left on the device; returns the `inserted` mask of the sampled frame."""
        loc = np.ascontiguousarray(location, dtype=np.float64)
        mask = np.zeros(max(1, getattr(self, "_last_n1", 0)), dtype=np.uint8)
        L.check(self._h, L.lib().ctgn_frame_update_map(self._h, loc.ctypes.data_as(C.POINTER(C.c_double)), float(max_distance),
                                                      1 if add_points else 0, mask.ctypes.data if add_points else None))
        return mask[:getattr(self, "_last_n1", 0)] if add_points else None

    def frame(self, raw, t, pose14, t_begin_end, options: CTICPOptions, max_distance: float, motion_model=None, order=None,
              override_timestamp=None, want_all=True, want_sampled=False, all_world_out=None, shuffle_seed=0) -> dict:
        """register() + update_map(end translation, max_distance, success) in one call (always_insert policy). On the GN route the map
        update is enqueued behind the undistortion and runs beside the hand-over of the outputs (ctgn_frame, round 4)."""
        return self._call(L.lib().ctgn_frame, raw, t, pose14, t_begin_end, options, motion_model, order, override_timestamp, want_all,
                          want_sampled, (C.c_double(float(max_distance)),), all_world_out, shuffle_seed)


class GnSolver:
    """Array-level access to the same entry points (resident keypoints, repeated solves, stepwise GN for the
    sharded multi-GPU mode, introspection). Used by bench.py, ct_icp_amd.distributed and the parity tests."""

    def __init__(self, voxel_map: GpuVoxelMap):
        self.map = voxel_map
        self._h = voxel_map.handle
        self._n = 0

    def set_keypoints(self, raw, world, t):
        if L.is_device_tensor(raw):                      # torch CUDA tensors: gathered on the device
            assert len(raw) == len(world) == len(t)
            self._n = len(t)
            L.check(self._h, L.lib().ctgn_set_keypoints(self._h, L.tensor_view(raw), L.tensor_view(world), L.tensor_view(t, 1), self._n))
            return
        raw = np.ascontiguousarray(raw, dtype=np.float64).reshape(-1, 3)
        world = np.ascontiguousarray(world, dtype=np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(t, dtype=np.float64).ravel()
        assert len(raw) == len(world) == len(t)
        self._n = len(t)
        L.check(self._h, L.lib().ctgn_set_keypoints(self._h, L.View(raw.ctypes.data, 24, L.CTGN_F64, 0),
                                                   L.View(world.ctypes.data, 24, L.CTGN_F64, 0),
                                                   L.View(t.ctypes.data, 8, L.CTGN_F64, 0), self._n))

    def set_keypoints_sharded(self, raw, world, t, rank: int, world_size: int) -> np.ndarray:
        """ctgn_set_keypoints_sharded: the WHOLE scan in, this rank's chunk of its home-voxel order resident; returns the indices (into the
        caller's arrays) of the resident keypoints, in resident order."""
        raw = np.ascontiguousarray(raw, dtype=np.float64).reshape(-1, 3)
        world = np.ascontiguousarray(world, dtype=np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(t, dtype=np.float64).ravel()
        assert len(raw) == len(world) == len(t)
        idx = np.zeros(len(t) // max(1, world_size) + 1, dtype=np.uint32)
        m = C.c_size_t()
        L.check(self._h, L.lib().ctgn_set_keypoints_sharded(self._h, L.View(raw.ctypes.data, 24, L.CTGN_F64, 0), L.View(world.ctypes.data, 24, L.CTGN_F64, 0),
                                                           L.View(t.ctypes.data, 8, L.CTGN_F64, 0), len(t), int(rank), int(world_size),
                                                           idx.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(m)))
        self._n = int(m.value)
        return idx[:self._n].copy()

    def last_upload_bytes(self) -> int:
        """Host-to-device bytes of the last set_keypoints_sharded on this rank (ctgn_internal.h: measurement hook)."""
        out = C.c_uint64()
        L.check(self._h, L.lib().ctgn_last_upload_bytes(self._h, C.byref(out)))
        return int(out.value)

    def set_rewind(self, on=True):
        """ctgn_set_rewind: later uploads keep a device copy of their world points for rewind()."""
        L.check(self._h, L.lib().ctgn_set_rewind(self._h, int(on)))

    def rewind(self):
        """ctgn_rewind_keypoints: world points back to what the last set_keypoints uploaded (enqueued, no synchronisation)."""
        L.check(self._h, L.lib().ctgn_rewind_keypoints(self._h))

    def solve(self, pose14, t_begin_end, options: CTICPOptions, motion_model=None):
        pose = np.ascontiguousarray(pose14, dtype=np.float64).copy()
        tbe = np.ascontiguousarray(t_begin_end, dtype=np.float64)
        opts, prior, s = _c_options(options), _c_prior(motion_model), L.Summary()
        dp = C.POINTER(C.c_double)
        st = L.lib().ctgn_solve(self._h, pose.ctypes.data_as(dp), tbe.ctypes.data_as(dp), C.byref(opts),
                                C.byref(prior) if prior is not None else None, C.byref(s))
        L.check(self._h, st)
        return pose, _summary(s), s

    # ---- keypoint-sharded mode: the collective is issued by the library (ctgn_solve_sharded), RCCL bound at run time
    @staticmethod
    def dist_unique_id() -> bytes:
        out = (C.c_uint8 * 128)()
        st = L.lib().ctgn_dist_unique_id(out)
        if st != 0:
            raise L.CtgnError(st, "ctgn_dist_unique_id failed (is librccl.so.1 loadable?)")
        return bytes(out)

    def dist_init(self, rank: int, world_size: int, unique_id: bytes):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        L.check(self._h, L.lib().ctgn_dist_init(self._h, int(rank), int(world_size), buf))

    def dist_shutdown(self):
        L.check(self._h, L.lib().ctgn_dist_shutdown(self._h))

    def dist_overheads(self, reps: int = 1000):
        """(microseconds of one bare ncclAllReduce of the packed system, microseconds of one sharded iteration whose kernels have nothing to
        do) — ctgn_dist_overheads (ctgn_internal.h); collective."""
        out = (C.c_double * 2)()
        L.check(self._h, L.lib().ctgn_dist_overheads(self._h, int(reps), out))
        return float(out[0]), float(out[1])

    def solve_sharded(self, pose14, t_begin_end, options: CTICPOptions, motion_model=None):
        """The sharded GN loop on this rank's resident keypoints; one ncclAllReduce of the packed system per iteration, issued in C."""
        pose = np.ascontiguousarray(pose14, dtype=np.float64).copy()
        tbe = np.ascontiguousarray(t_begin_end, dtype=np.float64)
        opts, prior, s = _c_options(options), _c_prior(motion_model), L.Summary()
        dp = C.POINTER(C.c_double)
        st = L.lib().ctgn_solve_sharded(self._h, pose.ctypes.data_as(dp), tbe.ctypes.data_as(dp), C.byref(opts),
                                        C.byref(prior) if prior is not None else None, C.byref(s))
        L.check(self._h, st)
        return pose, _summary(s), s

    def solve_robust(self, pose14, t_begin_end, options: CTICPOptions, motion_model=None):
        """DoRegisterCeres on the resident keypoints (their world coordinates are ignored and rewritten)."""
        pose = np.ascontiguousarray(pose14, dtype=np.float64).copy()
        tbe = np.ascontiguousarray(t_begin_end, dtype=np.float64)
        opts, prior, s = _c_robust_options(options), _c_robust_prior(motion_model), L.Summary()
        dp = C.POINTER(C.c_double)
        st = L.lib().ctgn_solve_robust(self._h, pose.ctypes.data_as(dp), tbe.ctypes.data_as(dp), C.byref(opts),
                                       C.byref(prior) if prior is not None else None, C.byref(s))
        L.check(self._h, st)
        return pose, _summary(s), s

    def robust_report(self) -> dict:
        r = L.RobustReport()
        L.check(self._h, L.lib().ctgn_robust_get_report(self._h, C.byref(r)))
        return dict(cost=r.cost, radius=r.radius, diff_rot_deg=r.diff_rot_deg, diff_trans=r.diff_trans,
                    num_residuals=r.num_residuals, ls_iterations=r.ls_iterations, ls_accepted=r.ls_accepted,
                    converged=bool(r.converged), JtJ=np.array(r.JtJ).reshape(12, 12), Jtr=np.array(r.Jtr),
                    step_cycles=[int(c) for c in r.step_cycles])

    def robust_blocks(self) -> dict:
        n = self._n
        normal, weight, alpha, ref = np.zeros((n, 3)), np.zeros(n), np.zeros(n), np.zeros((n, 3))
        rank = np.zeros(n, dtype=np.int32)
        dp = C.POINTER(C.c_double)
        L.check(self._h, L.lib().ctgn_robust_get_blocks(self._h, normal.ctypes.data_as(dp), weight.ctypes.data_as(dp),
                                                       alpha.ctypes.data_as(dp), ref.ctypes.data_as(dp),
                                                       rank.ctypes.data_as(C.POINTER(C.c_int32)), n))
        return dict(normal=normal, weight=weight, alpha=alpha, ref=ref, rank=rank)

    def world_points(self, out=None):
        """Host copy of the (re-transformed) world points, or — given a torch CUDA tensor (N, 3) — written in place on the device."""
        if out is not None and L.is_device_tensor(out):
            v = L.tensor_view(out)
            L.check(self._h, L.lib().ctgn_get_world_points(self._h, v.base, v.stride_bytes, v.dtype, self._n))
            return out
        out = np.zeros((self._n, 3))
        L.check(self._h, L.lib().ctgn_get_world_points(self._h, out.ctypes.data, 24, L.CTGN_F64, self._n))
        return out

    # ---- stepwise ------------------------------------------------------------------------------------------
    def gn_begin(self, pose14, t_begin_end, options: CTICPOptions, motion_model=None):
        pose = np.ascontiguousarray(pose14, dtype=np.float64)
        tbe = np.ascontiguousarray(t_begin_end, dtype=np.float64)
        opts, prior = _c_options(options), _c_prior(motion_model)
        dp = C.POINTER(C.c_double)
        L.check(self._h, L.lib().ctgn_gn_begin(self._h, pose.ctypes.data_as(dp), tbe.ctypes.data_as(dp), C.byref(opts),
                                              C.byref(prior) if prior is not None else None))

    def gn_iterate(self, iterations: int, sharded: bool = False):
        """Enqueue whole GN iterations behind gn_begin (no synchronisation); see ctgn_gn_iterate."""
        L.check(self._h, L.lib().ctgn_gn_iterate(self._h, int(iterations), int(bool(sharded))))

    def gn_accumulate(self):
        L.check(self._h, L.lib().ctgn_gn_accumulate(self._h))

    def gn_solve_update(self):
        L.check(self._h, L.lib().ctgn_gn_solve_update(self._h))

    def gn_system_ptr(self) -> int:
        p = C.c_void_p()
        L.check(self._h, L.lib().ctgn_gn_system_device_ptr(self._h, C.byref(p)))
        return p.value

    def gn_set_system_buffer(self, device_ptr: int | None):
        L.check(self._h, L.lib().ctgn_gn_set_system_buffer(self._h, C.c_void_p(device_ptr or 0)))

    def gn_done(self) -> bool:
        d = C.c_int32()
        L.check(self._h, L.lib().ctgn_gn_done(self._h, C.byref(d)))
        return bool(d.value)

    def gn_end(self):
        pose, s = np.zeros(14), L.Summary()
        L.check(self._h, L.lib().ctgn_gn_end(self._h, pose.ctypes.data_as(C.POINTER(C.c_double)), C.byref(s)))
        return pose, _summary(s), s

    # ---- introspection ---------------------------------------------------------------------------------------
    def set_debug(self, on=True):
        L.check(self._h, L.lib().ctgn_set_debug(self._h, int(on)))

    def get_debug(self):
        n = self._n
        nn = np.zeros(n, dtype=np.int32)
        normal, a2d, far = np.zeros((n, 3)), np.zeros(n), np.zeros((n, 3))
        used = np.zeros(n, dtype=np.uint8)
        dp = C.POINTER(C.c_double)
        L.check(self._h, L.lib().ctgn_get_debug(self._h, nn.ctypes.data_as(C.POINTER(C.c_int32)), normal.ctypes.data_as(dp),
                                               a2d.ctypes.data_as(dp), far.ctypes.data_as(dp),
                                               used.ctypes.data_as(C.POINTER(C.c_uint8)), n))
        return dict(n_neighbors=nn, normal=normal, a2d=a2d, farthest=far, used=used.astype(bool))

    def get_system(self):
        """(A 12x12, b 12, n_used) unpacked from the 96-double packed system of the last accumulate."""
        s = np.zeros(L.CTGN_SYSTEM_DOUBLES)
        L.check(self._h, L.lib().ctgn_get_system(self._h, s.ctypes.data_as(C.POINTER(C.c_double))))
        A = np.zeros((12, 12))
        iu = np.triu_indices(12)
        A[iu] = s[:78]
        A = A + np.triu(A, 1).T
        return A, s[78:90].copy(), int(round(s[90]))

    def count_traffic(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        L.check(self._h, L.lib().ctgn_count_traffic(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def path_counters(self):
        """(residual launches with per-XCD pre-sums, 1 if the last solve launch summed the group records) — ctgn_path_counters."""
        out = (C.c_uint64 * 2)()
        L.check(self._h, L.lib().ctgn_path_counters(self._h, out))
        return int(out[0]), int(out[1])

    def measure_hbm(self, nbytes=1 << 30, reps=10):
        """(copy, triad, runtime device-to-device copy) GB/s of bytes read + written on this device (ctgn_measure_hbm, measurement hook)."""
        out = (C.c_double * 3)()
        L.check(self._h, L.lib().ctgn_measure_hbm(self._h, int(nbytes), int(reps), out))
        return float(out[0]), float(out[1]), float(out[2])

    def set_profiling(self, on=True):
        L.check(self._h, L.lib().ctgn_set_profiling(self._h, int(on)))

    def kernel_timing(self, reset=False):
        ms, n = C.c_double(), C.c_int32()
        L.check(self._h, L.lib().ctgn_kernel_timing(self._h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def kernel_timing_split(self, reset=False):
        """((ms, launches) of the first search of a solve, (ms, launches) of the searches with a carried-over bound)."""
        ms, n = (C.c_double * 2)(), (C.c_int32 * 2)()
        L.check(self._h, L.lib().ctgn_kernel_timing_split(self._h, ms, n, int(reset)))
        return (ms[0], n[0]), (ms[1], n[1])

    def phase_cycles(self, reset=False):
        out = (C.c_uint64 * 12)()
        L.check(self._h, L.lib().ctgn_phase_cycles(self._h, out, int(reset)))
        return [int(x) for x in out]

    def traffic_counters(self, reset=False):
        """(hash probes issued, map points streamed) by the instrumented row kernel (variant 3) since the last reset."""
        out = (C.c_uint64 * 2)()
        L.check(self._h, L.lib().ctgn_traffic_counters(self._h, out, int(reset)))
        return int(out[0]), int(out[1])

    def wave_timeline(self, max_waves: int = 8192) -> np.ndarray:
        """(waves, 4) uint64: start clock, end clock, fast-path rounds, rounds of the last variant-3 launch."""
        out = np.zeros((max_waves, 4), dtype=np.uint64)
        n = C.c_size_t()
        L.check(self._h, L.lib().ctgn_wave_timeline(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), max_waves, C.byref(n)))
        out = out[:n.value]
        return out[out[:, 1] > 0]

    def set_search_guess(self, factor: float):
        """< 0 automatic, 0 off, > 0 forced factor of the first search's guessed bound: ct_icp_amd/csrc/ctgn_internal.h."""
        L.check(self._h, L.lib().ctgn_set_search_guess(self._h, float(factor)))

    def set_normals(self, mode: int):
        """0 library default (hybrid), 1 exact (bit-identical normals, slower), 2 hybrid, 3 fast: ct_icp_amd/csrc/ctgn_internal.h."""
        L.check(self._h, L.lib().ctgn_set_normals(self._h, mode))

    def set_ablation(self, mask: int):
        L.check(self._h, L.lib().ctgn_set_ablation(self._h, mask))

    def set_ordering(self, mode: int):
        """-1 automatic (default), 0 never, 1 always: home-voxel ordering of the GN kernels' work (include/ctgn.h)."""
        L.check(self._h, L.lib().ctgn_set_ordering(self._h, int(mode)))

    def set_persistent(self, mode: int):
        """ctgn_set_persistent: 1 = small frames (<= 1 024 keypoints) run as one persistent launch, 0 (default) = the three-launch loop."""
        L.check(self._h, L.lib().ctgn_set_persistent(self._h, int(mode)))

    def set_pools(self, mode: int):
        """ctgn_set_pools: -1 = automatic (frames of >= 8 192 keypoints), 0 = every iteration searches, 1 = pools for every frame."""
        L.check(self._h, L.lib().ctgn_set_pools(self._h, int(mode)))

    def set_variant(self, v: int):
        L.check(self._h, L.lib().ctgn_set_variant(self._h, v))

    def set_stream(self, stream_ptr: int):
        L.check(self._h, L.lib().ctgn_set_stream(self._h, C.c_void_p(stream_ptr)))
