"""ctypes wrapper of oracle/_ref/libctgn_ref_odometry.so: the REFERENCE'S OWN ct_icp::Odometry (src/ct_icp/odometry.cpp compiled where it
lies under /root/reference, `make -C oracle odometry`) behind integration/glue_odometry.cpp, on either of two maps that both come out of
the reference's own factory call (odometry.cpp:700):

    RefOdometry(map_kind=CPU_MAP)   MULTI_RESOLUTION_VOXEL_HASHMAP: the reference's map and its CPU solver loops
    RefOdometry(map_kind=GPU_MAP)   GPU_VOXEL_HASHMAP: integration/gpu_map.h over libctgn.so, Register through the arms of integration/gn_gpu_arm.h
    RefOdometry(map_kind=GPU_MAP_ARMED)   the same map in oracle/_ref/libctgn_ref_odometry_armed.so, whose odometry.cpp was compiled with the four
                                    arms of integration/odometry_gpu_arm.h (`make -C oracle odometry-armed`): InitializeFrame, TryRegister, the
                                    undistortion loops and the map half of UpdateMap run on the device, the scan resident from upload to insert.
                                    RefOdometry(GPU_MAP, armed_library=True) = that library with the arms standing down (`frame_pipeline` off).
    RefOdometry(map_kind=GPU_MAP_ARMED_DEVICE_SHUFFLE)   ... with `frame_shuffle_on_device`: the shuffle in front of sub_sample_frame is a keyed
                                    permutation made on the GPU instead of std::shuffle on the host

TEST INFRASTRUCTURE ONLY (tests/, tests/odometry_vs_reference.py, bench.py's cpu_baseline leg); nothing under ct_icp_amd/ imports it.
The library is built in the CPU container and travels to the GPU box as a built file. Third-party arithmetic underneath the reference
(Eigen, Ceres, tsl::robin_map, glog) is oracle/shims/ — on BOTH map kinds, so the comparison isolates exactly the drop-in.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libctgn_ref_odometry.so")
_SO_ARMED = os.path.join(_HERE, "_ref", "libctgn_ref_odometry_armed.so")
REFERENCE_ROOT = "/root/reference"
CPU_MAP, GPU_MAP, GPU_MAP_ARMED, GPU_MAP_ARMED_DEVICE_SHUFFLE = 0, 1, 2, 3
GN, CERES = 0, 1                      # ct_icp::CT_ICP_SOLVER (include/ct_icp/ct_icp.h:35-39)
DRIVING_YAML, DEFAULT_DRIVING, ROBUST_DRIVING, ROBUST_LOW_INERTIA = 0, 1, 2, 3


class Result(C.Structure):
    _fields_ = [("pose", C.c_double * 14), ("initial_pose", C.c_double * 14), ("relative_distance", C.c_double),
                ("relative_orientation", C.c_double), ("ego_orientation", C.c_double), ("distance_correction", C.c_double),
                ("milliseconds", C.c_double), ("success", C.c_int32), ("points_added", C.c_int32), ("sample_size", C.c_int32),
                ("number_of_residuals", C.c_int32), ("number_of_attempts", C.c_int32), ("robust_level", C.c_int32),
                ("icp_num_iters", C.c_int32), ("num_corrected", C.c_int32), ("map_points", C.c_uint64), ("phase_ms", C.c_double * 6),
                ("gpu_ms", C.c_double * 6)]


PHASES = ("total", "initialize_frame", "try_register", "undistort", "map_update", "initialize_motion")   # Result.phase_ms (glue_odometry.cpp)
ARM_PHASES = ("host_shuffles", "frame_begin_call", "build_sampled_frame", "try_register_arm", "undistort_call", "undistort_arm")   # Result.gpu_ms


_libs = {}


def build():
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "ct_icp")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(_HERE), "ct_icp_amd", "csrc"), "all"])
        subprocess.check_call(["make", "-s", "-j8", "-C", _HERE, "_ref"])
        subprocess.check_call(["make", "-s", "-j8", "-C", _HERE, "odometry", "odometry-armed"])
    return _SO if os.path.exists(_SO) and os.path.exists(_SO_ARMED) else None


def available() -> bool:
    return (os.path.exists(_SO) and os.path.exists(_SO_ARMED)) or os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "ct_icp"))


def lib(armed: bool = False):
    if armed not in _libs:
        path = _SO_ARMED if armed else _SO
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        L.glue_odometry_last_error.restype = C.c_char_p
        L.glue_odometry_options.restype = C.c_void_p
        L.glue_odometry_options.argtypes = [C.c_int]
        L.glue_odometry_set.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.glue_odometry_set_resolutions.argtypes = [C.c_void_p, C.c_int, dp, dp, C.POINTER(C.c_int32)]
        L.glue_odometry_start.argtypes = [C.c_void_p, C.c_int]
        L.glue_odometry_destroy.argtypes = [C.c_void_p]
        L.glue_odometry_register_frame.argtypes = [C.c_void_p, dp, dp, C.c_size_t, C.c_int, C.POINTER(Result), dp, C.c_int, dp]
        L.glue_odometry_map_points.argtypes = [C.c_void_p, dp, C.c_uint64, C.POINTER(C.c_uint64)]
        assert L.glue_odometry_is_armed() == (1 if armed else 0)
        _libs[armed] = L
    return _libs[armed]


class NoDevice(RuntimeError):
    pass


class RefOdometry:
    """ct_icp::Odometry constructed from `profile` + `options` (the reference's YAML key names) on the given map kind."""

    def __init__(self, map_kind: int = CPU_MAP, profile: int = DRIVING_YAML, resolutions=((0.8, 0.1, 30),), armed_library: bool = False, **options):
        self._armed = armed_library or map_kind >= GPU_MAP_ARMED
        L = self._lib = lib(self._armed)
        self._h = L.glue_odometry_options(profile)
        for k, v in options.items():
            if L.glue_odometry_set(self._h, k.encode(), float(v)) != 0:
                raise KeyError(L.glue_odometry_last_error().decode())
        res = np.ascontiguousarray([r[0] for r in resolutions], dtype=np.float64)
        md = np.ascontiguousarray([r[1] for r in resolutions], dtype=np.float64)
        mp = np.ascontiguousarray([r[2] for r in resolutions], dtype=np.int32)
        dp = C.POINTER(C.c_double)
        L.glue_odometry_set_resolutions(self._h, len(res), res.ctypes.data_as(dp), md.ctypes.data_as(dp), mp.ctypes.data_as(C.POINTER(C.c_int32)))
        if L.glue_odometry_start(self._h, map_kind) != 0:
            msg = L.glue_odometry_last_error().decode()
            L.glue_odometry_destroy(self._h)
            self._h = None
            raise (NoDevice if "libctgn" in msg else RuntimeError)(msg)
        self.map_kind = map_kind
        self._frames = 0

    def register_frame(self, raw: np.ndarray, t: np.ndarray, want_world: bool = False, want_map_points: bool = False, want_sampled: bool = False):
        """Odometry::RegisterFrame(PointCloud, frame_id) (odometry.cpp:209-224). Returns a dict of the RegistrationSummary's fields (+ 'world':
        all_corrected_points' world points, 'sampled_raw': the raw points of corrected_points = the sampled frame, on request)."""
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        t = np.ascontiguousarray(t, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        res = Result()
        world = np.empty_like(raw) if want_world else None
        sampled = np.empty_like(raw) if want_sampled else None
        rc = self._lib.glue_odometry_register_frame(self._h, raw.ctypes.data_as(dp), t.ctypes.data_as(dp), len(t), self._frames, C.byref(res),
                                                    world.ctypes.data_as(dp) if want_world else None, 1 if want_map_points else 0,
                                                    sampled.ctypes.data_as(dp) if want_sampled else None)
        if rc != 0:
            raise RuntimeError(self._lib.glue_odometry_last_error().decode())
        self._frames += 1
        out = {k: getattr(res, k) for k, _ in Result._fields_ if k not in ("pose", "initial_pose", "phase_ms", "gpu_ms")}
        out["pose"] = np.array(res.pose)
        out["initial_pose"] = np.array(res.initial_pose)
        out["phase_ms"] = dict(zip(PHASES, res.phase_ms))
        out["arm_ms"] = dict(zip(ARM_PHASES, res.gpu_ms))
        out["success"] = bool(res.success)
        out["points_added"] = bool(res.points_added)
        if want_world:
            out["world"] = world
        if want_sampled:
            out["sampled_raw"] = sampled[:res.num_corrected]
        return out

    def map_points(self) -> np.ndarray:
        n = C.c_uint64(0)
        dp = C.POINTER(C.c_double)
        if self._lib.glue_odometry_map_points(self._h, None, 0, C.byref(n)) != 0:
            raise RuntimeError(self._lib.glue_odometry_last_error().decode())
        out = np.empty((n.value, 3))
        self._lib.glue_odometry_map_points(self._h, out.ctypes.data_as(dp), n.value, C.byref(n))
        return out

    def close(self):
        if self._h is not None:
            self._lib.glue_odometry_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
