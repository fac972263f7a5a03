"""ctypes wrapper of oracle/_ref/libctgn_ref.so: the REFERENCE'S OWN SOURCES (compiled from /root/reference by
`make -C oracle _ref`, see oracle/Makefile and oracle/ref_wrap.cpp) against the third-party shims of oracle/shims/.

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; nothing under
ct_icp_amd/ may import it (tests/test_layout.py).  The library is built in the CPU container (where /root/reference exists)
and travels to the GPU box as a built file; `available()` says whether it is there.

What it pins: control flow, gates, visit order, heap behaviour, residual / Jacobian statements, motion-model terms and
the pose update of the reference are its literal code.  What it does not pin: the arithmetic of Eigen / Ceres, restated
in the shims (rounding-level differences only for Eigen; algorithm-level restatement for Ceres' minimiser).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libctgn_ref.so")
REFERENCE_ROOT = "/root/reference"

# numpy mirror of slam::WPoint3D as compiled (include/SlamCore/types.h:35-60); checked against ref_layout() on load
WPOINT3D_DTYPE = np.dtype({"names": ["raw_point", "t", "world_point", "index_frame"],
                           "formats": [("<f8", 3), "<f8", ("<f8", 3), "<u4"],
                           "offsets": [0, 24, 32, 56], "itemsize": 64})

LOSS = {"STANDARD": 0, "CAUCHY": 1, "HUBER": 2, "TOLERANT": 3, "TRUNCATED": 4}


class _Opts(C.Structure):
    _fields_ = [("solver", C.c_int), ("num_iters_icp", C.c_int), ("min_number_neighbors", C.c_int),
                ("max_number_neighbors", C.c_int), ("max_dist_to_plane_ct_icp", C.c_double),
                ("threshold_orientation_norm", C.c_double), ("threshold_translation_norm", C.c_double),
                ("loss_function", C.c_int), ("ls_max_num_iters", C.c_int), ("ls_num_threads", C.c_int),
                ("max_num_residuals", C.c_int), ("num_closest_neighbors", C.c_int), ("ls_sigma", C.c_double),
                ("ls_tolerant_min_threshold", C.c_double), ("weight_alpha", C.c_double), ("weight_neighborhood", C.c_double),
                ("power_planarity", C.c_double), ("point_to_plane_with_distortion", C.c_int)]


class _Prior(C.Structure):
    _fields_ = [("beta_location_consistency", C.c_double), ("beta_constant_velocity", C.c_double),
                ("beta_small_velocity", C.c_double), ("beta_orientation_consistency", C.c_double),
                ("previous_pose", C.c_double * 14), ("previous_t_begin_end", C.c_double * 2)]


class _Summary(C.Structure):
    _fields_ = [("success", C.c_int), ("num_residuals_used", C.c_int), ("num_iters", C.c_int), ("error_log", C.c_char * 256)]


_lib = None


def build() -> str | None:
    """(Re)build the library when the reference sources are present; otherwise keep whatever was prebuilt."""
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "ct_icp")):
        subprocess.check_call(["make", "-s", "-j8", "-C", _HERE, "_ref"])
    return _SO if os.path.exists(_SO) else None


def available() -> bool:
    return os.path.exists(_SO) or os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "ct_icp"))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    dp, sz = C.POINTER(C.c_double), C.c_size_t
    L.ref_last_error.restype = C.c_char_p
    L.ref_layout.argtypes = [C.POINTER(sz)]
    L.ref_map_create.restype = C.c_void_p
    L.ref_map_create.argtypes = [C.c_int, dp, dp, C.POINTER(C.c_int), C.c_double]
    L.ref_map_destroy.argtypes = [C.c_void_p]
    L.ref_map_insert.argtypes = [C.c_void_p, dp, sz, C.c_int, C.POINTER(C.c_uint8)]
    L.ref_map_remove_far.argtypes = [C.c_void_p, dp, C.c_double]
    L.ref_map_clear.argtypes = [C.c_void_p]
    L.ref_map_num_points.restype = sz
    L.ref_map_num_points.argtypes = [C.c_void_p, C.c_int]
    L.ref_map_export.argtypes = [C.c_void_p, C.c_int, dp, sz, C.POINTER(sz)]
    L.ref_map_search_params.argtypes = [C.c_void_p, C.c_double, dp]
    L.ref_radius_search.argtypes = [C.c_void_p, dp, sz, C.c_double, C.c_int, C.POINTER(C.c_int), dp]
    L.ref_neighborhood.argtypes = [dp, sz, dp, dp]
    L.ref_voxel_coordinates.argtypes = [dp, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong)]
    L.ref_alpha_timestamp.restype = C.c_double
    L.ref_alpha_timestamp.argtypes = [C.c_double, C.c_double, C.c_double]
    L.ref_transform_points.argtypes = [dp, dp, dp, dp, sz, dp]
    L.ref_sub_sample_frame.argtypes = [dp, sz, C.c_double, C.POINTER(C.c_uint32), C.POINTER(sz)]
    L.ref_register.argtypes = [C.c_void_p, C.c_void_p, sz, dp, dp, C.POINTER(_Opts), C.POINTER(_Prior), C.POINTER(_Summary)]
    lay = (sz * 16)()
    L.ref_layout(lay)
    assert lay[0] == WPOINT3D_DTYPE.itemsize and (lay[1], lay[2], lay[3], lay[4]) == (0, 24, 32, 56), list(lay)
    _lib = L
    return L


def layout() -> dict:
    lay = (C.c_size_t * 16)()
    lib().ref_layout(lay)
    names = ["sizeof_WPoint3D", "off_raw_point", "off_timestamp", "off_world_point", "off_index_frame", "sizeof_Pose",
             "off_pose", "off_ref_timestamp", "off_dest_timestamp", "off_ref_frame_id", "off_dest_frame_id",
             "sizeof_TrajectoryFrame", "sizeof_SE3", "off_quat", "off_tr", "sizeof_Voxel"]
    return dict(zip(names, (int(v) for v in lay)))


class RefError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise RefError(f"reference raised (rc={rc}): {lib().ref_last_error().decode(errors='replace')}")


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Map:
    """ct_icp::MultipleResolutionVoxelMap -- the reference's class (include/ct_icp/map.h:100-607)."""

    def __init__(self, resolutions=((0.2, 0.03, 50), (0.5, 0.1, 40), (1.5, 0.15, 40)), default_radius=0.8):
        res = _f64([r[0] for r in resolutions]); md = _f64([r[1] for r in resolutions])
        mp = np.ascontiguousarray([r[2] for r in resolutions], dtype=np.int32)
        self._h = lib().ref_map_create(len(resolutions), _dp(res), _dp(md), mp.ctypes.data_as(C.POINTER(C.c_int)), float(default_radius))
        self.resolutions = [tuple(r) for r in resolutions]
        self.default_radius = default_radius

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.ref_map_destroy(self._h)
            self._h = None

    def insert(self, xyz, via_pointcloud=False) -> np.ndarray | None:
        """via_pointcloud=False: InsertPointInVoxelMap per point and level, returns the (n, n_levels) inserted mask;
        True: the whole InsertPointCloud entry point (no mask)."""
        xyz = _f64(xyz).reshape(-1, 3)
        if via_pointcloud:
            _check(lib().ref_map_insert(self._h, _dp(xyz), len(xyz), 1, None))
            return None
        out = np.zeros((len(xyz), len(self.resolutions)), dtype=np.uint8)
        _check(lib().ref_map_insert(self._h, _dp(xyz), len(xyz), 0, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out.astype(bool)

    def remove_far(self, location, distance):
        _check(lib().ref_map_remove_far(self._h, _dp(_f64(location)), float(distance)))

    def clear(self):
        lib().ref_map_clear(self._h)

    def num_points(self, res_index=0) -> int:
        return int(lib().ref_map_num_points(self._h, res_index))

    def export(self, res_index=0) -> np.ndarray:
        n = C.c_size_t(0)
        cap = self.num_points(res_index)
        out = np.zeros((max(cap, 1), 3))
        _check(lib().ref_map_export(self._h, res_index, _dp(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def search_params(self, radius=None):
        out = np.zeros(4)
        lib().ref_map_search_params(self._h, self.default_radius if radius is None else float(radius), _dp(out))
        return int(out[2]), float(out[1]), int(out[3])        # (map_id, voxel_resolution, voxel_neighborhood) like oracle.Map

    def radius_search(self, queries, radius=0.0, max_num_neighbors=20):
        """radius == 0: ComputeNeighborhood (default radius). Returns (counts[n], xyz[n, k, 3]) farthest first."""
        q = _f64(queries).reshape(-1, 3)
        cnt = np.zeros(len(q), dtype=np.int32)
        out = np.zeros((len(q), max_num_neighbors, 3))
        _check(lib().ref_radius_search(self._h, _dp(q), len(q), float(radius), int(max_num_neighbors),
                                       cnt.ctypes.data_as(C.POINTER(C.c_int)), _dp(out)))
        return cnt, out


@dataclass
class Options:
    """CTICPOptions fields the GN and CERES routes read (include/ct_icp/ct_icp.h:56-153), reference defaults."""
    solver: str = "GN"
    num_iters_icp: int = 5
    min_number_neighbors: int = 20
    max_number_neighbors: int = 20
    max_dist_to_plane_ct_icp: float = 0.3
    threshold_orientation_norm: float = 1e-4
    threshold_translation_norm: float = 1e-3
    loss_function: str = "CAUCHY"
    ls_max_num_iters: int = 1
    ls_num_threads: int = 1
    max_num_residuals: int = -1
    num_closest_neighbors: int = 1
    ls_sigma: float = 0.1
    ls_tolerant_min_threshold: float = 0.05
    weight_alpha: float = 0.9
    weight_neighborhood: float = 0.1
    power_planarity: float = 2.0
    point_to_plane_with_distortion: bool = True

    def c(self) -> _Opts:
        return _Opts({"GN": 0, "CERES": 1}[self.solver], self.num_iters_icp, self.min_number_neighbors, self.max_number_neighbors,
                     self.max_dist_to_plane_ct_icp, self.threshold_orientation_norm, self.threshold_translation_norm,
                     LOSS[self.loss_function], self.ls_max_num_iters, self.ls_num_threads, self.max_num_residuals,
                     self.num_closest_neighbors, self.ls_sigma, self.ls_tolerant_min_threshold, self.weight_alpha,
                     self.weight_neighborhood, self.power_planarity, int(self.point_to_plane_with_distortion))


@dataclass
class Prior:
    """PreviousFrameMotionModel (include/ct_icp/motion_model.h:33-84) with its previous frame."""
    beta_location_consistency: float = 0.001
    beta_constant_velocity: float = 0.001
    beta_small_velocity: float = 0.0
    beta_orientation_consistency: float = 0.0
    previous_pose: np.ndarray = field(default_factory=lambda: np.array([0, 0, 0, 1, 0, 0, 0] * 2, dtype=np.float64))
    previous_t_begin_end: tuple = (0.0, 0.0)

    def c(self) -> _Prior:
        p = _Prior(self.beta_location_consistency, self.beta_constant_velocity, self.beta_small_velocity,
                   self.beta_orientation_consistency)
        for i in range(14):
            p.previous_pose[i] = float(np.asarray(self.previous_pose).ravel()[i])
        p.previous_t_begin_end[0], p.previous_t_begin_end[1] = map(float, self.previous_t_begin_end)
        return p


@dataclass
class Summary:
    success: bool
    num_residuals_used: int
    num_iters: int
    error_log: str


def register(m: Map, raw, world, t, pose, t_begin_end, opts: Options, prior: Prior | None = None):
    """CT_ICP_Registration::Register(map, std::vector<WPoint3D>&, TrajectoryFrame&, motion_model) -- ct_icp.cpp:1026-1038.
    Returns (pose14, world[n, 3], Summary)."""
    raw, world, t = _f64(raw).reshape(-1, 3), _f64(world).reshape(-1, 3), _f64(t).ravel()
    kp = np.zeros(len(t), dtype=WPOINT3D_DTYPE)
    kp["raw_point"], kp["t"], kp["world_point"], kp["index_frame"] = raw, t, world, 0
    pose, tbe = _f64(pose).ravel().copy(), _f64(t_begin_end)
    o = opts.c()
    p = prior.c() if prior is not None else None
    s = _Summary()
    _check(lib().ref_register(m._h, kp.ctypes.data_as(C.c_void_p), len(kp), _dp(pose), _dp(tbe), C.byref(o),
                              C.byref(p) if p is not None else None, C.byref(s)))
    return pose, kp["world_point"].copy(), Summary(bool(s.success), s.num_residuals_used, s.num_iters, s.error_log.decode())


def neighborhood(points):
    pts = _f64(points).reshape(-1, 3)
    normal, a2d = np.zeros(3), C.c_double(0)
    ok = lib().ref_neighborhood(_dp(pts), len(pts), _dp(normal), C.byref(a2d))
    return (normal, a2d.value) if ok else None


def voxel_coordinates(p, voxel_size):
    out = (C.c_int * 3)(); h = C.c_ulonglong(0)
    lib().ref_voxel_coordinates(_dp(_f64(p)), float(voxel_size), out, C.byref(h))
    return (out[0], out[1], out[2]), int(h.value)


def alpha_timestamp(t, tb, te) -> float:
    return float(lib().ref_alpha_timestamp(float(t), float(tb), float(te)))


def transform_points(pose, t_begin_end, t, raw) -> np.ndarray:
    raw, t = _f64(raw).reshape(-1, 3), _f64(t).ravel()
    out = np.zeros_like(raw)
    _check(lib().ref_transform_points(_dp(_f64(pose).ravel()), _dp(_f64(t_begin_end)), _dp(t), _dp(raw), len(t), _dp(out)))
    return out


def sub_sample_frame(raw, voxel_size) -> np.ndarray:
    """Indices of the surviving points, in the (shimmed) container's iteration order -- compare as a set."""
    raw = _f64(raw).reshape(-1, 3)
    out = np.zeros(max(len(raw), 1), dtype=np.uint32)
    n = C.c_size_t(0)
    _check(lib().ref_sub_sample_frame(_dp(raw), len(raw), float(voxel_size), out.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n)))
    return out[:n.value].copy()
