"""ctypes wrapper of the CPU oracle (oracle/ctgn_oracle.c).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under ct_icp_amd/ may import this module (tests/test_layout.py enforces it).
Parity pinned against oracle/_ref (the reference's own sources) — see oracle/ctgn_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libctgn_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (a few seconds). Returns the .so path."""
    srcs = [os.path.join(_HERE, f) for f in ("ctgn_oracle.c", "ctgn_oracle_robust.c", "ref_shaped.cpp", "ctgn_oracle.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-B", "-C", _HERE])
    return _SO


class _Res(C.Structure):
    _fields_ = [("resolution", C.c_double), ("min_distance_between_points", C.c_double),
                ("max_num_points", C.c_int)]


class _Opts(C.Structure):
    _fields_ = [("num_iters_icp", C.c_int), ("min_number_neighbors", C.c_int),
                ("max_number_neighbors", C.c_int), ("debug_print", C.c_int),
                ("max_dist_to_plane_ct_icp", C.c_double), ("threshold_orientation_norm", C.c_double)]


class _Prior(C.Structure):
    _fields_ = [("beta_location_consistency", C.c_double), ("beta_constant_velocity", C.c_double),
                ("previous_begin_tr", C.c_double * 3), ("previous_end_tr", C.c_double * 3)]


class _Summary(C.Structure):
    _fields_ = [("success", C.c_int), ("num_residuals_used", C.c_int), ("num_iters", C.c_int),
                ("last_step_norm", C.c_double), ("error_log", C.c_char * 256),
                ("t_neighbors", C.c_double), ("t_normals", C.c_double), ("t_jacobian", C.c_double),
                ("t_solve", C.c_double), ("t_update", C.c_double)]


class _RobustOpts(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("num_iters_icp", "min_number_neighbors", "max_number_neighbors", "debug_print",
                                       "max_num_residuals", "loss_function", "ls_max_num_iters",
                                       "num_closest_neighbors")] + \
               [(k, C.c_double) for k in ("weight_alpha", "weight_neighborhood", "power_planarity",
                                          "max_dist_to_plane_ct_icp", "ls_sigma", "ls_tolerant_min_threshold",
                                          "threshold_orientation_norm", "threshold_translation_norm")]


class _RobustPrior(C.Structure):
    _fields_ = [("beta_location_consistency", C.c_double), ("beta_constant_velocity", C.c_double),
                ("beta_small_velocity", C.c_double), ("beta_orientation_consistency", C.c_double),
                ("previous_begin_tr", C.c_double * 3), ("previous_end_tr", C.c_double * 3),
                ("previous_end_quat", C.c_double * 4)]


class _LmReport(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("final_radius", C.c_double),
                ("iterations", C.c_int), ("num_successful_steps", C.c_int), ("num_unsuccessful_steps", C.c_int),
                ("termination", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        L.orc_map_create.restype = C.c_void_p
        L.orc_map_create.argtypes = [C.POINTER(_Res), C.c_int, C.c_double]
        L.orc_map_destroy.argtypes = [C.c_void_p]
        L.orc_map_clear.argtypes = [C.c_void_p]
        L.orc_map_insert.argtypes = [C.c_void_p, dp, C.c_size_t, C.POINTER(C.c_uint8)]
        L.orc_map_remove_far.argtypes = [C.c_void_p, dp, C.c_double]
        L.orc_map_num_points.restype = C.c_uint64
        L.orc_map_num_points.argtypes = [C.c_void_p]
        L.orc_map_num_voxels.restype = C.c_uint64
        L.orc_map_num_voxels.argtypes = [C.c_void_p, C.c_int]
        L.orc_map_export.restype = C.c_uint64
        L.orc_map_export.argtypes = [C.c_void_p, C.c_int, dp, C.c_uint64]
        L.orc_map_search_params.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_int), dp, C.POINTER(C.c_int)]
        L.orc_map_radius_search.restype = C.c_int
        L.orc_map_radius_search.argtypes = [C.c_void_p, dp, C.c_double, C.c_int, C.c_int, dp]
        L.orc_map_count.argtypes = [C.c_void_p, dp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                    C.POINTER(C.c_uint64)]
        L.orc_voxel_coord.restype = C.c_int
        L.orc_voxel_coord.argtypes = [C.c_double, C.c_double]
        L.orc_alpha_timestamp.restype = C.c_double
        L.orc_alpha_timestamp.argtypes = [C.c_double] * 3
        L.orc_quat_normalize.argtypes = [dp]
        L.orc_quat_rotate.argtypes = [dp, dp, dp]
        L.orc_quat_slerp.argtypes = [dp, dp, C.c_double, dp]
        L.orc_quat_to_matrix.argtypes = [dp, dp]
        L.orc_matrix_to_quat.argtypes = [dp, dp]
        L.orc_transform_point.argtypes = [dp, dp, C.c_double, dp, dp]
        L.orc_transform_points.argtypes = [dp, dp, dp, dp, C.c_size_t, dp, C.c_int]
        L.orc_neighborhood.restype = C.c_int
        L.orc_neighborhood.argtypes = [dp, C.c_int, dp, dp]
        L.orc_sym_eigen3.argtypes = [dp, dp, dp]
        L.orc_ldlt_solve12.argtypes = [dp, dp, dp]
        L.orc_gn_accumulate.argtypes = [C.c_void_p, dp, dp, dp, C.c_size_t, dp, dp, C.POINTER(_Opts), C.c_int,
                                        C.c_int, dp, dp, C.POINTER(C.c_int), C.POINTER(C.c_int32), dp, dp, dp,
                                        C.POINTER(C.c_uint8)]
        L.orc_gn_solve_update.restype = C.c_double
        L.orc_gn_solve_update.argtypes = [dp, dp, C.c_int, C.POINTER(_Prior), dp, dp]
        L.orc_register_gn.restype = C.c_int
        L.orc_register_gn.argtypes = [C.c_void_p, dp, dp, dp, C.c_size_t, dp, dp, C.POINTER(_Opts),
                                      C.POINTER(_Prior), C.c_int, C.c_int, C.POINTER(_Summary)]
        L.orc_grid_sampling.restype = C.c_size_t
        L.orc_grid_sampling.argtypes = [dp, C.c_size_t, C.c_double, C.POINTER(C.c_uint32)]
        L.orc_adaptive_sampling.restype = C.c_size_t
        L.orc_adaptive_sampling.argtypes = [dp, C.c_size_t, C.c_int, C.c_int, C.c_int, dp, dp, C.POINTER(C.c_uint32)]
        # reference-shaped containers (ref_shaped.cpp)
        L.orc_refshaped_create.restype = C.c_void_p
        L.orc_refshaped_create.argtypes = [dp, C.c_size_t, C.c_double]
        L.orc_refshaped_destroy.argtypes = [C.c_void_p]
        L.orc_refshaped_gn_accumulate.argtypes = [C.c_void_p, dp, dp, dp, C.c_size_t, dp, dp, C.POINTER(_Opts), C.c_double,
                                                  C.c_int, C.c_int, dp, dp, C.POINTER(C.c_int)]
        # robust-loss route (ctgn_oracle_robust.c)
        L.orc_loss_evaluate.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, dp]
        L.orc_ct_point_to_plane.argtypes = [dp, C.c_double, dp, dp, dp, C.c_double, dp, dp]
        L.orc_robust_build.restype = C.c_size_t
        L.orc_robust_build.argtypes = [C.c_void_p, dp, dp, dp, C.c_size_t, dp, C.POINTER(_RobustOpts), C.c_int, dp, dp,
                                       dp, dp, dp, C.POINTER(C.c_int32)]
        L.orc_robust_evaluate_fixed.restype = C.c_double
        L.orc_robust_evaluate_fixed.argtypes = [dp, dp, dp, dp, dp, C.c_size_t, C.POINTER(_RobustOpts),
                                                C.POINTER(_RobustPrior), dp, dp, dp]
        L.orc_robust_solve_fixed.restype = C.c_int
        L.orc_robust_solve_fixed.argtypes = [dp, dp, dp, dp, dp, C.c_size_t, C.POINTER(_RobustOpts),
                                             C.POINTER(_RobustPrior), dp, C.c_int, C.POINTER(_LmReport)]
        L.orc_register_robust.restype = C.c_int
        L.orc_register_robust.argtypes = [C.c_void_p, dp, dp, dp, C.c_size_t, dp, dp, C.POINTER(_RobustOpts),
                                          C.POINTER(_RobustPrior), C.c_int, C.POINTER(_Summary)]
        _lib = L
    return _lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


@dataclass
class Options:
    """GN fields of ct_icp::CTICPOptions (reference include/ct_icp/ct_icp.h:56-153)."""
    num_iters_icp: int = 5
    min_number_neighbors: int = 20
    max_number_neighbors: int = 20
    debug_print: bool = False
    max_dist_to_plane_ct_icp: float = 0.3
    threshold_orientation_norm: float = 1e-4

    def c(self) -> _Opts:
        return _Opts(self.num_iters_icp, self.min_number_neighbors, self.max_number_neighbors,
                     int(self.debug_print), self.max_dist_to_plane_ct_icp, self.threshold_orientation_norm)


@dataclass
class MotionPrior:
    beta_location_consistency: float = 0.001
    beta_constant_velocity: float = 0.001
    previous_begin_tr: np.ndarray = field(default_factory=lambda: np.zeros(3))
    previous_end_tr: np.ndarray = field(default_factory=lambda: np.zeros(3))

    def c(self) -> _Prior:
        p = _Prior()
        p.beta_location_consistency = self.beta_location_consistency
        p.beta_constant_velocity = self.beta_constant_velocity
        for i in range(3):
            p.previous_begin_tr[i] = float(self.previous_begin_tr[i])
            p.previous_end_tr[i] = float(self.previous_end_tr[i])
        return p


class Map:
    """ct_icp::MultipleResolutionVoxelMap restated (reference include/ct_icp/map.h)."""

    def __init__(self, resolutions=((0.2, 0.03, 50), (0.5, 0.1, 40), (1.5, 0.15, 40)), default_radius=0.8):
        arr = (_Res * len(resolutions))()
        for i, (r, d, m) in enumerate(resolutions):
            arr[i] = _Res(r, d, m)
        self._h = lib().orc_map_create(arr, len(resolutions), default_radius)
        if not self._h:
            raise ValueError("bad resolutions")
        self.resolutions = [tuple(r) for r in resolutions]
        self.default_radius = default_radius

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.orc_map_destroy(self._h)
            self._h = None

    def insert(self, xyz) -> np.ndarray:
        xyz = _f64(xyz).reshape(-1, 3)
        out = np.zeros(len(xyz), dtype=np.uint8)
        lib().orc_map_insert(self._h, _dp(xyz), len(xyz), out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out.astype(bool)

    def remove_far(self, location, distance):
        lib().orc_map_remove_far(self._h, _dp(_f64(location)), float(distance))

    def clear(self):
        lib().orc_map_clear(self._h)

    def num_points(self) -> int:
        return int(lib().orc_map_num_points(self._h))

    def num_voxels(self, res_index=0) -> int:
        return int(lib().orc_map_num_voxels(self._h, res_index))

    def export(self, res_index=0) -> np.ndarray:
        n = int(lib().orc_map_export(self._h, res_index, None, 0))
        out = np.zeros((n, 3))
        lib().orc_map_export(self._h, res_index, _dp(out), n)
        return out

    def search_params(self, radius=None):
        mid, nb, res = C.c_int(), C.c_int(), C.c_double()
        lib().orc_map_search_params(self._h, self.default_radius if radius is None else radius, C.byref(mid),
                                    C.byref(res), C.byref(nb))
        return mid.value, res.value, nb.value

    def radius_search(self, query, radius=0.0, max_num_neighbors=20, heap_mode=0) -> np.ndarray:
        q = _f64(query)
        out = np.zeros((max_num_neighbors, 3))
        n = lib().orc_map_radius_search(self._h, _dp(q), float(radius), max_num_neighbors, heap_mode, _dp(out))
        return out[:n].copy()

    def count_traffic(self, queries):
        q = _f64(queries).reshape(-1, 3)
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        for i in range(len(q)):
            lib().orc_map_count(self._h, _dp(q[i]), C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value


@dataclass
class Summary:
    success: bool
    num_residuals_used: int
    num_iters: int
    last_step_norm: float
    error_log: str
    t_accumulate: float
    t_solve: float
    t_update: float


def gn_accumulate(m: Map, raw, world, t, pose, t_begin_end, opts: Options, heap_mode=0, num_threads=1,
                  debug=False):
    raw, world, t = _f64(raw).reshape(-1, 3), _f64(world).reshape(-1, 3), _f64(t).ravel()
    n = len(t)
    pose, tbe = _f64(pose).ravel(), _f64(t_begin_end)
    A, b = np.zeros(144), np.zeros(12)
    nu = C.c_int(0)
    o = opts.c()
    if debug:
        nn = np.zeros(n, dtype=np.int32)
        normal, a2d, far = np.zeros((n, 3)), np.zeros(n), np.zeros((n, 3))
        used = np.zeros(n, dtype=np.uint8)
        lib().orc_gn_accumulate(m._h, _dp(raw), _dp(world), _dp(t), n, _dp(pose), _dp(tbe), C.byref(o), heap_mode,
                                num_threads, _dp(A), _dp(b), C.byref(nu), nn.ctypes.data_as(C.POINTER(C.c_int32)),
                                _dp(normal), _dp(a2d), _dp(far), used.ctypes.data_as(C.POINTER(C.c_uint8)))
        return A.reshape(12, 12), b, nu.value, dict(n_neighbors=nn, normal=normal, a2d=a2d, farthest=far,
                                                    used=used.astype(bool))
    lib().orc_gn_accumulate(m._h, _dp(raw), _dp(world), _dp(t), n, _dp(pose), _dp(tbe), C.byref(o), heap_mode,
                            num_threads, _dp(A), _dp(b), C.byref(nu), None, None, None, None, None)
    return A.reshape(12, 12), b, nu.value


def gn_solve_update(A, b, n_used, prior: MotionPrior | None, pose):
    A, b, pose = _f64(A).ravel().copy(), _f64(b).copy(), _f64(pose).ravel().copy()
    x = np.zeros(12)
    p = prior.c() if prior is not None else None
    nrm = lib().orc_gn_solve_update(_dp(A), _dp(b), int(n_used), C.byref(p) if p is not None else None, _dp(pose),
                                    _dp(x))
    return pose, x, nrm


def register_gn(m: Map, raw, world, t, pose, t_begin_end, opts: Options, prior: MotionPrior | None = None,
                heap_mode=0, num_threads=1):
    """DoRegisterGaussNewton (reference src/ct_icp/ct_icp.cpp:709-996). Returns (pose, world, Summary)."""
    raw, world, t = _f64(raw).reshape(-1, 3), _f64(world).reshape(-1, 3).copy(), _f64(t).ravel()
    pose, tbe = _f64(pose).ravel().copy(), _f64(t_begin_end)
    o = opts.c()
    p = prior.c() if prior is not None else None
    s = _Summary()
    rc = lib().orc_register_gn(m._h, _dp(raw), _dp(world), _dp(t), len(t), _dp(pose), _dp(tbe), C.byref(o),
                               C.byref(p) if p is not None else None, heap_mode, num_threads, C.byref(s))
    if rc != 0:
        raise ValueError(f"oracle: timestamp outside [t_begin, t_end] (rc={rc})")
    return pose, world, Summary(bool(s.success), s.num_residuals_used, s.num_iters, s.last_step_norm,
                                s.error_log.decode(), s.t_neighbors, s.t_solve, s.t_update)


def grid_sampling(raw, voxel_size) -> np.ndarray:
    raw = _f64(raw).reshape(-1, 3)
    out = np.zeros(len(raw), dtype=np.uint32)
    k = lib().orc_grid_sampling(_dp(raw), len(raw), float(voxel_size), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out[:k].copy()


# AdaptiveGridSamplingOptions::distance_voxel_size defaults (include/ct_icp/algorithm/sampling.h:18-25)
ADAPTIVE_DEFAULT_BANDS = ((0.5, 0.1), (2.0, 0.2), (4.0, 0.4), (8.0, 0.8), (16.0, 1.6), (200.0, -1.0))


def adaptive_sampling(raw, distance_voxel_size=ADAPTIVE_DEFAULT_BANDS, num_points_per_voxel=1, max_num_points=-1) -> np.ndarray:
    """AdaptiveSamplePointsInGrid (sampling.h:55-110); order: band, voxel (z, y, x), index."""
    raw = _f64(raw).reshape(-1, 3)
    bands = _f64(np.asarray(distance_voxel_size, dtype=np.float64).reshape(-1, 2))
    dist, size = np.ascontiguousarray(bands[:, 0]), np.ascontiguousarray(bands[:, 1])
    out = np.zeros(max(len(raw), 1), dtype=np.uint32)
    k = lib().orc_adaptive_sampling(_dp(raw), len(raw), int(num_points_per_voxel), int(max_num_points), len(dist), _dp(dist),
                                    _dp(size), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    if k == C.c_size_t(-1).value:
        raise ValueError("oracle: invalid adaptive sampling options")
    return out[:k].copy()


def transform_points(pose, t_begin_end, t, raw, num_threads=1, out=None) -> np.ndarray:
    raw, t = _f64(raw).reshape(-1, 3), _f64(t).ravel()
    pose, tbe = _f64(pose).ravel(), _f64(t_begin_end)
    if out is None:
        out = np.zeros_like(raw)
    lib().orc_transform_points(_dp(pose), _dp(tbe), _dp(t), _dp(raw), len(t), _dp(out), int(num_threads))
    return out


def neighborhood(points):
    pts = _f64(points).reshape(-1, 3)
    normal, a2d = np.zeros(3), C.c_double(0)
    ok = lib().orc_neighborhood(_dp(pts), len(pts), _dp(normal), C.byref(a2d))
    return bool(ok), normal, a2d.value


def quat_slerp(a, b, t):
    out = np.zeros(4)
    lib().orc_quat_slerp(_dp(_f64(a)), _dp(_f64(b)), float(t), _dp(out))
    return out


def quat_rotate(q, v):
    out = np.zeros(3)
    lib().orc_quat_rotate(_dp(_f64(q)), _dp(_f64(v)), _dp(out))
    return out


def quat_to_matrix(q):
    out = np.zeros(9)
    lib().orc_quat_to_matrix(_dp(_f64(q)), _dp(out))
    return out.reshape(3, 3)


def matrix_to_quat(R):
    out = np.zeros(4)
    lib().orc_matrix_to_quat(_dp(_f64(R).ravel()), _dp(out))
    return out


def sym_eigen3(Cm):
    ev, V = np.zeros(3), np.zeros(9)
    lib().orc_sym_eigen3(_dp(_f64(Cm).ravel()), _dp(ev), _dp(V))
    return ev, V.reshape(3, 3)


def ldlt_solve12(A, b):
    x = np.zeros(12)
    lib().orc_ldlt_solve12(_dp(_f64(A).ravel()), _dp(_f64(b)), _dp(x))
    return x


def alpha_timestamp(t, tb, te):
    return lib().orc_alpha_timestamp(float(t), float(tb), float(te))


def voxel_coord(p, size):
    return lib().orc_voxel_coord(float(p), float(size))


# ---------------------------------------------------------------------------------------------------------------------
# robust-loss (CERES-profile) route — DoRegisterCeres, reference src/ct_icp/ct_icp.cpp:457-707
# ---------------------------------------------------------------------------------------------------------------------
LOSS = {"STANDARD": 0, "CAUCHY": 1, "HUBER": 2, "TOLERANT": 3, "TRUNCATED": 4}


@dataclass
class RobustOptions:
    """Fields of ct_icp::CTICPOptions read by DoRegisterCeres (reference include/ct_icp/ct_icp.h:58-132); defaults
    are the reference's."""
    num_iters_icp: int = 5
    min_number_neighbors: int = 20
    max_number_neighbors: int = 20
    debug_print: bool = False
    max_num_residuals: int = -1
    loss_function: str = "CAUCHY"
    ls_max_num_iters: int = 1
    num_closest_neighbors: int = 1
    weight_alpha: float = 0.9
    weight_neighborhood: float = 0.1
    power_planarity: float = 2.0
    max_dist_to_plane_ct_icp: float = 0.3
    ls_sigma: float = 0.1
    ls_tolerant_min_threshold: float = 0.05
    threshold_orientation_norm: float = 1e-4
    threshold_translation_norm: float = 1e-3

    def c(self) -> _RobustOpts:
        return _RobustOpts(self.num_iters_icp, self.min_number_neighbors, self.max_number_neighbors,
                           int(self.debug_print), self.max_num_residuals, LOSS[self.loss_function],
                           self.ls_max_num_iters, self.num_closest_neighbors, self.weight_alpha,
                           self.weight_neighborhood, self.power_planarity, self.max_dist_to_plane_ct_icp,
                           self.ls_sigma, self.ls_tolerant_min_threshold, self.threshold_orientation_norm,
                           self.threshold_translation_norm)


@dataclass
class RobustPrior:
    """PreviousFrameMotionModel terms of the CERES route (reference src/ct_icp/motion_model.cpp:12-61)."""
    beta_location_consistency: float = 0.001
    beta_constant_velocity: float = 0.001
    beta_small_velocity: float = 0.0
    beta_orientation_consistency: float = 0.0
    previous_begin_tr: tuple = (0.0, 0.0, 0.0)
    previous_end_tr: tuple = (0.0, 0.0, 0.0)
    previous_end_quat: tuple = (0.0, 0.0, 0.0, 1.0)

    def c(self) -> _RobustPrior:
        p = _RobustPrior(self.beta_location_consistency, self.beta_constant_velocity, self.beta_small_velocity,
                         self.beta_orientation_consistency)
        for i in range(3):
            p.previous_begin_tr[i] = float(self.previous_begin_tr[i])
            p.previous_end_tr[i] = float(self.previous_end_tr[i])
        for i in range(4):
            p.previous_end_quat[i] = float(self.previous_end_quat[i])
        return p


def loss_evaluate(kind: str, sigma, tolerant_min, s):
    rho = np.zeros(3)
    lib().orc_loss_evaluate(LOSS[kind], float(sigma), float(tolerant_min), float(s), _dp(rho))
    return rho


def ct_point_to_plane(pose, alpha, raw, ref, normal, weight, jacobian=True):
    pose, raw, ref, normal = _f64(pose).ravel(), _f64(raw), _f64(ref), _f64(normal)
    r = np.zeros(1)
    J = np.zeros(12)
    lib().orc_ct_point_to_plane(_dp(pose), float(alpha), _dp(raw), _dp(ref), _dp(normal), float(weight), _dp(r),
                                _dp(J) if jacobian else None)
    return (r[0], J) if jacobian else r[0]


def robust_build(m: Map, raw, world, t, t_begin_end, opts: RobustOptions, heap_mode=0):
    """Residual blocks of one ICP iteration: dict(raw, ref, normal, weight, alpha, keypoint)."""
    raw, world, t = _f64(raw).reshape(-1, 3), _f64(world).reshape(-1, 3), _f64(t).ravel()
    cap = max(1, len(t) * opts.num_closest_neighbors)
    out = dict(raw=np.zeros((cap, 3)), ref=np.zeros((cap, 3)), normal=np.zeros((cap, 3)), weight=np.zeros(cap),
               alpha=np.zeros(cap), keypoint=np.zeros(cap, dtype=np.int32))
    o = opts.c()
    k = lib().orc_robust_build(m._h, _dp(raw), _dp(world), _dp(t), len(t), _dp(_f64(t_begin_end)), C.byref(o),
                               heap_mode, _dp(out["raw"]), _dp(out["ref"]), _dp(out["normal"]), _dp(out["weight"]),
                               _dp(out["alpha"]), out["keypoint"].ctypes.data_as(C.POINTER(C.c_int32)))
    return {key: v[:k].copy() for key, v in out.items()}


def _blocks(blocks):
    return [_f64(blocks[k]) for k in ("raw", "ref", "normal", "weight", "alpha")]


def robust_evaluate(blocks, opts: RobustOptions, prior: RobustPrior | None, pose, jacobian=True):
    """cost (= 1/2 sum rho) and, optionally, the loss-corrected J^T J and J^T r at `pose`."""
    raw, ref, normal, weight, alpha = _blocks(blocks)
    o = opts.c()
    p = prior.c() if prior is not None else None
    H, g = np.zeros(144), np.zeros(12)
    cost = lib().orc_robust_evaluate_fixed(_dp(raw), _dp(ref), _dp(normal), _dp(weight), _dp(alpha), len(weight),
                                           C.byref(o), C.byref(p) if p is not None else None,
                                           _dp(_f64(pose).ravel()), _dp(H) if jacobian else None,
                                           _dp(g) if jacobian else None)
    return (cost, H.reshape(12, 12), g) if jacobian else cost


def robust_solve_fixed(blocks, opts: RobustOptions, prior: RobustPrior | None, pose, max_num_iterations):
    """The inner ceres::Solve on fixed correspondences. Returns (pose, report dict)."""
    raw, ref, normal, weight, alpha = _blocks(blocks)
    o = opts.c()
    p = prior.c() if prior is not None else None
    pose = _f64(pose).ravel().copy()
    rep = _LmReport()
    term = lib().orc_robust_solve_fixed(_dp(raw), _dp(ref), _dp(normal), _dp(weight), _dp(alpha), len(weight),
                                        C.byref(o), C.byref(p) if p is not None else None, _dp(pose),
                                        int(max_num_iterations), C.byref(rep))
    return pose, dict(termination=term, initial_cost=rep.initial_cost, final_cost=rep.final_cost,
                      final_radius=rep.final_radius, iterations=rep.iterations,
                      successful=rep.num_successful_steps, unsuccessful=rep.num_unsuccessful_steps)


def register_robust(m: Map, raw, t, pose, t_begin_end, opts: RobustOptions, prior: RobustPrior | None = None,
                    heap_mode=0):
    """DoRegisterCeres (CONTINUOUS_TIME, POINT_TO_PLANE). Returns (pose, world, Summary)."""
    raw, t = _f64(raw).reshape(-1, 3), _f64(t).ravel()
    world = np.zeros_like(raw)
    pose, tbe = _f64(pose).ravel().copy(), _f64(t_begin_end)
    o = opts.c()
    p = prior.c() if prior is not None else None
    s = _Summary()
    rc = lib().orc_register_robust(m._h, _dp(raw), _dp(world), _dp(t), len(t), _dp(pose), _dp(tbe), C.byref(o),
                                   C.byref(p) if p is not None else None, heap_mode, C.byref(s))
    if rc != 0:
        raise ValueError(f"oracle: register_robust failed (rc={rc})")
    return pose, world, Summary(bool(s.success), s.num_residuals_used, s.num_iters, s.last_step_norm,
                                s.error_log.decode(), 0.0, 0.0, 0.0)


# ---------------------------------------------------------------------------------------------------------------------
# reference-shaped containers (ref_shaped.cpp): CPU-baseline honesty, SURVEY.md 8d
# ---------------------------------------------------------------------------------------------------------------------
class RefShapedMap:
    """One resolution level of an oracle Map re-bucketed into a node-based hash map of vectors of 80-byte records."""

    def __init__(self, m: Map, radius: float | None = None):
        map_id, self.resolution, self.nb = m.search_params(radius)
        self.radius = float(radius if radius else m.default_radius)
        pts = _f64(m.export(map_id))
        self._h = lib().orc_refshaped_create(_dp(pts), len(pts), float(self.resolution))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_refshaped_destroy(self._h)
            self._h = None

    def gn_accumulate(self, raw, world, t, pose, t_begin_end, opts: Options, num_threads=1):
        raw, world, t = _f64(raw).reshape(-1, 3), _f64(world).reshape(-1, 3), _f64(t).ravel()
        A, b = np.zeros(144), np.zeros(12)
        nu = C.c_int(0)
        o = opts.c()
        lib().orc_refshaped_gn_accumulate(self._h, _dp(raw), _dp(world), _dp(t), len(t), _dp(_f64(pose).ravel()),
                                          _dp(_f64(t_begin_end)), C.byref(o), self.radius, int(self.nb), int(num_threads),
                                          _dp(A), _dp(b), C.byref(nu))
        return A.reshape(12, 12), b, nu.value
