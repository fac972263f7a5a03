"""ctypes binding of libctgn.so (include/ctgn.h). The library is the product: there is no Python/CPU fallback —
loading fails loudly when the in-tree libctgn.so is missing, and every device entry point fails with
CTGN_ERR_NO_DEVICE when no gfx950 GPU is present."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CTGN_LIB_PATH") or os.path.join(_HERE, "libctgn.so")   # override: A/B timing of two builds

CTGN_MAX_RESOLUTIONS = 8
CTGN_SYSTEM_DOUBLES = 96
CTGN_MAX_NEIGHBORS = 32
CTGN_MIN_KEYPOINTS_USED = 100
CTGN_F32, CTGN_F64 = 0, 1

OK = 0
ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_OUT_OF_MEMORY = -1, -2, -3, -4
ERR_TIMESTAMP_RANGE, ERR_VOXEL_RANGE, ERR_UNSUPPORTED, ERR_SOLVER = -5, -6, -7, -8
LOSS = {"STANDARD": 0, "CAUCHY": 1, "HUBER": 2, "TOLERANT": 3, "TRUNCATED": 4}    # ct_icp::LEAST_SQUARES


class CtgnError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libctgn status {status}: {message}")
        self.status = status


class ResolutionParam(C.Structure):
    _fields_ = [("resolution", C.c_double), ("min_distance_between_points", C.c_double),
                ("max_num_points", C.c_int32), ("_pad", C.c_int32)]


class MapOptions(C.Structure):
    _fields_ = [("num_resolutions", C.c_int32), ("device", C.c_int32), ("default_radius", C.c_double),
                ("resolutions", ResolutionParam * CTGN_MAX_RESOLUTIONS), ("initial_voxel_capacity", C.c_uint64)]


class Options(C.Structure):
    _fields_ = [("num_iters_icp", C.c_int32), ("min_number_neighbors", C.c_int32),
                ("max_number_neighbors", C.c_int32), ("debug_print", C.c_int32),
                ("max_dist_to_plane_ct_icp", C.c_double), ("threshold_orientation_norm", C.c_double)]


class MotionPrior(C.Structure):
    _fields_ = [("beta_location_consistency", C.c_double), ("beta_constant_velocity", C.c_double),
                ("previous_begin_tr", C.c_double * 3), ("previous_end_tr", C.c_double * 3)]


class Summary(C.Structure):
    _fields_ = [("success", C.c_int32), ("num_residuals_used", C.c_int32), ("num_iters", C.c_int32),
                ("_pad", C.c_int32), ("duration_total_ms", C.c_double), ("duration_device_ms", C.c_double),
                ("last_step_norm", C.c_double), ("duration_init_ms", C.c_double), ("avg_duration_neighborhood_ms", C.c_double),
                ("avg_duration_solve_ms", C.c_double), ("avg_duration_iter_ms", C.c_double), ("error_log", C.c_char * 256)]


class RobustOptions(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("num_iters_icp", "min_number_neighbors", "max_number_neighbors", "debug_print",
                                         "max_num_residuals", "loss_function", "ls_max_num_iters",
                                         "num_closest_neighbors")] + \
               [(k, C.c_double) for k in ("weight_alpha", "weight_neighborhood", "power_planarity",
                                          "max_dist_to_plane_ct_icp", "ls_sigma", "ls_tolerant_min_threshold",
                                          "threshold_orientation_norm", "threshold_translation_norm")]


class RobustPrior(C.Structure):
    _fields_ = [("beta_location_consistency", C.c_double), ("beta_constant_velocity", C.c_double),
                ("beta_small_velocity", C.c_double), ("beta_orientation_consistency", C.c_double),
                ("previous_begin_tr", C.c_double * 3), ("previous_end_tr", C.c_double * 3),
                ("previous_end_quat", C.c_double * 4)]


class RobustReport(C.Structure):
    _fields_ = [("cost", C.c_double), ("radius", C.c_double), ("diff_rot_deg", C.c_double), ("diff_trans", C.c_double),
                ("num_residuals", C.c_int32), ("ls_iterations", C.c_int32), ("ls_accepted", C.c_int32),
                ("converged", C.c_int32), ("JtJ", C.c_double * 144), ("Jtr", C.c_double * 12),
                ("step_cycles", C.c_uint64 * 8)]


class AdaptiveSamplingOptions(C.Structure):
    _fields_ = [("num_points_per_voxel", C.c_int32), ("max_num_points", C.c_int32), ("num_bands", C.c_int32),
                ("reserved", C.c_int32), ("distance", C.c_double * 16), ("voxel_size", C.c_double * 16)]


class FrameOptions(C.Structure):
    _fields_ = [("frame_voxel_size", C.c_double), ("sample_voxel_size", C.c_double), ("max_num_keypoints", C.c_int32),
                ("override_timestamps", C.c_int32), ("override_timestamp", C.c_double), ("shuffle_seed", C.c_uint64)]


class FrameOutputs(C.Structure):
    _fields_ = [("all_world_base", C.c_void_p), ("all_world_stride_bytes", C.c_size_t), ("all_world_dtype", C.c_int32),
                ("_pad0", C.c_int32), ("sampled_indices", C.c_void_p), ("sampled_world_base", C.c_void_p),
                ("sampled_world_stride_bytes", C.c_size_t), ("sampled_world_dtype", C.c_int32), ("_pad1", C.c_int32),
                ("keypoint_indices", C.c_void_p), ("num_sampled", C.c_uint64), ("num_keypoints", C.c_uint64),
                ("keypoint_world_base", C.c_void_p), ("keypoint_world_stride_bytes", C.c_size_t), ("keypoint_world_dtype", C.c_int32),
                ("_pad2", C.c_int32), ("num_keypoint_candidates", C.c_uint64)]


class View(C.Structure):
    _fields_ = [("base", C.c_void_p), ("stride_bytes", C.c_size_t), ("dtype", C.c_int32), ("_pad", C.c_int32)]


# every symbol include/ctgn.h declares (+ the measurement / test hooks of ct_icp_amd/csrc/ctgn_internal.h): name -> (restype, argtypes)
_dp = C.POINTER(C.c_double)
_H = C.c_void_p
SYMBOLS = {
    "ctgn_abi_version": (C.c_int32, []),
    "ctgn_status_string": (C.c_char_p, [C.c_int]),
    "ctgn_last_error": (C.c_char_p, [_H]),
    "ctgn_map_options_default": (None, [C.POINTER(MapOptions)]),
    "ctgn_options_default": (None, [C.POINTER(Options)]),
    "ctgn_create": (C.c_int, [C.POINTER(MapOptions), C.POINTER(_H)]),
    "ctgn_destroy": (None, [_H]),
    "ctgn_map_insert": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.POINTER(C.c_uint8)]),
    "ctgn_map_remove_far": (C.c_int, [_H, _dp, C.c_double]),
    "ctgn_map_clear": (C.c_int, [_H]),
    "ctgn_map_num_points": (C.c_int, [_H, C.POINTER(C.c_uint64)]),
    "ctgn_map_num_voxels": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_uint64)]),
    "ctgn_map_search_params": (C.c_int, [_H, C.c_double, C.POINTER(C.c_int32), _dp, C.POINTER(C.c_int32)]),
    "ctgn_map_export": (C.c_int, [_H, C.c_int32, _dp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "ctgn_map_set_update_mode": (C.c_int, [_H, C.c_int32]),
    "ctgn_map_sync": (C.c_int, [_H]),
    "ctgn_map_radius_search": (C.c_int, [_H, _dp, C.c_size_t, C.c_double, C.c_int32, _dp, C.POINTER(C.c_int32)]),
    "ctgn_set_keypoints": (C.c_int, [_H, View, View, View, C.c_size_t]),
    "ctgn_set_rewind": (C.c_int, [_H, C.c_int32]),
    "ctgn_rewind_keypoints": (C.c_int, [_H]),
    "ctgn_solve": (C.c_int, [_H, _dp, _dp, C.POINTER(Options), C.POINTER(MotionPrior), C.POINTER(Summary)]),
    "ctgn_set_keypoints_sharded": (C.c_int, [_H, View, View, View, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_size_t)]),
    "ctgn_solve_sharded": (C.c_int, [_H, _dp, _dp, C.POINTER(Options), C.POINTER(MotionPrior), C.POINTER(Summary)]),
    "ctgn_dist_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "ctgn_dist_init": (C.c_int, [_H, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]),
    "ctgn_dist_shutdown": (C.c_int, [_H]),
    "ctgn_gn_iterate": (C.c_int, [_H, C.c_int32, C.c_int32]),
    "ctgn_get_world_points": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t]),
    "ctgn_grid_sampling": (C.c_int, [_H, View, C.c_size_t, C.c_double, C.POINTER(C.c_uint32), C.POINTER(C.c_size_t)]),
    "ctgn_adaptive_sampling_options_default": (None, [C.POINTER(AdaptiveSamplingOptions)]),
    "ctgn_adaptive_sampling": (C.c_int, [_H, View, C.c_size_t, C.POINTER(AdaptiveSamplingOptions), C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_size_t)]),
    "ctgn_transform_points": (C.c_int, [_H, View, View, C.c_size_t, _dp, _dp, C.c_void_p, C.c_size_t, C.c_int]),
    "ctgn_register": (C.c_int, [_H, View, C.c_void_p, C.c_size_t, C.c_int, View, C.c_size_t, _dp, _dp,
                                C.POINTER(Options), C.POINTER(MotionPrior), C.POINTER(Summary)]),
    "ctgn_frame_options_default": (None, [C.POINTER(FrameOptions)]),
    "ctgn_frame_register": (C.c_int, [_H, View, View, C.c_size_t, C.c_void_p, C.POINTER(FrameOptions), _dp, _dp, C.POINTER(Options),
                                      C.POINTER(MotionPrior), C.POINTER(RobustOptions), C.POINTER(RobustPrior),
                                      C.POINTER(FrameOutputs), C.POINTER(Summary)]),
    "ctgn_host_alloc": (C.c_int, [_H, C.c_size_t, C.POINTER(C.c_void_p)]),
    "ctgn_host_free": (C.c_int, [_H, C.c_void_p]),
    "ctgn_frame_stage": (C.c_int, [_H, View, View, C.c_size_t, C.POINTER(FrameOptions), _dp, _dp]),
    "ctgn_frame_begin": (C.c_int, [_H, View, View, C.c_size_t, C.c_void_p, C.POINTER(FrameOptions), _dp, _dp, C.POINTER(FrameOutputs)]),
    "ctgn_frame_try_register": (C.c_int, [_H, C.POINTER(FrameOptions), _dp, _dp, C.POINTER(Options), C.POINTER(MotionPrior),
                                          C.POINTER(RobustOptions), C.POINTER(RobustPrior), C.POINTER(FrameOutputs), C.POINTER(Summary)]),
    "ctgn_frame_undistort": (C.c_int, [_H, _dp, _dp, C.POINTER(FrameOutputs)]),
    "ctgn_frame_update_map": (C.c_int, [_H, _dp, C.c_double, C.c_int32, C.c_void_p]),
    "ctgn_frame": (C.c_int, [_H, View, View, C.c_size_t, C.c_void_p, C.POINTER(FrameOptions), _dp, _dp, C.POINTER(Options),
                             C.POINTER(MotionPrior), C.POINTER(RobustOptions), C.POINTER(RobustPrior), C.c_double,
                             C.POINTER(FrameOutputs), C.POINTER(Summary)]),
    "ctgn_robust_options_default": (None, [C.POINTER(RobustOptions)]),
    "ctgn_solve_robust": (C.c_int, [_H, _dp, _dp, C.POINTER(RobustOptions), C.POINTER(RobustPrior), C.POINTER(Summary)]),
    "ctgn_register_robust": (C.c_int, [_H, View, C.c_void_p, C.c_size_t, C.c_int, View, C.c_size_t, _dp, _dp,
                                       C.POINTER(RobustOptions), C.POINTER(RobustPrior), C.POINTER(Summary)]),
    "ctgn_robust_get_report": (C.c_int, [_H, C.POINTER(RobustReport)]),
    "ctgn_robust_get_blocks": (C.c_int, [_H, _dp, _dp, _dp, _dp, C.POINTER(C.c_int32), C.c_size_t]),
    "ctgn_gn_begin": (C.c_int, [_H, _dp, _dp, C.POINTER(Options), C.POINTER(MotionPrior)]),
    "ctgn_gn_accumulate": (C.c_int, [_H]),
    "ctgn_gn_system_device_ptr": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "ctgn_gn_set_system_buffer": (C.c_int, [_H, C.c_void_p]),
    "ctgn_gn_solve_update": (C.c_int, [_H]),
    "ctgn_gn_end": (C.c_int, [_H, _dp, C.POINTER(Summary)]),
    "ctgn_gn_done": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "ctgn_set_stream": (C.c_int, [_H, C.c_void_p]),
    "ctgn_get_stream": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "ctgn_set_debug": (C.c_int, [_H, C.c_int32]),
    "ctgn_get_debug": (C.c_int, [_H, C.POINTER(C.c_int32), _dp, _dp, _dp, C.POINTER(C.c_uint8), C.c_size_t]),
    "ctgn_get_system": (C.c_int, [_H, _dp]),
    "ctgn_count_traffic": (C.c_int, [_H, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ctgn_set_profiling": (C.c_int, [_H, C.c_int32]),
    "ctgn_kernel_timing": (C.c_int, [_H, _dp, C.POINTER(C.c_int32), C.c_int32]),
    "ctgn_kernel_timing_split": (C.c_int, [_H, _dp, C.POINTER(C.c_int32), C.c_int32]),
    "ctgn_set_variant": (C.c_int, [_H, C.c_int32]),
    "ctgn_set_ablation": (C.c_int, [_H, C.c_int32]),
    "ctgn_set_normals": (C.c_int, [_H, C.c_int32]),
    "ctgn_set_search_guess": (C.c_int, [_H, C.c_double]),
    "ctgn_set_ordering": (C.c_int, [_H, C.c_int32]),
    "ctgn_set_persistent": (C.c_int, [_H, C.c_int32]),
    "ctgn_set_pools": (C.c_int, [_H, C.c_int32]),
    "ctgn_traffic_counters": (C.c_int, [_H, C.POINTER(C.c_uint64), C.c_int32]),
    "ctgn_phase_cycles": (C.c_int, [_H, C.POINTER(C.c_uint64), C.c_int32]),
    "ctgn_test_sort_pairs": (C.c_int, [_H, C.POINTER(C.c_uint64), C.c_size_t, C.c_int32, C.c_int32, C.POINTER(C.c_uint32)]),
    "ctgn_test_compact": (C.c_int, [_H, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_size_t)]),
    "ctgn_last_upload_bytes": (C.c_int, [_H, C.POINTER(C.c_uint64)]),
    "ctgn_set_tuning": (C.c_int, [C.c_char_p, C.c_double]),
    "ctgn_path_counters": (C.c_int, [_H, C.POINTER(C.c_uint64)]),
    "ctgn_debug_pool_state": (C.c_int, [_H, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_size_t]),
    "ctgn_dist_overheads": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_double)]),
    "ctgn_measure_hbm": (C.c_int, [_H, C.c_uint64, C.c_int32, C.POINTER(C.c_double)]),
    "ctgn_wave_timeline": (C.c_int, [_H, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_size_t)]),
}

_lib = None


def lib() -> C.CDLL:
    """Load the in-tree libctgn.so. Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). ct_icp_amd has no CPU fallback.")
        # One process, one HIP runtime: PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64. If libctgn pulls in the
        # system copies first, a later `import torch` finds no GPU (measured on the MI355X box). Loading torch first makes
        # libctgn's dependencies resolve to the already-loaded copies. CTGN_NO_TORCH_PRELOAD=1 skips this (torch-free users).
        import sys
        if "torch" not in sys.modules and not os.environ.get("CTGN_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            if os.environ.get("CTGN_LIB_PATH") and not hasattr(L, name):
                continue               # an older build under A/B (the override only): hooks it predates stay unbound
            fn = getattr(L, name)      # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(handle, status: int):
    if status != OK:
        L = lib()
        msg = L.ctgn_last_error(handle).decode() if handle else L.ctgn_status_string(status).decode()
        if not msg:
            msg = L.ctgn_status_string(status).decode()
        raise CtgnError(status, msg)


def is_device_tensor(obj) -> bool:
    """A torch tensor living on a GPU (anything with data_ptr() and is_cuda): passed to the library as a device view."""
    return hasattr(obj, "data_ptr") and bool(getattr(obj, "is_cuda", False))


def tensor_view(t, comps: int = 3, offset_elems: int = 0) -> View:
    """ctgn_view of a device tensor of shape (N, >= comps) or (N,): row stride and dtype taken from the tensor."""
    import torch
    if t.dtype not in (torch.float32, torch.float64):
        raise TypeError("device views must be float32 or float64")
    if t.dim() == 2 and t.stride(1) != 1:
        raise ValueError("the components of a point must be contiguous")
    es = t.element_size()
    return View(t.data_ptr() + offset_elems * es, t.stride(0) * es, CTGN_F64 if t.dtype == torch.float64 else CTGN_F32, 0)
