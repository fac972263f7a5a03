"""CPU tests of the drop-in boundary: libctgn.so loads, exports every symbol include/ctgn.h declares, struct layouts
agree between the header and the ctypes mirror, and the product refuses to run without a gfx950 device."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import ct_icp_amd as cia
from ct_icp_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ctgn.h")
INTERNAL = os.path.join(ROOT, "ct_icp_amd", "csrc", "ctgn_internal.h")      # measurement / test hooks: exported, not part of the contract


def _declared_symbols(path=HEADER):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ctgn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = L.lib()
    declared = _declared_symbols()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ctgn.h but not exported by libctgn.so"
    internal = _declared_symbols(INTERNAL)
    for name in internal:
        assert hasattr(lib, name), f"{name} declared in ctgn_internal.h but not exported by libctgn.so"
    assert not set(internal) & set(declared)
    # the contract header carries no ablation / instrumentation / test entry points
    assert not [n for n in declared if re.search(r"ablation|phase_cycles|wave_timeline|traffic_counters|ctgn_test_|count_traffic|timing_split|set_variant|upload_bytes", n)]
    assert set(declared) | set(internal) == set(L.SYMBOLS), (set(declared) | set(internal)) ^ set(L.SYMBOLS)
    assert lib.ctgn_abi_version() == 6


def test_struct_layouts_match_the_header(tmp_path):
    """Compile a tiny C program against include/ctgn.h and compare sizeof/offsetof with the ctypes mirror."""
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ctgn.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                    'sizeof(ctgn_map_options),sizeof(ctgn_options),sizeof(ctgn_motion_prior),sizeof(ctgn_summary),'
                    'sizeof(ctgn_view),sizeof(ctgn_resolution_param),offsetof(ctgn_summary,error_log),'
                    'offsetof(ctgn_map_options,initial_voxel_capacity),sizeof(ctgn_adaptive_sampling_options),'
                    'offsetof(ctgn_adaptive_sampling_options,voxel_size),sizeof(ctgn_frame_outputs),offsetof(ctgn_frame_outputs,num_sampled),'
                    'offsetof(ctgn_frame_outputs,keypoint_world_dtype),sizeof(ctgn_frame_options));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = list(map(int, subprocess.check_output([str(exe)]).split()))
    want = [C.sizeof(L.MapOptions), C.sizeof(L.Options), C.sizeof(L.MotionPrior), C.sizeof(L.Summary), C.sizeof(L.View),
            C.sizeof(L.ResolutionParam), L.Summary.error_log.offset, L.MapOptions.initial_voxel_capacity.offset,
            C.sizeof(L.AdaptiveSamplingOptions), L.AdaptiveSamplingOptions.voxel_size.offset, C.sizeof(L.FrameOutputs),
            L.FrameOutputs.num_sampled.offset, L.FrameOutputs.keypoint_world_dtype.offset, C.sizeof(L.FrameOptions)]
    assert got == want


def test_defaults_match_the_reference():
    lib = L.lib()
    mo = L.MapOptions()
    lib.ctgn_map_options_default(C.byref(mo))
    assert mo.num_resolutions == 3 and mo.default_radius == 0.8          # map.h:117-125
    assert [(r.resolution, r.min_distance_between_points, r.max_num_points) for r in mo.resolutions[:3]] == \
           [(0.2, 0.03, 50), (0.5, 0.1, 40), (1.5, 0.15, 40)]
    o = L.Options()
    lib.ctgn_options_default(C.byref(o))
    d = cia.CTICPOptions()
    assert (o.num_iters_icp, o.min_number_neighbors, o.max_number_neighbors) == (d.num_iters_icp, d.min_number_neighbors, d.max_number_neighbors)
    assert (o.max_dist_to_plane_ct_icp, o.threshold_orientation_norm) == (d.max_dist_to_plane_ct_icp, d.threshold_orientation_norm)
    a = L.AdaptiveSamplingOptions()
    lib.ctgn_adaptive_sampling_options_default(C.byref(a))
    da = cia.AdaptiveGridSamplingOptions()                                # sampling.h:13-26
    assert (a.num_points_per_voxel, a.max_num_points, a.num_bands) == (da.num_points_per_voxel, da.max_num_points, 6)
    assert [(a.distance[j], a.voxel_size[j]) for j in range(6)] == [tuple(map(float, p)) for p in da.distance_voxel_size]
    assert cia.WPOINT3D_DTYPE.itemsize == 64 and cia.WPOINT3D_DTYPE.fields["world_point"][1] == 32   # types.h:35-41


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device behaviour")
def test_no_cpu_fallback_without_a_device():
    with pytest.raises(cia.CtgnError) as e:
        cia.GpuVoxelMap(cia.GpuVoxelMapOptions())                  # device 0 requested, none present
    assert e.value.status == L.ERR_NO_DEVICE
    m = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(device=-1))         # host-only mirror: map edits work, queries do not
    m.InsertPointCloud(np.random.default_rng(0).uniform(-3, 3, (500, 3)))
    with pytest.raises(cia.CtgnError) as e:
        m.ComputeNeighborhood([0, 0, 0], 20)
    assert e.value.status == L.ERR_NO_DEVICE
    with pytest.raises(cia.CtgnError) as e:
        cia.AdaptiveSamplePointsInGrid(m, np.ones((10, 3)))
    assert e.value.status == L.ERR_NO_DEVICE
    s = cia.GnSolver(m)
    with pytest.raises(cia.CtgnError) as e:
        s.set_keypoints(np.zeros((4, 3)), np.zeros((4, 3)), np.zeros(4))
    assert e.value.status == L.ERR_NO_DEVICE


def test_invalid_options_are_rejected():
    with pytest.raises(cia.CtgnError):
        cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(0.5, 0.1, 65)], device=-1))
    with pytest.raises(cia.CtgnError):
        cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(-1.0, 0.1, 20)], device=-1))
    with pytest.raises(RuntimeError, match="Unsupported Solver Type"):
        m = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(device=-1))
        cia.CT_ICP_Registration(cia.CTICPOptions(solver=cia.ROBUST)).Register(
            m, np.zeros(0, dtype=cia.WPOINT3D_DTYPE), cia.TrajectoryFrame())


def test_robust_route_fails_loudly_without_a_device():
    m = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(device=-1))
    with pytest.raises(cia.CtgnError) as e:
        cia.CT_ICP_Registration(cia.CTICPOptions(solver=cia.CERES)).Register(
            m, np.zeros(0, dtype=cia.WPOINT3D_DTYPE), cia.TrajectoryFrame())
    assert e.value.status == L.ERR_NO_DEVICE
    with pytest.raises(RuntimeError, match="CONTINUOUS_TIME"):
        cia.CT_ICP_Registration(cia.CTICPOptions(solver=cia.CERES, distance="POINT_TO_POINT")).Register(
            m, np.zeros(0, dtype=cia.WPOINT3D_DTYPE), cia.TrajectoryFrame())
