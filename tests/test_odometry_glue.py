"""The drop-in at the level north_star names: `ct_icp::Odometry::RegisterFrame` — the REFERENCE'S OWN odometry loop, compiled from
/root/reference/src/ct_icp/odometry.cpp where it lies (oracle/Makefile target `odometry`, integration/glue_odometry.cpp around it) —
constructed twice from the same OdometryOptions, once with `MultipleResolutionVoxelMap::Options` (MULTI_RESOLUTION_VOXEL_HASHMAP: the
reference's map, its CPU solver loops) and once with `GpuVoxelMap::Options` (GPU_VOXEL_HASHMAP: integration/gpu_map.h over libctgn.so;
`Register` reaches the GPU through the two one-line arms of integration/gn_gpu_arm.h). Everything else — InitializeMotion, the two shuffles
and sub_sample_frame, grid_sampling, AssessRegistration, the robust retry loop, undistortion, UpdateMap's insertion policy — is the
reference's code on both sides, fed the same scans.

CPU tier: the library builds and loads, the GPU map refuses without a device (no CPU fallback), the reference's loop runs on its own map.
GPU tier: >= 60 frames, solver GN and the CERES profile: identical success flags, keypoint counts, residual counts and insertion decisions,
trajectories equal to 1e-9 (the two solvers differ in rounding only: reduction order of the normal equations, ~1e-13 per frame)."""
import os

import numpy as np
import pytest

from ct_icp_amd import synthetic as syn
from oracle import ref_odometry as ro

pytestmark = pytest.mark.skipif(not ro.available(), reason="oracle/_ref/libctgn_ref_odometry.so absent and no /root/reference to build it from")


def street_sequence(frames: int, azimuth_steps: int, seed: int = 10, ramp_frames: int = 20):
    """Config-B generator: HDL-64E pattern over the procedural street; the vehicle pulls away from rest, because the
    reference's odometry starts from the identity with nothing to extrapolate (odometry.cpp:276-300)."""
    scene = syn.street_scene(max(300.0, frames * 1.2 + 60.0), seed=seed)
    dirs, rel_t = syn.lidar_pattern("hdl64", azimuth_steps=azimuth_steps, azimuth_offset=np.pi)     # the sweep starts / ends at the rear, as KITTI's does
    knots = syn.driving_trajectory(frames + 1, seed=seed, start_x=20.0, ramp_frames=ramp_frames, centered=True)
    scans = []
    for j in range(frames):
        sc = syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, j), 0.1 * j, 0.1 * (j + 1), noise=0.02, seed=1000 * seed + j,
                               use_torch=True)
        scans.append((sc.raw, sc.t))
    return scans, knots


def test_library_builds_and_the_gpu_map_refuses_without_a_device():
    assert ro.build() or os.path.exists(ro._SO)
    L = ro.lib()
    for sym in ("glue_odometry_options", "glue_odometry_set", "glue_odometry_start", "glue_odometry_register_frame", "glue_odometry_map_points"):
        assert hasattr(L, sym)
    with pytest.raises(KeyError):
        ro.RefOdometry(ro.CPU_MAP, no_such_option=1)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(ro.NoDevice, match="no CPU fallback"):
            ro.RefOdometry(ro.GPU_MAP)


def test_reference_odometry_runs_on_its_own_map():
    scans, knots = street_sequence(6, 250)
    od = ro.RefOdometry(ro.CPU_MAP, solver=ro.CERES, ls_num_threads=1)
    rs = [od.register_frame(raw, t, want_map_points=True) for raw, t in scans]
    assert all(r["success"] for r in rs)
    assert rs[0]["sample_size"] == 0 and rs[1]["sample_size"] > 300               # frame 0 is only inserted (odometry.cpp:405)
    assert rs[-1]["map_points"] > rs[0]["map_points"] > 1000
    assert len(od.map_points()) == rs[-1]["map_points"]
    # the loop tracks the vehicle: displacement of the end pose between frames 1 and 5 against ground truth (world frame = first scan's)
    moved = np.linalg.norm(rs[5]["pose"][11:14] - rs[1]["pose"][11:14])
    truth = np.linalg.norm(syn.frame_pose14(knots, 5)[11:14] - syn.frame_pose14(knots, 1)[11:14])
    assert abs(moved - truth) < 0.1, (moved, truth)


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["GN", "CERES"])
def test_reference_register_frame_gives_the_same_trajectory_on_the_gpu_map(solver):
    frames = 64
    scans, knots = street_sequence(frames, 500)
    kw = dict(solver=ro.GN if solver == "GN" else ro.CERES, ls_num_threads=1)
    cpu = ro.RefOdometry(ro.CPU_MAP, **kw)
    gpu = ro.RefOdometry(ro.GPU_MAP, **kw)
    worst, worst_world = 0.0, 0.0
    for j, (raw, t) in enumerate(scans):
        want_world = j % 16 == 5
        a = cpu.register_frame(raw, t, want_world=want_world, want_map_points=(j % 8 == 0))
        b = gpu.register_frame(raw, t, want_world=want_world, want_map_points=(j % 8 == 0))
        for key in ("success", "points_added", "sample_size", "number_of_residuals", "number_of_attempts", "num_corrected", "map_points"):
            assert a[key] == b[key], (solver, j, key, a[key], b[key])
        assert np.array_equal(a["initial_pose"], b["initial_pose"]) or np.abs(a["initial_pose"] - b["initial_pose"]).max() < 1e-9
        worst = max(worst, float(np.abs(a["pose"] - b["pose"]).max()))
        assert worst < 1e-9, (solver, j, worst)
        if want_world:
            worst_world = max(worst_world, float(np.abs(a["world"] - b["world"]).max()))
            assert worst_world < 1e-8, (solver, j, worst_world)
    # and the run is a registration, not two identical failures: the reference's loop followed the vehicle on both maps
    moved = np.linalg.norm(b["pose"][11:14])
    truth = np.linalg.norm(syn.frame_pose14(knots, frames - 1)[11:14] - syn.frame_pose14(knots, 0)[4:7])
    assert abs(moved - truth) < 1.5, (moved, truth)
    pa, pb = cpu.map_points(), gpu.map_points()
    assert len(pa) == len(pb)
    ka = np.lexsort(np.round(pa, 6).T)
    kb = np.lexsort(np.round(pb, 6).T)
    assert np.abs(pa[ka] - pb[kb]).max() < 1e-8
    print(f"Odometry::RegisterFrame x {frames} [{solver}]: max |pose(cpu map) - pose(gpu map)| = {worst:.2e}, world points {worst_world:.2e}, "
          f"{len(pa)} map points on both")
