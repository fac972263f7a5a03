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
    assert ro.build() or (os.path.exists(ro._SO) and os.path.exists(ro._SO_ARMED))
    for armed in (False, True):
        L = ro.lib(armed)
        for sym in ("glue_odometry_options", "glue_odometry_set", "glue_odometry_start", "glue_odometry_register_frame", "glue_odometry_map_points",
                    "glue_odometry_is_armed"):
            assert hasattr(L, sym)
        assert L.glue_odometry_is_armed() == int(armed)
    with pytest.raises(KeyError):
        ro.RefOdometry(ro.CPU_MAP, no_such_option=1)
    with pytest.raises(RuntimeError, match="armed library"):        # the un-armed library has no arms to switch on
        L = ro.lib(False)
        h = L.glue_odometry_options(0)
        try:
            if L.glue_odometry_start(h, ro.GPU_MAP_ARMED) != 0:
                raise RuntimeError(L.glue_odometry_last_error().decode())
        finally:
            L.glue_odometry_destroy(h)
    import torch
    if not torch.cuda.is_available():
        for kind in (ro.GPU_MAP, ro.GPU_MAP_ARMED):
            with pytest.raises(ro.NoDevice, match="no CPU fallback"):
                ro.RefOdometry(kind)


def test_the_armed_source_carries_exactly_the_documented_insertions():
    """oracle/Makefile's GLUE_PATCH2 / GLUE_PATCH3 (what the armed library's odometry.cpp and map.cpp were compiled from) against the reference's
    files: every line of the reference survives in order, and the inserted lines are the include, the five arm statements with the block one of them
    opens closed again, and the factory line — nothing else."""
    import difflib
    import subprocess
    if not os.path.isdir(os.path.join(ro.REFERENCE_ROOT, "src", "ct_icp")):
        pytest.skip("/root/reference absent: the patched stream is made from it at build time")
    oracle_dir = os.path.dirname(os.path.abspath(ro.__file__))
    for target, ref_file, want in (("glue-print2", "src/ct_icp/odometry.cpp",
                                    ["#include <ct_icp/odometry_gpu_arm.h>", "GpuFrameTimeRange(", "GpuInitializeFrame(", "GpuUndistortFrame(", "}", "GpuTryRegister(", "GpuUpdateMap("]),
                                   ("glue-print3", "src/ct_icp/map.cpp", ["#include <ct_icp/gpu_map.h>", "gpu_map_options_from_yaml(node)"])):
        patched = subprocess.check_output(["make", "-s", "-C", oracle_dir, target], text=True).splitlines()
        original = open(os.path.join(ro.REFERENCE_ROOT, ref_file)).read().splitlines()
        added = []
        for tag, i1, i2, j1, j2 in difflib.SequenceMatcher(None, original, patched, autojunk=False).get_opcodes():
            assert tag in ("equal", "insert"), (target, tag, original[i1:i2], patched[j1:j2])
            if tag == "insert":
                added += [line.strip() for line in patched[j1:j2]]
        assert len(added) == len(want), added
        for line, fragment in zip(added, want):
            assert fragment in line, (line, fragment)


def test_the_arms_stand_down_on_the_reference_map():
    """The armed library on MULTI_RESOLUTION_VOXEL_HASHMAP: every arm answers "not mine" and the reference's own code runs — same poses, bit
    for bit, same map, as the library compiled from the untouched odometry.cpp (the random stream included: both shuffle the same g_)."""
    scans, _ = street_sequence(6, 250)
    kw = dict(solver=ro.CERES, ls_num_threads=1)
    plain, armed = ro.RefOdometry(ro.CPU_MAP, **kw), ro.RefOdometry(ro.CPU_MAP, armed_library=True, **kw)
    for raw, t in scans:
        a, b = plain.register_frame(raw, t, want_sampled=True, want_map_points=True), armed.register_frame(raw, t, want_sampled=True, want_map_points=True)
        assert np.array_equal(a["pose"], b["pose"]) and np.array_equal(a["sampled_raw"], b["sampled_raw"])
        for key in ("success", "sample_size", "number_of_residuals", "map_points"):
            assert a[key] == b[key]
    assert set(a["phase_ms"]) == set(ro.PHASES) and a["phase_ms"]["total"] > 0 and a["phase_ms"]["initialize_frame"] > 0


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


def _rows_as_set(points: np.ndarray):
    return set(map(bytes, np.ascontiguousarray(points)))


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["GN", "CERES"])
def test_armed_register_frame_against_the_unarmed_one(solver):
    """Odometry::RegisterFrame compiled with the four arms of integration/odometry_gpu_arm.h (oracle/_ref/libctgn_ref_odometry_armed.so, map kind
    GPU_MAP_ARMED) beside the un-armed library on the same GPU map and on the reference's CPU map, 64 frames, GN and CERES profile:
      * the first shuffle is reproduced: the sampled frame of EVERY frame is the same point set (corrected_points' raw points, bit for bit) —
        which also proves g_ stays in step (the next frame's shuffle starts from the stream the previous frame left);
      * same success flags, attempts, insertion decisions; the keypoints are another first-per-voxel choice from the same sampled frame (the
        reference takes them in robin_map's iteration order, the device in processing order), so poses agree as two valid registrations of
        the same frame do: inside the envelope the un-armed run itself keeps to the ground truth;
      * all_corrected_points = the reference's own transform applied to the armed run's poses;
      * with `frame_pipeline` off the armed library IS the un-armed one, bit for bit."""
    frames = 64
    scans, knots = street_sequence(frames, 500)
    kw = dict(solver=ro.GN if solver == "GN" else ro.CERES, ls_num_threads=1)
    plain = ro.RefOdometry(ro.GPU_MAP, **kw)
    off = ro.RefOdometry(ro.GPU_MAP, armed_library=True, **kw)
    armed = ro.RefOdometry(ro.GPU_MAP_ARMED, **dict(kw, ls_num_threads=4))     # threads to spare: the scan is uploaded beside the shuffle
    from ct_icp_amd import se3
    gap, ms = 0.0, {"plain": [], "armed": []}
    for j, (raw, t) in enumerate(scans):
        want_world = j in (0, 1, 2, 21, 40)
        a = plain.register_frame(raw, t, want_sampled=True, want_map_points=(j % 8 == 0))
        o = off.register_frame(raw, t, want_sampled=True, want_map_points=(j % 8 == 0))
        b = armed.register_frame(raw, t, want_sampled=True, want_world=want_world, want_map_points=(j % 8 == 0))
        assert np.array_equal(a["pose"], o["pose"]) and np.array_equal(a["sampled_raw"], o["sampled_raw"]) and a["map_points"] == o["map_points"]
        assert _rows_as_set(a["sampled_raw"]) == _rows_as_set(b["sampled_raw"]), (solver, j, len(a["sampled_raw"]), len(b["sampled_raw"]))
        for key in ("success", "points_added", "number_of_attempts", "num_corrected", "robust_level"):
            assert a[key] == b[key], (solver, j, key, a[key], b[key])
        assert a["success"] and a["sample_size"] == b["sample_size"]            # one keypoint per occupied voxel of the same sampled frame
        gap = max(gap, float(np.abs(a["pose"][[4, 5, 6, 11, 12, 13]] - b["pose"][[4, 5, 6, 11, 12, 13]]).max()))
        if want_world:
            tbe = (t.min(), t.max())
            want = se3.ct_transform(b["pose"], tbe, t, raw)
            assert np.abs(b["world"] - want).max() < 1e-8, (solver, j)
        if j >= 25:
            ms["plain"].append(a["milliseconds"]); ms["armed"].append(b["milliseconds"])
    truth = np.linalg.norm(syn.frame_pose14(knots, frames - 1)[11:14] - syn.frame_pose14(knots, 0)[4:7])
    err_plain = abs(np.linalg.norm(a["pose"][11:14]) - truth)
    err_armed = abs(np.linalg.norm(b["pose"][11:14]) - truth)
    assert err_armed < max(2.0 * err_plain, 0.3), (err_plain, err_armed)
    assert gap < 0.3, gap
    pa, pb = plain.map_points(), armed.map_points()
    assert abs(len(pa) - len(pb)) < 0.02 * len(pa)
    print(f"Odometry::RegisterFrame x {frames} [{solver}] armed vs un-armed: identical sampled frames; max translation gap {gap:.3f} m; end-point error "
          f"{err_plain:.3f} / {err_armed:.3f} m; {np.median(ms['plain']):.2f} -> {np.median(ms['armed']):.2f} ms per frame (median, frames 25+)")
