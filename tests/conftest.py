import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) — run with `pytest -m gpu` on the GPU box")


def _gpu_present() -> bool:
    """True when libctgn.so loads and ctgn_create finds a device. The product fails loudly without one (CTGN_ERR_NO_DEVICE);
    this only decides whether gpu-marked tests are collected as runnable or skipped in a plain `pytest tests` on a CPU box."""
    try:
        import ct_icp_amd as cia
        cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(1.0, 0.1, 20)], default_radius=1.0))
        return True
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if not any("gpu" in it.keywords for it in items):
        return
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        return                                   # `-m gpu` was asked for explicitly: run them, fail loudly without a device
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no gfx950 device / libctgn.so: GPU parity tests need the MI355X box (`pytest -m gpu`)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "gn_small.npz")))


@pytest.fixture(scope="session")
def golden_frame_steps():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "frame_steps_small.npz")))


@pytest.fixture(scope="session")
def golden_robust():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "robust_small.npz")))


def _sphere_pattern(n_az, n_el, el_lim=75.0):
    el = np.radians(np.linspace(-el_lim, el_lim, n_el))
    az = np.linspace(0, 2 * np.pi, n_az, endpoint=False)
    dirs = np.stack([np.outer(np.cos(az), np.cos(el)), np.outer(np.sin(az), np.cos(el)),
                     np.outer(np.ones_like(az), np.sin(el))], -1).reshape(-1, 3)
    return dirs, np.repeat(np.arange(n_az) / n_az, n_el)


@pytest.fixture(scope="session")
def box_case():
    """Config-A-like: closed box + spheres, 0.5 m map, radius 0.8 (125 voxels / query), k = 20."""
    from ct_icp_amd import se3, synthetic as syn
    scene = syn.box_scene(8.0, n_spheres=3, seed=20240901)
    dirs, rel_t = _sphere_pattern(300, 48)
    knots = np.zeros((8, 7))
    for j in range(8):
        knots[j, :4] = se3.quat_from_rotvec(np.array([0.02 * j, -0.015 * j, 0.05 * j]))
        knots[j, 4:] = [0.12 * j, 0.06 * j, 0.02 * j]
    scans = [syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, j), 0.1 * j, 0.1 * (j + 1), max_range=40.0,
                               min_range=0.3, noise=0.01, seed=10 + j) for j in range(7)]
    return dict(scene=scene, knots=knots, scans=scans, resolutions=[(0.5, 0.05, 20)], default_radius=0.8)


@pytest.fixture(scope="session")
def street_case():
    """Config-B-like (driving profile, reduced azimuth resolution): 0.8 m map x 30 pts, radius 0.75 (27 voxels)."""
    from ct_icp_amd import synthetic as syn
    scene = syn.street_scene(200.0, seed=1)
    dirs, rel_t = syn.lidar_pattern("hdl64", azimuth_steps=700)
    knots = syn.driving_trajectory(12, seed=0, start_x=20.0)
    scans = [syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, j), 0.1 * j, 0.1 * (j + 1), noise=0.02,
                               seed=100 + j) for j in range(11)]
    return dict(scene=scene, knots=knots, scans=scans, resolutions=[(0.8, 0.1, 30)], default_radius=0.75)


@pytest.fixture(scope="session")
def nclt_case():
    """Config-C-like (BASELINE.json configs[2]): HDL-32E pattern, jittery Segway-like motion, the NCLT profile's three-resolution
    map (0.5 / 1 / 2 m x 30 pts; default radius 0.8 => the 0.5 m level, 125 voxels per query), min_number_neighbors 10,
    20 iterations, <= 1500 keypoints (reference config/odometry/nclt_config.yaml:19-104)."""
    from ct_icp_amd import synthetic as syn
    scene = syn.street_scene(150.0, seed=2, half_width=(7.0, 9.0))
    dirs, rel_t = syn.lidar_pattern("hdl32", azimuth_steps=900)
    knots = syn.driving_trajectory(10, dt=0.1, speed=2.0, yaw_rate=0.3, height=1.0, jitter=0.02, seed=2, start_x=20.0)
    scans = [syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, j), 0.1 * j, 0.1 * (j + 1), max_range=60.0, noise=0.02,
                               seed=200 + j) for j in range(9)]
    return dict(scene=scene, knots=knots, scans=scans, resolutions=[(0.5, 0.08, 30), (1.0, 0.08, 30), (2.0, 0.08, 30)],
                default_radius=0.8)


@pytest.fixture(scope="session")
def config_a_case():
    """BASELINE.json configs[0]: the reference's synthetic courtyard (tests/golden/config_a.npz, generated from the reference's
    own scene file by tests/golden/make_config_a.py) with the odometry values of config/synthetic_ct_icp_config.yaml as SURVEY.md
    section 8d reads them: voxel_size 0.1, sample_voxel_size 0.5, map 0.5 m x 20 pts, min distance 0.05, radius 0.8 (125 voxels)."""
    from ct_icp_amd import se3, synthetic as syn
    d = dict(np.load(os.path.join(ROOT, "tests", "golden", "config_a.npz")))
    offs = np.concatenate([[0], np.cumsum(d["counts"])])
    scans = []
    for j in range(len(d["counts"])):
        raw, t = d["raw"][offs[j]:offs[j + 1]], d["t"][offs[j]:offs[j + 1]]
        keep = np.sort(syn.grid_sample_indices(raw, 0.1))                         # odometry voxel_size
        raw, t = raw[keep], t[keep]
        scans.append(syn.Scan(raw=raw, t=t, world_gt=se3.ct_transform(d["pose_gt"][j], d["tbe"][j], t, raw), pose_gt=d["pose_gt"][j],
                              t_begin_end=d["tbe"][j]))
    return dict(scans=scans, resolutions=[(0.5, 0.05, 20)], default_radius=0.8, sample_voxel_size=0.5)


def build_maps(case, n_map_frames, with_gpu=False, device=0, subsample=None):
    """Insert the first n_map_frames scans (ground-truth world points) into an oracle map and, optionally, a
    GpuVoxelMap, through their own insert rules."""
    from oracle import oracle as orc
    from ct_icp_amd import synthetic as syn
    om = orc.Map(resolutions=case["resolutions"], default_radius=case["default_radius"])
    gm = None
    if with_gpu:
        import ct_icp_amd as cia
        gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(*r) for r in case["resolutions"]],
                                                    default_radius=case["default_radius"], device=device))
    for j in range(n_map_frames):
        pts = case["scans"][j].world_gt
        if subsample:
            pts = pts[syn.grid_sample_indices(case["scans"][j].raw, subsample)]
        om.insert(pts)
        if gm is not None:
            gm.InsertPointCloud(pts)
    return om, gm


def load_frame_steps_module():
    """tests/golden/make_golden_frame_steps.py as a module (its DictMap is the plain-Python model of the map rules)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_frame_steps", os.path.join(ROOT, "tests", "golden", "make_golden_frame_steps.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


MAP_FUZZ_LEVELS = ((0.5, 0.05, 4), (1.0, 0.2, 20), (0.3, 0.0, 2))


def map_fuzz_steps(seed, min_d, steps=12):
    """Random insert batches (clustered: full voxels, near-duplicates below the minimum distance, sign flips across the axis
    planes) with an eviction centre every third step."""
    rng = np.random.default_rng(100 + seed)
    centre = np.zeros(3)
    for step in range(steps):
        centre = centre + rng.normal(0, 1.5, 3)
        pts = centre + rng.normal(0, 2.0, (300, 3)) * np.array([1.0, 1.0, 0.3])
        pts[::7] = pts[1::7][:len(pts[::7])] + rng.normal(0, 0.3 * max(min_d, 0.01), (len(pts[::7]), 3))
        pts[::11, rng.integers(0, 3)] *= -1.0
        yield pts, (centre.copy() if step % 3 == 2 else None)
