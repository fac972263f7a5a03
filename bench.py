#!/usr/bin/env python3
"""bench.py — registered keypoints/s per Gauss–Newton iteration of the CT-ICP registration path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under torch.distributed.run, one
rank per GPU. A "step" is ONE GN iteration (neighbour search + covariance/normal + residual/Jacobian + reduction + 12x12 solve +
pose update, ct_icp.cpp:745-981) over the resident keypoint batch, and the K timed steps are FRESH solves of the profile's iteration
budget (driving profile: 5, config/odometry/driving_config.yaml; ct_icp.cpp:745): per solve the uploaded world points are put back
(device-to-device), `gn_begin`, 5 iterations. The first search of every solve therefore runs on the radius alone and the library
plans for a 5-iteration solve — what a frame pays. Round 2's figure (iterations 170.. of ONE running loop, every search bounded by a
converged solve's k-th distances) rides along as `roofline.steady_state_*`. Rank 0 prints one JSON line.

N = 1  workload B2 = BASELINE.json configs[1]: a KITTI-00-like HDL-64E sweep (~132 k returns, every return a keypoint: the
       throughput regime of SURVEY.md section 8d) registered against a STEADY-STATE driving-profile local map (0.8 m x 30 pts,
       radius 0.75 => 27 voxels per query, k = 20, everything within the 100 m eviction radius of an open residential scene:
       ~3.3 x 10^5 voxels, searched level ~270 MB — larger than L2 + Infinity Cache). The line also carries `workloads`: B1 (the
       reference's keypoint count), C (configs[2]: NCLT / HDL-32E profile) and D (configs[3] on one GPU), each with its own
       roofline, parity and cpu_baseline. `--workload X` makes X the headline; `--workload B2-small` is round 1's 5 MB map.
N > 1  config D (BASELINE.json configs[3]) STRONG scaling: ONE Ouster-128-style scan; every rank hands the library the whole scan,
       the library sorts it by home voxel and keeps the rank's contiguous chunk (ctgn_set_keypoints_sharded; map replicated), one
       ncclAllReduce of the 96-double packed system per iteration issued by the library (ctgn_solve_sharded's launch sequence).
       The same scan on one GPU and a weak-scaling line (one B2-small sweep per rank) ride along.
Inputs are synthetic (no dataset on the box) and resident in HBM before the timed region. Clocks: the GPU leaves its idle power
state only under sustained load (profiles/r01_launch_series.txt), so `CLOCK_WARM` untimed iterations of the same loop run
immediately before the W warm-up and the K timed steps — same launch sequence, same resident data, no upload in between.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEARCH_WAVES_PER_SIMD = 3     # k_accumulate_rows: __launch_bounds__(256, 3), profiles/r05_kernel_resources.txt
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
B_KP, B_SLOT, B_PT = 32, 16, 24    # algorithmic bytes: keypoint record, hash slot, map point (FP64 xyz storage)
CLOCK_WARM = 150               # untimed iterations that bring the clocks up before the W + K steps of the contract


def _rocprof():
    import shutil
    return shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)


SEARCH_KERNELS = ("k_accumulate_rows", "k_pool_check", "k_accumulate_lane")      # what one neighbour search of an iteration launches


def collect_pmc(args, workload=None, groups=None, steps=None, timeout: int = 150):
    """Counters of the neighbour search, collected LIVE by rocprofv3 passes (`--kernel-trace --pmc ...`, one pass per counter group,
    no other trace domain — MI355X_MICROARCH.md "rocprofv3 PMC slots") over a short inner run of this same script and workload.
    Per ITERATION: the counters of every search-kernel launch (k_accumulate_rows; from the third search of a solve on a 400 k-keypoint
    frame also k_pool_check in front of it) summed over the second half of the inner run's iterations and divided by their number (one
    k_residual_reduce launch = one iteration). Returns (dict of per-iteration means, source / reason string)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = _rocprof()
    if exe is None:
        return {}, "rocprofv3 not found"
    workload = workload or args.workload
    if groups is None:
        groups = [["FETCH_SIZE"], ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"],
                  ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAVES"],
                  ["SQ_INSTS_VALU", "SQ_INSTS_SALU"]]
    optional = ("SQ_INSTS", "TCC_")
    vals = {}
    t_all = time.perf_counter()
    for counters in groups:
        if time.perf_counter() - t_all > 2.5 * timeout:          # a profiler that crawls must not hold the bench line up
            return vals, "rocprofv3 passes took too long: the remaining counter groups were skipped"
        out_dir = tempfile.mkdtemp(prefix="ctgn_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", out_dir, "-o", "pmc", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--inner", "--steps", str(steps or min(args.steps, 20)), "--warmup", "0", "--workload", workload,
               "--variant", str(args.variant), "--map-frames", str(args.map_frames), "--order", args.order, "--d-sweeps", str(args.d_sweeps),
               "--d-radius", str(args.d_radius)]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=timeout, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, check=False)
            rows = []                                           # (dispatch id, kernel kind, counter, value)
            for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    name = r.get("Kernel_Name", "")
                    kind = "search" if any(k in name for k in SEARCH_KERNELS) else "residual" if "k_residual_reduce" in name else None
                    if kind:
                        rows.append((int(r.get("Dispatch_Id", 0) or 0), kind, r.get("Counter_Name"), float(r["Counter_Value"])))
            rows.sort(key=lambda x: x[0])
            res_ids = sorted({d for d, kind, _, _ in rows if kind == "residual"})
            if len(res_ids) < 2:
                if all(c.startswith(optional) for c in counters):
                    continue
                return vals, f"rocprofv3 --pmc {' '.join(counters)}: no rows"
            first = res_ids[len(res_ids) // 2 - 1]                 # iterations of the second half: clocks and caches are warm
            n_iter = len(res_ids) - len(res_ids) // 2
            for c in counters:
                tot = sum(v for d, kind, cn, v in rows if kind == "search" and cn == c and first < d <= res_ids[-1])
                if tot == 0 and not any(cn == c for _, _, cn, _ in rows):
                    if c.startswith(optional):
                        continue
                    return vals, f"rocprofv3 --pmc {c}: no rows"
                vals[c] = tot / n_iter
        except Exception as e:        # noqa: BLE001 — measurement nicety: never fail the bench over it
            return vals, f"rocprofv3 --pmc {' '.join(counters)}: {type(e).__name__}"
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    return vals, (f"live: rocprofv3 --kernel-trace --pmc, {len(groups)} passes ({' | '.join(' '.join(g) for g in groups)}), per search iteration "
                  "(all search-kernel launches of an iteration), means over the second half of the inner run")


def arrays_sha256(d) -> str:
    """SHA-256 over names, dtypes, shapes and bytes of a dict of arrays (sorted by name)."""
    import hashlib
    hs = hashlib.sha256()
    for k in sorted(d):
        a = np.ascontiguousarray(d[k])
        hs.update(f"{k}|{a.dtype.str}|{a.shape}|".encode())
        hs.update(a.tobytes())
    return hs.hexdigest()


def _generator_fingerprint() -> str:
    """What the cached inputs depend on besides their parameters: the generator's source."""
    import hashlib
    with open(os.path.join(ROOT, "ct_icp_amd", "synthetic.py"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def cache_load(path: str, params: dict):
    """A cached input set, or None when the file is absent, was written for other generator parameters (or another generator), or does not
    hash to what it says it holds: .bench_cache/ is git-ignored and travels to the GPU box with the push, so nothing in it is trusted —
    a mismatch means the inputs are ray-cast again."""
    if not os.path.exists(path):
        return None
    try:
        d = np.load(path, allow_pickle=False)
        want = json.dumps(dict(params, generator=_generator_fingerprint()), sort_keys=True)
        if "__params__" not in d.files or "__sha256__" not in d.files or str(d["__params__"]) != want:
            return None
        out = {k: d[k] for k in d.files if not k.startswith("__")}
        return out if arrays_sha256(out) == str(d["__sha256__"]) else None
    except Exception:        # noqa: BLE001 — a damaged cache file is a cache miss
        return None


def cache_save(path: str, out: dict, params: dict) -> None:
    try:
        np.savez(path, __params__=np.array(json.dumps(dict(params, generator=_generator_fingerprint()), sort_keys=True)),
                 __sha256__=np.array(arrays_sha256(out)), **out)
    except OSError:
        pass


def make_inputs(rank: int, map_frames: int, cache_dir: str = os.path.join(ROOT, ".bench_cache")):
    """Deterministic config-B inputs: map insert list (world points of `map_frames` preceding sweeps after the 0.5 m
    frame grid) + the sweep to register. Cached as .npz because ray-casting 21 sweeps in NumPy takes ~30 s."""
    from ct_icp_amd import synthetic as syn
    tag = f"ctgn_bench_B_v4_r{rank}_m{map_frames}.npz"
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, tag)
    params = dict(workload="B", rank=rank, map_frames=map_frames, scene="street_scene(400, seed 1)", pattern="hdl64", trajectory_seed=0, noise=0.02)
    cached = cache_load(path, params)
    if cached is not None:
        return cached
    scene = syn.street_scene(400.0, seed=1)
    dirs, rel_t = syn.lidar_pattern("hdl64")
    knots = syn.driving_trajectory(map_frames + 2, seed=0, start_x=20.0)
    map_pts = []
    for j in range(map_frames):
        sc = syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, j), 0.1 * j, 0.1 * (j + 1), noise=0.02, seed=100 + j)
        map_pts.append(sc.world_gt[syn.grid_sample_indices(sc.raw, 0.5)])
    j = map_frames
    # every rank registers the same frame geometry with its own noise realisation (a different shard of a denser scan)
    sc = syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, j), 0.1 * j, 0.1 * (j + 1), noise=0.02,
                           seed=1000 + 17 * rank)
    out = dict(map_points=np.concatenate(map_pts), map_counts=np.array([len(m) for m in map_pts]), raw=sc.raw, t=sc.t,
               pose_gt=sc.pose_gt, tbe=sc.t_begin_end, prev_b=knots[j - 1, 4:7], prev_e=knots[j, 4:7])
    cache_save(path, out, params)
    return out


def make_inputs_large(rank: int, cache_dir: str = os.path.join(ROOT, ".bench_cache")):
    """Workload B2: an open residential scene (ct_icp_amd.synthetic.suburb_scene) whose steady-state local map — every surface within
    the driving profile's 100 m eviction radius, sampled directly instead of ray-casting the few hundred sweeps that would have
    accumulated it, then passed through the map's own insert rule — has ~3.3 x 10^5 voxels of 0.8 m (searched level ~260 MB), and
    one ray-cast HDL-64E sweep of it to register. The sweep is cached (.npz, ~4 MB: ray-casting 133 k rays against 8 k primitives in
    NumPy takes ~30 s); the ~16 M map candidates are regenerated every run (~10 s)."""
    from ct_icp_amd import synthetic as syn
    scene = syn.suburb_scene(seed=7, n_buildings=220, n_trees=4000)
    knots = syn.driving_trajectory(3, seed=0, start_x=20.0)
    map_points = syn.sample_scene_surfaces(scene, knots[1, 4:7], radius=100.0, density=70.0, noise=0.02, seed=5)
    tag = f"ctgn_bench_B2L_v2_r{rank}.npz"
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, tag)
    params = dict(workload="B2", rank=rank, scene="suburb_scene(seed 7, 220 buildings, 4000 trees)", pattern="hdl64", trajectory_seed=0, noise=0.02,
                  scan_seed=1000 + 17 * rank)
    scan = cache_load(path, params)
    if scan is None:
        dirs, rel_t = syn.lidar_pattern("hdl64")
        sc = syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, 1), 0.1, 0.2, noise=0.02, seed=1000 + 17 * rank)
        scan = dict(raw=sc.raw, t=sc.t, pose_gt=sc.pose_gt, tbe=sc.t_begin_end)
        cache_save(path, scan, params)
    return dict(map_points=map_points, prev_b=knots[0, 4:7], prev_e=knots[1, 4:7], **scan)


def make_inputs_nclt(rank: int, cache_dir: str = os.path.join(ROOT, ".bench_cache")):
    """Workload C = BASELINE.json configs[2]: a synthetic HDL-32E (32 beams, +10.67 .. -30.67 deg) on a Segway-like platform (2 m/s,
    0.3 rad/s yaw, roll / pitch jitter) in a narrow campus street; the map is what the eight preceding sweeps left in the NCLT profile's
    three-resolution map (reference config/odometry/nclt_config.yaml:41-53; default radius 0.8 => the 0.5 m level, 125 voxels per query).
    Cached: nine small ray-cast sweeps."""
    from ct_icp_amd import synthetic as syn
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"ctgn_bench_C_v2_r{rank}.npz")
    params = dict(workload="C", rank=rank, scene="street_scene(150, seed 2, half width 7-9)", pattern="hdl32 x 1800", trajectory_seed=2, noise=0.02)
    cached = cache_load(path, params)
    if cached is not None:
        return cached
    scene = syn.street_scene(150.0, seed=2, half_width=(7.0, 9.0))
    dirs, rel_t = syn.lidar_pattern("hdl32", azimuth_steps=1800)
    knots = syn.driving_trajectory(10, dt=0.1, speed=2.0, yaw_rate=0.3, height=1.0, jitter=0.02, seed=2, start_x=20.0)
    map_pts = []
    for j in range(8):
        sc = syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, j), 0.1 * j, 0.1 * (j + 1), max_range=60.0, noise=0.02, seed=200 + j)
        map_pts.append(sc.world_gt[syn.grid_sample_indices(sc.raw, 0.5)])                   # odometry voxel_size 0.5 (nclt_config.yaml:32)
    sc = syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, 8), 0.8, 0.9, max_range=60.0, noise=0.02, seed=208 + 17 * rank)
    out = dict(map_points=np.concatenate(map_pts), raw=sc.raw, t=sc.t, pose_gt=sc.pose_gt, tbe=sc.t_begin_end, prev_b=knots[7, 4:7], prev_e=knots[8, 4:7])
    cache_save(path, out, params)
    return out


def make_inputs_ouster(sweeps: int = 8, density: float = 100.0, radius: float = 100.0):
    """Workload D = BASELINE.json configs[3], SURVEY.md section 8d: an Ouster-128-style scan (128 beams +-22.5 deg, 2048 columns x `sweeps`
    accumulated sub-sweeps = 2.1 M rays at 8) RAY-CAST against the residential scene of workload B2 with the sensor moving through the
    frame, keypoints = every return that survives a 0.05 m grid; the map is the scene's surfaces within `radius` sampled directly (what
    ~50 dense frames from many viewpoints leave behind) and passed through the map's own insert rule at {0.5 m, 40 pts, 0.03 m}:
    ~0.8 M voxels x 40 slots x 24 B — well beyond L2 + Infinity Cache. The ray-casting runs as broadcast torch operations on the GPU
    (ct_icp_amd.synthetic.raycast_torch: seconds instead of minutes), so nothing is cached."""
    from ct_icp_amd import synthetic as syn
    scene = syn.suburb_scene(seed=7, n_buildings=220, n_trees=4000)
    knots = syn.driving_trajectory(3, seed=0, start_x=20.0)
    dirs, rel_t = syn.lidar_pattern("os128", sweeps=sweeps)
    sc = syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, 1), 0.1, 0.2, noise=0.02, seed=3, use_torch=True)
    keep = syn.grid_sample_indices(sc.raw, 0.05)
    map_points = syn.sample_scene_surfaces(scene, knots[1, 4:7], radius=radius, density=density, noise=0.02, seed=6)
    return dict(map_points=map_points, raw=sc.raw[keep], t=sc.t[keep], pose_gt=sc.pose_gt, tbe=sc.t_begin_end, prev_b=knots[0, 4:7], prev_e=knots[1, 4:7],
                returns=len(sc.raw), rays=len(dirs))


# per workload: (resolutions [(size, min distance, points per voxel)], default radius, GN iterations per frame, min_number_neighbors)
PROFILES = {
    "B2": ([(0.8, 0.1, 30)], 0.75, 5, 20),         # config/odometry/driving_config.yaml:18-90 with solver GN
    "B2-small": ([(0.8, 0.1, 30)], 0.75, 5, 20),
    "B1": ([(0.8, 0.1, 30)], 0.75, 5, 20),
    "C": ([(0.5, 0.05, 30), (1.0, 0.1, 30), (2.0, 0.2, 30)], 0.8, 20, 10),      # config/odometry/nclt_config.yaml:41-53,69,83
    "D": ([(0.5, 0.03, 40)], 0.8, 5, 20),
}
NAMES = {
    "B2": "config B2 = BASELINE.json configs[1]: synthetic HDL-64E sweep (KITTI-00-like, every return a keypoint) over the steady-state "
          "driving-profile map of an open residential scene: 0.8 m x 30 pts, radius 0.75 (27 voxels), k=20, everything within the 100 m eviction radius",
    "B2-small": "config B2-small: the same sweep regime over round 1's 20-frame street-canyon map (L2-resident best case)",
    "B1": "config B1: street-canyon sweep and map, keypoints = 1.5 m grid of the 0.5 m-subsampled frame (the reference's keypoint count; latency regime)",
    "C": "config C = BASELINE.json configs[2]: synthetic HDL-32E on a Segway-like platform, NCLT profile: map 0.5 / 1 / 2 m x 30 pts, radius 0.8 => 0.5 m "
         "level, 125 voxels per query, min_number_neighbors 10, 20 GN iterations per frame, <= 1500 keypoints (0.8 m grid)",
    "D": "config D = BASELINE.json configs[3]: Ouster-128-style 2.1 M-ray scan ray-cast against the residential scene, keypoints = 0.05 m grid of the "
         "returns, map {0.5 m, 40 pts, 0.03 m} of every surface within 100 m (> L2 + Infinity Cache), radius 0.8 (125 voxels), k=20",
}


def build_workload(name, rank, world, args, cia, syn, se3):
    """Inputs + map + keypoints of one workload. Returns a dict the measuring functions share."""
    res_list, radius, ipf, min_nb = PROFILES[name]
    if name == "D":
        inp = make_inputs_ouster(sweeps=args.d_sweeps, radius=args.d_radius)      # ONE scan for all ranks (strong scaling)
    elif name == "B2":
        inp = make_inputs_large(rank)
    elif name == "C":
        inp = make_inputs_nclt(rank)
    else:
        inp = make_inputs(rank, args.map_frames)
    # what was measured, whatever it was loaded from: the scan to register, its poses and the map's insert list
    inputs_sha16 = arrays_sha256({k: inp[k] for k in ("map_points", "raw", "t", "pose_gt", "tbe")})[:16]
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(*r) for r in res_list], default_radius=radius, device=args.local_rank))
    for s0 in range(0, len(inp["map_points"]), 2_000_000):
        gm.InsertPointCloud(inp["map_points"][s0:s0 + 2_000_000])
    gm.Sync()
    pose0 = syn.perturb_pose(inp["pose_gt"], 0.003, 0.03, seed=4)
    raw, t = inp["raw"], inp["t"]
    if name == "B1":                                   # the reference's two-stage grid sampling (odometry.cpp:349,538)
        sel = syn.grid_sample_indices(raw, 0.5)
        sel = sel[syn.grid_sample_indices(raw[sel], 1.5)]
        raw, t = raw[sel], t[sel]
    elif name == "C":                                  # voxel_size 0.5, then the 0.8 m keypoint grid capped at max_num_keypoints 1500
        sel = syn.grid_sample_indices(raw, 0.5)
        sel = sel[syn.grid_sample_indices(raw[sel], 0.8)][:1500]
        raw, t = raw[sel], t[sel]
    world0 = se3.ct_transform(pose0, inp["tbe"], t, raw)
    mm = cia.PreviousFrameMotionModel()
    mm.previous_frame = cia.TrajectoryFrame.from_pose14(np.concatenate([[0, 0, 0, 1], inp["prev_b"], [0, 0, 0, 1], inp["prev_e"]]), 0, 0)
    searched = gm.SearchParamsFromRadiusSearch()
    lv = searched[0]
    level_mb = (gm.NumVoxels(lv) * res_list[lv][2] * 24 + (1 << int(np.ceil(np.log2(max(gm.NumVoxels(lv), 1) * 4)))) * 16) / 1e6
    return dict(name=name, inp=inp, gm=gm, res_list=res_list, radius=radius, ipf=ipf, min_nb=min_nb, raw=raw, t=t, world0=world0, pose0=pose0,
                mm=mm, level=lv, level_mb=level_mb, nb=searched[2], inputs_sha16=inputs_sha16)


def oracle_map(W):
    from oracle import oracle as orc
    om = orc.Map(resolutions=W["res_list"], default_radius=W["radius"])
    for s0 in range(0, len(W["inp"]["map_points"]), 2_000_000):
        om.insert(W["inp"]["map_points"][s0:s0 + 2_000_000])
    return om


class Runner:
    """The GN loops of one workload on one rank: fresh solves (the headline) and the long running loop (steady state)."""

    def __init__(self, W, args, cia, torch, dist, sharded):
        self.W, self.args, self.cia, self.torch, self.dist = W, args, cia, torch, dist
        self.sh = None
        if sharded:
            from ct_icp_amd.distributed import ShardedGnSolver
            self.sh = ShardedGnSolver(W["gm"], library_collective=not args.torch_collective)
            self.solver = self.sh.solver
        else:
            self.solver = cia.GnSolver(W["gm"])
        self.solver.set_variant(args.variant)
        self.solver.set_ordering({"auto": -1, "off": 0, "on": 1}[args.order])
        self.solver.set_ablation(args.ablate)
        self.solver.set_rewind(True)

    def options(self, iters):   # threshold 0: no early stop, exactly `iters` GN iterations
        return self.cia.CTICPOptions(solver=self.cia.GN, num_iters_icp=iters, min_number_neighbors=self.W["min_nb"], threshold_orientation_norm=0.0,
                                     debug_print=False)

    def sync_all(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def iterate(self, k):
        """k GN iterations of the running loop, enqueued without synchronising."""
        if self.sh is not None and self.args.torch_collective:
            from ct_icp_amd.distributed import allreduce_system
            for _ in range(k):
                self.solver.gn_accumulate()
                allreduce_system(self.sh.system)
                self.solver.gn_solve_update()
        else:
            self.solver.gn_iterate(k, sharded=self.sh is not None)

    def fresh(self, k, ipf=None):
        """k GN iterations as FRESH solves of the profile's budget (5 for the driving profile, ct_icp.cpp:745): world points back to the
        uploaded ones (device-to-device), gn_begin, `ipf` iterations — so the first search of every solve is unbounded and the library
        plans for a `ipf`-iteration solve. Nothing synchronises; the last solve may be shorter so that exactly k iterations run."""
        W, ipf = self.W, ipf or self.W["ipf"]
        left = k
        while left > 0:
            m = min(ipf, left)
            self.solver.rewind()
            self.solver.gn_begin(W["pose0"], W["inp"]["tbe"], self.options(ipf), W["mm"])
            self.iterate(m)
            left -= m
        return k

    def upload(self):
        """Keypoints resident on the device. Sharded config D: every rank hands the library the WHOLE scan; it sorts by home voxel and keeps
        this rank's contiguous chunk (ctgn_set_keypoints_sharded)."""
        W = self.W
        if "shard" in W:
            idx = self.solver.set_keypoints_sharded(W["all_raw"], W["all_world0"], W["all_t"], *W["shard"])
            W["upload_h2d_bytes"] = self.solver.last_upload_bytes()         # 24 B x scan + 56 B x chunk per rank (56 B x scan before round 5)
            W["raw"], W["t"], W["world0"] = W["all_raw"][idx], W["all_t"][idx], W["all_world0"][idx]
        else:
            self.solver.set_keypoints(W["raw"], W["world0"], W["t"])

    def settle_clocks(self, body, limit_s=2.0):
        """Untimed chunks of `body` (four solves' worth each) until two consecutive chunks take the same time to 3 %, at most `limit_s`:
        the GPU leaves its idle clocks only under sustained load, and how long that takes depends on what the process — and the box — did
        before (a run that started right behind another process's teardown measured 0.46 ms per step over K = 5 after the fixed 150
        warm-up iterations). Single-process runs only (every rank would have to agree on the chunk count). Returns the iterations run."""
        if self.dist is not None:
            return 0
        chunk = 4 * self.W["ipf"]
        self.solver.set_profiling(False)
        prev, good, done = None, 0, 0
        t_start = time.perf_counter()
        while time.perf_counter() - t_start < limit_s:
            self.sync_all()
            t0 = time.perf_counter()
            body(chunk)
            self.sync_all()
            dt = time.perf_counter() - t0
            done += chunk
            if prev is not None and abs(dt - prev) <= 0.03 * prev:
                good += 1
                if good >= 2:
                    break
            else:
                good = 0
            prev = dt
        return done

    def timed(self, body, steps, warmup, clock_warm, profile=True):
        """clock_warm + warmup untimed iterations of `body`, then exactly `steps` timed ones bracketed by barrier + synchronize.
        Returns (seconds, summary, HIP-event kernel times: all, first-of-solve, bounded). profile=False: no event pairs (and a small
        frame then runs as ONE persistent launch per solve, which has no separate search kernel to bracket)."""
        s = self.solver
        s.set_profiling(False)
        body(clock_warm)
        body(warmup)
        self.sync_all()
        s.set_profiling(profile)                               # HIP-event pair around every neighbour-search launch from here on
        s.kernel_timing(reset=True)
        self.sync_all()
        t0 = time.perf_counter()
        body(steps)
        self.sync_all()
        dt = time.perf_counter() - t0
        summ = s.gn_end()[1]
        kern = s.kernel_timing(reset=False)
        first, later = s.kernel_timing_split(reset=True)
        s.set_profiling(False)
        return dt, summ, kern, first, later

    def requested(self, steady_after=10):
        """Bytes the search kernel asks the memory system for (SURVEY.md 8d pricing: 32 B per keypoint + 16 B per hash probe it issues
        + 24 B per map point it streams), counted by the instrumented instantiation: first search of a solve, the later ones of a
        `ipf`-iteration solve, and one iteration of the steady state."""
        s, W = self.solver, self.W
        n = len(W["t"])
        s.set_variant(3)
        out = {}
        s.rewind()
        s.gn_begin(W["pose0"], W["inp"]["tbe"], self.options(W["ipf"]), W["mm"])
        s.traffic_counters(reset=True)
        self.iterate(1)
        p0, q0 = s.traffic_counters(reset=True)
        s.phase_cycles(reset=True)
        self.iterate(W["ipf"] - 1)
        p1, q1 = s.traffic_counters(reset=True)
        certified = int(s.phase_cycles(reset=True)[8])      # keypoints whose neighbours came from their pool (no search), summed over these launches
        s.gn_end()
        s.rewind()
        s.gn_begin(W["pose0"], W["inp"]["tbe"], self.options(steady_after + 1), W["mm"])
        self.iterate(steady_after)
        s.traffic_counters(reset=True)
        self.iterate(1)
        ps, qs = s.traffic_counters(reset=True)
        s.gn_end()
        s.set_variant(self.args.variant)
        later_n = max(1, W["ipf"] - 1)
        for key, (p_, q_, launches) in {"first": (p0, q0, 1), "later": (p1, q1, later_n), "steady": (ps, qs, 1)}.items():
            out[key] = {"bytes_per_launch": (n * B_KP * launches + p_ * B_SLOT + q_ * B_PT) / launches, "probes_per_keypoint": p_ / launches / n,
                        "points_per_keypoint": q_ / launches / n}
        out["later"]["pool_certified_frac"] = certified / later_n / n
        return out

    def close(self):
        if self.sh is not None:
            self.sh.close()


def roofline_object(W, n_kp, timing, req, traffic, pmc, pmc_src, alg_all, sweep, variant):
    """`roofline` of the bench line for the neighbour-search kernel over the timed fresh solves."""
    dt, summ, (kern_ms, kern_launches), (first_ms, first_n), (later_ms, later_n) = timing
    ipf = W["ipf"]
    t_k = kern_ms * 1e-3
    if req is not None:
        alg = (req["first"]["bytes_per_launch"] + (ipf - 1) * req["later"]["bytes_per_launch"]) / ipf
    else:
        alg = alg_all
    achieved = alg / t_k / 1e9 if t_k > 0 else 0.0
    roof = {"bound": "latency/issue", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": pmc_src,
            "achieved_definition": f"mean over the {ipf} searches of a fresh solve of: algorithmic bytes the kernel requests per launch (32 B/keypoint + 16 B per "
                                   "hash probe it issues + 24 B per map point it streams, counted by the instrumented instantiation) / HIP-event launch "
                                   "time; the first search of a solve is bounded by the radius only, the second by the carried-over k-th distance; from the "
                                   "third on (frames of >= 8192 keypoints) a keypoint whose pool certifies its neighbours fetches the pool's points "
                                   "(24 B each, counted) and is not searched: the requested bytes fall faster than the time, so this fraction FALLS "
                                   "when pools make the step faster (DESIGN.md section 17)",
            "frac_survey_8d_all_sweep_voxels": (alg_all / t_k / 1e9 / HBM_PEAK_GBS) if t_k > 0 else None,
            "frac_survey_8d_note": "SURVEY.md 8(d)'s literal per-keypoint figure (every voxel of the sweep, every point in them) over the same time: "
                                   "above 1 because the kernel culls what that formula charges - context, not evidence",
            "hbm_counter_gbs": (traffic / t_k / 1e9) if traffic and t_k > 0 else None,
            "hbm_counter_frac": (traffic / t_k / 1e9 / HBM_PEAK_GBS) if traffic and t_k > 0 else None,
            "kernel": "k_accumulate_rows (voxel-hash neighbour search + k-nearest selection)" if variant != 1 else "k_accumulate_lane",
            "kernel_ms_avg": kern_ms, "kernel_launches": kern_launches,
            "alg_bytes_per_launch": alg, "alg_bytes_per_keypoint": alg / n_kp,
            "alg_bytes_per_launch_all_sweep_voxels": alg_all,     # SURVEY.md 8d's formula with every voxel of the sweep: what an unculled search reads
            "voxels_in_sweep_per_keypoint": sweep[0] / n_kp, "voxels_occupied_per_keypoint": sweep[1] / n_kp, "points_in_sweep_per_keypoint": sweep[2] / n_kp}
    for key, ms, nl in (("first_iteration", first_ms, first_n), ("later_iterations", later_ms, later_n)):
        r = req[{"first_iteration": "first", "later_iterations": "later"}[key]] if req is not None else None
        roof[key] = {"kernel_ms": ms, "launches": nl}
        if r is not None and ms > 0:
            g = r["bytes_per_launch"] / (ms * 1e-3) / 1e9
            roof[key].update({"requested_bytes_per_launch": r["bytes_per_launch"], "requested_bytes_per_keypoint": r["bytes_per_launch"] / n_kp,
                              "probes_issued_per_keypoint": r["probes_per_keypoint"], "points_streamed_per_keypoint": r["points_per_keypoint"],
                              "achieved": g, "frac": g / HBM_PEAK_GBS})
            if "pool_certified_frac" in r:
                roof[key]["pool_certified_frac"] = r["pool_certified_frac"]
    if "SQ_WAVE_CYCLES" in pmc and pmc["SQ_WAVE_CYCLES"] > 0:
        wc = pmc["SQ_WAVE_CYCLES"]
        roof.update({"wait_frac": pmc["SQ_WAIT_ANY"] / wc, "valu_busy": pmc["SQ_ACTIVE_INST_VALU"] / wc, "waves_per_launch": pmc.get("SQ_WAVES")})
        # the roof that binds this kernel (HBM does not): share of a SIMD's time with a vector-ALU instruction of one of its resident waves in
        # flight = per-wave VALU-active share x the waves the kernel keeps resident per SIMD (compiled for 3: 168 registers, 12.9 KB of LDS per wave)
        roof["valu_issue_frac"] = min(1.0, SEARCH_WAVES_PER_SIMD * pmc["SQ_ACTIVE_INST_VALU"] / wc)
        roof["valu_issue_frac_definition"] = (f"SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x {SEARCH_WAVES_PER_SIMD} resident waves per SIMD: the kernel is VALU-issue + "
                                              "latency bound; `frac` (requested bytes against the HBM peak) is the contract's figure, this is the roof that binds")
        if "SQ_WAIT_INST_ANY" in pmc:
            roof.update({"issue_stall_frac": pmc["SQ_WAIT_INST_ANY"] / wc, "active_frac": pmc["SQ_ACTIVE_INST_ANY"] / wc})
        if "SQ_INSTS_VALU" in pmc:
            roof["valu_instructions_per_keypoint"] = pmc["SQ_INSTS_VALU"] / n_kp
    if pmc.get("TCC_HIT_sum", 0) + pmc.get("TCC_MISS_sum", 0) > 0:
        roof["tcc_hit_rate"] = pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])
    return roof


def measure_workload(W, args, cia, torch, dist, sharded, steps, warmup, clock_warm, want_steady=True, want_parity=True, want_cpu=True, cpu_seconds=10.0,
                     world=1, rank=0, pmc_live=False, register_extras=False):
    """Everything the bench line says about one workload: fresh-solve throughput (the headline definition), first / later iteration
    split, requested bytes + roofline, steady-state figure, pose parity against the oracle on the timed inputs, CPU baseline."""
    from ct_icp_amd import se3
    R = Runner(W, args, cia, torch, dist, sharded)
    R.upload()
    n_kp = len(W["t"])
    # two passes of the same loop: the VALUE from a pass without event pairs (what a caller runs; a small frame goes through the
    # one-launch persistent kernel there), the search-kernel times of the roofline from a pass with a HIP-event pair around every
    # search launch (always the three-launch loop)
    settled = R.settle_clocks(R.fresh)
    plain = R.timed(R.fresh, steps, warmup, clock_warm, profile=False)
    timing = R.timed(R.fresh, steps, warmup, 0, profile=True)
    prof_dt = timing[0]
    timing = (plain[0], plain[1]) + tuple(timing[2:])
    dt, summ = timing[0], timing[1]
    last = steps % W["ipf"] or W["ipf"]
    assert args.ablate or (summ.success and summ.num_iters == last), summ
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        nn = torch.tensor([n_kp], dtype=torch.float64, device="cuda")
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        total_kp = int(nn.item())
    else:
        total_kp = n_kp
    timing = (dt,) + tuple(timing[1:])
    out = {"value": total_kp * steps / dt, "unit": "keypoints/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "keypoints": n_kp,
           "keypoints_total": total_kp, "iterations_per_solve": W["ipf"], "solves_timed": (steps + W["ipf"] - 1) // W["ipf"],
           "clock_settle_iterations": settled,
           "n_used_last_iter": int(summ.num_residuals_used), "first_iteration_ms": None, "later_iteration_ms": None,
           "map_points": int(W["gm"].NumPoints()), "map_voxels": int(W["gm"].NumVoxels(W["level"])), "searched_level_mb": W["level_mb"],
           "frames_per_sec_equiv": 1.0 / (dt / steps * W["ipf"]) if dt > 0 else None}
    if "upload_h2d_bytes" in W:
        out["sharded_upload"] = {"h2d_bytes_this_rank": int(W["upload_h2d_bytes"]), "h2d_bytes_whole_scan_7_arrays": 56 * int(total_kp),
                                 "note": "per rank: 24 B x scan (world points: the home-voxel order every rank must agree on) + 56 B x chunk"}
    if R.sh is not None and not args.torch_collective:
        # what a sharded iteration costs before it does any work, so that a scaling record explains itself: the bare 768-byte all-reduce and
        # the chain of the iteration's five launches with nothing to do (collective: every rank measures)
        try:
            ar_us, chain_us = R.solver.dist_overheads(1000)
            out["sharded_overheads"] = {"allreduce_us": ar_us, "five_launch_overhead_us": chain_us, "iteration_us": dt / steps * 1e6,
                                        "what": "allreduce_us = one ncclAllReduce(sum, 96 f64) on the handle's stream, mean of 1000 back to back; "
                                                "five_launch_overhead_us = one sharded iteration (search, residual, reduce, all-reduce, solve) whose "
                                                "kernels return at once; iteration_us = the timed iteration of this line"}
        except Exception as e:        # noqa: BLE001 — a measurement nicety: never fail the bench over it
            out["sharded_overheads"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    # ---- steady state: ONE running loop (round 2's headline): every search after the first is bounded, the pose has converged
    steady = None
    if want_steady and not args.inner:
        begun = []

        def running(k):
            if not begun:
                R.solver.rewind()
                R.solver.gn_begin(W["pose0"], W["inp"]["tbe"], R.options(clock_warm + warmup + steps), W["mm"])
                begun.append(True)
            R.iterate(k)
        sdt, ssumm, skern, _, _ = R.timed(running, steps, warmup, clock_warm)
        if dist is not None:
            tt = torch.tensor([sdt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sdt = float(tt.item())
        steady = {"value": total_kp * steps / sdt, "ms_per_step": sdt / steps * 1e3, "kernel_ms_avg": skern[0], "kernel_launches": skern[1],
                  "what": f"iterations {clock_warm + warmup + 1}..{clock_warm + warmup + steps} of ONE running GN loop (round 2's headline): every search bounded by a "
                          "converged solve's k-th distances — the search kernel's best case, not what a frame pays"}
    # ---- accounting (untimed)
    R.solver.rewind()
    sweep = R.solver.count_traffic()                       # (voxels probed, voxels hit, points inside them) at the uploaded world points
    alg_all = n_kp * B_KP + sweep[0] * B_SLOT + sweep[2] * B_PT
    req = None
    if args.variant == 0 and not args.ablate and not args.inner and W["nb"] in (1, 2):
        req = R.requested()
    pmc, pmc_src = ({}, "not collected for this workload (the headline workload's passes: profiles/)")
    if pmc_live:
        if W["name"] == args.workload:
            pmc, pmc_src = collect_pmc(args)
        else:                                              # a sub-workload: the HBM and wait counters only (three short inner runs)
            pmc, pmc_src = collect_pmc(args, workload=W["name"], steps=10,
                                       groups=[["FETCH_SIZE"], ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"],
                                               ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAVES"]])
    traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0 if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc else None
    roof = roofline_object(W, n_kp, timing, req, traffic, pmc, pmc_src, alg_all, sweep, args.variant)
    prof_step = prof_dt / steps * 1e3                      # ms per step of the pass that carried the event pairs (this rank)
    out["first_iteration_ms"] = roof["first_iteration"]["kernel_ms"] + (prof_step - roof["kernel_ms_avg"])
    out["later_iteration_ms"] = roof["later_iterations"]["kernel_ms"] + (prof_step - roof["kernel_ms_avg"])
    out["ms_per_step_three_launch_loop_with_event_pairs"] = prof_step
    out["iteration_split_note"] = ("search-kernel HIP-event time of that kind of iteration + the rest of a mean iteration (residual + solve kernels, "
                                   "hand-overs) of the pass with event pairs; `value` / `ms_per_step` come from the pass without them")
    if steady is not None:
        if req is not None and steady["kernel_ms_avg"] > 0:
            g = req["steady"]["bytes_per_launch"] / (steady["kernel_ms_avg"] * 1e-3) / 1e9
            steady.update({"requested_bytes_per_keypoint": req["steady"]["bytes_per_launch"] / n_kp, "achieved": g, "frac": g / HBM_PEAK_GBS})
        for k_, v_ in steady.items():
            roof["steady_state_" + k_] = v_
    if W["name"] == args.workload and rank == 0 and not args.inner and not args.ablate:
        # what `peak` is measured against on THIS box: float4 copy and triad over 1 GiB arrays, ~60 ms in all
        try:
            copy_gbs, triad_gbs, memcpy_gbs = R.solver.measure_hbm(1 << 30, 10)
            roof["peak_measured"] = max(copy_gbs, triad_gbs, memcpy_gbs)
            roof["peak_measured_detail"] = {"copy_gbs": copy_gbs, "triad_gbs": triad_gbs, "runtime_d2d_memcpy_gbs": memcpy_gbs, "bytes_per_array": 1 << 30,
                                            "launches": 10, "definition": "bytes read + written / HIP-event time: float4 grid-stride copy and triad kernels "
                                                                          "(ctgn_measure_hbm) and hipMemcpyAsync device to device; peak_measured = the best of the three"}
            roof["frac_of_measured"] = roof["achieved"] / roof["peak_measured"] if roof["peak_measured"] > 0 else None
        except Exception as e:                             # a measurement beside the path: recorded, not raised
            roof["peak_measured"] = None
            roof["peak_measured_error"] = repr(e)[:200]
    out["roofline"] = roof
    # ---- parity on the very inputs that were timed: one fresh solve of the profile's budget on the GPU and through the oracle
    om = None
    if want_parity and not args.inner and not args.ablate:
        from oracle import oracle as orc
        R.upload()
        o = R.options(W["ipf"])
        pose_g, summ_g = (R.sh.solve(W["pose0"], W["inp"]["tbe"], o, W["mm"]) if R.sh is not None else R.solver.solve(W["pose0"], W["inp"]["tbe"], o, W["mm"]))[:2]
        if rank == 0:
            om = oracle_map(W)
            if world > 1:                                   # the oracle registers the whole scan, the ranks their shards of it
                o_raw, o_t, o_w0 = W["all_raw"], W["all_t"], W["all_world0"]
            else:
                o_raw, o_t, o_w0 = W["raw"], W["t"], W["world0"]
            op = orc.MotionPrior(previous_begin_tr=W["inp"]["prev_b"], previous_end_tr=W["inp"]["prev_e"])
            pose_o, _, so = orc.register_gn(om, o_raw, o_w0, o_t, W["pose0"], W["inp"]["tbe"],
                                            orc.Options(num_iters_icp=W["ipf"], min_number_neighbors=W["min_nb"], threshold_orientation_norm=0.0), op,
                                            heap_mode=0, num_threads=usable_cores())
            tr, rot = se3.pose_error(pose_g, pose_o)
            out["parity_m_rad"] = [tr, rot]
            out["parity"] = {"gpu_vs_oracle_m_rad": [tr, rot], "iterations": W["ipf"], "n_used_gpu": int(summ_g.num_residuals_used),
                             "n_used_oracle": int(so.num_residuals_used), "tolerance_m_rad": [1e-4, 1e-4]}
            assert tr < 1e-4 and rot < 1e-4 and summ_g.num_residuals_used == so.num_residuals_used, out["parity"]
    if want_cpu and rank == 0 and not args.no_cpu_baseline and world == 1:
        if om is None:
            om = oracle_map(W)
        out["cpu_baseline"] = cpu_baseline_gn(W, om, cpu_seconds)
        if out["cpu_baseline"]["value"]:
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if register_extras and rank == 0 and world == 1 and not args.ablate and not args.inner:
        out.update(register_calls(W, cia, om))
    R.close()
    return out, om


def cpu_baseline_gn(W, om, seconds: float = 10.0):
    """`cpu_baseline` of a workload: the oracle's GN loop (a port; OpenMP over keypoints) on this box's host cores, on a BOUNDED sample:
    the first keypoints of the workload, as many as ~`seconds` of CPU work allow, whole solves of the profile's iteration budget."""
    from oracle import oracle as orc
    cores = usable_cores()
    prior = orc.MotionPrior(previous_begin_tr=W["inp"]["prev_b"], previous_end_tr=W["inp"]["prev_e"])
    o = orc.Options(num_iters_icp=W["ipf"], min_number_neighbors=W["min_nb"], threshold_orientation_norm=0.0)
    n_all = len(W["t"])

    def timed(n, threads):
        t0 = time.perf_counter()
        _, _, s = orc.register_gn(om, W["raw"][:n], W["world0"][:n], W["t"][:n], W["pose0"], W["inp"]["tbe"], o, prior, heap_mode=0, num_threads=threads)
        return n * max(1, s.num_iters) / (time.perf_counter() - t0), s
    probe_n = min(n_all, 20_000)
    rate, _ = timed(probe_n, cores)                         # also warms the OpenMP pool and the caches
    n = int(min(n_all, max(probe_n, rate * seconds / W["ipf"])))
    v_n, s = timed(n, cores)
    n1 = int(min(n, max(2_000, v_n / cores * min(seconds, 5.0) / W["ipf"])))
    v_1, _ = timed(n1, 1)
    return {"value": v_n, "unit": "keypoints/s", "cores": cores, "kind": "port", "single_thread_value": v_1,
            "sample": f"oracle GN loop (OpenMP over keypoints, {cores} threads), the first {n} of the workload's {n_all} keypoints x {W['ipf']} iterations "
                      f"(one fresh solve; single thread: the first {n1})"}


def register_calls(W, cia, om):
    """Whole `CT_ICP_Registration::Register` calls through the C ABI on this workload's keypoints with the profile's own stop threshold
    (host WPoint3D buffer in, pose + world points out): GN and the robust-loss route, each next to the oracle on the host."""
    from oracle import oracle as orc
    from ct_icp_amd import se3
    name, inp = W["name"], W["inp"]
    gn_kw = dict(num_iters_icp=W["ipf"], min_number_neighbors=W["min_nb"], threshold_orientation_norm=0.1)
    rb_kw = dict(ROBUST_PROFILE) if name != "C" else dict(num_iters_icp=20, ls_max_num_iters=10, max_num_residuals=1500, loss_function="CAUCHY", ls_sigma=0.1,
                                                          threshold_orientation_norm=0.1, threshold_translation_norm=0.01)       # nclt_config.yaml:69-104
    rb_kw["min_number_neighbors"] = W["min_nb"]
    kps = np.zeros(len(W["t"]), dtype=cia.WPOINT3D_DTYPE)
    kps["raw_point"], kps["t"] = W["raw"], W["t"]
    res = {}
    for key, reg in (("frames_per_sec", cia.CT_ICP_Registration(cia.CTICPOptions(solver=cia.GN, debug_print=False, **gn_kw))),
                     ("robust_route", cia.CT_ICP_Registration(cia.CTICPOptions(solver=cia.CERES, debug_print=False, **rb_kw)))):
        times = []
        for _ in range(23):
            kps["world_point"] = W["world0"]
            frame = cia.TrajectoryFrame.from_pose14(W["pose0"], *inp["tbe"])
            t0 = time.perf_counter()
            summ = reg.Register(W["gm"], kps, frame, W["mm"])
            times.append(time.perf_counter() - t0)
        med = float(np.median(times[3:]))
        res[key] = {"value": 1.0 / med, "unit": "frames/s", "ms_per_frame": med * 1e3, "keypoints": int(len(kps)), "iterations": int(summ.num_iters),
                    "residuals": int(summ.num_residuals_used), "includes": "host WPoint3D buffer -> H2D -> all iterations to the stop test -> pose + world points D2H",
                    "pose": frame.pose14()}
    prior = orc.MotionPrior(previous_begin_tr=inp["prev_b"], previous_end_tr=inp["prev_e"])
    cores = usable_cores()
    for threads in (1, cores):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            pose_o, _, so = orc.register_gn(om, W["raw"], W["world0"], W["t"], W["pose0"], inp["tbe"], orc.Options(**gn_kw), prior, heap_mode=0, num_threads=threads)
            ts.append((time.perf_counter() - t0) * 1e3)
        res["frames_per_sec"][f"cpu_port_ms_per_frame_{threads}_threads"] = float(np.median(ts))
    tr, rot = se3.pose_error(res["frames_per_sec"].pop("pose"), pose_o)
    res["frames_per_sec"]["gpu_vs_oracle_m_rad"] = [tr, rot]
    rp = orc.RobustPrior(previous_begin_tr=tuple(inp["prev_b"]), previous_end_tr=tuple(inp["prev_e"]))
    t0 = time.perf_counter()
    pose_r, _, s_r = orc.register_robust(om, W["raw"], W["t"], W["pose0"], inp["tbe"], orc.RobustOptions(**rb_kw), rp, heap_mode=0)
    res["robust_route"]["cpu_port_ms_per_frame_1_thread"] = (time.perf_counter() - t0) * 1e3
    tr, rot = se3.pose_error(res["robust_route"].pop("pose"), pose_r)
    res["robust_route"]["gpu_vs_oracle_m_rad"] = [tr, rot]
    res["robust_route"]["gpu_over_cpu_1core"] = res["robust_route"]["cpu_port_ms_per_frame_1_thread"] / res["robust_route"]["ms_per_frame"]
    return res


def self_launch(n_gpus: int, backend: str, devices_visible: int) -> int:
    """Re-run this command line under torch.distributed.run with one rank per GPU (rendezvous on 127.0.0.1, a free port) and return its exit
    code. Fewer visible devices than ranks is an error (one JSON line, exit code 2) unless the gloo rehearsal backend was asked for, whose
    ranks share devices on purpose."""
    import socket
    import subprocess
    if backend == "nccl" and devices_visible < n_gpus:
        print(json.dumps({"error": f"--gpus {n_gpus} needs {n_gpus} visible devices, this box has {devices_visible}",
                          "n_gpus_requested": n_gpus, "devices_visible": devices_visible}))
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--map-frames", type=int, default=20, help="B2-small / B1: sweeps accumulated into the street-canyon map")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default=None, choices=["B2", "B2-small", "B1", "C", "D"],
                    help="B2 (default at --gpus 1): all returns of the HDL-64E sweep as keypoints over the steady-state ~270 MB map; "
                         "B2-small: the same regime over round 1's 20-frame, 5 MB street map; B1: the reference's keypoint count "
                         "(1.5 m grid of the 0.5 m-subsampled frame, latency regime) over the B2-small map; C: NCLT / HDL-32E profile; D (default at "
                         "--gpus > 1): Ouster-128-style ray-cast scan, 0.05 m grid keypoints, 0.5 m x 40-pt map of everything within 100 m")
    ap.add_argument("--sub", default=None, help="comma-separated workloads reported as sub-objects of the line (default at --gpus 1 with the default "
                                               "workload: B1,C,D; 'none' to skip)")
    ap.add_argument("--d-sweeps", type=int, default=8, help="workload D: accumulated sub-sweeps of the 128 x 2048 pattern (8 = 2.1 M rays)")
    ap.add_argument("--d-radius", type=float, default=100.0, help="workload D: radius of the sampled map (100 m = the eviction radius; smaller only for rehearsals)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="keypoints in the CPU baseline sample of B2 (0 = all)")
    ap.add_argument("--ablate", type=int, default=0, help="measurement hook: skip kernel phases (invalid results)")
    ap.add_argument("--force-dist", action="store_true", help="use the sharded (all-reduce) loop even with one rank")
    ap.add_argument("--torch-collective", action="store_true", help="sharded loop with torch.distributed.all_reduce between stepwise calls "
                                                                   "instead of the library's own ncclAllReduce")
    ap.add_argument("--order", default="auto", choices=["auto", "on", "off"],
                    help="home-voxel ordering of the GN kernels' work (ctgn_set_ordering); auto = the library's cost model")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 PMC passes (roofline.traffic, wait fractions)")
    ap.add_argument("--sub-pmc", default="D", help="comma-separated sub-workloads whose HBM / wait counters are collected too (three short "
                                                    "rocprofv3 passes each; default D — the configuration whose map exceeds the caches; 'C,D' for both)")
    ap.add_argument("--no-extras", action="store_true", help="skip frames/s, robust route, frame stages")
    ap.add_argument("--detail-stdout", action="store_true", help="also print the full detail object as an EARLIER stdout line (default: files only)")
    ap.add_argument("--config-e-scale", type=int, default=10,
                    help="config E (BASELINE.json configs[4]) on this one GPU inside the default line: 11 sequences, seeds 10-20, KITTI lengths / "
                         "this (10 = the size SURVEY.md 8d defines: 2 319 frames, ~1.7 s of loop + ~30 s of scan generation on the GPU, "
                         "profiles/r05_config_e_n1.json; 100 = 233 frames for a quick look); 0 = skip")
    ap.add_argument("--clock-warm", type=int, default=CLOCK_WARM)
    ap.add_argument("--inner", action="store_true", help="the run the PMC passes profile: timed loop only, no extras")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, the all-reduce goes "
                         "through torch.distributed on the host; numbers are not scaling numbers)")
    args = ap.parse_args()
    if args.inner:
        args.no_pmc = args.no_cpu_baseline = args.no_extras = True
        args.clock_warm = min(args.clock_warm, 30)
        args.sub = "none"

    import torch

    # `python bench.py --gpus N` with N > 1 and no launcher around it starts its own ranks (one process per GPU over RCCL, exactly the
    # command the driver uses for N > 1); it never prints a line whose n_gpus differs from --gpus with exit code 0.
    # (a stray WORLD_SIZE=1 — e.g. left in the environment by a single-rank rendezvous earlier in the same process tree — is not a launcher)
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and "TORCHELASTIC_RUN_ID" not in os.environ:
        raise SystemExit(self_launch(args.gpus, args.dist_backend, torch.cuda.device_count()))

    import ct_icp_amd as cia
    from ct_icp_amd import se3, synthetic as syn

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(json.dumps({"error": f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", "n_gpus_requested": args.gpus}))
        raise SystemExit(2)
    if args.dist_backend == "nccl" and torch.cuda.device_count() < (world if world > 1 else 1):
        if rank == 0:
            print(json.dumps({"error": f"--gpus {args.gpus} needs {world} visible devices, this box has {torch.cuda.device_count()}",
                              "n_gpus_requested": args.gpus, "devices_visible": torch.cuda.device_count()}))
        raise SystemExit(2)
    default_line = args.workload is None and world == 1
    if args.workload is None:
        args.workload = "B2" if world == 1 else "D"
    if args.sub is None:
        args.sub = "B1,C,D" if (default_line and not args.ablate and args.variant == 0) else "none"
    subs = [w for w in args.sub.split(",") if w and w != "none"]
    if args.dist_backend == "gloo":
        args.torch_collective = True
        local_rank = local_rank % max(1, torch.cuda.device_count())
    args.local_rank = local_rank
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world > 1 or args.force_dist
    if sharded:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", RANK="0", WORLD_SIZE="1")
        if args.dist_backend == "gloo":
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    # ------------------------------------------------------------------------------------------------ the headline workload
    W = build_workload(args.workload, rank, world, args, cia, syn, se3)
    if args.workload == "D" and world > 1:
        # config D sharding: global sort of the scan by home voxel, contiguous chunk per rank — done by the
        # library (ctgn_set_keypoints_sharded) from the WHOLE scan, which every rank holds
        W["all_raw"], W["all_t"], W["all_world0"] = W["raw"], W["t"], W["world0"]
        W["shard"] = (rank, world)
    live_pmc = world == 1 and not args.no_pmc and not args.ablate and dist is None
    res, om = measure_workload(W, args, cia, torch, dist, sharded, args.steps, args.warmup, args.clock_warm, world=world, rank=rank, pmc_live=live_pmc,
                               want_cpu=False, want_steady=not args.inner)
    result = None
    if rank == 0:
        result = {
            "metric": "registered keypoints/sec per GN iter", "value": res["value"], "unit": "keypoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
            "higher_is_better": True, "scaling": "strong" if (args.workload == "D" and world > 1) else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": NAMES[args.workload], "workload_id": args.workload,
                       "keypoints_per_gpu": res["keypoints"], "keypoints_total": res["keypoints_total"], "map_points": res["map_points"],
                       "map_voxels": res["map_voxels"], "searched_level_mb": res["searched_level_mb"], "n_used_last_iter": res["n_used_last_iter"],
                       "iterations_per_solve": res["iterations_per_solve"], "solves_timed": res["solves_timed"],
                       "parallelism": "single GPU" if world == 1 else f"keypoints sharded x{world} (home-voxel sort, contiguous chunks), map replicated, "
                                                                      + ("1 ncclAllReduce(96 f64) per iteration issued by the library" if not args.torch_collective else
                                                                         f"1 torch.distributed all_reduce(96 f64) per iteration, backend {args.dist_backend}"),
                       "kernel_variant": args.variant, "keypoint_ordering": args.order, "inputs_sha16": W["inputs_sha16"],
                       "inputs_note": "synthetic, deterministic from the seeds in bench.py; cached scans under .bench_cache/ carry their generator "
                                      "parameters + a SHA-256 of their arrays, verified on load (mismatch = regenerated); inputs_sha16 = SHA-256 of "
                                      "the arrays actually measured (map insert list, scan, timestamps, ground-truth pose)"
                                      + ("; the map is the scene's surfaces within the 100 m eviction radius sampled directly (6.9 M points after the "
                                         "insert rule = the steady-state driving-profile map), larger than SURVEY 8d's 20-frame map" if args.workload == "B2" else "")},
            "step_definition": f"one GN iteration; the K timed steps are FRESH solves of the profile's budget ({res['iterations_per_solve']} iterations each, "
                               "ct_icp.cpp:745): per solve the uploaded world points are put back on the device, gn_begin, then the iterations — the first search "
                               "of every solve has no carried-over bound, exactly as a frame pays it",
            "first_iteration_ms": res["first_iteration_ms"], "later_iteration_ms": res["later_iteration_ms"], "iteration_split_note": res["iteration_split_note"],
            "clock_warmup_iterations": args.clock_warm,
            "clock_settle_iterations": res.get("clock_settle_iterations"),
            "clock_warmup_note": "untimed iterations of the same fresh-solve loop immediately before the W warm-up and K timed steps (no upload in "
                                 "between): the GPU leaves its idle clocks only under sustained load",
            "frames_per_sec_equiv": res["frames_per_sec_equiv"],
            "roofline": res["roofline"],
        }
        for k in ("parity_m_rad", "parity", "sharded_upload", "sharded_overheads"):
            if k in res:
                result[k] = res[k]
        inp = W["inp"]
        if args.workload in ("B2", "B2-small") and world == 1 and not args.ablate and not args.no_extras and args.variant != 1:
            gm, mm = W["gm"], W["mm"]
            result["frames_per_sec"] = measure_frames_per_sec(cia, gm, inp, syn, se3, mm)
            result["robust_route"] = measure_robust_frames_per_sec(cia, gm, inp, syn, se3, mm)
            # (the measurements around the path must not cost the run its headline line: a failure is recorded, not raised)
            try:
                result["frame_stages"] = fs = measure_frame_stages(cia, inp, syn, se3, local_rank)
                # the whole per-frame loop of Odometry::DoRegister on this frame, every data-parallel step through the library with host
                # buffers in and out: the steps either side + one Register call on the sampled keypoints
                for route, reg_ms in (("gn", result["frames_per_sec"]["ms_per_frame"]), ("robust", result["robust_route"]["ms_per_frame"])):
                    ms = fs["grid_sampling_ms"] + fs["keypoint_sampling_ms"] + reg_ms + fs["undistortion_ms"] + fs["map_update_ms"]
                    result.setdefault("frame_by_stage_calls", {})[route] = {"ms_per_frame": ms, "frames_per_sec": 1e3 / ms}
                # the same frame as ONE ctgn_frame_register + ctgn_frame_update_map (scan resident on the device)
                result["frame_pipeline"] = fs.pop("frame_pipeline")
                result["frame_pipeline"]["frames_per_sec"] = 1e3 / result["frame_pipeline"]["frame_ms"]
            except Exception as e:
                result["frame_stages_error"] = f"{type(e).__name__}: {e}"[:300]
            if default_line and args.config_e_scale > 0:
                # config E on one GPU (no 8-GPU number is asked of this run): the 11 sequences back to back, one ctgn_frame call per frame
                import importlib.util
                spec = importlib.util.spec_from_file_location("ctgn_sequence_run", os.path.join(ROOT, "scripts", "sequence_run.py"))
                seq_mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(seq_mod)
                try:
                    ce = seq_mod.run_config_e(scale=args.config_e_scale, device=local_rank)
                    ce.pop("per_sequence_detail", None)
                    result["config_e"] = ce
                except Exception as e:
                    result["config_e_error"] = f"{type(e).__name__}: {e}"[:300]
                # ... and what the same scans cost through the reference's own Odometry::RegisterFrame (bounded: 100 frames of sequence 0)
                if not args.no_cpu_baseline and world == 1 and isinstance(result.get("config_e"), dict):
                    try:
                        ro_ = measure_reference_odometry(syn, 100, local_rank)
                        if ro_:
                            result["config_e"]["reference_odometry"] = ro_
                            result["config_e"]["reference_odometry_on_gpu_map_frames_per_sec"] = {
                                "unarmed": ro_["gpu_map"]["frames_per_sec"], "armed": ro_["gpu_map_armed"]["frames_per_sec"],
                                "armed_device_shuffle": ro_["gpu_map_armed_device_shuffle"]["frames_per_sec"], "cpu_map": ro_["cpu_map"]["frames_per_sec"]}
                    except Exception as e:
                        result["config_e"]["reference_odometry_error"] = f"{type(e).__name__}: {e}"[:300]
        if not args.no_cpu_baseline and args.workload in ("B2", "B2-small") and world == 1:      # rank 0 at N = 1 only
            result["cpu_baseline"] = cpu_baseline(inp, W["pose0"], W["world0"], args, om)
            if result["cpu_baseline"]["value"]:
                result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
                result["gpu_over_cpu_reference_shaped"] = result["value"] / result["cpu_baseline"]["reference_shaped"]["value"]
            if "robust_route" in result:                   # same frame, same settings: the two poses must agree
                rr, cr = result["robust_route"], result["cpu_baseline"]["robust_route"]
                tr, rot = se3.pose_error(np.array(rr.pop("pose")), np.array(cr.pop("pose")))
                rr["gpu_vs_cpu_pose_m_rad"] = [tr, rot]
                rr["gpu_over_cpu_1core"] = cr["ms_per_frame"] / rr["ms_per_frame"]
        elif not args.no_cpu_baseline and world == 1 and not args.inner:
            result["cpu_baseline"] = cpu_baseline_gn(W, om if om is not None else oracle_map(W), 10.0)
            result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
        if "robust_route" in result:
            result["robust_route"].pop("pose", None)
    del om

    # ------------------------------------------------------------------------------------------------ the other configurations, one GPU
    if subs and world == 1:
        W.clear()
        import gc
        gc.collect()
        result["workloads"] = {}
        for name in subs:
            t_sub = time.perf_counter()
            Ws = som = None
            try:                                          # a sub-workload that fails costs its own object, not the line
                Ws = build_workload(name, rank, world, args, cia, syn, se3)
                small = name in ("B1", "C")
                sub_steps = {"B1": 200, "C": 200, "D": 20}.get(name, 50)
                sres, som = measure_workload(Ws, args, cia, torch, None, False, sub_steps, 10 if small else 5, 100 if small else 10, want_steady=True,
                                             cpu_seconds=8.0, register_extras=small, pmc_live=(name in args.sub_pmc.split(",") and not args.no_pmc))
                sres["config"] = NAMES[name]
                sres["inputs_sha16"] = Ws["inputs_sha16"]
                if name == "D":
                    sres["config_detail"] = {"rays": int(Ws["inp"]["rays"]), "returns": int(Ws["inp"]["returns"])}
                sres["wall_seconds"] = time.perf_counter() - t_sub
                result["workloads"][name] = sres
            except Exception as e:
                result["workloads"][name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            del Ws, som
            gc.collect()

    # ------------------------------------------------------------------------------------------------ N > 1 extras
    if world > 1 and not args.inner:
        from ct_icp_amd.distributed import ShardedGnSolver, allreduce_system
        gm, mm, pose0, inp = W["gm"], W["mm"], W["pose0"], W["inp"]
        # (a) the same scan on ONE GPU (rank 0, the others wait): the denominator of the strong-scaling efficiency
        # (neither may cost the run its line: rank 0's own measurement is wrapped; the collective one starts only if every rank got
        # its inputs, so no rank waits in an exchange its peers never reach)
        single = None
        if rank == 0:
            try:
                W1 = dict(W, raw=W["all_raw"], t=W["all_t"], world0=W["all_world0"])
                W1.pop("shard", None)
                R1 = Runner(W1, args, cia, torch, None, False)
                R1.upload()
                d1 = R1.timed(R1.fresh, args.steps, args.warmup, args.clock_warm)[0]
                R1.close()
                single = {"value": len(W1["t"]) * args.steps / d1, "ms_per_step": d1 / args.steps * 1e3, "keypoints": int(len(W1["t"])),
                          "note": "the same scan, unsharded, on rank 0's GPU; same fresh-solve loop"}
            except Exception as e:
                result["strong_scaling_single_gpu_reference_error"] = f"{type(e).__name__}: {e}"[:300]
        dist.barrier()
        # (b) weak-scaling line: one B2-small sweep per rank (its own noise realisation), same sharded loop
        Ww = None
        try:
            Ww = build_workload("B2-small", rank, world, args, cia, syn, se3)
        except Exception as e:
            if rank == 0:
                result["weak_scaling_line_error"] = f"{type(e).__name__}: {e}"[:300]
        ready = torch.tensor([1.0 if Ww is not None else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(ready, op=dist.ReduceOp.MIN)
        if rank == 0:
            result["strong_scaling_single_gpu_reference"] = single
            if single:
                result["strong_scaling_efficiency"] = result["value"] / (world * single["value"])
        if ready.item() > 0.5:
            Rw = Runner(Ww, args, cia, torch, dist, True)
            Rw.upload()
            dw = Rw.timed(Rw.fresh, args.steps, args.warmup, args.clock_warm)[0]
            Rw.close()
            tw = torch.tensor([dw], dtype=torch.float64, device="cuda")
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            nw = torch.tensor([len(Ww["t"])], dtype=torch.float64, device="cuda")
            dist.all_reduce(nw, op=dist.ReduceOp.SUM)
            if rank == 0:
                result["weak_scaling_line"] = {"value": float(nw.item()) * args.steps / float(tw.item()), "ms_per_step": float(tw.item()) / args.steps * 1e3,
                                               "keypoints_per_gpu": int(len(Ww["t"])), "workload": "B2-small sweep per rank, sharded loop", "scaling": "weak"}
    if rank == 0:
        emit(result, args)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


DETAIL_NAME = "bench_detail.json"
LINE_LIMIT = 4096              # the driver keeps a tail of its child's output: the line it parses must fit well inside it


def _round(o, sig=5):
    """Numbers to `sig` significant digits (the line is a record, not an archive); NaN / inf -> None: strict JSON only."""
    if isinstance(o, bool) or o is None or isinstance(o, str):
        return o
    if isinstance(o, (int, np.integer)):
        return int(o)
    if isinstance(o, (float, np.floating)):
        f = float(o)
        if f != f or f in (float("inf"), float("-inf")):
            return None
        return float(f"{f:.{sig}g}")
    if isinstance(o, dict):
        return {k: _round(v, sig) for k, v in o.items() if v is not None or k in ("vs_baseline", "traffic")}
    if isinstance(o, (list, tuple, np.ndarray)):
        return [_round(v, sig) for v in o]
    return str(o)


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _compact_roofline(r):
    if not isinstance(r, dict):
        return None
    out = _pick(r, "bound", "achieved", "peak", "peak_measured", "unit", "frac", "frac_of_measured", "traffic", "hbm_counter_frac", "kernel_ms_avg",
                "alg_bytes_per_launch", "wait_frac", "valu_issue_frac", "tcc_hit_rate", "valu_instructions_per_keypoint")
    out.setdefault("traffic", None)
    out["kernel"] = str(r.get("kernel", "")).split(" ")[0]
    if isinstance(r.get("first_iteration"), dict):
        out["first_iteration"] = _pick(r["first_iteration"], "kernel_ms", "frac")
    if isinstance(r.get("later_iterations"), dict):
        out["later_iterations"] = _pick(r["later_iterations"], "kernel_ms", "frac", "pool_certified_frac")
    return out


def _compact_cpu(c):
    return _pick(c, "value", "unit", "cores", "kind", "single_thread_value") if isinstance(c, dict) else None


def compact_line(result) -> str:
    """The ONE stdout line of the contract: strict JSON, < LINE_LIMIT bytes. Headline keys, a compact `roofline` and `cpu_baseline`,
    parity, frames/s, and per sub-workload {value, ms_per_step, frac, parity, cpu}. Everything else (definitions, notes, per-stage tables)
    is in the detail file (`bench_detail.json`, next to this script and under gpurun_out/ when that directory exists)."""
    line = _pick(result, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    cfg = result.get("config", {})
    line["config"] = _pick(cfg, "workload_id", "keypoints_per_gpu", "keypoints_total", "map_points", "searched_level_mb", "iterations_per_solve", "inputs_sha16")
    line["config"]["workload"] = {"B2": "config B2 = BASELINE.json configs[1]: synthetic HDL-64E sweep, every return a keypoint, sampled steady-state driving-profile map "
                                        "(6.9 M pts, 0.8 m x 30 pts), radius 0.75 (27 voxels), k=20, 5-iteration fresh solves",
                                  "D": "config D = BASELINE.json configs[3]: Ouster-128-style 2.1 M-ray scan, 0.05 m grid keypoints, map 0.5 m x 40 pts within "
                                       "100 m, radius 0.8 (125 voxels), k=20"}.get(cfg.get("workload_id"), str(cfg.get("workload", ""))[:160])
    if "parallelism" in cfg:
        line["config"]["parallelism"] = str(cfg["parallelism"])[:120]
    line["roofline"] = _compact_roofline(result.get("roofline"))
    line["cpu_baseline"] = _compact_cpu(result.get("cpu_baseline"))
    if isinstance(result.get("cpu_baseline"), dict) and "sample" in result["cpu_baseline"]:
        line["cpu_baseline"]["sample"] = str(result["cpu_baseline"]["sample"])[:150]
    for k in ("parity_m_rad", "gpu_over_cpu", "first_iteration_ms", "later_iteration_ms", "strong_scaling_efficiency"):
        if k in result:
            line[k] = result[k]
    if isinstance(result.get("sharded_overheads"), dict):
        line["sharded_overheads"] = _pick(result["sharded_overheads"], "allreduce_us", "five_launch_overhead_us", "iteration_us", "error")
    if isinstance(result.get("frames_per_sec"), dict):
        line["frames_per_sec"] = _pick(result["frames_per_sec"], "value", "ms_per_frame", "keypoints")
    if isinstance(result.get("robust_route"), dict):
        line["robust_route"] = _pick(result["robust_route"], "value", "ms_per_frame")
    if isinstance(result.get("frame_pipeline"), dict):
        line["frame_pipeline"] = _pick(result["frame_pipeline"], "frame_ms", "register_ms", "update_map_ms", "frames_per_sec")
        if isinstance(result["frame_pipeline"].get("page_locked_arrays"), dict):
            line["frame_pipeline"]["page_locked_frame_ms"] = _round(result["frame_pipeline"]["page_locked_arrays"].get("frame_ms"))
    if isinstance(result.get("config_e"), dict):
        line["config_e"] = _pick(result["config_e"], "frames_per_sec", "frames", "sequences", "failures", "scale",
                                 "reference_odometry_on_gpu_map_frames_per_sec")
    for k in ("strong_scaling_single_gpu_reference", "weak_scaling_line"):
        if isinstance(result.get(k), dict):
            line[k] = _pick(result[k], "value", "ms_per_step", "keypoints", "keypoints_per_gpu", "scaling")
    subs = {}
    for name, w in (result.get("workloads") or {}).items():
        s = _pick(w, "value", "ms_per_step", "keypoints", "parity_m_rad", "gpu_over_cpu")
        r = w.get("roofline") or {}
        s.update(_pick(r, "frac", "kernel_ms_avg", "traffic", "hbm_counter_frac", "wait_frac", "tcc_hit_rate"))
        if isinstance(w.get("cpu_baseline"), dict):
            s["cpu"] = _pick(w["cpu_baseline"], "value", "cores", "kind")
        if isinstance(w.get("frames_per_sec"), dict):
            s["frame_ms"] = w["frames_per_sec"].get("ms_per_frame")
        if isinstance(w.get("robust_route"), dict):
            s["robust_frame_ms"] = w["robust_route"].get("ms_per_frame")
        subs[name] = s
    if subs:
        line["workloads"] = subs
    line["detail"] = DETAIL_NAME
    line = _round(line)
    for k in ("value",):                                 # the headline keeps its digits
        if isinstance(result.get(k), (int, float)):
            line[k] = float(result[k])
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    while len(text) >= LINE_LIMIT and line.get("workloads"):      # never over the limit: shed sub-workloads last in, first out
        line["workloads"].popitem()
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, len(text)
    return text


def emit(result, args):
    """Detail to files, the compact line — and nothing else — to stdout."""
    detail = json.dumps(result, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAIL_NAME if not args.inner else "bench_detail_inner.json"), "w") as f:
                    f.write(detail + "\n")
            except OSError:
                pass
    if args.detail_stdout:
        print(detail, flush=True)
    print(compact_line(result), flush=True)


def usable_cores() -> int:
    """Threads the CPU baseline may really use: the affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max") and txt[0] != "max":
                n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            elif path.endswith("quota_us") and int(txt[0]) > 0:
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                n = min(n, max(1, int(txt[0]) // period))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def measure_reference_odometry(syn, frames: int = 100, device: int = 0):
    """What a caller of the REFERENCE gets: `ct_icp::Odometry::RegisterFrame` — the reference's own src/ct_icp/odometry.cpp, compiled where
    it lies (oracle/_ref, the checker-side build; the cpu_baseline leg of this file is the one place outside tests/ that may load it) — fed
    the first `frames` scans of config E's sequence 0 (133 k points each; the vehicle pulls away over 20 frames because that loop starts
    from the identity, odometry.cpp:276-300), GN solver, driving profile, on four maps: its own MULTI_RESOLUTION_VOXEL_HASHMAP (CPU);
    GPU_VOXEL_HASHMAP with only Register and the map on the device (integration/gpu_map.h + gn_gpu_arm.h); the same with the four arms of
    integration/odometry_gpu_arm.h compiled into odometry.cpp (scan resident on the device from InitializeFrame to UpdateMap), once with the
    reference's own std::shuffle reproduced on the host and once with the shuffle made on the device. Milliseconds per RegisterFrame call
    past the 20-frame start-up regime (frames 25+)."""
    from oracle import ref_odometry as ro
    if not ro.available():
        return None
    seed = 10
    scene = syn.street_scene(max(300.0, frames * 1.2 + 60.0), seed=seed)
    dirs, rel_t = syn.lidar_pattern("hdl64", azimuth_offset=np.pi)
    knots = syn.driving_trajectory(frames + 1, seed=seed, start_x=20.0, ramp_frames=20, centered=True)
    scans = []
    for j in range(frames):
        sc = syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, j), 0.1 * j, 0.1 * (j + 1), noise=0.02, seed=1000 * seed + j, use_torch=True)
        scans.append((sc.raw, sc.t))
    out = {"frames": frames, "steady_state_from_frame": 25, "points_per_frame": float(np.mean([len(t) for _, t in scans])), "solver": "GN",
           "scans": "config E sequence 0 (seed 10), 20-frame ramp from rest"}
    end = {}
    for name, kind in (("cpu_map", ro.CPU_MAP), ("gpu_map", ro.GPU_MAP), ("gpu_map_armed", ro.GPU_MAP_ARMED),
                       ("gpu_map_armed_device_shuffle", ro.GPU_MAP_ARMED_DEVICE_SHUFFLE)):
        od = ro.RefOdometry(kind, solver=ro.GN, device=device)
        rs = [od.register_frame(raw, t) for raw, t in scans]
        od.close()
        ms = np.array([r["milliseconds"] for r in rs[25:]])
        end[name] = rs[-1]["pose"][11:14]
        out[name] = {"ms_per_frame": float(ms.mean()), "ms_per_frame_median": float(np.median(ms)), "frames_per_sec": float(1e3 / ms.mean()),
                     "failures": int(sum(0 if r["success"] else 1 for r in rs)), "keypoints_mean": float(np.mean([r["sample_size"] for r in rs[25:]]))}
    out["end_position_gap_to_cpu_map_m"] = {k: float(np.linalg.norm(v - end["cpu_map"])) for k, v in end.items() if k != "cpu_map"}
    return out


def measure_frames_per_sec(cia, gm, inp, syn, se3, mm, reps: int = 30):
    """M2 of SURVEY.md 8d: frames/s = 1 / wall time of one whole `CT_ICP_Registration::Register` call through the C ABI
    (host WPoint3D buffer in, H2D, all GN iterations to the stop test, pose + world points out) with the driving profile
    (5 iterations, stop at ||x|| < 0.1, config/odometry/driving_config.yaml:58-83) on the reference's keypoint count
    (1.5 m grid of the 0.5 m-subsampled sweep)."""
    raw, t = inp["raw"], inp["t"]
    sel = syn.grid_sample_indices(raw, 0.5)
    sel = sel[syn.grid_sample_indices(raw[sel], 1.5)]
    pose0 = syn.perturb_pose(inp["pose_gt"], 0.003, 0.03, seed=4)
    kps = np.zeros(len(sel), dtype=cia.WPOINT3D_DTYPE)
    kps["raw_point"], kps["t"] = raw[sel], t[sel]
    world0 = se3.ct_transform(pose0, inp["tbe"], t[sel], raw[sel])
    reg = cia.CT_ICP_Registration(cia.CTICPOptions(solver=cia.GN, num_iters_icp=5, threshold_orientation_norm=0.1, debug_print=False))
    times, iters = [], []
    for _ in range(reps):
        kps["world_point"] = world0
        frame = cia.TrajectoryFrame.from_pose14(pose0, *inp["tbe"])
        t0 = time.perf_counter()
        summ = reg.Register(gm, kps, frame, mm)
        times.append(time.perf_counter() - t0)
        iters.append(summ.num_iters)
    med = float(np.median(times[3:]))
    return {"value": 1.0 / med, "unit": "frames/s", "ms_per_frame": med * 1e3, "keypoints": int(len(sel)),
            "gn_iterations": int(np.median(iters)), "includes": "host WPoint3D buffer -> H2D -> GN loop -> pose + world points D2H"}


def measure_frame_stages(cia, inp, syn, se3, device: int):
    """The steps either side of the path on the frame being registered, host buffers in and out:
    frame grid sampling (0.5 m), keypoint grid sampling (1.5 m), full-scan undistortion, far-voxel eviction + insertion of the
    sampled frame into a device-resident map that already holds the 20 preceding frames. The map update changes its map, so
    every repetition gets a fresh one: one warm-up map, then the median over five timed maps."""
    raw, t = inp["raw"], inp["t"]
    maps = []
    for _ in range(10):                                  # 0-3: stage by stage, 4-6: the frame pipeline, 7-9: the same from page-locked arrays
        m = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(0.8, 0.1, 30)], default_radius=0.75,
                                                    device=device, device_updates=True))
        # frame-sized batches: the hash table is sized for the worst case of a batch (every point a new voxel), and the eviction scan
        # reads the whole table — a map built from 2 M-point batches carries a 0.5 GB table no odometry run would have
        for s0 in range(0, len(inp["map_points"]), 100_000):
            m.InsertPointCloud(inp["map_points"][s0:s0 + 100_000])
        maps.append(m)
    times, counts = [], {}
    # the same frame through ctgn_frame_register + ctgn_frame_update_map (scan resident on the device): 5 GN iterations from a
    # perturbed pose on the 1.5 m keypoints, every scan point undistorted and returned, then evict + insert
    pose0 = syn.perturb_pose(inp["pose_gt"], 0.003, 0.03, seed=4)
    o5 = cia.CTICPOptions(solver=cia.GN, num_iters_icp=5, debug_print=False)
    ptimes, pcounts = [], {}
    world_all = np.zeros_like(raw)                       # the caller's own array for the undistorted scan, as the reference's in-place loop has
    for rep, m in enumerate(maps[4:7]):
        fp = cia.FramePipeline(m, frame_voxel_size=0.5, sample_voxel_size=1.5)
        regs = []
        for _ in range(4):                               # the registration does not change the map: repeat on the same one
            t0 = time.perf_counter()
            r = fp.register(raw, t, pose0, inp["tbe"], o5, want_all=True, want_sampled=False, all_world_out=world_all)
            regs.append((time.perf_counter() - t0) * 1e3)
        lean = []
        for _ in range(4):
            t0 = time.perf_counter()
            fp.register(raw, t, pose0, inp["tbe"], o5, want_all=False, want_sampled=False)
            lean.append((time.perf_counter() - t0) * 1e3)
        fp.update_map(r["pose"][11:14], 100.0, False)    # every frame evicts: the timed update is not the table's first scan
        # the whole frame as ONE ctgn_frame call (round 4: the map update is enqueued behind the undistortion, with the new pose read on
        # the device, while the outputs travel home on a second stream): sampling, keypoints, 5 GN iterations, every scan point
        # undistorted and returned, eviction + insertion. It changes the map, hence once per fresh map.
        pts_before = m.NumPoints()
        t0 = time.perf_counter()
        rf = fp.frame(raw, t, pose0, inp["tbe"], o5, 100.0, want_all=True, want_sampled=False, all_world_out=world_all)
        whole = (time.perf_counter() - t0) * 1e3
        inserted = int(m.NumPoints() - pts_before)       # (the far voxels were evicted by the untimed update above: the difference is the insertion)
        assert np.array_equal(rf["pose"], r["pose"])
        if rep > 0:
            ptimes.append([min(regs[1:]), min(lean[1:]), whole])
        tr, rot = se3.pose_error(r["pose"], inp["pose_gt"])
        pcounts = {"sampled": int(len(r["sampled_indices"])), "keypoints": int(len(r["keypoint_indices"])),
                   "inserted": inserted, "gn_iterations": int(r["summary"].num_iters),
                   "error_vs_ground_truth_m_rad": [tr, rot], "gn_iteration_device_ms": float(r["summary"].avg_duration_iter)}
    pm = np.median(np.array(ptimes), axis=0)
    pipeline = {"register_ms": float(pm[0]), "register_without_full_scan_output_ms": float(pm[1]), "update_map_ms": float(pm[2] - pm[0]),
                "frame_ms": float(pm[2]),
                "includes": "ONE ctgn_frame call: host scan (xyz f64 + t) -> one H2D -> frame + keypoint grid sampling -> 5 GN iterations -> "
                            "undistortion of the sampled frame and of every scan point -> pose, summary, indices, N x 3 f64 world points D2H "
                            "and handed over, beside far-voxel eviction + insertion of the device-resident sampled frame (update_map_ms = what "
                            "the call costs beyond a register call)"}
    pipeline.update(pcounts)
    # the same call with the scan, its timestamps and the output array in page-locked memory (what a driver that fills a pinned buffer
    # from its sensor packets hands over): no staging copy on the way in, no hand-over copy on the way out
    try:
        raw_p, t_p, world_p = cia.pinned_array(raw.shape), cia.pinned_array(t.shape), cia.pinned_array(raw.shape)
    except Exception as e:                                # no page-locked memory to be had: the staged numbers stand alone
        raw_p = None
        pipeline["page_locked_arrays"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    locked = []
    for rep, m in enumerate(maps[7:10] if raw_p is not None else []):
        if rep == 0:
            raw_p[:] = raw
            t_p[:] = t
        fp = cia.FramePipeline(m, frame_voxel_size=0.5, sample_voxel_size=1.5)
        regs = []
        for _ in range(4):
            t0 = time.perf_counter()
            r = fp.register(raw_p, t_p, pose0, inp["tbe"], o5, want_all=True, want_sampled=False, all_world_out=world_p)
            regs.append((time.perf_counter() - t0) * 1e3)
        fp.update_map(r["pose"][11:14], 100.0, False)
        m.NumPoints()                                    # (an update without an insert mask returns when it is enqueued: wait for it outside the timed call)
        t0 = time.perf_counter()
        rp = fp.frame(raw_p, t_p, pose0, inp["tbe"], o5, 100.0, want_all=True, want_sampled=False, all_world_out=world_p)
        whole = (time.perf_counter() - t0) * 1e3
        assert np.array_equal(rp["pose"], rf["pose"]) and np.array_equal(world_p, world_all)
        if rep > 0:
            locked.append([min(regs[1:]), whole])
    if locked:
        lm = np.median(np.array(locked), axis=0)
        pipeline["page_locked_arrays"] = {"register_ms": float(lm[0]), "frame_ms": float(lm[1]), "frames_per_sec": 1e3 / float(lm[1]),
                                          "what": "scan rows, timestamps and the output array in page-locked host memory (ct_icp_amd.pinned_array): "
                                                  "DMA from / to the caller's arrays, no staging; same results bit for bit"}
    world_buf = np.zeros_like(raw)
    for rep, m in enumerate(maps[:4]):
        cia.grid_sampling(m, raw, 0.5)                   # a fresh handle sizes its scratch on the first scan-sized call: not a per-frame cost
        cia.transform_points(m, raw, t, inp["pose_gt"], inp["tbe"], out=world_buf)     # likewise its pinned staging buffers and helper threads
        m.RemoveElementsFarFromLocation(inp["pose_gt"][11:14], 100.0)     # every frame evicts: the timed update is not the table's first scan
        t0 = time.perf_counter()
        keep = np.sort(cia.grid_sampling(m, raw, 0.5))
        t1 = time.perf_counter()
        kp = cia.grid_sampling(m, raw[keep], 1.5)                    # keypoint selection (odometry.cpp:538)
        t1b = time.perf_counter()
        world = cia.transform_points(m, raw, t, inp["pose_gt"], inp["tbe"], out=world_buf)     # in place, like the reference's loop (odometry.cpp:461-486)
        t2 = time.perf_counter()
        m.RemoveElementsFarFromLocation(inp["pose_gt"][11:14], 100.0)
        kept = m.InsertPointCloud(world[keep])
        t3 = time.perf_counter()
        if rep > 0:
            times.append([(t1 - t0) * 1e3, (t1b - t1) * 1e3, (t2 - t1b) * 1e3, (t3 - t2) * 1e3])
        counts = {"points": int(len(t)), "sampled": int(len(keep)), "keypoints": int(len(kp)), "inserted": int(np.count_nonzero(kept)),
                  "map_points_after": int(m.NumPoints())}
    med = np.median(np.array(times), axis=0)
    out = {"grid_sampling_ms": float(med[0]), "keypoint_sampling_ms": float(med[1]), "undistortion_ms": float(med[2]),
           "map_update_ms": float(med[3]), "repetitions": len(times)}
    out.update(counts)
    out["frame_pipeline"] = pipeline
    return out


ROBUST_PROFILE = dict(num_iters_icp=5, ls_max_num_iters=5, max_num_residuals=900, loss_function="CAUCHY", ls_sigma=0.1)


def robust_keypoints(inp, syn):
    sel = syn.grid_sample_indices(inp["raw"], 0.5)
    return sel[syn.grid_sample_indices(inp["raw"][sel], 1.5)]


def measure_robust_frames_per_sec(cia, gm, inp, syn, se3, mm, reps: int = 20):
    """The robust-loss (CERES-profile) route, SURVEY.md 8f row 4: whole `Register(solver=CERES)` calls through the C ABI
    with the driving profile's solver settings (config/odometry/driving_config.yaml:52-89: 5 ICP x 5 LM iterations, Cauchy
    0.1, at most 900 residuals) on the reference's keypoint count."""
    sel = robust_keypoints(inp, syn)
    pose0 = syn.perturb_pose(inp["pose_gt"], 0.003, 0.03, seed=4)
    kps = np.zeros(len(sel), dtype=cia.WPOINT3D_DTYPE)
    kps["raw_point"], kps["t"] = inp["raw"][sel], inp["t"][sel]
    reg = cia.CT_ICP_Registration(cia.CTICPOptions(solver=cia.CERES, debug_print=False, **ROBUST_PROFILE))
    times = []
    for _ in range(reps):
        frame = cia.TrajectoryFrame.from_pose14(pose0, *inp["tbe"])
        t0 = time.perf_counter()
        summ = reg.Register(gm, kps, frame, mm)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times[3:]))
    s = cia.GnSolver(gm)
    rep = s.robust_report()
    tr, rot = se3.pose_error(frame.pose14(), inp["pose_gt"])
    return {"value": 1.0 / med, "unit": "frames/s", "ms_per_frame": med * 1e3, "keypoints": int(len(sel)),
            "residual_blocks": int(summ.num_residuals_used), "icp_iterations": int(summ.num_iters),
            "lm_iterations": int(rep["ls_iterations"]), "lm_accepted": int(rep["ls_accepted"]),
            "error_vs_ground_truth_m_rad": [tr, rot], "pose": [float(v) for v in frame.pose14()],
            "includes": "host WPoint3D buffer -> H2D -> 5 x (search, weights, cap, 5 x LM) -> pose + world points D2H"}


def cpu_baseline(inp, pose0, world0, args, om=None):
    """The oracle (a port, not the reference: it cannot be built here) timed on this box's host cores on the same
    workload: CPU-N = OpenMP over keypoints on all cores, plus the faithful serial CPU-1 the reference actually runs
    (its GN keypoint loop has no `#pragma omp`, ct_icp.cpp:753)."""
    from oracle import oracle as orc
    cores = usable_cores()
    if om is None:
        om = orc.Map(resolutions=[(0.8, 0.1, 30)], default_radius=0.75)
        for s0 in range(0, len(inp["map_points"]), 2_000_000):
            om.insert(inp["map_points"][s0:s0 + 2_000_000])
    n = len(inp["t"]) if args.cpu_sample <= 0 else min(args.cpu_sample, len(inp["t"]))
    raw, t, w0 = inp["raw"][:n], inp["t"][:n], world0[:n]
    prior = orc.MotionPrior(previous_begin_tr=inp["prev_b"], previous_end_tr=inp["prev_e"])

    def timed(threads, iters):
        o = orc.Options(num_iters_icp=iters, threshold_orientation_norm=0.0)
        t0 = time.perf_counter()
        _, _, s = orc.register_gn(om, raw, w0, t, pose0, inp["tbe"], o, prior, heap_mode=0, num_threads=threads)
        return n * s.num_iters / (time.perf_counter() - t0)

    timed(cores, 1)                                        # warm the caches / OpenMP pool
    iters_n = 10
    v_n, best_threads = 0.0, cores
    for th in sorted({cores, max(1, cores // 2), max(1, cores // 4), min(cores, 64), min(cores, 32)}):
        v = timed(th, iters_n if th == cores else 4)       # oversubscribed SMT boxes peak below the full count
        if v > v_n:
            v_n, best_threads = v, th
    cores = best_threads
    v_1 = timed(1, 2)
    # the same accumulation on reference-shaped containers (node-based hash map of vectors of 80-byte records,
    # std::priority_queue of tuples, one heap-allocated neighbour vector per keypoint): the flat-hash oracle above is faster
    # than the reference's own implementation, this variant is what the ">= 50x" target is fair against
    rm = orc.RefShapedMap(om)
    o1 = orc.Options(num_iters_icp=1, threshold_orientation_norm=0.0)

    def timed_ref(threads, reps):
        t0 = time.perf_counter()
        for _ in range(reps):
            rm.gn_accumulate(raw, w0, t, pose0, inp["tbe"], o1, num_threads=threads)
        return n * reps / (time.perf_counter() - t0)

    timed_ref(cores, 1)
    ref_n, ref_1 = timed_ref(cores, 5), timed_ref(1, 1)
    # the robust-loss route on one core (the oracle's restatement of DoRegisterCeres is serial)
    from ct_icp_amd import synthetic as syn
    sel = robust_keypoints(inp, syn)
    ro = orc.RobustOptions(**ROBUST_PROFILE)
    rp = orc.RobustPrior(previous_begin_tr=tuple(inp["prev_b"]), previous_end_tr=tuple(inp["prev_e"]))
    t0 = time.perf_counter()
    pose_r, _, s_r = orc.register_robust(om, inp["raw"][sel], inp["t"][sel], pose0, inp["tbe"], ro, rp, heap_mode=0)
    robust_ms = (time.perf_counter() - t0) * 1e3
    # M2 (frames/s): one whole Register(solver GN) on the reference's keypoint count (1.5 m grid of the 0.5 m-subsampled sweep), driving
    # profile (5 iterations, stop at ||x|| < 0.1) — the same call measure_frames_per_sec times through the C ABI. Port: the oracle on 1
    # thread (the reference's GN keypoint loop is serial) and on `cores`; reference: the reference's own Register compiled from its
    # sources (oracle/_ref, built by __graft_entry__.build(); GN runs on one thread there too, ct_icp.cpp:753).
    se3_ = __import__("ct_icp_amd.se3", fromlist=["x"])
    pose_f = syn.perturb_pose(inp["pose_gt"], 0.003, 0.03, seed=4)
    raw_f, t_f = inp["raw"][sel], inp["t"][sel]
    world_f = se3_.ct_transform(pose_f, inp["tbe"], t_f, raw_f)
    of = orc.Options(num_iters_icp=5, threshold_orientation_norm=0.1)

    def frame_ms(threads, reps):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            _, _, sf = orc.register_gn(om, raw_f, world_f, t_f, pose_f, inp["tbe"], of, prior, heap_mode=0, num_threads=threads)
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts)), sf
    f1, sf1 = frame_ms(1, 5)
    fn, _ = frame_ms(cores, 5)
    frames = {"keypoints": int(len(sel)), "gn_iterations": int(sf1.num_iters),
              "port": {"ms_per_frame_1_thread": f1, "frames_per_sec_1_thread": 1e3 / f1, "ms_per_frame": fn, "frames_per_sec": 1e3 / fn, "cores": cores}}
    try:
        from oracle import ref as oref
        if oref.available():
            t0 = time.perf_counter()
            rmap = oref.Map(resolutions=[(0.8, 0.1, 30)], default_radius=0.75)
            for s0 in range(0, len(inp["map_points"]), 2_000_000):
                rmap.insert(inp["map_points"][s0:s0 + 2_000_000])
            build_s = time.perf_counter() - t0
            ropt = oref.Options(solver="GN", num_iters_icp=5, threshold_orientation_norm=0.1)
            rprior = oref.Prior(previous_pose=np.concatenate([[0, 0, 0, 1], inp["prev_b"], [0, 0, 0, 1], inp["prev_e"]]).astype(np.float64))
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                pose_ref, _, s_ref = oref.register(rmap, raw_f, world_f, t_f, pose_f, inp["tbe"], ropt, rprior)
                ts.append((time.perf_counter() - t0) * 1e3)
            frames["reference"] = {"ms_per_frame": float(np.median(ts)), "frames_per_sec": 1e3 / float(np.median(ts)), "cores": 1,
                                   "gn_iterations": int(s_ref.num_iters), "map_build_seconds": build_s,
                                   "what": "CT_ICP_Registration::Register of the reference's own sources (oracle/_ref/libctgn_ref.so, "
                                           "compiled against header shims), its MultipleResolutionVoxelMap holding the same map points",
                                   "caveat": "the linear algebra underneath is oracle/shims/mini_eigen.h — a plain-C++ stand-in for Eigen, NOT vectorised "
                                             "(a stock Eigen build runs SSE2 packets): the reference's own arithmetic would be somewhat faster than this"}
    except Exception as e:        # noqa: BLE001 — the reference build is optional on a box
        frames["reference"] = {"unavailable": f"{type(e).__name__}: {e}"}
    # the steps either side of the path on one core, as the reference runs them (only its undistortion loop is OpenMP)
    om2 = orc.Map(resolutions=[(0.8, 0.1, 30)], default_radius=0.75)
    om2.insert(inp["map_points"])
    world_cpu = np.zeros_like(inp["raw"])
    orc.transform_points(inp["pose_gt"], inp["tbe"], inp["t"], inp["raw"], num_threads=cores, out=world_cpu)      # threads started, output pages touched: as for the GPU call
    t0 = time.perf_counter()
    keep = np.sort(orc.grid_sampling(inp["raw"], 0.5))
    t1 = time.perf_counter()
    world = orc.transform_points(inp["pose_gt"], inp["tbe"], inp["t"], inp["raw"], num_threads=cores, out=world_cpu)
    t2 = time.perf_counter()
    om2.remove_far(inp["pose_gt"][11:14], 100.0)
    kept = om2.insert(world[keep])
    t3 = time.perf_counter()
    stages = {"grid_sampling_ms": (t1 - t0) * 1e3, "undistortion_ms": (t2 - t1) * 1e3, "map_update_ms": (t3 - t2) * 1e3,
              "cores": f"1 (undistortion: {cores}, the reference's OpenMP loop)", "sampled": int(len(keep)), "inserted": int(np.count_nonzero(kept)), "map_points_after": int(om2.num_points())}
    return {"value": v_n, "unit": "keypoints/s", "cores": cores, "kind": "port", "frame_stages": stages, "frames_per_sec": frames,
            "reference_shaped": {"value": ref_n, "single_thread_value": ref_1, "cores": cores,
                                 "note": "accumulation pass only (search + normal + residual + sums), std::unordered_map<Voxel, "
                                         "vector<80 B record>> + std::priority_queue: the reference's container shapes"},
            "robust_route": {"ms_per_frame": robust_ms, "cores": 1, "icp_iterations": s_r.num_iters,
                             "residual_blocks": s_r.num_residuals_used, "pose": [float(v) for v in pose_r]},
            "sample": f"oracle GN loop, {n} keypoints x {iters_n} iterations, OpenMP over keypoints on {cores} threads "
                      f"(same map and sweep as the GPU run)",
            "single_thread_value": v_1,
            "single_thread_note": "serial keypoint loop, as the reference executes GN (no OpenMP at ct_icp.cpp:753)"}


if __name__ == "__main__":
    main()
