#!/usr/bin/env python3
"""bench.py — registered keypoints/s per Gauss–Newton iteration of the CT-ICP registration path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under torch.distributed.run, one
rank per GPU. A "step" is ONE GN iteration (neighbour search + covariance/normal + residual/Jacobian + reduction + 12x12 solve +
pose update, ct_icp.cpp:745-981) over the resident keypoint batch. Rank 0 prints one JSON line.

N = 1  workload B2 = BASELINE.json configs[1]: a KITTI-00-like HDL-64E sweep (~132 k returns, every return a keypoint: the
       throughput regime of SURVEY.md section 8d) registered against a STEADY-STATE driving-profile local map (0.8 m x 30 pts,
       radius 0.75 => 27 voxels per query, k = 20, everything within the 100 m eviction radius of an open residential scene:
       ~3.3 x 10^5 voxels, searched level ~260 MB — larger than L2 + Infinity Cache). `--workload B2-small` keeps round 1's
       20-frame street-canyon map (6.8 k voxels, 5 MB: an L2-resident best case).
N > 1  config D (BASELINE.json configs[3]) STRONG scaling: ONE dense 2 M-keypoint scan, sorted by home voxel, cut into N
       contiguous chunks (map replicated), one ncclAllReduce of the 96-double packed system per iteration issued by the
       library (ctgn_solve_sharded's launch sequence). A weak-scaling line (one B2 sweep per rank) rides along.
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

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
B_KP, B_SLOT, B_PT = 32, 16, 24    # algorithmic bytes: keypoint record, hash slot, map point (FP64 xyz storage)
CLOCK_WARM = 150               # untimed iterations that bring the clocks up before the W + K steps of the contract


def _rocprof():
    import shutil
    return shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)


def collect_pmc(args, timeout: int = 300):
    """Counters of the dominant kernel, collected LIVE by rocprofv3 passes (`--kernel-trace --pmc ...`, one pass per counter group,
    no other trace domain — MI355X_MICROARCH.md "rocprofv3 PMC slots") over a short inner run of this same script and workload.
    Returns (dict of per-launch means, source / reason string)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = _rocprof()
    if exe is None:
        return {}, "rocprofv3 not found"
    kernel = "k_accumulate_lane" if args.variant == 1 else "k_accumulate_rows"
    groups = [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAVES"]]
    vals = {}
    for counters in groups:
        out_dir = tempfile.mkdtemp(prefix="ctgn_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", out_dir, "-o", "pmc", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--inner", "--steps", str(min(args.steps, 20)), "--warmup", "0", "--workload", args.workload,
               "--variant", str(args.variant), "--map-frames", str(args.map_frames), "--order", args.order]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=timeout, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, check=False)
            got = {c: [] for c in counters}
            for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") in got:
                        got[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for c, rows in got.items():
                if not rows:
                    return vals, f"rocprofv3 --pmc {c}: no rows for {kernel}"
                rows = rows[len(rows) // 2:]                 # the second half of the inner run: clocks and caches are warm
                vals[c] = sum(rows) / len(rows)
        except Exception as e:        # noqa: BLE001 — measurement nicety: never fail the bench over it
            return vals, f"rocprofv3 --pmc {' '.join(counters)}: {type(e).__name__}"
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    return vals, "live: rocprofv3 --kernel-trace --pmc, 3 passes (FETCH_SIZE | WRITE_SIZE | SQ_*), means over the second half of the launches"


def make_inputs(rank: int, map_frames: int, cache_dir: str = os.path.join(ROOT, ".bench_cache")):
    """Deterministic config-B inputs: map insert list (world points of `map_frames` preceding sweeps after the 0.5 m
    frame grid) + the sweep to register. Cached as .npz because ray-casting 21 sweeps in NumPy takes ~30 s."""
    from ct_icp_amd import synthetic as syn
    tag = f"ctgn_bench_B_v3_r{rank}_m{map_frames}.npz"
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, tag)
    if os.path.exists(path):
        d = np.load(path)
        return {k: d[k] for k in d.files}
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
    try:
        np.savez(path, **out)
    except OSError:
        pass
    return out


def make_inputs_dense(rank: int, length: float = 2400.0, seed: int = 3, n_kp: int = 1_000_000):
    """Config-D-like dense workload built analytically (no ray casting): a 2.4 km street whose ground and two facades are
    sampled densely enough to fill the 0.5 m x 40-point voxels, so the device map is ~0.5 GB (>> 256 MB Infinity Cache), and
    ~1 M keypoints spread over the WHOLE map, so one accumulate launch touches the whole working set."""
    rng = np.random.default_rng(seed)
    def plane(n, fixed_axis, fixed_val, r0, r1):
        p = np.empty((n, 3))
        free = [a for a in range(3) if a != fixed_axis]
        p[:, free[0]] = rng.uniform(r0[0], r0[1], n)
        p[:, free[1]] = rng.uniform(r1[0], r1[1], n)
        p[:, fixed_axis] = fixed_val + rng.normal(0, 0.01, n)
        return p
    dens = 260                                                  # points / m^2: ~65 per 0.5 m voxel face before the min-distance rule
    ground = plane(int(length * 24 * dens), 2, 0.0, (0, length), (-12, 12))
    wall_l = plane(int(length * 8 * dens), 1, 12.0, (0, length), (0, 8))
    wall_r = plane(int(length * 8 * dens), 1, -12.0, (0, length), (0, 8))
    map_points = np.concatenate([ground, wall_l, wall_r])
    kp = np.concatenate([plane(n_kp // 2, 2, 0.0, (5, length - 5), (-11.5, 11.5)),
                         plane(n_kp // 4, 1, 12.0, (5, length - 5), (0.3, 7.7)),
                         plane(n_kp // 4, 1, -12.0, (5, length - 5), (0.3, 7.7))])
    kp = kp[np.random.default_rng(seed + 17 * rank + 1).permutation(len(kp))]
    # identity begin/end pose at the origin: raw == world, the solver then estimates a small correction
    pose = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0], float)
    t = np.linspace(0.0, 1.0, len(kp))
    return dict(map_points=map_points, raw=kp, t=t, pose_gt=pose, tbe=np.array([0.0, 1.0]), prev_b=np.zeros(3), prev_e=np.zeros(3))


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
    tag = f"ctgn_bench_B2L_v1_r{rank}.npz"
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, tag)
    if os.path.exists(path):
        d = np.load(path)
        scan = {k: d[k] for k in d.files}
    else:
        dirs, rel_t = syn.lidar_pattern("hdl64")
        sc = syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, 1), 0.1, 0.2, noise=0.02, seed=1000 + 17 * rank)
        scan = dict(raw=sc.raw, t=sc.t, pose_gt=sc.pose_gt, tbe=sc.t_begin_end)
        try:
            np.savez(path, **scan)
        except OSError:
            pass
    return dict(map_points=map_points, prev_b=knots[0, 4:7], prev_e=knots[1, 4:7], **scan)


def shard_of(inp, resolution, rank, world, pose0):
    """Config D sharding: global sort of the keypoints by home voxel, contiguous chunk per rank."""
    from ct_icp_amd import se3
    from ct_icp_amd.distributed import home_voxel_order, shard_bounds
    world0 = se3.ct_transform(pose0, inp["tbe"], inp["t"], inp["raw"])
    order = home_voxel_order(world0, resolution)
    lo, hi = shard_bounds(len(order), world, rank)
    idx = order[lo:hi]
    return inp["raw"][idx], inp["t"][idx], world0[idx]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--map-frames", type=int, default=20, help="B2-small / B1: sweeps accumulated into the street-canyon map")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default=None, choices=["B2", "B2-small", "B1", "D"],
                    help="B2 (default at --gpus 1): all returns of the HDL-64E sweep as keypoints over the steady-state ~260 MB map; "
                         "B2-small: the same regime over round 1's 20-frame, 5 MB street map; B1: the reference's keypoint count "
                         "(1.5 m grid of the 0.5 m-subsampled frame, latency regime) over the B2-small map; D (default at --gpus > 1): "
                         "dense analytic street, 0.5 m x 40-pt map ~0.5 GB, 125 voxels per query, 1 M keypoints (2 M when sharded)")
    ap.add_argument("--ablate", type=int, default=0, help="measurement hook: skip kernel phases (invalid results)")
    ap.add_argument("--force-dist", action="store_true", help="use the sharded (all-reduce) loop even with one rank")
    ap.add_argument("--torch-collective", action="store_true", help="sharded loop with torch.distributed.all_reduce between stepwise calls "
                                                                   "instead of the library's own ncclAllReduce")
    ap.add_argument("--cpu-sample", type=int, default=0, help="keypoints in the CPU baseline sample (0 = all)")
    ap.add_argument("--order", default="auto", choices=["auto", "on", "off"],
                    help="home-voxel ordering of the GN kernels' work (ctgn_set_ordering); auto = the library's cost model")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 PMC passes (roofline.traffic, wait fractions)")
    ap.add_argument("--no-extras", action="store_true", help="skip frames/s, robust route, frame stages")
    ap.add_argument("--clock-warm", type=int, default=CLOCK_WARM)
    ap.add_argument("--inner", action="store_true", help="the run the PMC passes profile: timed loop only, no extras")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, the all-reduce goes "
                         "through torch.distributed on the host; numbers are not scaling numbers)")
    args = ap.parse_args()
    if args.inner:
        args.no_pmc = args.no_cpu_baseline = args.no_extras = True
        args.clock_warm = min(args.clock_warm, 30)

    import torch
    import ct_icp_amd as cia
    from ct_icp_amd import se3, synthetic as syn

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.workload is None:
        args.workload = "B2" if world == 1 else "D"
    if args.dist_backend == "gloo":
        args.torch_collective = True
        local_rank = local_rank % max(1, torch.cuda.device_count())
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

    # ------------------------------------------------------------------------------------------------ inputs
    if args.workload == "D":
        inp = make_inputs_dense(0, n_kp=2_000_000 if world > 1 else 1_000_000)      # ONE scan for all ranks (strong scaling)
        res_param, radius = cia.ResolutionParam(0.5, 0.03, 40), 0.8          # config D map: {0.5 m, 40 pts, 0.03 m}
    elif args.workload == "B2":
        inp = make_inputs_large(rank)
        res_param, radius = cia.ResolutionParam(0.8, 0.1, 30), 0.75         # driving profile
    else:
        inp = make_inputs(rank, args.map_frames)
        res_param, radius = cia.ResolutionParam(0.8, 0.1, 30), 0.75
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[res_param], default_radius=radius, device=local_rank))
    for s0 in range(0, len(inp["map_points"]), 2_000_000):
        gm.InsertPointCloud(inp["map_points"][s0:s0 + 2_000_000])
    gm.Sync()
    pose0 = syn.perturb_pose(inp["pose_gt"], 0.003, 0.03, seed=4)
    raw, t = inp["raw"], inp["t"]
    if args.workload == "B1":                                  # the reference's two-stage grid sampling (odometry.cpp:349,538)
        sel = syn.grid_sample_indices(raw, 0.5)
        sel = sel[syn.grid_sample_indices(raw[sel], 1.5)]
        raw, t = raw[sel], t[sel]
    if args.workload == "D" and world > 1:
        raw, t, world0 = shard_of(inp, res_param.resolution, rank, world, pose0)
    else:
        world0 = se3.ct_transform(pose0, inp["tbe"], t, raw)
    n_kp = len(t)
    mm = cia.PreviousFrameMotionModel()
    mm.previous_frame = cia.TrajectoryFrame.from_pose14(np.concatenate([[0, 0, 0, 1], inp["prev_b"], [0, 0, 0, 1], inp["prev_e"]]), 0, 0)

    def options(iters):   # threshold 0: no early stop, exactly `iters` GN iterations
        return cia.CTICPOptions(solver=cia.GN, num_iters_icp=iters, threshold_orientation_norm=0.0, debug_print=False)

    sh = None
    if sharded:
        from ct_icp_amd.distributed import ShardedGnSolver, allreduce_system
        sh = ShardedGnSolver(gm, library_collective=not args.torch_collective)
        solver = sh.solver
    else:
        solver = cia.GnSolver(gm)
    solver.set_variant(args.variant)
    solver.set_ordering({"auto": -1, "off": 0, "on": 1}[args.order])
    solver.set_ablation(args.ablate)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def iterate(k):
        """k GN iterations of the running loop, enqueued without synchronising."""
        if sh is not None and args.torch_collective:
            for _ in range(k):
                solver.gn_accumulate()
                allreduce_system(sh.system)
                solver.gn_solve_update()
        else:
            solver.gn_iterate(k, sharded=sh is not None)

    # ------------------------------------------------------------------------------------------------ the timed loop
    # one upload, then ONE running GN loop: CLOCK_WARM iterations (clocks), W warm-up steps (contract), K timed steps — same launch
    # sequence and resident data throughout, nothing but a barrier + device synchronisation between the three segments
    total_iters = args.clock_warm + args.warmup + args.steps
    solver.set_keypoints(raw, world0, t)                       # inputs resident in HBM before the timed region
    solver.set_profiling(False)
    solver.gn_begin(pose0, inp["tbe"], options(total_iters), mm)
    iterate(args.clock_warm)
    iterate(args.warmup)
    sync_all()
    solver.set_profiling(True)                                 # HIP-event pair around every neighbour-search launch from here on
    solver.kernel_timing(reset=True)
    sync_all()
    t0 = time.perf_counter()
    iterate(args.steps)
    sync_all()
    dt = time.perf_counter() - t0
    pose1, summ = solver.gn_end()[:2]
    kern_ms, kern_launches = solver.kernel_timing(reset=True)
    solver.set_profiling(False)
    assert args.ablate or (summ.success and summ.num_iters == total_iters), summ

    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        nn = torch.tensor([n_kp], dtype=torch.float64, device="cuda")
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        total_kp = int(nn.item())
    else:
        total_kp = n_kp

    # ------------------------------------------------------------------------------------------------ accounting (untimed)
    # algorithmic bytes of one neighbour-search launch: B_kp per keypoint + B_slot per hash probe + B_pt per map point.
    # "all" charges every point of all 27 / 125 sweep voxels (round 1's figure); "requested" charges what the kernel asks the
    # memory system for — probes of voxels it did not cull, points it streamed — counted by its instrumented instantiation.
    solver.set_keypoints(raw, world0, t)
    probed, hit, points = solver.count_traffic()
    alg_all = n_kp * B_KP + probed * B_SLOT + points * B_PT
    alg_req, req_probes, req_points = None, None, None
    if args.variant == 0 and not args.ablate and not args.inner and gm.SearchParamsFromRadiusSearch()[2] in (1, 2):
        solver.set_variant(3)
        solver.traffic_counters(reset=True)
        # counted on an iteration of the steady state the timed loop is in: every search but the first of a solve is bounded by the
        # previous one's k-th neighbour distance, which culls most of the sweep
        solver.gn_begin(pose0, inp["tbe"], options(total_iters), mm)
        iterate(10)
        solver.traffic_counters(reset=True)
        iterate(1)
        solver.gn_end()
        req_probes, req_points = solver.traffic_counters(reset=True)
        solver.set_variant(0)
        alg_req = n_kp * B_KP + req_probes * B_SLOT + req_points * B_PT

    # parity on the very workload that was timed: 5 GN iterations (the driving profile's budget) on the GPU and through the oracle
    parity = None
    if rank == 0 and not args.inner and not args.ablate:
        from oracle import oracle as orc
        om = orc.Map(resolutions=[(res_param.resolution, res_param.min_distance_between_points, res_param.max_num_points)], default_radius=radius)
        for s0 in range(0, len(inp["map_points"]), 2_000_000):
            om.insert(inp["map_points"][s0:s0 + 2_000_000])
        solver.set_keypoints(raw, world0, t)
        if sh is not None and world > 1:
            pose_g, summ_g = None, None                        # every rank has to take part: done below, collectively
        else:
            pose_g, summ_g = (sh.solve(pose0, inp["tbe"], options(5), mm) if sh is not None else solver.solve(pose0, inp["tbe"], options(5), mm))[:2]
        parity = dict(om=om)
    if sh is not None and world > 1 and not args.inner and not args.ablate:
        solver.set_keypoints(raw, world0, t)
        pose_g, summ_g = sh.solve(pose0, inp["tbe"], options(5), mm)[:2]
    if parity is not None:
        om = parity.pop("om")
        if args.workload == "D" and world > 1:
            o_raw, o_t = inp["raw"], inp["t"]
            o_w0 = se3.ct_transform(pose0, inp["tbe"], o_t, o_raw)
        else:
            o_raw, o_t, o_w0 = raw, t, world0
        op = orc.MotionPrior(previous_begin_tr=inp["prev_b"], previous_end_tr=inp["prev_e"])
        pose_o, _, so = orc.register_gn(om, o_raw, o_w0, o_t, pose0, inp["tbe"], orc.Options(num_iters_icp=5, threshold_orientation_norm=0.0), op,
                                        heap_mode=0, num_threads=usable_cores())
        tr, rot = se3.pose_error(pose_g, pose_o)
        parity = {"gpu_vs_oracle_m_rad": [tr, rot], "iterations": 5, "n_used_gpu": int(summ_g.num_residuals_used),
                  "n_used_oracle": int(so.num_residuals_used), "tolerance_m_rad": [1e-4, 1e-4]}
        assert tr < 1e-4 and rot < 1e-4 and summ_g.num_residuals_used == so.num_residuals_used, parity
        parity["om"] = om

    result = None
    if rank == 0:
        value = total_kp * args.steps / dt
        t_k = kern_ms * 1e-3
        pmc, pmc_src = ({}, "skipped")
        if world == 1 and not args.no_pmc and not args.ablate and dist is None:
            pmc, pmc_src = collect_pmc(args)
        traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0 if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc else None
        alg = alg_req if alg_req is not None else alg_all
        achieved = alg / t_k / 1e9 if t_k > 0 else 0.0
        level_mb = (gm.NumVoxels(0) * res_param.max_num_points * 24 + (1 << int(np.ceil(np.log2(max(gm.NumVoxels(0), 1) * 4)))) * 16) / 1e6
        roof = {"bound": "latency/issue", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": pmc_src,
                "achieved_definition": "algorithmic bytes the kernel requests per launch (32 B/keypoint + 16 B per hash probe it issues + 24 B per map "
                                       "point it streams, counted by the instrumented instantiation) / HIP-event launch time; served mostly by "
                                       "L2 / Infinity Cache, hence the separate HBM counter figure",
                "hbm_counter_gbs": (traffic / t_k / 1e9) if traffic and t_k > 0 else None,
                "hbm_counter_frac": (traffic / t_k / 1e9 / HBM_PEAK_GBS) if traffic and t_k > 0 else None,
                "kernel": "k_accumulate_rows (voxel-hash neighbour search + k-nearest selection)" if args.variant != 1 else "k_accumulate_lane",
                "kernel_ms_avg": kern_ms, "kernel_launches": kern_launches,
                "alg_bytes_per_launch": alg, "alg_bytes_per_keypoint": alg / n_kp,
                "alg_bytes_per_launch_all_sweep_voxels": alg_all,     # SURVEY.md 8d's formula with every voxel of the sweep: what an unculled search reads
                "probes_issued_per_keypoint": (req_probes / n_kp) if req_probes is not None else None,
                "points_streamed_per_keypoint": (req_points / n_kp) if req_points is not None else None,
                "voxels_in_sweep_per_keypoint": probed / n_kp, "voxels_occupied_per_keypoint": hit / n_kp,
                "points_in_sweep_per_keypoint": points / n_kp}
        if "SQ_WAVE_CYCLES" in pmc and pmc["SQ_WAVE_CYCLES"] > 0:
            wc = pmc["SQ_WAVE_CYCLES"]
            roof.update({"wait_frac": pmc["SQ_WAIT_ANY"] / wc, "issue_stall_frac": pmc["SQ_WAIT_INST_ANY"] / wc,
                         "active_frac": pmc["SQ_ACTIVE_INST_ANY"] / wc, "valu_busy": pmc["SQ_ACTIVE_INST_VALU"] / wc,
                         "waves_per_launch": pmc.get("SQ_WAVES")})
        names = {"B2": "config B2: synthetic HDL-64E sweep (KITTI-00-like, every return a keypoint) over the steady-state driving-profile map "
                       "of an open residential scene: 0.8 m x 30 pts, radius 0.75 (27 voxels), k=20, everything within the 100 m eviction radius",
                 "B2-small": "config B2-small: the same sweep regime over round 1's street-canyon map of "
                             f"{args.map_frames} frames (L2-resident best case)",
                 "B1": "config B1: street-canyon sweep and map, keypoints = 1.5 m grid of the 0.5 m-subsampled frame (the reference's keypoint "
                       "count; latency regime)",
                 "D": "config D: dense analytic street, map {0.5 m, 40 pts} ~0.5 GB (> Infinity Cache), keypoints spread over the whole map, "
                      "radius 0.8 (125 voxels), k=20" + ("; ONE 2 M-keypoint scan sorted by home voxel and cut into contiguous chunks" if world > 1 else "")}
        result = {
            "metric": "registered keypoints/sec per GN iter", "value": value, "unit": "keypoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if (args.workload == "D" and world > 1) else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": names[args.workload], "workload_id": args.workload,
                       "keypoints_per_gpu": n_kp, "keypoints_total": total_kp, "map_points": int(gm.NumPoints()),
                       "map_voxels": int(gm.NumVoxels(0)), "searched_level_mb": level_mb, "n_used_last_iter": summ.num_residuals_used,
                       "parallelism": "single GPU" if world == 1 else f"keypoints sharded x{world} (home-voxel sort, contiguous chunks), map replicated, "
                                                                      + ("1 ncclAllReduce(96 f64) per iteration issued by the library" if not args.torch_collective else
                                                                         f"1 torch.distributed all_reduce(96 f64) per iteration, backend {args.dist_backend}"),
                       "kernel_variant": args.variant, "keypoint_ordering": args.order},
            "clock_warmup_iterations": args.clock_warm,
            "clock_warmup_note": "untimed iterations of the same running GN loop immediately before the W warm-up and K timed steps (no upload in "
                                 "between): the GPU leaves its idle clocks only under sustained load",
            "frames_per_sec_equiv": 1.0 / (dt / args.steps * 5) if dt > 0 else None,   # 5 GN iterations per frame (driving profile)
            "roofline": roof,
        }
        om = None
        if parity is not None:
            om = parity.pop("om")
            result["parity_m_rad"] = parity["gpu_vs_oracle_m_rad"]
            result["parity"] = parity
        if args.workload in ("B2", "B2-small") and world == 1 and not args.ablate and not args.no_extras and args.variant != 1:
            result["frames_per_sec"] = measure_frames_per_sec(cia, gm, inp, syn, se3, mm)
            result["robust_route"] = measure_robust_frames_per_sec(cia, gm, inp, syn, se3, mm)
            result["frame_stages"] = fs = measure_frame_stages(cia, inp, syn, se3, local_rank)
            # the whole per-frame loop of Odometry::DoRegister on this frame, every data-parallel step through the library with host
            # buffers in and out: the steps either side + one Register call on the sampled keypoints
            for route, reg_ms in (("gn", result["frames_per_sec"]["ms_per_frame"]), ("robust", result["robust_route"]["ms_per_frame"])):
                ms = fs["grid_sampling_ms"] + fs["keypoint_sampling_ms"] + reg_ms + fs["undistortion_ms"] + fs["map_update_ms"]
                result.setdefault("frame_by_stage_calls", {})[route] = {"ms_per_frame": ms, "frames_per_sec": 1e3 / ms}
            # the same frame as ONE ctgn_frame_register + ctgn_frame_update_map (scan resident on the device)
            result["frame_pipeline"] = fs.pop("frame_pipeline")
            result["frame_pipeline"]["frames_per_sec"] = 1e3 / result["frame_pipeline"]["frame_ms"]
        if not args.no_cpu_baseline and args.workload in ("B2", "B2-small") and world == 1:      # rank 0 at N = 1 only
            result["cpu_baseline"] = cpu_baseline(inp, pose0, world0, args, om)
            if result["cpu_baseline"]["value"]:
                result["gpu_over_cpu"] = value / world / result["cpu_baseline"]["value"]
                result["gpu_over_cpu_reference_shaped"] = value / world / result["cpu_baseline"]["reference_shaped"]["value"]
            if "robust_route" in result:                   # same frame, same settings: the two poses must agree
                rr, cr = result["robust_route"], result["cpu_baseline"]["robust_route"]
                tr, rot = se3.pose_error(np.array(rr.pop("pose")), np.array(cr.pop("pose")))
                rr["gpu_vs_cpu_pose_m_rad"] = [tr, rot]
                rr["gpu_over_cpu_1core"] = cr["ms_per_frame"] / rr["ms_per_frame"]
        elif "robust_route" in result:
            result["robust_route"].pop("pose", None)

    # ------------------------------------------------------------------------------------------------ N > 1 extras
    if world > 1 and not args.inner:
        # (a) the same scan on ONE GPU (rank 0, the others wait): the denominator of the strong-scaling efficiency
        single = None
        if rank == 0:
            s1 = cia.GnSolver(gm)
            w_all = se3.ct_transform(pose0, inp["tbe"], inp["t"], inp["raw"])
            s1.set_keypoints(inp["raw"], w_all, inp["t"])
            s1.gn_begin(pose0, inp["tbe"], options(args.clock_warm + args.steps), mm)
            s1.gn_iterate(args.clock_warm)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s1.gn_iterate(args.steps)
            torch.cuda.synchronize()
            d1 = time.perf_counter() - t0
            s1.gn_end()
            single = {"value": len(inp["t"]) * args.steps / d1, "ms_per_step": d1 / args.steps * 1e3, "keypoints": int(len(inp["t"])),
                      "note": "the same 2 M-keypoint scan, unsharded, on rank 0's GPU"}
        dist.barrier()
        # (b) weak-scaling line: one B2-small sweep per rank (its own noise realisation), same sharded loop
        winp = make_inputs(rank, args.map_frames)
        gw = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(0.8, 0.1, 30)], default_radius=0.75, device=local_rank))
        gw.InsertPointCloud(winp["map_points"]); gw.Sync()
        shw = ShardedGnSolver(gw, library_collective=not args.torch_collective)
        wp0 = syn.perturb_pose(winp["pose_gt"], 0.003, 0.03, seed=4)
        shw.set_keypoints(winp["raw"], se3.ct_transform(wp0, winp["tbe"], winp["t"], winp["raw"]), winp["t"])
        shw.solver.gn_begin(wp0, winp["tbe"], options(args.clock_warm + args.steps), None)
        def iterate_w(k):
            if args.torch_collective:
                for _ in range(k):
                    shw.solver.gn_accumulate()
                    allreduce_system(shw.system)
                    shw.solver.gn_solve_update()
            else:
                shw.solver.gn_iterate(k, sharded=True)
        iterate_w(args.clock_warm)
        sync_all()
        t0 = time.perf_counter()
        iterate_w(args.steps)
        sync_all()
        dw = time.perf_counter() - t0
        shw.solver.gn_end()
        tw = torch.tensor([dw], dtype=torch.float64, device="cuda")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        nw = torch.tensor([len(winp["t"])], dtype=torch.float64, device="cuda")
        dist.all_reduce(nw, op=dist.ReduceOp.SUM)
        shw.close()
        if rank == 0:
            result["strong_scaling_single_gpu_reference"] = single
            if single:
                result["strong_scaling_efficiency"] = result["value"] / (world * single["value"])
            result["weak_scaling_line"] = {"value": float(nw.item()) * args.steps / float(tw.item()), "ms_per_step": float(tw.item()) / args.steps * 1e3,
                                           "keypoints_per_gpu": int(len(winp["t"])), "workload": "B2-small sweep per rank, sharded loop", "scaling": "weak"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        if sh is not None:
            sh.close()
        dist.barrier()
        dist.destroy_process_group()


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
    for _ in range(7):                                   # 0-3: stage by stage, 4-6: the frame pipeline
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
    for rep, m in enumerate(maps[4:]):
        fp = cia.FramePipeline(m, frame_voxel_size=0.5, sample_voxel_size=1.5)
        regs = []
        for _ in range(4):                               # the registration does not change the map: repeat on the same one
            t0 = time.perf_counter()
            r = fp.register(raw, t, pose0, inp["tbe"], o5, want_all=True, want_sampled=False)
            regs.append((time.perf_counter() - t0) * 1e3)
        lean = []
        for _ in range(4):
            t0 = time.perf_counter()
            fp.register(raw, t, pose0, inp["tbe"], o5, want_all=False, want_sampled=False)
            lean.append((time.perf_counter() - t0) * 1e3)
        fp.update_map(r["pose"][11:14], 100.0, False)    # every frame evicts: the timed update is not the table's first scan
        t0 = time.perf_counter()
        mask = fp.update_map(r["pose"][11:14], 100.0, True)
        upd = (time.perf_counter() - t0) * 1e3
        if rep > 0:
            ptimes.append([min(regs[1:]), min(lean[1:]), upd])
        tr, rot = se3.pose_error(r["pose"], inp["pose_gt"])
        pcounts = {"sampled": int(len(r["sampled_indices"])), "keypoints": int(len(r["keypoint_indices"])),
                   "inserted": int(np.count_nonzero(mask)), "gn_iterations": int(r["summary"].num_iters),
                   "error_vs_ground_truth_m_rad": [tr, rot], "gn_iteration_device_ms": float(r["summary"].avg_duration_iter)}
    pm = np.median(np.array(ptimes), axis=0)
    pipeline = {"register_ms": float(pm[0]), "register_without_full_scan_output_ms": float(pm[1]), "update_map_ms": float(pm[2]),
                "frame_ms": float(pm[0] + pm[2]),
                "includes": "host scan (xyz f64 + t) -> one H2D -> frame + keypoint grid sampling -> 5 GN iterations -> undistortion of "
                            "the sampled frame and of every scan point -> pose, summary, indices, N x 3 f64 world points D2H; then "
                            "far-voxel eviction + insertion of the device-resident sampled frame"}
    pipeline.update(pcounts)
    for rep, m in enumerate(maps[:4]):
        cia.grid_sampling(m, raw, 0.5)                   # a fresh handle sizes its scratch on the first scan-sized call: not a per-frame cost
        m.RemoveElementsFarFromLocation(inp["pose_gt"][11:14], 100.0)     # every frame evicts: the timed update is not the table's first scan
        t0 = time.perf_counter()
        keep = np.sort(cia.grid_sampling(m, raw, 0.5))
        t1 = time.perf_counter()
        kp = cia.grid_sampling(m, raw[keep], 1.5)                    # keypoint selection (odometry.cpp:538)
        t1b = time.perf_counter()
        world = cia.transform_points(m, raw, t, inp["pose_gt"], inp["tbe"])
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
    pose_r, _, s_r = orc.register_robust(om, inp["raw"][sel], inp["t"][sel], pose0, inp["tbe"], ro, rp, heap_mode=1)
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
                                           "compiled against header shims), its MultipleResolutionVoxelMap holding the same map points"}
    except Exception as e:        # noqa: BLE001 — the reference build is optional on a box
        frames["reference"] = {"unavailable": f"{type(e).__name__}: {e}"}
    # the steps either side of the path on one core, as the reference runs them (only its undistortion loop is OpenMP)
    om2 = orc.Map(resolutions=[(0.8, 0.1, 30)], default_radius=0.75)
    om2.insert(inp["map_points"])
    t0 = time.perf_counter()
    keep = np.sort(orc.grid_sampling(inp["raw"], 0.5))
    t1 = time.perf_counter()
    world = orc.transform_points(inp["pose_gt"], inp["tbe"], inp["t"], inp["raw"], num_threads=cores)
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
