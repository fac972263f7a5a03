#!/usr/bin/env python3
"""bench.py — registered keypoints/s per Gauss–Newton iteration of the CT-ICP registration path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run, one rank per GPU over RCCL. A "step" is ONE GN iteration (neighbour search + covariance/normal +
residual/Jacobian + reduction + 12x12 solve + pose update) over the resident keypoint batch. Rank 0 prints one JSON line.

Workload at N = 1: BASELINE.json configs[1] — KITTI-00-like HDL-64E sweep (~130 k returns) over a procedural street,
driving profile (0.8 m map x 30 pts, radius 0.75 => 27 voxels / query, k = 20), every return used as a keypoint
(the throughput regime B2 of SURVEY.md section 8d). Inputs are synthetic (no dataset on the box) and already resident
in HBM when the timed region starts. At N > 1 every rank holds its own ~130 k-keypoint shard of a denser scan of the same
scene (weak scaling) and the packed normal equations are all-reduced once per iteration.
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


def collect_pmc_traffic(args, timeout: int = 240):
    """HBM bytes per launch of the dominant kernel, collected LIVE: two rocprofv3 passes (`--kernel-trace --pmc FETCH_SIZE`,
    then `... WRITE_SIZE`; separate passes and no other trace domain, as MI355X_MICROARCH.md prescribes) over a short inner
    run of this same script and workload. bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE under-counts wide reads by
    half on gfx950. Returns (bytes or None, source / reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    kernel = "k_accumulate_lane" if args.variant == 1 else "k_accumulate_rows"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out_dir = tempfile.mkdtemp(prefix="ctgn_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out_dir, "-o", "pmc", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--inner", "--steps", str(args.steps), "--warmup", "0", "--workload", args.workload,
               "--variant", str(args.variant), "--map-frames", str(args.map_frames), "--order", args.order]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=timeout, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, check=False)
            rows = []
            for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                rows += [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                         if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter]
            if not rows:
                return None, f"rocprofv3 --pmc {counter}: no rows for {kernel}"
            vals[counter] = sum(rows) / len(rows)
        except Exception as e:        # noqa: BLE001 — measurement nicety: never fail the bench over it
            return None, f"rocprofv3 --pmc {counter}: {type(e).__name__}"
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, \
        "live: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes), 2*FETCH_SIZE + WRITE_SIZE"


def pmc_traffic_bytes():
    """Fallback when the live collection is unavailable: the committed rocprofv3 PMC passes of the B2 workload
    (profiles/). Returns (bytes or None, source string)."""
    import glob
    import re
    fetch = write = None
    src = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_pass*.txt"))):
        txt = open(path).read()
        blk = txt.split("ctgn::k_reduce_solve")[0]
        if "k_accumulate_rows" not in blk:
            continue
        m = re.search(r"FETCH_SIZE\s+n=\s*\d+\s+mean=([0-9.e+]+)", blk)
        if m:
            fetch, _ = float(m.group(1)), src.append(os.path.basename(path))
        m = re.search(r"WRITE_SIZE\s+n=\s*\d+\s+mean=([0-9.e+]+)", blk)
        if m:
            write, _ = float(m.group(1)), src.append(os.path.basename(path))
    if fetch is None or write is None:
        return None, None
    return (2.0 * fetch + write) * 1024.0, "profiles/" + "+".join(sorted(set(src[-2:]))) + " (2*FETCH_SIZE + WRITE_SIZE)"


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


def make_inputs_dense(rank: int, length: float = 2400.0, seed: int = 3):
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
    n_kp = 1_000_000
    kp = np.concatenate([plane(n_kp // 2, 2, 0.0, (5, length - 5), (-11.5, 11.5)),
                         plane(n_kp // 4, 1, 12.0, (5, length - 5), (0.3, 7.7)),
                         plane(n_kp // 4, 1, -12.0, (5, length - 5), (0.3, 7.7))])
    kp = kp[np.random.default_rng(seed + 17 * rank + 1).permutation(len(kp))]
    # identity begin/end pose at the origin: raw == world, the solver then estimates a small correction
    pose = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0], float)
    t = np.linspace(0.0, 1.0, len(kp))
    return dict(map_points=map_points, raw=kp, t=t, pose_gt=pose, tbe=np.array([0.0, 1.0]), prev_b=np.zeros(3), prev_e=np.zeros(3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--map-frames", type=int, default=20)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="B2", choices=["B2", "B1", "D"],
                    help="B2: all returns of the HDL-64E sweep as keypoints (default, throughput regime); B1: the reference's "
                         "keypoint count (1.5 m grid of the 0.5 m-subsampled frame, latency regime); D: dense synthetic workload "
                         "whose map working set exceeds the 256 MB Infinity Cache (HBM-bound evidence)")
    ap.add_argument("--ablate", type=int, default=0, help="measurement hook: skip kernel phases (invalid results)")
    ap.add_argument("--force-dist", action="store_true", help="use the sharded (all-reduce) loop even with one rank")
    ap.add_argument("--presort", action="store_true", help="experiment: sort the keypoints by home voxel on the host")
    ap.add_argument("--cpu-sample", type=int, default=0, help="keypoints in the CPU baseline sample (0 = all)")
    ap.add_argument("--order", default="auto", choices=["auto", "on", "off"],
                    help="home-voxel ordering of the GN kernels' work (ctgn_set_ordering); auto = the library's cost model")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the two rocprofv3 PMC passes for roofline.traffic")
    ap.add_argument("--inner", action="store_true", help="the run the PMC passes profile: timed loop only, no extras")
    args = ap.parse_args()
    if args.inner:
        args.no_pmc = args.no_cpu_baseline = True

    import torch
    import ct_icp_amd as cia
    from ct_icp_amd import se3, synthetic as syn

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", RANK="0", WORLD_SIZE="1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    if args.workload == "D":
        inp = make_inputs_dense(rank)
        res_param, radius = cia.ResolutionParam(0.5, 0.03, 40), 0.8          # config D map: {0.5 m, 40 pts, 0.03 m}
    else:
        inp = make_inputs(rank, args.map_frames)
        res_param, radius = cia.ResolutionParam(0.8, 0.1, 30), 0.75         # driving profile
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[res_param], default_radius=radius, device=local_rank))
    for s0 in range(0, len(inp["map_points"]), 2_000_000):
        gm.InsertPointCloud(inp["map_points"][s0:s0 + 2_000_000])
    gm.Sync()
    raw, t = inp["raw"], inp["t"]
    if args.workload == "B1":                                  # the reference's two-stage grid sampling (odometry.cpp:349,538)
        sel = syn.grid_sample_indices(raw, 0.5)
        sel = sel[syn.grid_sample_indices(raw[sel], 1.5)]
        raw, t = raw[sel], t[sel]
    n_kp = len(t)
    pose0 = syn.perturb_pose(inp["pose_gt"], 0.003, 0.03, seed=4)
    world0 = se3.ct_transform(pose0, inp["tbe"], t, raw)
    mm = cia.PreviousFrameMotionModel()
    mm.previous_frame = cia.TrajectoryFrame.from_pose14(np.concatenate([[0, 0, 0, 1], inp["prev_b"], [0, 0, 0, 1], inp["prev_e"]]), 0, 0)

    def options(iters):   # threshold 0: no early stop, exactly `iters` GN iterations
        return cia.CTICPOptions(solver=cia.GN, num_iters_icp=iters, threshold_orientation_norm=0.0, debug_print=False)

    if dist is not None:
        from ct_icp_amd.distributed import ShardedGnSolver
        sh = ShardedGnSolver(gm)
        solver = sh.solver
        run = lambda iters: sh.solve(pose0, inp["tbe"], options(iters), mm)
    else:
        solver = cia.GnSolver(gm)
        run = lambda iters: solver.solve(pose0, inp["tbe"], options(iters), mm)[:2] + (None,)
    if args.presort:
        vox = np.trunc(world0 / 0.8).astype(np.int64)
        order = np.lexsort((vox[:, 2], vox[:, 1], vox[:, 0]))
        raw, t, world0 = raw[order], t[order], world0[order]
    solver.set_variant(args.variant)
    solver.set_ordering({"auto": -1, "off": 0, "on": 1}[args.order])
    solver.set_ablation(args.ablate)
    solver.set_keypoints(raw, world0, t)                       # inputs resident in HBM before the timed region
    probed, hit, points = solver.count_traffic()
    alg_bytes = n_kp * B_KP + probed * B_SLOT + points * B_PT   # per accumulate launch

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        # W untimed steps through the stepwise entry points, announced with the timed run's iteration budget so that they take
        # the path the timed steps take (the library orders an upload only when the budget covers the sort)
        solver.set_keypoints(raw, world0, t)
        solver.gn_begin(pose0, inp["tbe"], options(args.steps), mm)
        for _ in range(args.warmup):
            solver.gn_accumulate()
            if dist is not None:
                from ct_icp_amd.distributed import allreduce_system
                allreduce_system(sh.system)
            solver.gn_solve_update()
        solver.gn_end()
    solver.set_keypoints(raw, world0, t)
    solver.set_profiling(True)
    solver.kernel_timing(reset=True)
    sync_all()
    t0 = time.perf_counter()
    pose1, summ, _ = run(args.steps)
    sync_all()
    dt = time.perf_counter() - t0
    kern_ms, kern_launches = solver.kernel_timing(reset=True)
    solver.set_profiling(False)
    if args.variant == 3 and rank == 0:
        pcs_all = solver.phase_cycles(reset=True)
        pcs, fast_rounds, all_rounds = pcs_all[:8], pcs_all[8], pcs_all[9]
        print(f"fast-path rounds: {fast_rounds} of {all_rounds}; slowest wave {pcs_all[10] / 1e3:.0f} kclk vs mean "
              f"{sum(pcs) / max(pcs_all[11], 1) / 1e3:.0f} kclk over {pcs_all[11]} wave-launches", file=sys.stderr)
        tl = solver.wave_timeline().astype(np.float64)
        if len(tl):
            dur = tl[:, 1] - tl[:, 0]
            frac_fast = tl[:, 2] / np.maximum(tl[:, 3], 1)
            qd = np.percentile(dur, [1, 5, 25, 50, 75, 95, 99, 100]) / 1e3
            # the clock is per XCD: group the waves by start clock (gaps >> kernel length) and look at each group's own timeline
            order = np.argsort(tl[:, 0])
            cuts = np.nonzero(np.diff(tl[order, 0]) > 50 * dur.max())[0] + 1
            groups = np.split(order, cuts)
            spans = [(tl[g, 1].max() - tl[g, 0].min()) / 1e3 for g in groups]
            starts = [(tl[g, 0].max() - tl[g, 0].min()) / 1e3 for g in groups]
            print(f"wave timeline (last launch, {len(tl)} waves, kclk): duration percentiles 1/5/25/50/75/95/99/100 = " +
                  "/".join(f"{v:.0f}" for v in qd) + f"; corr(duration, fast-path share) = {np.corrcoef(dur, frac_fast)[0, 1]:.2f}; "
                  f"{len(groups)} clock domains, span first-start..last-end per domain = " + "/".join(f"{v:.0f}" for v in spans) +
                  ", start spread per domain = " + "/".join(f"{v:.0f}" for v in starts), file=sys.stderr)
        tot = float(sum(pcs)) or 1.0
        names = ["A transform", "B1 probes", "B2 stream", "B2 prunes", "B3 select", "B4 sums", "C normal/jac", "D accumulate"]
        print("phase cycles: " + ", ".join(f"{n}={100 * c / tot:.1f}%" for n, c in zip(names, pcs)) +
              f" | total Mcycles/launch={tot / 1e6 / max(kern_launches, 1):.1f} abs=" +
              ",".join(f"{c / 1e6 / max(kern_launches, 1):.1f}" for c in pcs), file=sys.stderr)
    assert args.ablate or (summ.success and summ.num_iters == args.steps), summ

    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        nn = torch.tensor([n_kp], dtype=torch.float64, device="cuda")
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        total_kp = int(nn.item())
    else:
        total_kp = n_kp

    result = None
    if rank == 0:
        value = total_kp * args.steps / dt
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic, traffic_src = (None, None)
        if world == 1 and not args.no_pmc and not args.ablate and dist is None:
            traffic, traffic_src = collect_pmc_traffic(args)
        if traffic is None and args.variant == 0 and world == 1 and args.workload == "B2":
            why = traffic_src
            traffic, traffic_src = pmc_traffic_bytes()
            if why and traffic_src:
                traffic_src += f" [live collection unavailable: {why}]"
        result = {
            "metric": "registered keypoints/sec per GN iter", "value": value, "unit": "keypoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "config B2: synthetic HDL-64E sweep over procedural street (KITTI-00-like), all returns "
                                   "as keypoints, driving profile map 0.8 m x 30 pts, radius 0.75 (27 voxels), k=20, "
                                   f"{args.map_frames} map frames",
                       "keypoints_per_gpu": n_kp, "keypoints_total": total_kp, "map_points": int(gm.NumPoints()),
                       "map_voxels": int(gm.NumVoxels(0)), "n_used_last_iter": summ.num_residuals_used,
                       "parallelism": "single GPU" if world == 1 else f"keypoints sharded x{world}, 1 all-reduce(96 f64)/iter",
                       "kernel_variant": args.variant, "keypoint_ordering": args.order},
            "frames_per_sec_equiv": 1.0 / (dt / args.steps * 5) if dt > 0 else None,   # 5 GN iterations per frame (driving profile)
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_accumulate_rows (voxel-hash neighbour search + k-nearest selection)" if args.variant != 1
                                   else "k_accumulate_lane",
                         "kernel_ms_avg": kern_ms, "kernel_launches": kern_launches,
                         "alg_bytes_per_launch": alg_bytes,
                         "alg_bytes_per_keypoint": alg_bytes / n_kp,
                         "voxels_probed_per_keypoint": probed / n_kp, "voxels_hit_per_keypoint": hit / n_kp,
                         "points_scanned_per_keypoint": points / n_kp},
        }
        result["config"]["workload_id"] = args.workload
        if args.workload != "B2":
            result["config"]["workload"] = {"B1": "config B1: same sweep and map, keypoints = 1.5 m grid of the 0.5 m-subsampled frame "
                                                  "(the reference's keypoint count; latency regime)",
                                            "D": "config D-like: dense analytic street, map {0.5 m, 40 pts} ~0.5 GB (> Infinity Cache), "
                                                 "1 M keypoints spread over the whole map, radius 0.8 (125 voxels), k=20"}[args.workload]
        if args.workload == "B2" and world == 1 and not args.ablate and not args.inner and args.variant != 1:   # variant 1 has no robust route
            result["frames_per_sec"] = measure_frames_per_sec(cia, gm, inp, syn, se3, mm)
            result["robust_route"] = measure_robust_frames_per_sec(cia, gm, inp, syn, se3, mm)
            result["frame_stages"] = fs = measure_frame_stages(cia, inp, syn, se3, local_rank)
            # the whole per-frame loop of Odometry::DoRegister on this 132 k-point frame, every data-parallel step through the
            # library with host buffers in and out: the steps either side + one Register call on the sampled keypoints
            for route, reg_ms in (("gn", result["frames_per_sec"]["ms_per_frame"]), ("robust", result["robust_route"]["ms_per_frame"])):
                ms = fs["grid_sampling_ms"] + fs["keypoint_sampling_ms"] + reg_ms + fs["undistortion_ms"] + fs["map_update_ms"]
                result.setdefault("frame_pipeline", {})[route] = {"ms_per_frame": ms, "frames_per_sec": 1e3 / ms}
        if not args.no_cpu_baseline and args.workload == "B2" and world == 1:      # rank 0 at N = 1 only
            result["cpu_baseline"] = cpu_baseline(inp, pose0, world0, args)
            if result["cpu_baseline"]["value"]:
                result["gpu_over_cpu"] = value / world / result["cpu_baseline"]["value"]
                result["gpu_over_cpu_reference_shaped"] = value / world / result["cpu_baseline"]["reference_shaped"]["value"]
            if "robust_route" in result:                   # same frame, same settings: the two poses must agree
                rr, cr = result["robust_route"], result["cpu_baseline"]["robust_route"]
                tr, rot = se3.pose_error(np.array(rr.pop("pose")), np.array(cr.pop("pose")))
                rr["gpu_vs_cpu_pose_m_rad"] = [tr, rot]
                rr["gpu_over_cpu_1core"] = cr["ms_per_frame"] / rr["ms_per_frame"]
        print(json.dumps(result), flush=True)
    if dist is not None:
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
    for _ in range(6):
        m = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(0.8, 0.1, 30)], default_radius=0.75,
                                                    device=device, device_updates=True))
        m.InsertPointCloud(inp["map_points"])
        maps.append(m)
    times, counts = [], {}
    for rep, m in enumerate(maps):
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


def cpu_baseline(inp, pose0, world0, args):
    """The oracle (a port, not the reference: it cannot be built here) timed on this box's host cores on the same
    workload: CPU-N = OpenMP over keypoints on all cores, plus the faithful serial CPU-1 the reference actually runs
    (its GN keypoint loop has no `#pragma omp`, ct_icp.cpp:753)."""
    from oracle import oracle as orc
    cores = usable_cores()
    om = orc.Map(resolutions=[(0.8, 0.1, 30)], default_radius=0.75)
    om.insert(inp["map_points"])
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
    return {"value": v_n, "unit": "keypoints/s", "cores": cores, "kind": "port", "frame_stages": stages,
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
