#!/usr/bin/env python3
"""Where the host time of a small-frame `Register` goes: 50 calls of CT_ICP_Registration::Register on the bench's
1 679-keypoint frame (bench.measure_frames_per_sec) under `rocprofv3 --hip-trace --kernel-trace`, then every HIP API call and kernel of ONE
steady-state call on a common time line, and the per-call means.

  run:     (cd /tmp && rocprofv3 --hip-trace --kernel-trace --output-format csv -d OUT -o t -- python scripts/register_api_trace.py run [--robust])
  report:  python scripts/register_api_trace.py report OUT/.../t_hip_api_trace.csv OUT/.../t_kernel_trace.csv
Measurement script."""
import collections
import csv
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(robust):
    import argparse
    import numpy as np
    import bench
    import ct_icp_amd as cia
    from ct_icp_amd import se3, synthetic as syn
    args = argparse.Namespace(map_frames=20, d_sweeps=8, d_radius=100.0, local_rank=0)
    W = bench.build_workload("B1", 0, 1, args, cia, syn, se3)
    inp = W["inp"]
    raw, t = inp["raw"], inp["t"]
    sel = syn.grid_sample_indices(raw, 0.5)
    sel = sel[syn.grid_sample_indices(raw[sel], 1.5)]
    pose0 = syn.perturb_pose(inp["pose_gt"], 0.003, 0.03, seed=4)
    kps = np.zeros(len(sel), dtype=cia.WPOINT3D_DTYPE)
    kps["raw_point"], kps["t"] = raw[sel], t[sel]
    world0 = se3.ct_transform(pose0, inp["tbe"], t[sel], raw[sel])
    opts = cia.CTICPOptions(solver=cia.CERES if robust else cia.GN, num_iters_icp=5, threshold_orientation_norm=0.1, debug_print=False) if not robust else \
        cia.CTICPOptions(solver=cia.CERES, num_iters_icp=5, ls_max_num_iters=5, debug_print=False)
    reg = cia.CT_ICP_Registration(opts)
    times = []
    for i in range(60):
        kps["world_point"] = world0
        frame = cia.TrajectoryFrame.from_pose14(pose0, *inp["tbe"])
        t0 = time.perf_counter()
        reg.Register(W["gm"], kps, frame, W["mm"])
        times.append(time.perf_counter() - t0)
        if i % 10 == 9:
            time.sleep(0.002)          # a gap in the time line between groups of calls
    print(f"Register x 60, {len(sel)} keypoints: median of the last 50 {1e3 * np.median(times[10:]):.4f} ms (under the tracer)")


def report(api_csv, kernel_csv):
    api = list(csv.DictReader(open(api_csv)))
    ker = list(csv.DictReader(open(kernel_csv)))
    name_k = "Function" if "Function" in api[0] else "Name"
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[name_k], "api") for r in api]
    ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60], "kernel") for r in ker]
    ev.sort()
    # a Register call = from one ctgn-side hipSetDevice ... to its hipStreamSynchronize: cut the API stream at every gap > 20 us with no call in flight
    apis = [e for e in ev if e[3] == "api"]
    calls, cur = [], [apis[0]]
    for e in apis[1:]:
        if e[0] - max(x[1] for x in cur) > 15_000:
            calls.append(cur)
            cur = []
        cur.append(e)
    calls.append(cur)
    reg = [c for c in calls if sum(1 for e in c if e[2] == "hipLaunchKernel" or e[2] == "hipModuleLaunchKernel" or "Launch" in e[2]) >= 10]
    steady = reg[len(reg) // 3:]
    print(f"{len(reg)} Register-sized bursts of HIP API calls found; statistics over the last {len(steady)}")
    span = [c[-1][1] - c[0][0] for c in steady]
    print(f"first API call -> end of the last one: mean {sum(span) / len(span) / 1e3:.1f} us, min {min(span) / 1e3:.1f} us")
    agg = collections.OrderedDict()
    for c in steady:
        for e in c:
            a = agg.setdefault(e[2], [0, 0])
            a[0] += 1
            a[1] += e[1] - e[0]
    print(f"\nper Register call (means over {len(steady)} calls):")
    print(f"  {'HIP API':40s}{'calls':>8s}{'us in the call':>16s}")
    tot = 0.0
    for k, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:40s}{n / len(steady):8.1f}{ns / len(steady) / 1e3:16.2f}")
        tot += ns / len(steady) / 1e3
    print(f"  {'sum of API time':40s}{'':8s}{tot:16.2f}")
    c = steady[len(steady) // 2]
    t0 = c[0][0]
    print("\none call on the time line (us from its first API call; kernels indented):")
    lo, hi = c[0][0], c[-1][1]
    last_end = t0
    for e in ev:
        if e[1] < lo or e[0] > hi:
            continue
        if e[3] == "api":
            gap = (e[0] - last_end) / 1e3
            print(f"  {(e[0] - t0) / 1e3:9.2f} +{(e[1] - e[0]) / 1e3:7.2f}  {e[2]}" + (f"      (host gap before: {gap:.1f})" if gap > 2.0 else ""))
            last_end = max(last_end, e[1])
        else:
            print(f"  {'':9s}        {'':4s}[{(e[0] - t0) / 1e3:8.2f} .. {(e[1] - t0) / 1e3:8.2f}] {e[2]}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run("--robust" in sys.argv)
    else:
        report(sys.argv[2], sys.argv[3])
