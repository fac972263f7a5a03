#!/usr/bin/env python3
"""Phase clocks of k_accumulate_rows (variant 3 = instrumented instantiation) on a bench workload. Measurement script.
usage: rows_prof.py [B2|B2-small|D] [iterations] [order: -1|0|1]"""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ct_icp_amd as cia
from ct_icp_amd import se3, synthetic as syn

wl = sys.argv[1] if len(sys.argv) > 1 else "B2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
order = int(sys.argv[3]) if len(sys.argv) > 3 else -1
if wl == "D":
    inp = bench.make_inputs_dense(0); rp, radius = cia.ResolutionParam(0.5, 0.03, 40), 0.8
elif wl == "B2":
    inp = bench.make_inputs_large(0); rp, radius = cia.ResolutionParam(0.8, 0.1, 30), 0.75
else:
    inp = bench.make_inputs(0, 20); rp, radius = cia.ResolutionParam(0.8, 0.1, 30), 0.75
gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[rp], default_radius=radius))
for s0 in range(0, len(inp["map_points"]), 2_000_000):
    gm.InsertPointCloud(inp["map_points"][s0:s0 + 2_000_000])
gm.Sync()
raw, t = inp["raw"], inp["t"]
pose0 = syn.perturb_pose(inp["pose_gt"], 0.003, 0.03, seed=4)
world0 = se3.ct_transform(pose0, inp["tbe"], t, raw)
o = cia.CTICPOptions(solver=cia.GN, num_iters_icp=iters, threshold_orientation_norm=0.0, debug_print=False)
s = cia.GnSolver(gm)
s.set_ordering(order); s.set_variant(3); s.set_profiling(True)
s.set_keypoints(raw, world0, t)
s.phase_cycles(reset=True); s.traffic_counters(reset=True)
pose, summ, _ = s.solve(pose0, inp["tbe"], o)
pc = s.phase_cycles()
probes, points = s.traffic_counters()
ms, n = s.kernel_timing()
names = ["A transform", "B1 probes", "B2 stream", "B2 prunes", "B3 select", "B4 handover", "C", "D"]
tot = float(sum(pc[:8])) or 1.0
print(json.dumps(dict(workload=wl, order=order, kernel_ms=round(ms, 4), launches=n, waves=pc[11], fast_path_rounds=round(pc[8] / max(pc[9], 1), 3),
                      slowest_over_mean=round(pc[10] / (tot / max(pc[11], 1)), 2), probes_per_kp=round(probes / n / len(t), 2),
                      points_per_kp=round(points / n / len(t), 1), phases={k: round(v / tot, 3) for k, v in zip(names, pc[:8])})))
if os.environ.get("TIMELINE"):
    tl = s.wave_timeline()
    tl = tl[(tl[:, 1] > 0)]
    t0 = tl[:, 0].min()
    start, end = (tl[:, 0] - t0).astype(float), (tl[:, 1] - t0).astype(float)
    dur = end - start
    pct = lambda a: [round(float(np.percentile(a, q)), 0) for q in (0, 10, 50, 90, 99, 100)]
    print("waves", len(tl), "start pct", pct(start), "end pct", pct(end), "dur pct", pct(dur))
    blk = np.arange(len(tl)) // 4
    for x in range(8):
        m = (blk % 8) == x
        print("xcd", x, "n", int(m.sum()), "dur med", float(np.median(dur[m])), "end max", float(end[m].max()), "fast rounds", float(tl[m, 2].sum() / max(1, tl[m, 3].sum())))
