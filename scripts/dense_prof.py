#!/usr/bin/env python3
"""Phase clocks of k_search_dense on the bench's B2 workload (variant 3 = instrumented instantiation). Measurement script."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ct_icp_amd as cia
from ct_icp_amd import se3, synthetic as syn

inp = bench.make_inputs(0, 20)
gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(0.8, 0.1, 30)], default_radius=0.75))
gm.InsertPointCloud(inp["map_points"]); gm.Sync()
raw, t = inp["raw"], inp["t"]
pose0 = syn.perturb_pose(inp["pose_gt"], 0.003, 0.03, seed=4)
world0 = se3.ct_transform(pose0, inp["tbe"], t, raw)
o = cia.CTICPOptions(solver=cia.GN, num_iters_icp=int(sys.argv[1]) if len(sys.argv) > 1 else 20, threshold_orientation_norm=0.0, debug_print=False)
names = ["phaseA", "probes", "mirror", "pass1", "pivot", "pass2", "rank", "sums", "fallback"]
for ab in [int(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"])]:
    s = cia.GnSolver(gm)
    s.set_ordering(1); s.set_search_kernel(1); s.set_variant(3); s.set_profiling(True); s.set_ablation(ab)
    s.set_keypoints(raw, world0, t)
    s.phase_cycles(reset=True)
    pose, summ, _ = s.solve(pose0, inp["tbe"], o)
    pc = s.phase_cycles()
    ms, n = s.kernel_timing(reset=True)
    tot = sum(pc[:9])
    print(json.dumps(dict(ablate=ab, kernel_ms=round(ms, 4), launches=n, waves=pc[11], runs_per_wave=round(pc[9] / max(pc[11], 1), 2), slowest_wave_clk=pc[10],
                          mean_wave_clk=int(tot / max(pc[11], 1)), clk_per_run={k: int(v / max(pc[9], 1)) for k, v in zip(names, pc[:9])})), flush=True)
