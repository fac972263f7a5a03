"""Phase timings of the frame pipeline on the bench's B2 frame (CTGN_TUNING="frame_timing=1" marks on stderr)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CTGN_TUNING"] = (os.environ.get("CTGN_TUNING", "") + ",frame_timing=1").lstrip(",")
import bench
import ct_icp_amd as cia
from ct_icp_amd import synthetic as syn

small = len(sys.argv) > 1 and sys.argv[1] == "small"
inp = bench.make_inputs(0, 20) if small else bench.make_inputs_large(0)
m = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(0.8, 0.1, 30)], default_radius=0.75, device=0, device_updates=True))
for s0 in range(0, len(inp["map_points"]), 100_000):
    m.InsertPointCloud(inp["map_points"][s0:s0 + 100_000])
fp = cia.FramePipeline(m, 0.5, 1.5)
pose0 = syn.perturb_pose(inp["pose_gt"], 0.003, 0.03, seed=4)
o5 = cia.CTICPOptions(solver=cia.GN, num_iters_icp=5, debug_print=False)
for want_all in (True, False):
    for _ in range(4):
        t0 = time.perf_counter()
        r = fp.register(inp["raw"], inp["t"], pose0, inp["tbe"], o5, want_all=want_all, want_sampled=False)
        print(f"python-side total {1e6 * (time.perf_counter() - t0):.0f} us (want_all={want_all})", file=sys.stderr)
fp.update_map(r["pose"][11:14], 100.0, True)
for _ in range(3):
    fp.update_map(r["pose"][11:14], 100.0, False)
for _ in range(3):
    fp.update_map(r["pose"][11:14], 100.0, True)

# the whole frame as one ctgn_frame call, on fresh maps (it changes the map): fused map update (default) unless CTGN_FRAME_UNFUSED is set
for rep in range(3):
    m2 = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(0.8, 0.1, 30)], default_radius=0.75, device=0, device_updates=True))
    for s0 in range(0, len(inp["map_points"]), 100_000):
        m2.InsertPointCloud(inp["map_points"][s0:s0 + 100_000])
    fp2 = cia.FramePipeline(m2, 0.5, 1.5)
    for _ in range(2):
        fp2.register(inp["raw"], inp["t"], pose0, inp["tbe"], o5, want_all=True, want_sampled=False)
    fp2.update_map(pose0[11:14], 100.0, False)
    for want_all in (True,):
        t0 = time.perf_counter()
        fp2.frame(inp["raw"], inp["t"], pose0, inp["tbe"], o5, 100.0, want_all=want_all, want_sampled=False)
        print(f"python-side ctgn_frame {1e6 * (time.perf_counter() - t0):.0f} us (want_all={want_all}, unfused={'CTGN_FRAME_UNFUSED' in os.environ})", file=sys.stderr)
