#!/usr/bin/env python3
"""Where does a fresh solve of a small frame spend its time? Host time of each call and the device time it leaves behind. Measurement script."""
import time, argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, ct_icp_amd as cia
from ct_icp_amd import se3, synthetic as syn
import torch
wl = sys.argv[1] if len(sys.argv) > 1 else "B1"
args = argparse.Namespace(map_frames=20, d_sweeps=8, d_radius=100.0, local_rank=0)
W = bench.build_workload(wl, 0, 1, args, cia, syn, se3)
s = cia.GnSolver(W["gm"]); s.set_rewind(True)
s.set_keypoints(W["raw"], W["world0"], W["t"])
o = cia.CTICPOptions(solver=cia.GN, num_iters_icp=W["ipf"], min_number_neighbors=W["min_nb"], threshold_orientation_norm=0.0, debug_print=2 if os.environ.get('SOLVE_CLOCKS') else False)
sync = torch.cuda.synchronize
def timed(f):
    sync(); t0 = time.perf_counter(); f(); t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    return (t1 - t0) * 1e6, (t2 - t0) * 1e6
for rep in range(3):
    for _ in range(30):
        s.rewind(); s.gn_begin(W["pose0"], W["inp"]["tbe"], o, W["mm"]); s.gn_iterate(W["ipf"])
    a = [timed(s.rewind) for _ in range(5)][-1]
    b = timed(lambda: s.gn_begin(W["pose0"], W["inp"]["tbe"], o, W["mm"]))
    c = timed(lambda: s.gn_iterate(W["ipf"]))
    out = []
    d = timed(lambda: out.append(s.gn_end()))
    sm = out[0][1]
    print("   device stamps per iteration (us): neighbourhood(search+residual+barrier) %.1f  solve(reduce+solve) %.1f  iteration %.1f" %
          (sm.avg_duration_neighborhood * 1e3, sm.avg_duration_solve * 1e3, sm.avg_duration_iter * 1e3))
    print(wl, "us host/total: rewind %.0f/%.0f  gn_begin %.0f/%.0f  gn_iterate(%d) %.0f/%.0f  gn_end %.0f/%.0f" % (*a, *b, W["ipf"], *c, *d))
    sync(); t0 = time.perf_counter()
    for _ in range(50):
        s.rewind(); s.gn_begin(W["pose0"], W["inp"]["tbe"], o, W["mm"]); s.gn_iterate(W["ipf"])
    sync(); print("  50 back-to-back solves: %.1f us per solve" % ((time.perf_counter() - t0) / 50 * 1e6))
    s.gn_end()

if "persist_times=1" in os.environ.get("CTGN_TUNING", ""):
    tl = s.wave_timeline(256).astype(np.int64)          # rows: iteration start, arrive, barrier passed, reduce done (10 ns ticks)
    t0 = tl[:, 0].min()
    print("blocks", len(tl), "start spread %.1f us" % ((tl[:, 0].max() - t0) / 100), " arrive (us after first start): min %.1f med %.1f max %.1f" %
          tuple((np.percentile(tl[:, 1] - t0, q) / 100) for q in (0, 50, 100)), " passed: min %.1f max %.1f" % ((tl[:, 2].min() - t0) / 100, (tl[:, 2].max() - t0) / 100),
          " reduce done: min %.1f max %.1f" % ((tl[:, 3].min() - t0) / 100, (tl[:, 3].max() - t0) / 100))
    order = np.argsort(tl[:, 1])
    print("slowest blocks", order[-6:], "their arrive", (tl[order[-6:], 1] - t0) / 100, "their start", (tl[order[-6:], 0] - t0) / 100)
