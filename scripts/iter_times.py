#!/usr/bin/env python3
"""Search-kernel time of EACH iteration of fresh solves (HIP events of the profiling mode, one iteration per gn_iterate call), for a
list of ablation masks: which iteration pays what. Measurement script.
usage: iter_times.py [B2|B2-small|B1|C|D] [mask ...]      (mask 2048 = no pools; bits 12-15 = spare pool members)"""
import argparse, sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ct_icp_amd as cia
from ct_icp_amd import se3, synthetic as syn
import torch

wl = sys.argv[1] if len(sys.argv) > 1 else "B2"
masks = [int(a) for a in sys.argv[2:]] or [0]
args = argparse.Namespace(map_frames=20, d_sweeps=8, d_radius=100.0, local_rank=0)
W = bench.build_workload(wl, 0, 1, args, cia, syn, se3)
s = cia.GnSolver(W["gm"])
s.set_rewind(True)
s.set_keypoints(W["raw"], W["world0"], W["t"])
ipf = W["ipf"]
o = cia.CTICPOptions(solver=cia.GN, num_iters_icp=ipf, min_number_neighbors=W["min_nb"], threshold_orientation_norm=0.0, debug_print=False)
sync = torch.cuda.synchronize
for _ in range(150):                                  # clocks
    s.rewind(); s.gn_begin(W["pose0"], W["inp"]["tbe"], o, W["mm"]); s.gn_iterate(ipf)
s.gn_end()
for mask in masks:
    s.set_ablation(mask)
    # un-profiled step time
    for _ in range(20):
        s.rewind(); s.gn_begin(W["pose0"], W["inp"]["tbe"], o, W["mm"]); s.gn_iterate(ipf)
    sync(); t0 = time.perf_counter()
    for _ in range(60):
        s.rewind(); s.gn_begin(W["pose0"], W["inp"]["tbe"], o, W["mm"]); s.gn_iterate(ipf)
    sync(); step_ms = (time.perf_counter() - t0) / 60 / ipf * 1e3
    pose, summ, _ = s.gn_end()
    s.set_profiling(True)
    # the events are read when a solve ends: solves of 1, 2, ... iterations, the totals differenced
    solves, shown_n = 10, min(ipf, 6)
    totals = []
    for j in range(1, shown_n + 1):
        oj = cia.CTICPOptions(solver=cia.GN, num_iters_icp=j, min_number_neighbors=W["min_nb"], threshold_orientation_norm=0.0, debug_print=False)
        for k in range(solves + 2):
            if k == 2:
                s.kernel_timing(reset=True)
            s.rewind(); s.gn_begin(W["pose0"], W["inp"]["tbe"], oj, W["mm"]); s.gn_iterate(j); s.gn_end()
        ms, launches = s.kernel_timing(reset=True)
        totals.append(ms * launches / solves)
    acc = [totals[0]] + [totals[i] - totals[i - 1] for i in range(1, shown_n)]
    s.set_profiling(False)
    print(json.dumps(dict(workload=wl, mask=mask, step_ms=round(step_ms, 4), n_used=summ.num_residuals_used,
                          search_ms_by_iteration=[round(a, 4) for a in acc], search_ms_mean_shown=round(sum(acc) / shown_n, 4))))
