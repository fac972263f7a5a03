#!/usr/bin/env python3
"""Phase clocks of k_accumulate_rows (variant 3 = instrumented instantiation) on FRESH solves of a bench workload, split into the first
search of a solve (radius only) and the later ones (carried-over bound). Measurement script.
usage: rows_prof3.py [B2|B2-small|B1|C|D] [solves]"""
import argparse, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ct_icp_amd as cia
from ct_icp_amd import se3, synthetic as syn

wl = sys.argv[1] if len(sys.argv) > 1 else "B2"
solves = int(sys.argv[2]) if len(sys.argv) > 2 else 6
args = argparse.Namespace(map_frames=20, d_sweeps=8, d_radius=100.0, local_rank=0)
W = bench.build_workload(wl, 0, 1, args, cia, syn, se3)
s = cia.GnSolver(W["gm"])
s.set_variant(3); s.set_rewind(True)
s.set_keypoints(W["raw"], W["world0"], W["t"])
o = cia.CTICPOptions(solver=cia.GN, num_iters_icp=W["ipf"], min_number_neighbors=W["min_nb"], threshold_orientation_norm=0.0, debug_print=False)
names = ["A transform", "B1 probes+lists", "B2 stream", "B2 prunes", "B3 select", "B4 handover", "C", "D"]
acc = {"first": [0] * 12, "later": [0] * 12}
n = len(W["t"])
for k in range(solves + 2):
    s.rewind(); s.gn_begin(W["pose0"], W["inp"]["tbe"], o, W["mm"])
    s.phase_cycles(reset=True)
    s.gn_iterate(1)
    p0 = s.phase_cycles(reset=True)
    s.gn_iterate(W["ipf"] - 1)
    p1 = s.phase_cycles(reset=True)
    s.gn_end()
    if k >= 2:
        for i in range(12):
            acc["first"][i] += p0[i]; acc["later"][i] += p1[i]
for key, launches in (("first", solves), ("later", solves * (W["ipf"] - 1))):
    pc = acc[key]
    tot = float(sum(pc[:8])) or 1.0
    per_kp = {k_: round(v / launches / n, 1) for k_, v in zip(names, pc[:8])}
    print(json.dumps(dict(workload=wl, kind=key, launches=launches, wave_cycles_per_keypoint=round(tot / launches / n, 1), waves=pc[11] // max(launches, 1),
                          phases_frac={k_: round(v / tot, 3) for k_, v in zip(names, pc[:8])}, wave_cycles_per_keypoint_by_phase=per_kp)))
