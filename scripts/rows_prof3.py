#!/usr/bin/env python3
"""Phase clocks of k_accumulate_rows (variant 3 = instrumented instantiation) on FRESH solves of a bench workload, per iteration of the
solve: the first search (radius only), then the pool checks + the searches of the keypoints they do not certify. Measurement script.
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
names = ["A transform", "B1 probes+lists", "B2 stream", "B2 prunes", "B3 select", "B4 handover", "V pool check"]
ipf = W["ipf"]
acc = [[0] * 12 for _ in range(ipf)]
n = len(W["t"])
for k in range(solves + 2):
    s.rewind(); s.gn_begin(W["pose0"], W["inp"]["tbe"], o, W["mm"])
    s.phase_cycles(reset=True)
    for it in range(ipf):
        s.gn_iterate(1)
        pc = s.phase_cycles(reset=True)
        if k >= 2:
            for i in range(12):
                acc[it][i] += pc[i]
    s.gn_end()
shown = list(range(min(ipf, 5))) + ([ipf - 1] if ipf > 5 else [])
for it in shown:
    pc = acc[it]
    tot = float(sum(pc[:7])) or 1.0
    per_kp = {k_: round(v / solves / n, 1) for k_, v in zip(names, pc[:7])}
    print(json.dumps(dict(workload=wl, iteration=it, launches=solves, wave_cycles_per_keypoint=round(tot / solves / n, 1), waves=pc[11] // max(solves, 1),
                          certified_frac=round(pc[8] / solves / n, 4), certified_same_order_frac=round(pc[7] / solves / n, 4), search_rounds_per_wave=round(pc[9] / max(pc[11], 1), 2),
                          phases_frac={k_: round(v / tot, 3) for k_, v in zip(names, pc[:7])}, wave_cycles_per_keypoint_by_phase=per_kp)))
