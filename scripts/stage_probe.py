#!/usr/bin/env python3
"""How often the home-voxel-group stage (tuning stage_lds, rows_tiles STAGE) runs on a bench workload: table fills, rounds that streamed
the table, search rounds — per iteration of one fresh solve. Measurement script.   usage: stage_probe.py [D|C] [tile_chunk]"""
import argparse, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ct_icp_amd as cia
from ct_icp_amd import se3, synthetic as syn, _lib as L

wl = sys.argv[1] if len(sys.argv) > 1 else "D"
chunk = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
args = argparse.Namespace(map_frames=20, d_sweeps=8, d_radius=100.0, local_rank=0)
W = bench.build_workload(wl, 0, 1, args, cia, syn, se3)
L.lib().ctgn_set_tuning(b"stage_lds", 1.0)
L.lib().ctgn_set_tuning(b"tile_chunk", chunk)
s = cia.GnSolver(W["gm"])
s.set_rewind(True)
s.set_keypoints(W["raw"], W["world0"], W["t"])
ipf = W["ipf"]
o = cia.CTICPOptions(solver=cia.GN, num_iters_icp=ipf, min_number_neighbors=W["min_nb"], threshold_orientation_norm=0.0, debug_print=False)
s.rewind(); s.gn_begin(W["pose0"], W["inp"]["tbe"], o, W["mm"])
s.phase_cycles(reset=True)
for it in range(ipf):
    s.gn_iterate(1)
    c = s.phase_cycles(reset=True)
    print(json.dumps(dict(workload=wl, tile_chunk=chunk, iteration=it, table_fills=c[7], staged_rounds=c[8], search_rounds=c[9], eligible_rounds=c[4], fills_given_up=c[6], points_per_fill=round(c[5] / max(c[7], 1), 1))))
s.gn_end()
