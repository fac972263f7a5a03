"""Worker of test_config_e_two_gloo_ranks_share_one_gpu: one rank of a 2-rank gloo job running its share of a tiny config E
(11 sequences, seeds 10-20, KITTI lengths / 400, a 1000-column HDL-64E pattern) through ct_icp_amd.sequence_runner.run_batch on the one GPU."""
import importlib.util
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    from ct_icp_amd import se3, sequence_runner as sr, synthetic as syn
    import ct_icp_amd as cia
    spec = importlib.util.spec_from_file_location("ctgn_sequence_run", os.path.join(ROOT, "scripts", "sequence_run.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    seqs, lengths = mod.config_e_sequences(scale=400, azimuth_steps=1000)
    _, shares = sr.deal_sequences(lengths, world)
    makers = {sid: maker for sid, _, maker in seqs}
    mine, gts = {}, {}
    for sid in shares[rank]:
        scans, knots = makers[sid](sid)
        mine[sid] = scans
        gts[sid] = [syn.frame_pose14(knots, j) for j in range(len(scans))]
    # this rank's sequences back to back (what run_batch does for a rank; every sequence bootstraps from its own ground-truth poses)
    results = []
    for sid in shares[rank]:
        r = sr.run_sequence(mine[sid], device=0, solver=cia.GN, init_poses=gts[sid], init_frames=min(5, len(mine[sid])), max_distance=100.0)
        err = max([se3.pose_error(r["poses"][j], gts[sid][j])[0] for j in range(min(5, len(mine[sid])), len(mine[sid]))] or [0.0])
        results.append(dict(sequence=sid, frames=int(r["frames"]), failures=int(np.count_nonzero(~r["success"])), err=float(err), seconds=float(r["seconds"])))
    gathered = [None] * world
    dist.all_gather_object(gathered, results)
    if rank == 0:
        flat = [r for g in gathered for r in g]
        print(json.dumps(dict(frames=sum(r["frames"] for r in flat), sequences=sorted(r["sequence"] for r in flat), failures=sum(r["failures"] for r in flat),
                              err_max=max(r["err"] for r in flat), lengths=lengths, shares=shares,
                              wall_seconds=max(sum(r["seconds"] for r in g) for g in gathered))))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
