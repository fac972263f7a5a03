#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "^FAILED|passed|failed|^E  " gpurun_out/pytest_gpu.log | cut -c1-260 | head -40
for wl in B1; do for p in 1 0; do
  CTGN_PERSISTENT=$p timeout 300 python bench.py --workload $wl --sub none --no-pmc --steps 200 --warmup 20 2>gpurun_out/b1_$p.err | tail -1 > gpurun_out/b1_$p.json
  python -c "import sys,json; d=json.loads(open('gpurun_out/b1_$p.json').read()); r=d['roofline']; print('$wl persistent=$p step_ms=%.4f kernel=%.4f first=%.4f later=%.4f steady_step=%.4f steadyk=%.4f' % (d['ms_per_step'], r['kernel_ms_avg'], r['first_iteration']['kernel_ms'], r['later_iterations']['kernel_ms'], r['steady_state_ms_per_step'], r['steady_state_kernel_ms_avg']))"
done; done
python - <<'PY'
import time, numpy as np, argparse, sys
sys.path.insert(0, '.')
import bench, ct_icp_amd as cia
from ct_icp_amd import se3, synthetic as syn
import torch
args = argparse.Namespace(map_frames=20, d_sweeps=8, d_radius=100.0, local_rank=0)
for wl in ("B1", "C"):
    W = bench.build_workload(wl, 0, 1, args, cia, syn, se3)
    s = cia.GnSolver(W["gm"]); s.set_rewind(True)
    s.set_keypoints(W["raw"], W["world0"], W["t"])
    o = cia.CTICPOptions(solver=cia.GN, num_iters_icp=W["ipf"], min_number_neighbors=W["min_nb"], threshold_orientation_norm=0.0, debug_print=False)
    def loop(k):
        for _ in range(k):
            s.rewind(); s.gn_begin(W["pose0"], W["inp"]["tbe"], o, W["mm"]); s.gn_iterate(W["ipf"])
    loop(60); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop(100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pose, summ, _ = s.gn_end()
    print(wl, "fresh solves, no profiling: ms per iteration %.4f  (iters %d, n_used %d)" % (dt / 100 / W["ipf"] * 1e3, summ.num_iters, summ.num_residuals_used))
PY
