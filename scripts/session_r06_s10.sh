cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_robust.py tests/test_odometry_glue.py -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for rep in 1 2; do python scripts/register_time.py B1 stop_poll=0,1 2>&1 | tail -4; done > $O/register_stop_poll.txt; cat $O/register_stop_poll.txt
python - <<'PY' > $O/register_1679.txt 2>&1
import argparse, sys, os
sys.path.insert(0, os.getcwd())
import bench, ct_icp_amd as cia
from ct_icp_amd import se3, synthetic as syn, _lib as L
args = argparse.Namespace(map_frames=20, d_sweeps=8, d_radius=100.0, local_rank=0)
W = bench.build_workload("B2", 0, 1, args, cia, syn, se3)
for rep in range(2):
    for v in (0, 1):
        L.lib().ctgn_set_tuning(b"stop_poll", float(v))
        r = bench.measure_frames_per_sec(cia, W["gm"], W["inp"], syn, se3, W["mm"], reps=80)
        print(f"B2 frame, stop_poll={v}: Register {r['ms_per_frame']:.4f} ms ({r['keypoints']} keypoints, {r['gn_iterations']} iterations)")
PY
cat $O/register_1679.txt
timeout 900 python tests/odometry_vs_reference.py --sequence 0 --frames 150 --solver GN --impl ref-gpu-armed,ref-gpu-armed-device-shuffle,ctgn --out $O/vs_reference_150.json > /dev/null 2> $O/vs_reference.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/s10/vs_reference_150.json"))
for k,v in r["runs"].items():
    print(k, round(v["ms_per_frame_mean"],3), v.get("ms_per_frame_mean_after_startup"), v.get("host_time_table_ms"), v.get("arm_time_table_ms"))
PY
