cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s15; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -k "not rehearsal" > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 900 python tests/odometry_vs_reference.py --sequence 0 --frames 150 --solver GN --impl ref-gpu-armed,ref-gpu-armed-device-shuffle,ctgn --out $O/vs_reference_150.json > /dev/null 2> $O/vs_reference.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/s15/vs_reference_150.json"))
for k,v in r["runs"].items():
    print(k, round(v["ms_per_frame_mean"],3), v.get("ms_per_frame_mean_after_startup"), v.get("host_time_table_ms"), v.get("arm_time_table_ms"))
PY
python scripts/sequence_run.py --help 2>&1 | head -5
