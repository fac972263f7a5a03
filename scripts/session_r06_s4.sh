cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -k "frame_steps or frame_pipeline or device_shuffle" > $O/pytest_steps.log 2>&1; echo "rc=$?" >> $O/pytest_steps.log
timeout 1200 python -m pytest tests/test_odometry_glue.py -m gpu -q -s --timeout 900 -p no:cacheprovider > $O/pytest_odo.log 2>&1; echo "rc=$?" >> $O/pytest_odo.log
timeout 1500 python tests/odometry_vs_reference.py --sequence 0 --frames 150 --solver GN --impl ref-gpu,ref-gpu-armed,ref-gpu-armed-device-shuffle,ctgn --out $O/vs_reference_150.json > /dev/null 2> $O/vs_reference.err
tail -15 $O/pytest_steps.log; tail -25 $O/pytest_odo.log; tail -6 $O/vs_reference.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/s4/vs_reference_150.json"))
for k,v in r["runs"].items():
    print(k, round(v["ms_per_frame_mean"],3), v.get("ms_per_frame_mean_after_startup"), v.get("host_time_table_ms"), v.get("arm_time_table_ms"), v["failures"], round(v["err_tr_max"],3))
PY
