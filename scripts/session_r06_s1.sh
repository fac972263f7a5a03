cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/s1
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc) > gpurun_out/s1/box.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s1/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/s1/smoke.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -k "frame_steps or frame_pipeline" > gpurun_out/s1/pytest_steps.log 2>&1; echo "rc=$?" >> gpurun_out/s1/pytest_steps.log
timeout 1200 python -m pytest tests/test_odometry_glue.py -m gpu -q -s --timeout 900 -p no:cacheprovider > gpurun_out/s1/pytest_odo.log 2>&1; echo "rc=$?" >> gpurun_out/s1/pytest_odo.log
timeout 1500 python tests/odometry_vs_reference.py --sequence 0 --frames 150 --solver GN --impl ref-gpu,ref-gpu-armed,ctgn --out gpurun_out/s1/vs_reference_150.json > /dev/null 2> gpurun_out/s1/vs_reference.err
tail -3 gpurun_out/s1/smoke.log; tail -15 gpurun_out/s1/pytest_steps.log; tail -25 gpurun_out/s1/pytest_odo.log; tail -5 gpurun_out/s1/vs_reference.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/s1/vs_reference_150.json"))
for k,v in r["runs"].items():
    print(k, v["ms_per_frame_mean"], v.get("ms_per_frame_mean_after_startup"), v.get("host_time_table_ms"), v["failures"], v["err_tr_max"])
print(r["between_runs"].keys())
PY
