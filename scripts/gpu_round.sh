#!/bin/bash
# One GPU-box session: smoke, parity tests, bench for the kernel variants, rocprofv3 kernel trace. Logs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os;print(len(os.sched_getaffinity(0)))") > gpurun_out/box.log 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
fi
for v in ${VARIANTS:-0 2 1}; do
  extra="--no-cpu-baseline --no-pmc"; [ "$v" = "0" ] && [ "${SKIP_CPU:-0}" != "1" ] && extra=""
  timeout 600 python bench.py --steps ${STEPS:-200} --warmup 20 --variant $v $extra > gpurun_out/bench_v$v.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_v$v.log
done
if [ "${PROFILE:-1}" = "1" ]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o trace -- python "$OLDPWD/bench.py" --steps 200 --warmup 0 --inner) > gpurun_out/rocprof.log 2>&1
  find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r cat > gpurun_out/kernel_stats.csv
  find gpurun_out/prof -name "*kernel_trace.csv" -delete
  cut -c1-160 gpurun_out/kernel_stats.csv
fi
tail -n 5 gpurun_out/smoke.log; tail -n 30 gpurun_out/pytest_gpu.log; for f in gpurun_out/bench_v*.log; do tail -n 2 $f | cut -c1-1500; done; true
