#!/bin/bash
# One GPU-box session: smoke, parity tests, bench for the kernel variants. Logs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc) > gpurun_out/box.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
for v in 0 2 1; do
  extra="--no-cpu-baseline"; [ "$v" = "0" ] && extra=""
  timeout 600 python bench.py --steps 10 --warmup 2 --variant $v $extra > gpurun_out/bench_v$v.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_v$v.log
done
tail -5 gpurun_out/smoke.log; tail -30 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/bench_v*.log
