#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 40 gpurun_out/pytest_gpu.log
REPS=2 bash scripts/ab3.sh > gpurun_out/ab3.txt 2>&1
cat gpurun_out/ab3.txt
timeout 300 python scripts/rows_prof3.py B2 6 > gpurun_out/rows_prof3_B2.txt 2>&1; cat gpurun_out/rows_prof3_B2.txt | tail -3
