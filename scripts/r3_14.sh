#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "^FAILED|passed|failed|^E  " gpurun_out/pytest_gpu.log | cut -c1-260 | head -30
REPS=2 bash scripts/ab3.sh > gpurun_out/ab3.txt 2>&1; cat gpurun_out/ab3.txt
for lib in libctgn_base.so libctgn.so; do for wl in B1 C; do echo "$lib $wl persistent=0"; CTGN_LIB_PATH=$PWD/ct_icp_amd/$lib CTGN_PERSISTENT=0 timeout 300 python scripts/fresh_probe.py $wl 2>&1 | grep -E "back-to-back|device stamps" | tail -2; done; done
