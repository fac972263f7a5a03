#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "^FAILED|passed|failed|^E  " gpurun_out/pytest_gpu.log | cut -c1-260 | head -30
for p in 1 0; do echo "CTGN_PERSISTENT=$p"; CTGN_PERSISTENT=$p timeout 300 python scripts/fresh_probe.py B1 2>&1 | grep -E "back-to-back|device stamps" | tail -2; CTGN_PERSISTENT=$p timeout 300 python scripts/fresh_probe.py C 2>&1 | grep -E "back-to-back|device stamps" | tail -2; done
