#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "radix or ordering or config_d or device_map or adaptive" 2>&1 | tail -3
timeout 300 python scripts/sort_time.py 8 2>&1 | grep ordering
