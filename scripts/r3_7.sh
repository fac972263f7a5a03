#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
CTGN_PERSIST_TIMES=1 CTGN_PERSISTENT=1 timeout 300 python scripts/fresh_probe.py B1 2>&1 | grep -E "blocks|slowest|device stamps" | tail -5
